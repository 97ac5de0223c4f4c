"""PlyData / PlyElement look-alikes on top of the native reader/writer (gsx_ply_*), covering what the
reference uses of `plyfile`: PlyData.read, plydata['vertex'], vertices['x'], len(vertices),
vertices.data.dtype, assignment vertices['x'] = array, and writing the vertex element back
(deep_learning_segmentation.py:25-40, 311-332; ply_handler.py:5-37)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check

_NP_TYPES = ["<i1", "<u1", "<i2", "<u2", "<i4", "<u4", "<f4", "<f8"]


class PlyElement:
    """The vertex element.  `.data` is a numpy structured array viewing the native rows (zero copy)."""

    def __init__(self, owner):
        self._owner = owner
        lib = owner._lib
        h = owner._h
        n = lib.gsx_ply_num_vertices(h)
        stride = lib.gsx_ply_row_stride(h)
        names, formats, offsets = [], [], []
        for i in range(lib.gsx_ply_num_properties(h)):
            name, typ, off = C.c_char_p(), C.c_int32(), C.c_int64()
            check(lib.gsx_ply_property(h, i, C.byref(name), C.byref(typ), C.byref(off)))
            names.append(name.value.decode())
            formats.append(_NP_TYPES[typ.value])
            offsets.append(off.value)
        self.name = "vertex"
        dtype = np.dtype({"names": names, "formats": formats, "offsets": offsets, "itemsize": int(stride)})
        if n > 0:
            raw = (C.c_ubyte * (n * stride)).from_address(lib.gsx_ply_rows(h))
            # every numpy view derived from .data keeps `raw` alive, and `raw` keeps the PlyData (hence the native
            # mapping) alive: `x = PlyData.read(p)['vertex']['x']` must not dangle once the PlyData goes out of scope
            raw._gsx_owner = owner
            self.data = np.frombuffer(raw, dtype=dtype, count=n)
        else:
            self.data = np.zeros(0, dtype=dtype)

    def __len__(self):
        return len(self.data)

    def __getitem__(self, key):
        return self.data[key]

    def __setitem__(self, key, value):
        self.data[key] = value

    @property
    def properties(self):
        return list(self.data.dtype.names)

    def column_f32(self, name):
        """Multi-threaded native column extraction (any scalar type -> float32)."""
        out = np.empty(len(self), np.float32)
        check(self._owner._lib.gsx_ply_read_f32(self._owner._h, name.encode(), out.ctypes.data))
        return out


class PlyData:
    def __init__(self, handle):
        self._lib = _lib.lib()
        self._h = handle
        self.elements = [PlyElement(self)]

    @classmethod
    def read(cls, path_or_file):
        path = getattr(path_or_file, "name", path_or_file)
        h = C.c_void_p()
        check(_lib.lib().gsx_ply_open(str(path).encode(), C.byref(h)))
        return cls(h)

    def __getitem__(self, name):
        if name != "vertex":
            raise KeyError(name)
        return self.elements[0]

    def write(self, path_or_file, labels=None, text=False):
        """Vertex element only (+ `int label` when labels is given), binary little endian by default."""
        path = getattr(path_or_file, "name", path_or_file)
        lab = None if labels is None else np.ascontiguousarray(labels, dtype=np.int32)
        if lab is not None and len(lab) != len(self.elements[0]):
            raise ValueError("labels must have one entry per vertex")
        check(self._lib.gsx_ply_write(self._h, str(path).encode(), None if lab is None else lab.ctypes.data, 1 if text else 0))

    def close(self):
        if getattr(self, "_h", None):
            for e in self.elements:
                e.data = None
            self._lib.gsx_ply_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def write_vertex_ply(path, columns, text=False):
    """Write a new PLY from a dict name -> 1-D array (dtype decides the PLY type).  Pure numpy: used by
    tools and tests to create inputs; the product path writes through PlyData.write."""
    names = list(columns)
    arrs = [np.asarray(columns[k]) for k in names]
    n = len(arrs[0])
    tn = {"i1": "char", "u1": "uchar", "i2": "short", "u2": "ushort", "i4": "int", "u4": "uint", "f4": "float", "f8": "double"}
    dt = np.dtype([(k, a.dtype.newbyteorder("<")) for k, a in zip(names, arrs)])
    rec = np.empty(n, dt)
    for k, a in zip(names, arrs):
        rec[k] = a
    hdr = f"ply\nformat {'ascii' if text else 'binary_little_endian'} 1.0\nelement vertex {n}\n"
    hdr += "".join(f"property {tn[a.dtype.str[1:]]} {k}\n" for k, a in zip(names, arrs)) + "end_header\n"
    with open(path, "wb") as f:
        f.write(hdr.encode("ascii"))
        if text:
            for row in rec:
                f.write((" ".join(repr(v.item()) if isinstance(v.item(), float) else str(v.item()) for v in row) + "\n").encode())
        else:
            f.write(rec.tobytes())
