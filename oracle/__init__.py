"""ORACLE — CPU restatement of the reference algorithms (test infrastructure, not the product).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The C sources cite the reference lines they follow; this file is the ctypes glue.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "libgsx_oracle.so")
_lib = None


class Camera(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("width", C.c_int32), ("height", C.c_int32),
                ("R", C.c_double * 9), ("p", C.c_double * 3)]


class View(C.Structure):
    _fields_ = [("cam", Camera), ("seg", C.c_void_p), ("seg_w", C.c_int32), ("seg_h", C.c_int32),
                ("img_w", C.c_int32), ("img_h", C.c_int32)]


def build(force=False):
    srcs = [os.path.join(_DIR, f) for f in os.listdir(_DIR) if f.endswith(".c")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", _DIR, "-B", "libgsx_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.gsxo_project.restype = C.c_int
        _lib.gsxo_assign_labels.restype = C.c_int
        _lib.gsxo_assign_labels.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_int]
        _lib.gsxo_project_many.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.gsxo_project_many.restype = None
        _lib.gsxo_max_threads.restype = C.c_int
    return _lib


def camera_struct(cam):
    """cam: a cameras.json-style dict (fx, fy, width, height, rotation 3x3, position 3)."""
    c = Camera()
    c.fx, c.fy = float(cam["fx"]), float(cam["fy"])
    c.width, c.height = int(cam["width"]), int(cam["height"])
    R = np.asarray(cam["rotation"], dtype=np.float64).reshape(9)
    p = np.asarray(cam["position"], dtype=np.float64).reshape(3)
    for i in range(9):
        c.R[i] = R[i]
    for i in range(3):
        c.p[i] = p[i]
    return c


def project_many(positions, cam):
    """positions (N,3) f32 -> (x, y) int32 arrays, -1 where the reference returns None."""
    pos = np.ascontiguousarray(positions, dtype=np.float32)
    n = len(pos)
    x = np.empty(n, np.int32)
    y = np.empty(n, np.int32)
    c = camera_struct(cam)
    lib().gsxo_project_many(pos.ctypes.data, n, C.addressof(c), x.ctypes.data, y.ctypes.data)
    return x, y


def assign_labels(positions, cams, segmaps, img_sizes, threads=1):
    """Majority vote over the given (already filtered, in-order) views -> int32 labels (N,)."""
    pos = np.ascontiguousarray(positions, dtype=np.float32)
    n = len(pos)
    segs = [np.ascontiguousarray(s, dtype=np.int32) for s in segmaps]
    views = (View * max(1, len(cams)))()
    for v, (cam, seg, (iw, ih)) in enumerate(zip(cams, segs, img_sizes)):
        views[v].cam = camera_struct(cam)
        views[v].seg = seg.ctypes.data
        views[v].seg_h, views[v].seg_w = seg.shape
        views[v].img_w, views[v].img_h = int(iw), int(ih)
    labels = np.empty(n, np.int32)
    used = lib().gsxo_assign_labels(pos.ctypes.data, n, C.addressof(views), len(cams), labels.ctypes.data, threads)
    assign_labels.threads_used = used
    return labels


def max_threads():
    return lib().gsxo_max_threads()
