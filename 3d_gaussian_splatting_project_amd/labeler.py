"""Host-side mirror of the reference's labeler interface on top of libgsx.so.

Same names, argument meaning and results as /root/reference/deep_learning_segmentation.py:
load_cameras (:17), project_gaussian (:43), assign_labels (:241) — except that assign_labels
takes its segmentation maps from a callable / directory instead of running a model, because the
2-D segmentation networks are outside this library.
"""
import ctypes as C
import json
import os

import numpy as np

from . import _lib
from ._lib import Camera, check

try:  # CPython binding of the per-view hand-over (csrc/gsxfast.c); without it the ctypes path below does the same
    from . import _gsxfast as _fast
except ImportError:
    _fast = None
if os.environ.get("GSX_NO_FAST_BINDING"):
    _fast = None


def camera_array(cameras):
    """ctypes array of gsx_camera from Camera structs and/or cameras.json dicts"""
    return (Camera * len(cameras))(*[c if isinstance(c, Camera) else Camera.from_dict(c) for c in cameras])


def load_cameras(camera_file):
    """cameras.json -> list of dicts (reference: deep_learning_segmentation.py:17-22)."""
    with open(camera_file, "r") as f:
        return json.load(f)


class Context:
    """One GPU (gsx_ctx).  device: HIP ordinal; defaults to LOCAL_RANK or 0."""

    def __init__(self, device=None):
        self._lib = _lib.lib()
        if device is None:  # one process per GPU: LOCAL_RANK picks it (wrapping when ranks share a GPU in a rehearsal)
            device = int(os.environ.get("LOCAL_RANK", "0")) % max(1, self._lib.gsx_device_count())
        h = C.c_void_p()
        check(self._lib.gsx_create(int(device), C.byref(h)))
        self.h = h
        self._vote_view_addr = C.cast(self._lib.gsx_vote_view, C.c_void_p).value   # for the CPython fast path
        self.device = int(device)
        self._keep_alive = []   # device maps handed to vote_view until the ctx stream has consumed them

    def close(self):
        if getattr(self, "h", None):
            self._lib.gsx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_option(self, name, value):
        """Tuning knobs of gsx_set_option (spatial_sort, xcd_swizzle, vote_unroll); never change results."""
        check(self._lib.gsx_set_option(self.h, name.encode(), int(value)), self.h)

    # -- scene ---------------------------------------------------------------------------------
    def upload_positions(self, positions):
        """positions: (N,3) float array, or a structured array with a 'position' field
        (the reference's `gaussians`, deep_learning_segmentation.py:33-38)."""
        if getattr(positions, "dtype", None) is not None and positions.dtype.names and "position" in positions.dtype.names:
            positions = positions["position"]
        pos = np.asarray(positions, dtype=np.float32)
        if pos.ndim != 2 or pos.shape[1] != 3:
            raise ValueError("positions must have shape (N, 3)")
        x = np.ascontiguousarray(pos[:, 0])
        y = np.ascontiguousarray(pos[:, 1])
        z = np.ascontiguousarray(pos[:, 2])
        check(self._lib.gsx_upload_positions(self.h, len(pos), x.ctypes.data, y.ctypes.data, z.ctypes.data), self.h)
        self.n = len(pos)

    def upload_rows(self, rows, off_x, off_y, off_z):
        """AoS rows (e.g. PLY vertex records as a 2-D uint8 array or structured array)."""
        rows = np.ascontiguousarray(rows)
        check(self._lib.gsx_upload_positions_strided(self.h, len(rows), rows.ctypes.data, rows.strides[0], off_x, off_y,
                                                     off_z), self.h)
        self.n = len(rows)

    # -- projection probe ------------------------------------------------------------------------
    def project_one(self, position, camera):
        p = (C.c_float * 3)(*[float(v) for v in position])
        cam = Camera.from_dict(camera)
        x, y, vis = C.c_int32(), C.c_int32(), C.c_int32()
        check(self._lib.gsx_project_one(self.h, p, C.byref(cam), C.byref(x), C.byref(y), C.byref(vis)), self.h)
        return (x.value, y.value) if vis.value else None

    def project_all(self, camera):
        cam = Camera.from_dict(camera)
        x = np.empty(self.n, np.int32)
        y = np.empty(self.n, np.int32)
        check(self._lib.gsx_project_all(self.h, C.byref(cam), x.ctypes.data, y.ctypes.data), self.h)
        return x, y

    # -- vote --------------------------------------------------------------------------------------
    def vote_begin(self, n_classes, first_view=0, total_views=255):
        self._keep_alive.clear()
        check(self._lib.gsx_vote_begin(self.h, n_classes, first_view, total_views), self.h)

    _TORCH_DT = {"torch.int32": _lib.GSX_SEG_I32, "torch.int64": _lib.GSX_SEG_I64, "torch.uint8": _lib.GSX_SEG_U8_LABELS}
    _NP_DT = {np.dtype(np.int32): _lib.GSX_SEG_I32, np.dtype(np.int64): _lib.GSX_SEG_I64, np.dtype(np.uint8): _lib.GSX_SEG_U8_LABELS}

    def vote_view(self, camera, seg_map, image_size=None, packed_u8=False):
        """seg_map: 2-D int array of labels (-1..n_classes-1; any integer dtype, as the reference accepts) on the host,
        or a torch tensor on this GPU.  image_size: (width, height) of the input image; defaults to the map's own size.
        packed_u8: a uint8 map already holds label+1 (0 = label -1), the library's own compact form.
        Host maps are range-checked here (ValueError); device maps when the labels are fetched."""
        cam = camera if isinstance(camera, Camera) else Camera.from_dict(camera)
        if _fast is not None and type(seg_map) is np.ndarray and self.h is not None:
            # the usual case in C (csrc/gsxfast.c): a contiguous 2-D int32 / int64 / uint8 array; anything else is declined
            iw, ih = (-1, -1) if image_size is None else (int(image_size[0]), int(image_size[1]))
            rc = _fast.vote_view(self._vote_view_addr, self.h.value, C.addressof(cam), seg_map, packed_u8, iw, ih)
            if rc != _fast.DECLINED:
                check(rc, self.h)
                return
        if type(seg_map) is not np.ndarray and hasattr(seg_map, "data_ptr"):  # torch tensor on the device
            t = seg_map.contiguous()
            h, w = t.shape
            if not t.is_cuda:
                raise ValueError("tensor seg maps must live on the GPU; pass numpy arrays for host maps")
            dt = self._TORCH_DT.get(str(t.dtype))
            if dt is None:
                import torch
                t, dt = t.to(torch.int32), _lib.GSX_SEG_I32
            if packed_u8 and dt == _lib.GSX_SEG_U8_LABELS:
                dt = _lib.GSX_SEG_U8
            iw, ih = image_size if image_size is not None else (w, h)
            check(self._lib.gsx_vote_view_device(self.h, C.byref(cam), t.data_ptr(), dt, w, h, int(iw), int(ih)), self.h)
            self._keep_alive.append(t)   # the pack kernel runs asynchronously on the ctx stream
            return
        # (this runs once per view, 200 times in the 10 ms of a run: the usual case - a contiguous int32 / int64 / uint8 array -
        # takes the short way through, ~1.5 us of Python instead of 3)
        seg = seg_map if type(seg_map) is np.ndarray else np.asarray(seg_map)
        if seg.ndim != 2:
            raise ValueError("seg_map must be 2-D")
        dt = self._NP_DT.get(seg.dtype)
        if dt is None:
            seg = seg.astype(np.int32, copy=False)
            dt = _lib.GSX_SEG_I32
        elif packed_u8 and dt == _lib.GSX_SEG_U8_LABELS:
            dt = _lib.GSX_SEG_U8
        if not seg.flags.c_contiguous:
            seg = np.ascontiguousarray(seg)
        h, w = seg.shape
        iw, ih = (w, h) if image_size is None else (int(image_size[0]), int(image_size[1]))
        try:
            ptr = C.addressof(C.c_char.from_buffer(seg))   # 3x cheaper than seg.ctypes.data
        except (TypeError, ValueError):                    # a read-only or empty array
            ptr = seg.ctypes.data
        check(self._lib.gsx_vote_view(self.h, C.byref(cam), ptr, dt, w, h, iw, ih), self.h)

    def vote_views_device(self, cameras, seg_tensors, image_size=None, packed_u8=False):
        """Several device-resident maps of one shape and dtype in one call (gsx_vote_views_device: 16 maps per kernel
        launch).  seg_tensors: list of 2-D torch tensors on this GPU, or one 3-D tensor (views, H, W)."""
        ts = [t.contiguous() for t in seg_tensors]
        if not ts:
            return
        h, w = ts[0].shape
        dt = self._TORCH_DT[str(ts[0].dtype)]
        if packed_u8 and dt == _lib.GSX_SEG_U8_LABELS:
            dt = _lib.GSX_SEG_U8
        if any(t.shape != ts[0].shape or t.dtype != ts[0].dtype or not t.is_cuda for t in ts):
            raise ValueError("vote_views_device: all maps must share shape and dtype and live on the GPU")
        cams = (Camera * len(ts))(*[c if isinstance(c, Camera) else Camera.from_dict(c) for c in cameras])
        ptrs = (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        iw, ih = image_size if image_size is not None else (w, h)
        check(self._lib.gsx_vote_views_device(self.h, len(ts), cams, ptrs, dt, w, h, int(iw), int(ih)), self.h)
        self._keep_alive.extend(ts)

    def vote_rewind(self):
        check(self._lib.gsx_vote_rewind(self.h), self.h)

    def vote_finalize(self, to_host=True, out=None):
        """out: optional preallocated int32 array of N labels (saves the page faults of a fresh 12 MB array)."""
        if to_host and out is None:
            out = np.empty(self.n, np.int32)
        if to_host and (out.dtype != np.int32 or out.shape != (self.n,) or not out.flags.c_contiguous):
            raise ValueError("out must be a contiguous int32 array of N labels")
        try:
            check(self._lib.gsx_vote_finalize(self.h, out.ctypes.data if to_host else None), self.h)
        finally:
            self._keep_alive.clear()     # the stream has been synchronised
        return out if to_host else None

    def vote_flush(self):
        check(self._lib.gsx_vote_flush(self.h), self.h)

    def vote_tiebreak_keys(self):
        check(self._lib.gsx_vote_tiebreak_keys(self.h), self.h)

    def vote_labels_from_keys(self, to_host=True, out=None):
        if to_host and out is None:
            out = np.empty(self.n, np.int32)
        try:
            check(self._lib.gsx_vote_labels_from_keys(self.h, out.ctypes.data if to_host else None), self.h)
        finally:
            self._keep_alive.clear()
        return out if to_host else None

    def counts_device(self):
        n = C.c_int64()
        p = self._lib.gsx_vote_counts_device(self.h, C.byref(n))
        return p, n.value

    def keys_device(self):
        n = C.c_int64()
        p = self._lib.gsx_vote_keys_device(self.h, C.byref(n))
        return p, n.value

    def first_device(self):
        n = C.c_int64()
        p = self._lib.gsx_vote_first_device(self.h, C.byref(n))
        return p, n.value

    def slab_size(self):
        return self._lib.gsx_vote_slab_size(self.h)

    def vote_slab_reduce(self, recv_counts_ptr, recv_first_ptr):
        check(self._lib.gsx_vote_slab_reduce(self.h, C.c_void_p(recv_counts_ptr), C.c_void_p(recv_first_ptr)), self.h)

    def vote_flush_counts(self):
        check(self._lib.gsx_vote_flush_counts(self.h), self.h)

    def vote_slab_totals(self, recv_counts_ptr):
        check(self._lib.gsx_vote_slab_totals(self.h, C.c_void_p(recv_counts_ptr)), self.h)

    def cand_device(self):
        n = C.c_int64()
        p = self._lib.gsx_vote_cand_device(self.h, C.byref(n))
        return p, n.value

    def vote_tie_codes(self, cand_all_ptr):
        check(self._lib.gsx_vote_tie_codes(self.h, C.c_void_p(cand_all_ptr)), self.h)

    def codes_device(self):
        n = C.c_int64()
        p = self._lib.gsx_vote_codes_device(self.h, C.byref(n))
        return p, n.value

    def vote_tie_resolve(self, recv_codes_ptr):
        check(self._lib.gsx_vote_tie_resolve(self.h, C.c_void_p(recv_codes_ptr)), self.h)

    def vote_labels_from_sorted(self, sorted_ptr, to_host=True, out=None):
        if to_host and out is None:
            out = np.empty(self.n, np.int32)
        try:
            check(self._lib.gsx_vote_labels_from_sorted(self.h, C.c_void_p(sorted_ptr), out.ctypes.data if to_host else None),
                  self.h)
        finally:
            self._keep_alive.clear()
        return out if to_host else None

    def kmeans(self, points, colors, k, init_index, max_iter=100, tol=1e-4):
        """k_means_with_color (3D_clustering/k_means.py:107-151) with injected initial rows.
        Returns (centroids float32 (k, 6), labels int32 (n,), iterations, converged)."""
        pts = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
        col = np.ascontiguousarray(colors, np.float32).reshape(-1, 3)
        if len(pts) != len(col):
            raise ValueError("points and colors differ in length")
        init = np.ascontiguousarray(init_index, np.int64)
        if init.shape != (int(k),):
            raise ValueError("init_index must hold k row indices")
        labels = np.empty(len(pts), np.int32)
        cent = np.empty((int(k), 6), np.float32)
        iters, conv = C.c_int32(0), C.c_int32(0)
        check(self._lib.gsx_kmeans(self.h, len(pts), pts.ctypes.data, col.ctypes.data, int(k), init.ctypes.data, int(max_iter),
                                   float(tol), labels.ctypes.data, cent.ctypes.data, C.byref(iters), C.byref(conv)), self.h)
        return cent, labels, int(iters.value), bool(conv.value)

    # -- exchange protocol v4 (see include/gsx.h) ------------------------------------------------------------------------
    def vote_num_views(self):
        return int(self._lib.gsx_vote_num_views(self.h))

    def vote_export(self, reserve_bytes=0, blobs=True):
        """-> (pool device pointer, bytes in use, blobs uint8 (views, 256) or None)."""
        nv = self.vote_num_views()
        b = np.empty((nv, 256), np.uint8) if blobs else None
        ptr, used = C.c_void_p(), C.c_int64()
        check(self._lib.gsx_vote_export(self.h, int(reserve_bytes), b.ctypes.data if blobs and nv else None, C.byref(ptr),
                                        C.byref(used)), self.h)
        return ptr.value, int(used.value), b

    def vote_pool_bytes(self):
        return int(self._lib.gsx_vote_pool_bytes(self.h))

    def vote_link_bytes(self):
        """bytes of host maps sent over PCIe since vote_begin (compact records, or maps in pool form)"""
        return int(self._lib.gsx_vote_link_bytes(self.h))

    def vote_early_views(self):
        """views of this run that were voted on the second stream while the rest was handed over (0: one-piece vote)"""
        return int(self._lib.gsx_vote_early_views(self.h))

    def vote_import(self, part_views, part_offsets, blobs, pool_all_ptr, pool_all_bytes):
        pv = np.ascontiguousarray(part_views, np.int32)
        po = np.ascontiguousarray(part_offsets, np.int64)
        bl = np.ascontiguousarray(blobs, np.uint8)
        check(self._lib.gsx_vote_import(self.h, len(pv), pv.ctypes.data, po.ctypes.data, bl.ctypes.data if bl.size else None,
                                        C.c_void_p(pool_all_ptr), int(pool_all_bytes)), self.h)

    def vote_import_uniform(self, part_views, part_offsets, cams, map_size, image_size, pool_all_ptr, pool_all_bytes):
        """gsx_vote_import_uniform: cams = the cameras of ALL views in global order (Camera structs, a ctypes array of them, or
        dicts); map_size / image_size = (width, height) shared by every view of the run."""
        pv = np.ascontiguousarray(part_views, np.int32)
        po = np.ascontiguousarray(part_offsets, np.int64)
        arr = cams if isinstance(cams, C.Array) else camera_array(cams)
        check(self._lib.gsx_vote_import_uniform(self.h, len(pv), pv.ctypes.data, po.ctypes.data, C.addressof(arr) if len(arr) else None,
                                                int(map_size[0]), int(map_size[1]), int(image_size[0]), int(image_size[1]),
                                                C.c_void_p(pool_all_ptr), int(pool_all_bytes)), self.h)

    def vote_import_undo(self):
        check(self._lib.gsx_vote_import_undo(self.h), self.h)

    def vote_slab_labels(self, slab, slabs):
        """-> slab size S; the slab's labels (Morton order) are the first S words at keys_device()."""
        sn = C.c_int64()
        check(self._lib.gsx_vote_slab_labels(self.h, int(slab), int(slabs), C.byref(sn)), self.h)
        return int(sn.value)

    def host_threads(self):
        return int(self._lib.gsx_host_threads(self.h))

    def vote_culled(self, reset=False):
        """(wave, view) pairs skipped by the wave culling so far (gsx_vote_culled)."""
        out = C.c_int64(0)
        check(self._lib.gsx_vote_culled(self.h, C.byref(out), 1 if reset else 0), self.h)
        return int(out.value)

    def debug_filter_check(self):
        """gsx_debug_filter_check: [max relative error of v_rcp_f32 in units of 2^-24, fract x 8, floor-convert x 8]."""
        out = np.zeros(17, np.float64)
        check(self._lib.gsx_debug_filter_check(self.h, out.ctypes.data), self.h)
        return out

    def debug_planes(self, bins):
        cnt = np.empty((bins, self.n), np.uint16)
        fv = np.empty((bins, self.n), np.uint16)
        check(self._lib.gsx_vote_debug_planes(self.h, cnt.ctypes.data, fv.ctypes.data), self.h)
        return cnt, fv

    # -- rasterizer ---------------------------------------------------------------------------------
    def upload_splats(self, xyz, scale, rot, opacity, f_dc, labels=None):
        """3DGS PLY attributes as float arrays (see include/gsx.h).  scale/opacity may be None."""
        f32 = lambda a, w: None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(-1, w))
        xyz, scale, rot, f_dc = f32(xyz, 3), f32(scale, 3), f32(rot, 4), f32(f_dc, 3)
        opacity = None if opacity is None else np.ascontiguousarray(np.asarray(opacity, dtype=np.float32).reshape(-1))
        lab = None if labels is None else np.ascontiguousarray(labels, dtype=np.int32)
        n = len(xyz)
        for a in (scale, rot, opacity, f_dc, lab):
            if a is not None and len(a) != n:
                raise ValueError("all splat attribute arrays must have the same number of rows")
        ptr = lambda a: None if a is None else a.ctypes.data
        check(self._lib.gsx_upload_splats(self.h, n, ptr(xyz), ptr(scale), ptr(rot), ptr(opacity), ptr(f_dc), ptr(lab)),
              self.h)
        self.n_splats = n

    def upload_sh(self, f_rest, degree):
        """Switch the colour path to float spherical harmonics of `degree` (see gsx_upload_sh)."""
        k1 = (degree + 1) ** 2 - 1
        fr = None
        if k1:
            fr = np.ascontiguousarray(np.asarray(f_rest, dtype=np.float32).reshape(self.n_splats, 3 * k1))
        check(self._lib.gsx_upload_sh(self.h, None if fr is None else fr.ctypes.data, int(degree)), self.h)

    def render_view(self, camera, width, height, to_host=True):
        """One frame of the viewer's pipeline -> (height, width, 4) float32 premultiplied RGBA."""
        cam = camera if isinstance(camera, Camera) else Camera.from_dict(camera)
        out = np.empty((height, width, 4), np.float32) if to_host else None
        check(self._lib.gsx_render_view(self.h, C.byref(cam), int(width), int(height), out.ctypes.data if to_host else None),
              self.h)
        return out

    def render_views(self, cameras, width, height, to_host=True):
        """Several frames of one size, up to four in flight on HIP streams of their own (gsx_render_views, option
        render_frames) -> (n, height, width, 4) float32, or None with to_host=False.  Same pixels as render_view, about half
        as many views per second again."""
        cams = (Camera * len(cameras))(*[c if isinstance(c, Camera) else Camera.from_dict(c) for c in cameras])
        out = np.empty((len(cameras), height, width, 4), np.float32) if to_host else None
        ptrs = (C.c_void_p * len(cameras))(*[out[k].ctypes.data for k in range(len(cameras))]) if to_host else None
        check(self._lib.gsx_render_views(self.h, len(cameras), cams, int(width), int(height), ptrs), self.h)
        return out

    def render_num_pairs(self):
        return self._lib.gsx_render_num_pairs(self.h)

    def render_num_pairs_consumed(self):
        return self._lib.gsx_render_num_pairs_consumed(self.h)

    def hit_test(self, camera, width, height, x, y):
        """performHitTesting (gs.js:361-395) -> (label or -999999, importance-order row or -1)."""
        cam = camera if isinstance(camera, Camera) else Camera.from_dict(camera)
        label, index = C.c_int32(), C.c_int64()
        check(self._lib.gsx_hit_test(self.h, C.byref(cam), int(width), int(height), float(x), float(y), C.byref(label),
                                     C.byref(index)), self.h)
        return label.value, index.value

    def render_debug(self, buckets=False):
        """(buffer (n,32) u8, order (n,) u32, texdata (8n,) u32[, buckets (n,) u32]) — test hooks."""
        n = self.n_splats
        buf = np.empty((n, 32), np.uint8)
        order = np.empty(n, np.uint32)
        tex = np.empty(8 * n, np.uint32)
        bk = np.empty(n, np.uint32) if buckets else None
        check(self._lib.gsx_render_debug(self.h, buf.ctypes.data, order.ctypes.data, tex.ctypes.data,
                                         bk.ctypes.data if buckets else None), self.h)
        return (buf, order, tex, bk) if buckets else (buf, order, tex)

    def export_splat(self, path):
        """Write the uploaded splats as a `.splat` file: the viewer's 32-byte rows (position f32x3, exp(scale)
        f32x3, rgba u8x4, quaternion u8x4; gs.js:237, 539-542) in importance order.  The reference viewer takes
        such a file by drag and drop without re-parsing a 250-byte-per-vertex PLY in JavaScript."""
        buf, _, _ = self.render_debug()
        with open(path, "wb") as f:
            f.write(buf.tobytes())
        return len(buf)

    def sort_pairs(self, keys, values, bits=32):
        """Stable GPU radix sort by key bits [0, bits) (test hook, gsx_debug_sort_pairs)."""
        k = np.ascontiguousarray(keys, dtype=np.uint32).copy()
        v = np.ascontiguousarray(values, dtype=np.uint32).copy()
        check(self._lib.gsx_debug_sort_pairs(self.h, k.ctypes.data, v.ctypes.data, len(k), bits), self.h)
        return k, v

    def sort_pairs_drop(self, keys, values, bits=16):
        """The rasterizer's level-1 sort: pairs with key 0xffffffff are left out by the first pass; returns the kept pairs, sorted
        (stable) by key bits [0, bits) (test hook, gsx_debug_sort_pairs_drop)."""
        k = np.ascontiguousarray(keys, dtype=np.uint32).copy()
        v = np.ascontiguousarray(values, dtype=np.uint32).copy()
        kept = C.c_int64()
        check(self._lib.gsx_debug_sort_pairs_drop(self.h, k.ctypes.data, v.ctypes.data, len(k), bits, C.byref(kept)), self.h)
        return k[:kept.value], v[:kept.value]

    def synchronize(self):
        check(self._lib.gsx_synchronize(self.h), self.h)
        self._keep_alive.clear()

    @property
    def stream(self):
        return self._lib.gsx_stream(self.h)

    # -- profiling ---------------------------------------------------------------------------------
    def profile(self, on=True):
        check(self._lib.gsx_profile_enable(self.h, 1 if on else 0), self.h)
        check(self._lib.gsx_profile_reset(self.h), self.h)

    def profile_names(self):
        out, i = [], 0
        while True:
            nm = self._lib.gsx_profile_name(self.h, i)
            if nm is None:
                return out
            out.append(nm.decode())
            i += 1

    def profile_get(self, name):
        n, ms = C.c_int64(), C.c_double()
        check(self._lib.gsx_profile_get(self.h, name.encode(), C.byref(n), C.byref(ms)), self.h)
        return n.value, ms.value


def bind_to_gpu_numa_node(device=0):
    """One process per GPU: keep this process (and every thread it starts from now on) on the CPUs of the NUMA node the
    GPU hangs off (/sys/bus/pci/devices/<bdf>/local_cpulist), so that the segmentation maps it allocates, the pinned
    staging ring and the packer threads all sit next to the GPU's PCIe root.  Returns the CPU set, or None when the
    kernel does not expose the topology or the process may not run there (then nothing is changed)."""
    try:
        import torch
        p = torch.cuda.get_device_properties(device)
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        txt = open(f"/sys/bus/pci/devices/{bdf}/local_cpulist").read().strip()
        cpus = set()
        for part in txt.split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus or cpus == os.sched_getaffinity(0):
            return None
        os.sched_setaffinity(0, cpus)
        return cpus
    except Exception:
        return None


def cull_planes(camera):
    """The five world-space culling planes of one view (test hook, host only): array (5, 5) = unit normal A, offset B,
    margin slope M; a sphere (c, r) is skipped when A.c + B > r + M * (|c|_1 + r) for one of them."""
    cam = camera if isinstance(camera, Camera) else Camera.from_dict(camera)
    out = np.empty((5, 5), np.float64)
    check(_lib.lib().gsx_debug_cull_planes(C.byref(cam), out.ctypes.data))
    return out


def project_gaussian(position, camera, ctx=None):
    """Device-side project_gaussian (reference :43-82): (x, y) ints or None."""
    own = ctx is None
    ctx = ctx or Context()
    try:
        return ctx.project_one(position, camera)
    finally:
        if own:
            ctx.close()


def assign_labels_from_maps(gaussians, cameras, segmaps, image_sizes=None, n_classes=150, ctx=None):
    """Majority vote given per-camera segmentation maps.

    cameras / segmaps / image_sizes are parallel lists over the cameras that the reference would
    actually process (missing images already skipped, deep_learning_segmentation.py:256-259).
    Returns int32 labels (N,), bit-identical to the reference's assign_labels (:297-308)."""
    own = ctx is None
    ctx = ctx or Context()
    try:
        ctx.upload_positions(gaussians)
        ctx.vote_begin(n_classes, 0, max(1, len(cameras)))
        for k, (cam, seg) in enumerate(zip(cameras, segmaps)):
            ctx.vote_view(cam, seg, None if image_sizes is None else image_sizes[k])
        return ctx.vote_finalize()
    finally:
        if own:
            ctx.close()


def host_pack(seg_map, n_classes, tiled=True, coarse=True, threads=1, packed_u8=False):
    """Test hook, no GPU: the packed form gsx_vote_view stages for a host map -> (bytes u8, coarse_off or -1, bad)."""
    seg = np.asarray(seg_map)
    if seg.dtype == np.int64:
        dt = _lib.GSX_SEG_I64
    elif seg.dtype == np.uint8:
        dt = _lib.GSX_SEG_U8 if packed_u8 else _lib.GSX_SEG_U8_LABELS
    else:
        seg, dt = seg.astype(np.int32, copy=False), _lib.GSX_SEG_I32
    seg = np.ascontiguousarray(seg)
    h, w = seg.shape
    nbytes, coff, bad = C.c_int64(), C.c_int64(), C.c_int32()
    L = _lib.lib()
    check(L.gsx_debug_host_pack(seg.ctypes.data, dt, w, h, n_classes, int(tiled), int(coarse), threads, None, 0, C.byref(nbytes),
                                C.byref(coff), C.byref(bad)))
    out = np.zeros(nbytes.value, np.uint8)
    check(L.gsx_debug_host_pack(seg.ctypes.data, dt, w, h, n_classes, int(tiled), int(coarse), threads, out.ctypes.data, out.size,
                                C.byref(nbytes), C.byref(coff), C.byref(bad)))
    return out, coff.value, bool(bad.value)
