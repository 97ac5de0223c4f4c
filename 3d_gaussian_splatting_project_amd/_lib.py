"""ctypes binding of libgsx.so (include/gsx.h).  There is no fallback: if the library or a gfx950
GPU is missing, loading / creating a context raises."""
import ctypes as C
import os
import subprocess

_DIR = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_DIR, "libgsx.so")
CSRC = os.path.join(_DIR, "csrc")

GSX_OK = 0
GSX_E_INVALID, GSX_E_HIP, GSX_E_STATE, GSX_E_RANGE, GSX_E_UNSUPPORTED, GSX_E_IO = -1, -2, -3, -4, -5, -6
GSX_SEG_I32, GSX_SEG_I64, GSX_SEG_U8, GSX_SEG_U8_LABELS = 0, 1, 2, 3


class GsxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libgsx error {code}: {msg}")
        self.code = code


class Camera(C.Structure):
    """gsx_camera: one cameras.json entry, fp64."""
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("width", C.c_int32), ("height", C.c_int32),
                ("R", C.c_double * 9), ("p", C.c_double * 3)]

    @classmethod
    def from_dict(cls, cam):
        c = cls()
        c.fx, c.fy = float(cam["fx"]), float(cam["fy"])
        c.width, c.height = int(cam["width"]), int(cam["height"])
        rot = cam["rotation"]
        flat = [float(v) for row in rot for v in row] if hasattr(rot[0], "__len__") else [float(v) for v in rot]
        if len(flat) != 9 or len(cam["position"]) != 3:
            raise ValueError("camera needs a 3x3 rotation and a 3-vector position")
        for i, v in enumerate(flat):
            c.R[i] = v
        for i, v in enumerate(cam["position"]):
            c.p[i] = float(v)
        return c


_SIGS = {
    "gsx_abi_version": (C.c_int, []),
    "gsx_device_count": (C.c_int, []),
    "gsx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "gsx_destroy": (None, [C.c_void_p]),
    "gsx_last_error": (C.c_char_p, [C.c_void_p]),
    "gsx_stream": (C.c_void_p, [C.c_void_p]),
    "gsx_synchronize": (C.c_int, [C.c_void_p]),
    "gsx_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "gsx_upload_positions": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gsx_upload_positions_strided": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                               C.c_int64]),
    "gsx_num_gaussians": (C.c_int64, [C.c_void_p]),
    "gsx_project_one": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(Camera), C.POINTER(C.c_int32),
                                  C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "gsx_project_all": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.c_void_p, C.c_void_p]),
    "gsx_vote_begin": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "gsx_vote_view": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                C.c_int32]),
    "gsx_vote_view_device": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_int32, C.c_int32]),
    "gsx_vote_views_device": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_int32]),
    "gsx_debug_workers_stress": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "gsx_debug_host_pack_compact": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64,
                                              C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "gsx_debug_widen_labels": (C.c_int, [C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "gsx_debug_host_pack": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "gsx_vote_num_views": (C.c_int32, [C.c_void_p]),
    "gsx_vote_rewind": (C.c_int, [C.c_void_p]),
    "gsx_vote_finalize": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gsx_vote_labels_device": (C.c_void_p, [C.c_void_p]),
    "gsx_vote_flush": (C.c_int, [C.c_void_p]),
    "gsx_vote_counts_device": (C.c_void_p, [C.c_void_p, C.POINTER(C.c_int64)]),
    "gsx_vote_tiebreak_keys": (C.c_int, [C.c_void_p]),
    "gsx_vote_keys_device": (C.c_void_p, [C.c_void_p, C.POINTER(C.c_int64)]),
    "gsx_vote_labels_from_keys": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gsx_vote_first_device": (C.c_void_p, [C.c_void_p, C.POINTER(C.c_int64)]),
    "gsx_vote_slab_size": (C.c_int64, [C.c_void_p]),
    "gsx_vote_slab_reduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "gsx_vote_flush_counts": (C.c_int, [C.c_void_p]),
    "gsx_vote_slab_totals": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gsx_vote_cand_device": (C.c_void_p, [C.c_void_p, C.POINTER(C.c_int64)]),
    "gsx_vote_tie_codes": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gsx_vote_codes_device": (C.c_void_p, [C.c_void_p, C.POINTER(C.c_int64)]),
    "gsx_vote_tie_resolve": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gsx_vote_labels_from_sorted": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "gsx_vote_export": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "gsx_vote_pool_bytes": (C.c_int64, [C.c_void_p]),
    "gsx_vote_link_bytes": (C.c_int64, [C.c_void_p]),
    "gsx_vote_early_views": (C.c_int64, [C.c_void_p]),
    "gsx_vote_import": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "gsx_vote_import_undo": (C.c_int, [C.c_void_p]),
    "gsx_vote_import_uniform": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                          C.c_void_p, C.c_int64]),
    "gsx_vote_slab_labels": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int64)]),
    "gsx_host_threads": (C.c_int, [C.c_void_p]),
    "gsx_default_host_threads": (C.c_int, []),
    "gsx_profile_name": (C.c_char_p, [C.c_void_p, C.c_int32]),
    "gsx_vote_debug_planes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "gsx_upload_splats": (C.c_int, [C.c_void_p, C.c_int64] + [C.c_void_p] * 6),
    "gsx_upload_sh": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "gsx_num_splats": (C.c_int64, [C.c_void_p]),
    "gsx_render_view": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.c_int32, C.c_int32, C.c_void_p]),
    "gsx_render_views": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "gsx_render_image_device": (C.c_void_p, [C.c_void_p]),
    "gsx_render_num_pairs": (C.c_int64, [C.c_void_p]),
    "gsx_render_num_pairs_consumed": (C.c_int64, [C.c_void_p]),
    "gsx_hit_test": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.c_int32, C.c_int32, C.c_double, C.c_double, C.POINTER(C.c_int32),
                               C.POINTER(C.c_int64)]),
    "gsx_render_debug": (C.c_int, [C.c_void_p] + [C.c_void_p] * 4),
    "gsx_ply_open": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "gsx_ply_close": (None, [C.c_void_p]),
    "gsx_ply_num_vertices": (C.c_int64, [C.c_void_p]),
    "gsx_ply_num_properties": (C.c_int32, [C.c_void_p]),
    "gsx_ply_property": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "gsx_ply_row_stride": (C.c_int64, [C.c_void_p]),
    "gsx_ply_rows": (C.c_void_p, [C.c_void_p]),
    "gsx_ply_read_f32": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p]),
    "gsx_ply_set_f32": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p]),
    "gsx_ply_write": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int32]),
    "gsx_kmeans": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_double,
                             C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "gsx_vote_culled": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_int32]),
    "gsx_debug_filter_check": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gsx_debug_cull_planes": (C.c_int, [C.POINTER(Camera), C.c_void_p]),
    "gsx_debug_sort_pairs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]),
    "gsx_debug_sort_pairs_drop": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_int64)]),
    "gsx_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "gsx_profile_reset": (C.c_int, [C.c_void_p]),
    "gsx_profile_get": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
}

_lib = None


def build(force=False):
    """Compile libgsx.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC] + (["-B"] if force else [])
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return SO_PATH


def declared_symbols():
    return sorted(_SIGS)


def lib():
    """Load libgsx.so.  Raises if it has not been built: the HIP library IS the product."""
    global _lib
    if _lib is None:
        path = os.environ.get("GSX_LIBRARY") or SO_PATH   # GSX_LIBRARY: another build of the same C ABI (tools/ablate.sh)
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing: build it with `make -C {CSRC}` "
                              "(or __graft_entry__.build()); there is no CPU fallback")
        l = C.CDLL(path)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc, ctx=None):
    if rc != GSX_OK:
        msg = lib().gsx_last_error(ctx)
        msg = msg.decode("utf-8", "replace") if msg else ""
        if rc in (GSX_E_INVALID, GSX_E_RANGE):
            raise ValueError(f"libgsx error {rc}: {msg}")
        raise GsxError(rc, msg)
