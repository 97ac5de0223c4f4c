"""Host side of gsx_vote_view (no GPU): the packed form of a segmentation map that the worker threads write into the
pinned staging ring — u8 bins (label + 1) in strips of 16 pixel columns plus the 4x4-coarsened level — against a
numpy model of the layout the vote kernels read (csrc/vote.hip gather_chunk), and the label range check
(the reference indexes its vote dict with whatever the map holds, dls.py:288-295; the library supports
-1 .. n_classes-1 and must refuse anything else loudly)."""
import importlib

import numpy as np
import pytest

labeler = importlib.import_module("3d_gaussian_splatting_project_amd.labeler")


def model(seg_bins, tiled, coarse):
    """seg_bins: (h, w) array of bins (label + 1).  Returns (fine index map, coarse expectations)."""
    h, w = seg_bins.shape
    ys, xs = np.mgrid[0:h, 0:w]
    if tiled:
        strip = (h + 7) // 8 * 128
        fine_off = (xs >> 4) * strip + ys * 16 + (xs & 15)
        fine_bytes = strip * ((w + 15) // 16)
    else:
        fine_off = ys * w + xs
        fine_bytes = w * h + 4
    cw, ch = (w + 3) // 4, (h + 3) // 4
    cexp = np.full((ch, cw), 255, np.uint8)
    for cy in range(ch):
        for cx in range(cw):
            if 4 * cx + 4 <= w and 4 * cy + 4 <= h:
                cell = seg_bins[4 * cy:4 * cy + 4, 4 * cx:4 * cx + 4]
                if (cell == cell[0, 0]).all():
                    cexp[cy, cx] = cell[0, 0]
    cys, cxs = np.mgrid[0:ch, 0:cw]
    cstrip = (ch + 7) // 8 * 128
    coarse_idx = (cxs >> 4) * cstrip + (cxs & 15) + cys * 16
    coarse_off = (fine_bytes + 255) // 256 * 256
    total = coarse_off + cstrip * ((cw + 15) // 16) if (tiled and coarse) else fine_bytes
    return fine_off, cexp, coarse_idx, coarse_off, total


@pytest.mark.parametrize("w,h", [(64, 32), (61, 35), (16, 8), (3, 3), (1, 1), (130, 17), (257, 64), (20, 1081 % 97)])
@pytest.mark.parametrize("dtype", ["i32", "i64", "u8", "u8p", "i16"])
def test_packed_layout_matches_model(w, h, dtype):
    rng = np.random.default_rng(w * 1000 + h)
    n_classes = 150
    # piecewise-constant with noise: both uniform and mixed 4x4 cells occur
    base = rng.integers(-1, n_classes, size=((h + 7) // 8, (w + 7) // 8))
    lab = np.kron(base, np.ones((8, 8), np.int64))[:h, :w]
    noise = rng.random((h, w)) < 0.02
    lab[noise] = rng.integers(-1, n_classes, size=int(noise.sum()))
    if dtype in ("u8",):
        lab = np.maximum(lab, 0)       # raw uint8 labels cannot express -1
    bins = lab + 1
    arr = {"i32": lab.astype(np.int32), "i64": lab.astype(np.int64), "u8": lab.astype(np.uint8), "u8p": bins.astype(np.uint8),
           "i16": lab.astype(np.int16)}[dtype]
    for tiled in (True, False):
        for coarse in (True, False):
            for threads in ((1, 3) if (tiled and coarse) else (1,)):
                out, coff, bad = labeler.host_pack(arr, n_classes, tiled, coarse, threads, packed_u8=(dtype == "u8p"))
                fine_off, cexp, cidx, coarse_off, total = model(bins, tiled, coarse)
                assert not bad and out.size == total
                assert np.array_equal(out[fine_off], bins.astype(np.uint8))
                if tiled and coarse:
                    assert coff == coarse_off
                    assert np.array_equal(out[coarse_off + cidx], cexp)
                else:
                    assert coff == -1


def test_range_check():
    ok = np.full((9, 21), 149, np.int32)
    assert not labeler.host_pack(ok, 150)[2]
    for bad_value, dtype in ((150, np.int32), (-2, np.int32), (2 ** 31 - 1, np.int32), (-2 ** 31, np.int32), (150, np.int64),
                             (-2, np.int64), (2 ** 40, np.int64), (2 ** 32 - 1, np.int64), (-2 ** 63, np.int64), (150, np.uint8), (255, np.uint8)):
        for pos in ((0, 0), (8, 20), (4, 16), (3, 15)):
            m = ok.astype(dtype)
            m[pos] = bad_value
            assert labeler.host_pack(m, 150, threads=2)[2], (bad_value, dtype, pos)
    packed = np.full((9, 21), 150, np.uint8)          # label+1 form: 150 = label 149 is fine, 151 is not
    assert not labeler.host_pack(packed, 150, packed_u8=True)[2]
    packed[5, 5] = 151
    assert labeler.host_pack(packed, 150, packed_u8=True)[2]
    packed[5, 5] = 255
    assert labeler.host_pack(packed, 150, packed_u8=True)[2]
    # 8-bit sources with 255 classes: every byte is a valid label (bin = label + 1 <= 255 < 256 bins) except 255 itself
    wide = np.arange(9 * 32, dtype=np.int64).reshape(9, 32) % 255
    assert not labeler.host_pack(wide.astype(np.uint8), 255)[2]
    out8 = labeler.host_pack(wide.astype(np.uint8), 255)[0]
    assert np.array_equal(out8, labeler.host_pack(wide.astype(np.int32), 255)[0])
    wide[7, 3] = 255
    assert labeler.host_pack(wide.astype(np.uint8), 255)[2]
    # 255 classes: bin 255 is a real class, so no coarse level is built (255 could not mean "mixed")
    out, coff, bad = labeler.host_pack(np.full((8, 16), 254, np.int32), 255)
    assert coff == -1 and not bad and (out[:128] == 255).all()


def test_full_hd_map_many_threads():
    rng = np.random.default_rng(5)
    sites = rng.integers(0, 1080, size=(60, 2)) * [1, 1920 / 1080]
    ys, xs = np.mgrid[0:1080, 0:1920]
    lab = np.argmin((ys[..., None] - sites[:, 0]) ** 2 + (xs[..., None] - sites[:, 1]) ** 2, axis=-1).astype(np.int32) - 1
    a, coff, bad = labeler.host_pack(lab, 150, threads=4)
    b, _, _ = labeler.host_pack(lab, 150, threads=1)
    assert not bad and np.array_equal(a, b)
    fine_off, cexp, cidx, coarse_off, total = model(lab + 1, True, True)
    assert np.array_equal(a[fine_off], (lab + 1).astype(np.uint8)) and np.array_equal(a[coarse_off + cidx], cexp)


@pytest.mark.parametrize("threads", [1, 2, 5])
def test_worker_pool_runs_every_part_exactly_once(threads):
    """The persistent fork-join pool behind gsx_vote_view (own-share-then-help scheduling, per-thread claim counters):
    thousands of back-to-back runs of varying size on one pool, every part executed exactly once in every run."""
    lib = labeler._lib.lib()
    assert lib.gsx_debug_workers_stress(threads, 3000, 150) == 0
    assert lib.gsx_debug_workers_stress(threads, 500, 1) == 0


def test_labels_cross_pcie_as_bytes_widened_on_the_host():
    """The D2H epilogue (vote.hip labels_to_host): bins u8 -> labels int32, every value, ragged sizes, 1 and 5 threads."""
    lib = labeler._lib.lib()
    rng = np.random.default_rng(7)
    for n in (0, 1, 7, 8, 255, 65536, 65537, 200_003):
        bins = rng.integers(0, 256, size=n, dtype=np.uint8)
        if n >= 256:
            bins[:256] = np.arange(256, dtype=np.uint8)
        for threads in (1, 5):
            out = np.full(n + 1, 77, np.int32)
            assert lib.gsx_debug_widen_labels(threads, bins.ctypes.data, n, out.ctypes.data) == 0
            assert np.array_equal(out[:n], bins.astype(np.int32) - 1) and out[n] == 77
    assert lib.gsx_debug_widen_labels(1, None, 4, None) == labeler._lib.GSX_E_INVALID


@pytest.mark.parametrize("w,h", [(1, 1), (3, 5), (4, 4), (16, 8), (17, 9), (61, 35), (64, 64), (65, 33), (130, 71), (330, 40), (64, 81), (48, 1080)])
def test_compact_transfer_form_expands_to_the_pool_form(w, h):
    """What gsx_vote_view sends over PCIe (coarse level + the mixed cells' blocks) rebuilds, cell by cell, exactly the pool form:
    the numpy restatement of the GPU's expansion applied to the host packer's record == the numpy restatement of the layout.
    Thread counts that divide the bands evenly and ones that do not."""
    import ctypes as C
    import oracle
    lib = labeler._lib.lib()
    rng = np.random.default_rng(w * 1000 + h)

    def record(seg, threads):
        nb, tb, so, bad = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int32()
        assert lib.gsx_debug_host_pack_compact(seg.ctypes.data, 0, w, h, 150, threads, None, 0, C.byref(nb), C.byref(tb), C.byref(so),
                                               C.byref(bad)) == 0
        rec = np.full(nb.value, 0xAB, np.uint8)
        assert lib.gsx_debug_host_pack_compact(seg.ctypes.data, 0, w, h, 150, threads, rec.ctypes.data, rec.size, C.byref(nb), C.byref(tb),
                                               C.byref(so), C.byref(bad)) == 0
        assert nb.value <= rec.size
        return rec[:nb.value], tb.value, so.value, bad.value

    for trial in range(3):
        blocks = rng.integers(-1, 150, size=((h + 7) // 8, (w + 7) // 8), dtype=np.int32)
        seg = np.repeat(np.repeat(blocks, 8, 0), 8, 1)[:h, :w].copy()
        noise = rng.random((h, w)) < (0.0, 0.03, 0.5)[trial]
        seg[noise] = rng.integers(-1, 150, size=int(noise.sum()))
        ref, _ = oracle.pack_map_numpy(seg, 150)
        for threads in (1, 2, 3, 4, 7):
            rec, tb, so, bad = record(seg, threads)
            assert bad == 0
            assert np.array_equal(oracle.expand_compact_numpy(rec, w, h, tb, so), ref), (w, h, trial, threads)
            if trial == 0 and w % 8 == 0 and h % 8 == 0:
                assert rec.size == so                            # no mixed cell: the coarse level is all that travels
    assert record(np.full((h, w), 150, np.int32), 2)[3] == 1     # a label out of range is reported
