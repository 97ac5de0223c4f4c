"""CPU: the native PLY reader/writer under AddressSanitizer + UBSan (sanitizers run on the CPU build only;
the GPU pool has no ASan).  Builds a small host-only test program from csrc/ply_io.cpp."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_pkg

CSRC = os.path.join(ROOT, "3d_gaussian_splatting_project_amd", "csrc")

DRIVER = r'''
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/gsx.h"
int main(int argc, char** argv) {
    gsx_ply* p = nullptr;
    if (gsx_ply_open("/nonexistent/file.ply", &p) != GSX_E_IO) return 10;
    if (gsx_ply_open(argv[1], &p) != GSX_OK) { std::fprintf(stderr, "%s\n", gsx_last_error(nullptr)); return 11; }
    const long long n = gsx_ply_num_vertices(p);
    std::vector<float> x((size_t)n);
    if (gsx_ply_read_f32(p, "x", x.data()) != GSX_OK) return 12;
    if (gsx_ply_read_f32(p, "no_such", x.data()) != GSX_E_INVALID) return 13;
    for (auto& v : x) v += 1.0f;
    if (gsx_ply_set_f32(p, "x", x.data()) != GSX_OK) return 14;
    std::vector<int32_t> lab((size_t)n);
    for (long long i = 0; i < n; ++i) lab[(size_t)i] = (int32_t)(i % 151) - 1;
    if (gsx_ply_write(p, argv[2], lab.data(), 0) != GSX_OK) return 15;
    if (gsx_ply_write(p, argv[3], lab.data(), 1) != GSX_OK) return 16;
    gsx_ply_close(p);
    if (gsx_ply_open(argv[3], &p) != GSX_OK) return 17;     // ascii round trip
    if (gsx_ply_num_vertices(p) != n) return 18;
    gsx_ply_close(p);
    if (gsx_ply_open(argv[4], &p) == GSX_OK) return 19;     // truncated file must be refused
    // hostile headers: counts whose n * stride wraps around size_t, negative and absurd counts (binary and ascii)
    for (int k = 5; k < argc - 1; ++k)
        if (gsx_ply_open(argv[k], &p) == GSX_OK) return 20 + k;
    // in-place write: the destination IS the mapped source (plyfile users do this; it must neither fault nor eat the input)
    const char* inplace = argv[argc - 1];
    if (gsx_ply_open(inplace, &p) != GSX_OK) return 40;
    const long long m = gsx_ply_num_vertices(p);
    std::vector<int32_t> lab2((size_t)m, 7);
    if (gsx_ply_write(p, inplace, lab2.data(), 0) != GSX_OK) { std::fprintf(stderr, "%s\n", gsx_last_error(nullptr)); return 41; }
    std::vector<float> y((size_t)m);
    if (gsx_ply_read_f32(p, "y", y.data()) != GSX_OK) return 42;   // the old mapping is still intact after the write
    gsx_ply_close(p);
    if (gsx_ply_open(inplace, &p) != GSX_OK) return 43;
    std::vector<float> lab3((size_t)m), y2((size_t)m);
    if (gsx_ply_num_vertices(p) != m || gsx_ply_read_f32(p, "label", lab3.data()) != GSX_OK || gsx_ply_read_f32(p, "y", y2.data()) != GSX_OK) return 44;
    for (long long i = 0; i < m; ++i) if (lab3[(size_t)i] != 7.0f || y2[(size_t)i] != y[(size_t)i]) return 45;
    gsx_ply_close(p);
    std::printf("ok %lld\n", n);
    return 0;
}
'''

STUB = r'''
#include <cstdarg>
#include <cstdio>
struct gsx_ctx;
namespace gsx {
struct Ctx;
static char g_buf[512] = "";
int fail(Ctx*, int code, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_buf, sizeof g_buf, fmt, ap); va_end(ap); return code; }
}
extern "C" const char* gsx_last_error(const gsx_ctx*) { return gsx::g_buf; }
'''


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_ply_io_under_asan(tmp_path):
    pio = __import__("importlib").import_module("3d_gaussian_splatting_project_amd.ply_io")
    n = 20_000
    rng = np.random.default_rng(0)
    cols = {k: rng.normal(size=n).astype(np.float32) for k in ("x", "y", "z", "opacity")}
    cols["red"] = rng.integers(0, 255, n).astype(np.uint8)
    src = str(tmp_path / "in.ply")
    pio.write_vertex_ply(src, cols)
    raw = open(src, "rb").read()
    open(tmp_path / "trunc.ply", "wb").write(raw[: len(raw) // 2])
    (tmp_path / "driver.cpp").write_text(DRIVER.replace('"../../include/gsx.h"', f'"{ROOT}/include/gsx.h"'))
    (tmp_path / "stub.cpp").write_text(STUB)
    exe = str(tmp_path / "ply_asan")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
           str(tmp_path / "driver.cpp"), str(tmp_path / "stub.cpp"), os.path.join(CSRC, "ply_io.cpp"), "-o", exe, "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("sanitizer build not available here: " + r.stderr[-300:])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0")
    hostile = []
    body = raw.split(b"end_header\n", 1)[1]
    for k, count in enumerate(("1085102592571150096", "18446744073709551615", "-5", "9223372036854775807", "99999999999999999999", "abc")):
        for fmt in (b"binary_little_endian", b"ascii"):
            hp = tmp_path / f"hostile_{k}_{fmt.decode()[:3]}.ply"
            hdr = b"ply\nformat " + fmt + b" 1.0\nelement vertex " + count.encode() + b"\nproperty float x\nproperty float y\nproperty float z\nproperty float opacity\nproperty uchar red\nend_header\n"
            hp.write_bytes(hdr + (body[:4096] if fmt.startswith(b"binary") else b"1 2 3 4 5\n" * 50))
            hostile.append(str(hp))
    inplace = str(tmp_path / "inplace.ply")
    shutil.copy(src, inplace)
    run = subprocess.run([exe, src, str(tmp_path / "out.ply"), str(tmp_path / "out_ascii.ply"), str(tmp_path / "trunc.ply")] + hostile + [inplace],
                         capture_output=True, text=True, env=env)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "ok 20000" in run.stdout and "ERROR: AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr
    out = pio.PlyData.read(str(tmp_path / "out.ply"))["vertex"]
    assert np.array_equal(out["x"], cols["x"] + 1) and np.array_equal(out["label"], np.arange(n) % 151 - 1)
    assert not [f for f in os.listdir(tmp_path) if "gsx-tmp" in f]       # no temporary file is left behind


def test_views_outlive_the_plydata_object(tmp_path):
    """`x = PlyData.read(p)['vertex']['x']`: the reference's plyfile returns arrays that own their memory; here they view
    a native mapping, which must stay alive as long as any view does (use-after-unmap)."""
    import gc
    pio = __import__("importlib").import_module("3d_gaussian_splatting_project_amd.ply_io")
    n = 300_000
    x = np.arange(n, dtype=np.float32)
    src = str(tmp_path / "v.ply")
    pio.write_vertex_ply(src, {"x": x, "y": x * 2})
    col = pio.PlyData.read(src)["vertex"]["x"]          # the PlyData temporary is unreachable from here on
    elem = pio.PlyData.read(src)["vertex"]
    gc.collect()
    junk = [np.ones(1 << 20) for _ in range(8)]          # churn the allocator / address space
    assert np.array_equal(col, x) and np.array_equal(elem["y"], x * 2) and len(junk) == 8
    # same destination as the source through the Python front-end (save_labeled_ply on --ply_file == --output_file)
    ply = pio.PlyData.read(src)
    ply.write(src, labels=np.full(n, 3, np.int32))
    again = pio.PlyData.read(src)["vertex"]
    assert np.array_equal(again["x"], x) and (again["label"] == 3).all() and np.array_equal(ply["vertex"]["x"], x)
