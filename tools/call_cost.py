import importlib, sys, time, os
import numpy as np
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("3d_gaussian_splatting_project_amd")
scene = pkg.scene
ctx = pkg.Context(0)
ctx.upload_positions(np.zeros((1000,3),np.float32))
cams = [pkg.Camera.from_dict(c) for c in scene.make_cameras(200, 64, 48, convention="w2c")]
seg = np.zeros((48,64), np.int32)
for rep in range(3):
    ctx.vote_begin(150, 0, 200)
    t0=time.perf_counter()
    for v in range(200): ctx.vote_view(cams[v], seg)
    t1=time.perf_counter()
    ctx.vote_finalize()
    print(f"tiny maps: {(t1-t0)/200*1e6:.2f} us per vote_view call (python + C fixed cost)")
import ctypes as C
lib = pkg.lib()
for rep in range(2):
    ctx.vote_begin(150, 0, 200)
    t0=time.perf_counter()
    p = seg.ctypes.data
    for v in range(200): lib.gsx_vote_view(ctx.h, C.byref(cams[v]), p, 0, 64, 48, 64, 48)
    t1=time.perf_counter()
    ctx.vote_finalize()
    print(f"tiny maps, raw ctypes: {(t1-t0)/200*1e6:.2f} us per call")
for ht in (1, 4, 16):
    ctx.set_option("host_threads", ht)
    ctx.vote_begin(150, 0, 200); [ctx.vote_view(cams[v], seg) for v in range(200)]; ctx.vote_finalize()
    ctx.vote_begin(150, 0, 200)
    t0=time.perf_counter()
    for v in range(200): ctx.vote_view(cams[v], seg)
    t1=time.perf_counter()
    ctx.vote_finalize()
    print(f"host_threads={ht}: {(t1-t0)/200*1e6:.2f} us per call")
