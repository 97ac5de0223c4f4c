#!/bin/bash
# GPU box, round 3: the experimental LDS layout of the one-piece kernel (labels checked first), the early vote's two forms, the
# protocol's host-side cost with a one-rank RCCL group, a randomised soak
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r03; mkdir -p $OUT; cd $ROOT
python - <<'PY' || exit 1
import importlib, sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle
pkg = importlib.import_module("3d_gaussian_splatting_project_amd"); scene = pkg.scene
for n, V, C in ((70_001, 9, 150), (30_000, 40, 254), (5_000, 3, 3)):
    pos, cams, segs = scene.make_scene(n, V, 480, 270, n_classes=C, config_id=12, convention="w2c")
    want = oracle.assign_labels(pos, cams, segs, [(480, 270)] * V, threads=0)
    with pkg.Context(0) as c:
        c.set_option("lds_wave_layout", 1); c.set_option("early_vote", 0)
        c.upload_positions(pos); c.vote_begin(C, 0, V)
        for cam, seg in zip(cams, segs): c.vote_view(cam, seg)
        assert np.array_equal(c.vote_finalize(), want), (n, V, C)
print("lds_wave_layout labels ok")
PY
tools/ab.sh wl "" "--opt lds_wave_layout=1" "" "--opt lds_wave_layout=1" || exit 1
tools/ab.sh replay "--opt early_replay=0" "--opt early_replay=1" "--opt early_replay=0" "--opt early_replay=1" || exit 1
timeout -k 10 300 python tools/pipeline_cost.py 2>/dev/null | tee $OUT/pipeline_cost.txt || exit 1
for seed in 311 312 313; do timeout -k 10 400 python tests/soak.py $seed 1500 2>&1 | tail -1 | tee -a $OUT/soak.txt || exit 1; done
