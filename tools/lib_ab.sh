#!/bin/bash
# GPU box: views/s of the rasterizer (defaults, four frames in flight) per library given as argument (another build of the same
# C ABI, selected through GSX_LIBRARY), interleaved twice:  tools/lib_ab.sh libgsx_base.so libgsx.so
for rep in 1 2; do for lib in "$@"; do echo "== $lib"; GSX_LIBRARY=$PWD/3d_gaussian_splatting_project_amd/$lib timeout -k 10 120 python tools/render_phase_sweep.py one 2>&1 | grep views; done; done
