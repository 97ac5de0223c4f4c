#!/bin/bash
# Build timing-only variants of libgsx.so with parts of the vote kernel removed (GSX_ABLATE bit mask:
# 1 = no LDS histogram, 2 = no seg-map gather, 4 = no per-view scalar loads).  RESULTS ARE INVALID by design;
# the variants only answer "where does the time go".  Run here (build container); bench them on the GPU box with
#   GSX_LIBRARY=tools/ablate/libgsx_<mask>.so python bench.py --cpu-sample 0 --render-views 0 --no-verify
set -e
cd "$(dirname "$0")/../3d_gaussian_splatting_project_amd/csrc"
make experiments >/dev/null
mkdir -p ../../tools/ablate
for m in ${ABLATE_MASKS:-2 32}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -Wno-unused-result --offload-arch=gfx950 -ffp-contract=off -DGSX_EXPERIMENTS -DGSX_ABLATE=$m -c vote.hip -o /tmp/vote_ablate_$m.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ablate/libgsx_$m.so exp/gsx_api.o /tmp/vote_ablate_$m.o exp/sort.o exp/render.o exp/blend.o exp/kmeans.o exp/ply_io.o exp/host_pack.o -lpthread
done
ls -la ../../tools/ablate
