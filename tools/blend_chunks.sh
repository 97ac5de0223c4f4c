#!/bin/bash
# GPU box: staging chunk / opacity-vote distance of the blend kernels: variant libraries (built by hand: hipcc -DGSX_BLEND4_CHUNK=..
# -DGSX_BLEND4_GROUP=.. -c blend.hip, linked with the product's other objects) against the product library, interleaved
D=3d_gaussian_splatting_project_amd
for rep in 1 2; do
for lib in $D/libgsx.so $D/libgsx_c*.so; do
  echo "== $(basename $lib) (rep $rep)"
  GSX_LIBRARY=$PWD/$lib python tools/render_phase_sweep.py blend1 2>&1 | grep -v amdgpu.ids | grep "blend_pk2': 2"
done; done
