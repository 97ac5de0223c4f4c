# GPU box: views/s of the rasterizer per variant library of the radix sort (make O=var8 OUT=../libgsx_sort8.so EXTRA=-DGSX_SORT_ITEMS=8 ../libgsx_sort8.so; likewise 4)
for rep in 1 2; do for v in "" _sort8 _sort4; do echo "== libgsx$v.so"; GSX_LIBRARY=$PWD/3d_gaussian_splatting_project_amd/libgsx$v.so timeout -k 10 120 python tools/render_phase_sweep.py one 2>&1 | grep views; done; done
