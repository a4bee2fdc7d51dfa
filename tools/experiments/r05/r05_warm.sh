#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
bash tools/variant_build.sh wnr lsd_kernels.hip=-DPLF_WARM_NOREFINE > /tmp/vb.log 2>&1 || tail -20 /tmp/vb.log
for fam in polygons natural; do for B in 1 8; do
  python tools/sweep_few2.py $fam $B spec_clip=16 spec_clip=32 spec_halo=2 spec_halo=3 spec_halo=6 spec_bands=40 spec_bands=56 spec_bands=64
  PLF_LIB_PATH=tools/scratch/libplf_wnr.so python tools/sweep_few2.py $fam $B spec_clip=16 spec_halo=6
done; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_warm_sweep.txt
