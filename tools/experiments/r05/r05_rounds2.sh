#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
bash tools/variant_build.sh rl lsd_kernels.hip=-DPLF_ROUND_LOG line_host.hip=-DPLF_ROUND_LOG > /tmp/vb.log 2>&1 || tail -20 /tmp/vb.log
export PLF_LIB_PATH=tools/scratch/libplf_rl.so PLF_LSD_ROUND_LOG=1 ROUND_LOG_BANDS=1
( python tools/round_log.py polygons 1 1; python tools/round_log.py natural 1 1 ) 2>&1 | grep -v amdgpu.ids
