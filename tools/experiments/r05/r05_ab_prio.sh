#!/bin/bash
# issue priorities of the line stream's throughput kernels vs the ORB tiles (large batch).  Run ON the GPU box.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
bash tools/variant_build.sh lp3 lsd_kernels.hip="-DPLF_LINE_PRIO=3" > /tmp/vb1.log 2>&1 || tail -5 /tmp/vb1.log
bash tools/variant_build.sh lp3o1 lsd_kernels.hip="-DPLF_LINE_PRIO=3" orb_front.hip="-DPLF_ORB_PRIO=1" > /tmp/vb2.log 2>&1 || tail -5 /tmp/vb2.log
bash tools/variant_build.sh lp2 lsd_kernels.hip="-DPLF_LINE_PRIO=2" > /tmp/vb3.log 2>&1 || tail -5 /tmp/vb3.log
bash tools/variant_build.sh lp3o0 lsd_kernels.hip="-DPLF_LINE_PRIO=3" orb_front.hip="-DPLF_ORB_PRIO=0" > /tmp/vb4.log 2>&1 || tail -5 /tmp/vb4.log
AB_BATCH=8192 bash tools/ab.sh lp3 lp3o1 lp2 lp3o0
