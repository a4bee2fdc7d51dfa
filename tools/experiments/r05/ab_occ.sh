#!/bin/bash
# region kernel at 5 / 6 / 7 / 8 waves per SIMD (96 / 80 / 72 / 64 VGPRs) with the matching number of frames in flight.  Run ON the GPU box.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for w in 5 6 7; do bash tools/variant_build.sh w$w lsd_kernels.hip="-DPLF_REGIONS_WPE=$w" > /dev/null 2>&1; done
run() { # name lib batch
  if [ "$2" = base ]; then unset PLF_LIB_PATH; else export PLF_LIB_PATH=$2; fi
  python bench.py --batch $3 --no-extras --cpu-seconds 0 --steps 6 --warmup 2 2>/dev/null | V="$1" python -c "
import json,sys,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s %9.1f fps %8.3f ms/step  regions %7.3f ms' % (os.environ['V'], d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
}
run "8 waves (64 VGPR) x 8192" base 8192
run "7 waves (72 VGPR) x 7168" tools/scratch/libplf_w7.so 7168
run "6 waves (80 VGPR) x 6144" tools/scratch/libplf_w6.so 6144
run "5 waves (96 VGPR) x 5120" tools/scratch/libplf_w5.so 5120
run "5 waves (96 VGPR) x 8192" tools/scratch/libplf_w5.so 8192
run "6 waves (80 VGPR) x 8192" tools/scratch/libplf_w6.so 8192
run "8 waves (64 VGPR) x 8192" base 8192
