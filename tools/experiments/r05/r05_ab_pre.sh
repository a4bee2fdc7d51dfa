#!/bin/bash
# solo duration of k_lsd_pre (rocprofv3 kernel stats of the serial bench step, 8192 frames) for the in-tree library and the named scratch variants.  Run ON the GPU box.
cd /tmp; export TMPDIR=/tmp
ls $GRAFT_REPO_ROOT/tools/scratch/*.so | head -20
for v in base "$@" base; do
  if [ $v = base ]; then unset PLF_LIB_PATH; else export PLF_LIB_PATH=$GRAFT_REPO_ROOT/tools/scratch/libplf_$v.so; fi
  for fam in polygons natural; do
    rm -rf /tmp/abp; rocprofv3 --kernel-trace --stats -d /tmp/abp --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-extras --cpu-seconds 0 --steps 4 --warmup 1 --serial --family $fam > /dev/null 2>&1
    python3 - $v $fam <<'PY'
import csv, glob, sys
fn = glob.glob('/tmp/abp/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(fn)):
    if r['Name'].startswith(('k_lsd_pre', 'k_lsd_regions2')): print("%-10s %-9s %-16s avg %8.3f ms" % (sys.argv[1], sys.argv[2], r['Name'][:16], float(r['AverageNs']) / 1e6))
PY
  done
done
