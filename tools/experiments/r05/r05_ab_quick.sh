#!/bin/bash
# quick A/B of the region kernel: solo (--serial) step of 8192 frames, polygons and natural, in-tree library against the named scratch variants.  Run ON the GPU box.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
[ -n "$AB_TESTS" ] && timeout 900 python -m pytest $AB_TESTS -m gpu -x -q 2>&1 | tail -2
run() { python bench.py "$@" --no-extras --cpu-seconds 0 --steps ${AB_STEPS:-5} --warmup 2 2>/dev/null | V="$PLF_LIB_PATH $*" python -c "
import json,sys,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-70s %9.1f fps %8.3f ms/step  regions %7.3f ms' % (os.environ['V'], d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"; }
for v in base "$@" base "$@"; do
  if [ $v = base ]; then unset PLF_LIB_PATH; else export PLF_LIB_PATH=tools/scratch/libplf_$v.so; fi
  run --family polygons --batch 8192 --serial
  run --family natural --batch 8192 --serial
  [ -n "$AB_OVERLAP" ] && run --family polygons --batch 8192
  [ -n "$AB_OVERLAP" ] && run --family natural --batch 8192
done
