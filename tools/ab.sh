#!/bin/bash
# tools/ab.sh <variant names ...>: bench (4096 in flight, 8 steps, no extras) of the in-tree library ("base") and of tools/scratch/libplf_<name>.so, in turn.  Run ON the GPU box.
for v in base "$@" base; do
  if [ $v = base ]; then unset PLF_LIB_PATH; else export PLF_LIB_PATH=tools/scratch/libplf_$v.so; fi
  python bench.py --batch ${AB_BATCH:-4096} --no-extras --cpu-seconds 0 --steps 8 2>/dev/null | V=$v python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print('%-10s %9.1f fps %8.3f ms/step  regions %7.3f ms  lines_map %d' % (os.environ['V'], d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['matches_frame0']['lines_map']))"
done
