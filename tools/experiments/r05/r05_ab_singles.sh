#!/bin/bash
# A/B of the single-pixel-seed fast path (PLF_LSD_SINGLES): exactness tests first, then the large-batch step (polygons / natural) and the few-frames latency, in-tree
# library against tools/scratch/libplf_nosingles.so (built with -DPLF_LSD_SINGLES=0).  Run ON the GPU box.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_lines.py tests/test_gpu_random.py tests/test_gpu_batch.py -m gpu -x -q 2>&1 | tail -3
run() { python bench.py "$@" --no-extras --cpu-seconds 0 --steps 6 --warmup 2 2>/dev/null | V="$PLF_LIB_PATH $*" python -c "
import json,sys,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-70s %9.1f fps %8.3f ms/step  regions %7.3f ms' % (os.environ['V'], d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"; }
for v in base nosingles base nosingles; do
  if [ $v = base ]; then unset PLF_LIB_PATH; else export PLF_LIB_PATH=tools/scratch/libplf_$v.so; fi
  run --family natural --batch 8192
  run --family polygons --batch 8192
  run --family polygons --batch 8192 --serial
  run --family natural --batch 8192 --serial
done
for rep in 1 2; do for v in base nosingles; do
  if [ $v = base ]; then unset PLF_LIB_PATH; else export PLF_LIB_PATH=tools/scratch/libplf_$v.so; fi
  for fam in polygons natural; do for B in 1 8; do echo -n "$v: "; python tools/latency_family.py $fam $B 10 2>&1 | grep LSD; done; done
done; done
