#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_lines.py tests/test_gpu_random.py -x -q -m gpu 2>&1 | tail -3
for fam in polygons natural; do for B in 1 8; do python tools/latency_family.py $fam $B 10 2>&1 | grep LSD; done; done
python bench.py --no-extras --cpu-seconds 0 --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench %9.1f fps %8.3f ms/step  regions %7.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
