#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_lines.py tests/test_gpu_random.py tests/test_gpu_batch.py -x -q -m gpu 2>&1 | tail -3
run() { python bench.py "$@" --no-extras --cpu-seconds 0 --steps 6 --warmup 2 2>/dev/null | V="$*" python -c "
import json,sys,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-50s %9.1f fps %8.3f ms/step  regions %7.3f ms' % (os.environ['V'], d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"; }
run --family polygons
PLF_LSD_BALANCE=0 run --family polygons
run --family natural
PLF_LSD_BALANCE=0 run --family natural
run --family polygons --batch 4096
PLF_LSD_BALANCE=0 run --family polygons --batch 4096
python tools/balance_probe.py natural 8192 1024 2>&1 | grep -v amdgpu.ids
