#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { python bench.py "$@" --no-extras --cpu-seconds 0 --steps 6 --warmup 2 2>/dev/null | V="$*" python -c "
import json,sys,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-60s %9.1f fps %8.3f ms/step  regions %7.3f ms' % (os.environ['V'], d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"; }
run --family natural --batch 8192
run --family natural --batch 4096 --line-handles 2
run --family natural --batch 4096
run --family natural --batch 6144 --line-handles 2
run --family polygons --batch 8192
run --family polygons --batch 4096 --line-handles 2
