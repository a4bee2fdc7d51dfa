#!/bin/bash
# frames (waves) per workgroup of the large-batch region kernel at 8192 in flight on real texture (the default 8 was chosen on the polygon scenes).  Run ON the GPU box.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for fam in photo natural; do
  for k in 2 4 8 16; do
    v=$(PLF_LSD_WPG=$k timeout 400 python bench.py --no-extras --cpu-seconds 0 --family $fam --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f fps, %.2f ms/step, region kernel %.2f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))")
    echo "$fam 8192 in flight, $k frames per workgroup: $v"
  done
done
