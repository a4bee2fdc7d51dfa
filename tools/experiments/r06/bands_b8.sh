#!/bin/bash
# BASELINE configs[2] (8 frames in flight, 2000 + 200): band count of the speculative schedule, three image families.  Run ON the GPU box.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for fam in polygons natural photo; do
  for k in 24 32 40 48 56 64; do
    v=$(PLF_LSD_SPEC_BANDS=$k timeout 300 python bench.py --no-extras --cpu-seconds 0 --family $fam --config 3 --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f fps, %.2f ms/step, region stage %.2f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))")
    echo "$fam 8 in flight, $k bands: $v"
  done
done
