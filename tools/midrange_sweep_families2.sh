#!/bin/bash
# 384-2048 frames in flight: banded validation rounds against the one-wave-per-frame kernel on the three image families.  Run ON the GPU box.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { local fam=$1 label=$2 b=$3; shift 3
  local v=$(env "$@" timeout 600 python bench.py --no-extras --cpu-seconds 0 --family $fam --batch $b --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f fps, %.2f ms/step, region stage %.2f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))")
  echo "$fam B=$b  $label: $v"; }
for fam in photo natural polygons; do
for b in 384 512 768 1024 2048; do
  run $fam "default" $b X=1
  for k in 2 4 8; do
    [ $((b * k)) -le 4096 ] && run $fam "rounds, $k bands" $b PLF_LSD_SPEC_MAX=4096 PLF_LSD_SPEC_Z=4096 PLF_LSD_SPEC_BANDS=$k
  done
done
done
