#!/bin/bash
# mid-range batches on REAL texture: is the band table that was swept on the polygon scenes (tools/midrange_sweep2.sh) also the optimum for windows of the real
# photographs and for natural-image-like frames?  Run ON the GPU box.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { local fam=$1 label=$2 b=$3; shift 3
  local v=$(env "$@" timeout 600 python bench.py --no-extras --cpu-seconds 0 --family $fam --batch $b --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f fps, %.2f ms/step, region stage %.2f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))")
  echo "$fam B=$b  $label: $v"; }
for fam in photo natural; do
for b in 8 16 32 64 128 256 512; do
  run $fam "default" $b X=1
  for k in 2 4 8 12 16 24 32 48 64; do
    [ $((b * k)) -le 4096 ] && [ $((b * k)) -ge 256 ] && run $fam "rounds, $k bands" $b PLF_LSD_SPEC_Z=1024 PLF_LSD_SPEC_BANDS=$k
  done
  [ $b -ge 64 ] && run $fam "one wave per frame" $b PLF_LSD_SPEC_MAX=0 PLF_LSD_LAT_MAX=0
done
done
