#!/bin/bash
# mid-range batches, second sweep: validation rounds (PLF_LSD_SPEC_Z raised) with different band counts.  Run ON the GPU box.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { local label=$1 b=$2; shift 2
  local v=$(env "$@" timeout 600 python bench.py --no-extras --cpu-seconds 0 --batch $b --steps 8 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f fps, %.2f ms/step, region stage %.2f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))")
  echo "B=$b  $label: $v"; }
for b in 16 32 64 128 256 512; do
  run "default" $b X=1
  for k in 4 8 12 16 24 32; do
    [ $((b * k)) -le 6144 ] && run "rounds, $k bands" $b PLF_LSD_SPEC_Z=1024 PLF_LSD_SPEC_BANDS=$k
  done
done
run "regions2 wpg=2" 512 PLF_LSD_SPEC_MAX=0 PLF_LSD_LAT_MAX=0 PLF_LSD_WPG=2
run "regions2 wpg=4" 512 PLF_LSD_SPEC_MAX=0 PLF_LSD_LAT_MAX=0 PLF_LSD_WPG=4
run "regions2 wpg=4" 1024 PLF_LSD_SPEC_MAX=0 PLF_LSD_LAT_MAX=0 PLF_LSD_WPG=4
