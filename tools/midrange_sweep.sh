#!/bin/bash
# mid-range batches: the default step at 64 / 256 / 512 / 1024 / 2048 frames in flight under different line-extractor schedules (environment knobs are
# read when the handle is created).  Run ON the GPU box: bash tools/midrange_sweep.sh > gpurun_out/midrange.txt
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { # label, batch, env...
  local label=$1 b=$2; shift 2
  local v=$(env "$@" timeout 600 python bench.py --no-extras --cpu-seconds 0 --batch $b --steps 8 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f fps, %.2f ms/step, region stage %.2f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))")
  echo "B=$b  $label: $v"
}
for b in 64 256 512 1024 2048; do
  run "default" $b X=1
  run "regions2 wpg=8 (no speculation)" $b PLF_LSD_SPEC_MAX=0 PLF_LSD_LAT_MAX=0
  run "regions2 wpg=2" $b PLF_LSD_SPEC_MAX=0 PLF_LSD_LAT_MAX=0 PLF_LSD_WPG=2
  run "regions2 wpg=1" $b PLF_LSD_SPEC_MAX=0 PLF_LSD_LAT_MAX=0 PLF_LSD_WPG=1
  run "spec 4 bands" $b PLF_LSD_SPEC_MAX=4096 PLF_LSD_SPEC_BANDS=4
  run "spec 8 bands" $b PLF_LSD_SPEC_MAX=4096 PLF_LSD_SPEC_BANDS=8
  [ $b -le 256 ] && run "spec 16 bands" $b PLF_LSD_SPEC_MAX=4096 PLF_LSD_SPEC_BANDS=16
  [ $b -le 256 ] && run "rounds 8 bands (spec_z)" $b PLF_LSD_SPEC_Z=256 PLF_LSD_SPEC_BANDS=8
done
