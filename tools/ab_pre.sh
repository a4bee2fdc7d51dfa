#!/bin/bash
# tools/ab_pre.sh <variant names ...>: for the in-tree library and each tools/variants/ (copied from tools/scratch, which does not travel) libplf_<name>.so -- line exactness suite, then solo kernel times of the pre-pass. Run ON the GPU box.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in base "$@"; do
  if [ $v = base ]; then unset PLF_LIB_PATH; else export PLF_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/libplf_$v.so; fi
  echo "== $v"
  if [ $v != base ]; then timeout 900 python -m pytest tests/test_gpu_lines.py tests/test_gpu_random.py -x -q -m gpu 2>&1 | tail -2; fi
  bash tools/prof_serial.sh ab_$v 2>&1 | grep -E "sum of solo|k_lsd_pre|k_blur5|k_lsd_regions2|k_orb_level"
  python bench.py --no-extras --cpu-seconds 0 --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench %.0f fps %.2f ms/step' % (d['value'], d['ms_per_step']))"
done
