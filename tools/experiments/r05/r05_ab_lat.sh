#!/bin/bash
# few-frames latency A/B (polygons / natural, 1 and 8 in flight; config 3 as specified) of the in-tree library against the named scratch variants.  Run ON the GPU box.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
[ -n "$AB_TESTS" ] && timeout 1200 python -m pytest $AB_TESTS -m gpu -x -q 2>&1 | tail -3
for rep in 1 2; do for v in base "$@"; do
  if [ $v = base ]; then unset PLF_LIB_PATH; else export PLF_LIB_PATH=tools/scratch/libplf_$v.so; fi
  for fam in polygons natural; do for B in 1 8; do echo -n "$v: "; python tools/latency_family.py $fam $B ${AB_CALLS:-10} 2>&1 | grep LSD; done; done
done; done
