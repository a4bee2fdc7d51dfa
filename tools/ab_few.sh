#!/bin/bash
# tools/ab_few.sh <name>="<flags for lsd_kernels.hip>" ...: few-frames latency (polygons / natural, 1 and 8 in flight) of the in-tree library and of variants built here.  Run ON the GPU box.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
names=""
for kv in "$@"; do n=${kv%%=*}; fl=${kv#*=}; bash tools/variant_build.sh $n lsd_kernels.hip="$fl" > /tmp/vb_$n.log 2>&1 || tail -5 /tmp/vb_$n.log; names="$names $n"; done
for rep in 1 2; do for v in base $names; do
  if [ $v = base ]; then unset PLF_LIB_PATH; else export PLF_LIB_PATH=tools/scratch/libplf_$v.so; fi
  for fam in polygons natural; do for B in 1 8; do echo -n "$v: "; python tools/latency_family.py $fam $B ${AB_CALLS:-10} 2>&1 | grep LSD; done; done
done; done
