#!/bin/bash
# large-batch A/B of lsd_kernels.hip variants: tools/r05_ab_big.sh name="flags" ...   Run ON the GPU box.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
names=""
for kv in "$@"; do n=${kv%%=*}; fl=${kv#*=}; bash tools/variant_build.sh $n lsd_kernels.hip="$fl" > /tmp/vb_$n.log 2>&1 || tail -5 /tmp/vb_$n.log; names="$names $n"; done
AB_BATCH=8192 bash tools/ab.sh $names
