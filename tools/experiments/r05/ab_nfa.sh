#!/bin/bash
# k_nfa_small at different occupancies (solo duration from a kernel trace).  Run ON the GPU box.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for w in "$@"; do
  if [ "$w" = base ]; then unset PLF_LIB_PATH; else export PLF_LIB_PATH=$GRAFT_REPO_ROOT/tools/scratch/libplf_$w.so; fi
  echo "== $w"; bash tools/nfa_trace.sh 8192 1024 | grep "nfa_small\|NFA kernels\|^frames"
done
