#!/bin/bash
# (PLF_NFA_EVAL_BLOCKS and the ev64 / evw8 libraries belong to an experiment patch that was not kept -- profiles/r04_prepass_ab.txt has its record; the script documents how it was measured)
# tools/ab_eval.sh: the starved k_nfa_eval launch of the overlapped step (profiles/r04_timeline.txt) under different launch shapes.  Run ON the GPU box.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() {  # name, env..., then bench
  name=$1; shift
  r=$(env "$@" python bench.py --no-extras --cpu-seconds 0 --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f fps %.2f ms/step' % (d['value'], d['ms_per_step']))")
  echo "$name: $r"
}
run base X=1
run blocks512 PLF_NFA_EVAL_BLOCKS=512
run blocks128 PLF_NFA_EVAL_BLOCKS=128
run nfa_list PLF_NFA_LIST=1
run ev64 PLF_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/libplf_ev64.so
run ev64_b512 PLF_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/libplf_ev64.so PLF_NFA_EVAL_BLOCKS=512
run evw8 PLF_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/libplf_evw8.so
run ob2_wide_eval PLF_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/libplf_ob2.so
run head PLF_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/libplf_head.so
run base X=1
for cfg in "X=1" "PLF_NFA_EVAL_BLOCKS=128"; do echo "== timeline $cfg"; env $cfg bash tools/timeline.sh abev 2>&1 | grep -E "k_nfa_eval|k_match_project_lines|k_nfa_count1_w" | head -12; done
