#!/bin/bash
# per-launch solo durations of the NFA kernels of one large batch: tools/nfa_trace.sh [N]   (run ON the GPU box)
N=${1:-2048}
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/nfa_tr
rocprofv3 --kernel-trace --output-format csv -d /tmp/nfa_tr -- python $GRAFT_REPO_ROOT/tools/nfa_stats.py $N ${2:-256} > /tmp/nfa_tr.log 2>&1
grep -a "^frames\|^per frame\|Error" /tmp/nfa_tr.log
python3 - <<PY
import csv, glob
fn = glob.glob('/tmp/nfa_tr/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(fn)) if r['Kernel_Name'].startswith('k_')]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
last = max(i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('k_lsd_pre'))
tot = 0
for r in rows[last:]:
    n = r['Kernel_Name'].split('(')[0]; d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    if n.startswith('k_nfa'): tot += d
    print("  %-22s %8.3f ms" % (n, d))
print("NFA kernels: %.3f ms" % tot)
PY
