#!/bin/bash
# tools/latency_profile.sh <tag>: rocprofv3 kernel trace of the few-frames schedule (1 and 8 VGA frames in flight, polygon and natural-image-like frames):
# per-kernel statistics + the launch-by-launch timeline of the LAST call of each run.  Run ON the GPU box; writes gpurun_out/<tag>_latency_*.
tag=$1
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
: > $OUT/${tag}_latency_timeline.txt
for fam in polygons natural; do for B in 1 8; do
  d=/tmp/lp_${tag}_${fam}_$B; rm -rf $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $GRAFT_REPO_ROOT/tools/latency_family.py $fam $B > /tmp/lp.log 2>&1
  grep 'LSD+LBD' /tmp/lp.log >> $OUT/${tag}_latency_timeline.txt
  cp $(find $d -name '*kernel_stats.csv' | head -1) $OUT/${tag}_latency_kernel_stats_${fam}_B$B.csv
  python3 - $d $fam $B >> $OUT/${tag}_latency_timeline.txt <<'PY'
import csv, glob, sys
fn = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(fn)), key=lambda r: int(r['Start_Timestamp']))
idx = max(i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('k_lsd_pre'))
t0 = int(rows[idx]['Start_Timestamp'])
print("  timeline of the last call (%s, %s in flight), us after k_lsd_pre started:" % (sys.argv[2], sys.argv[3]))
out = []
for r in rows[idx:]:
    n = r['Kernel_Name'].split('(')[0]; s = int(r['Start_Timestamp']); e = int(r['End_Timestamp'])
    base = n
    if out and out[-1][0] == base: out[-1][2] = e; out[-1][3] += 1; out[-1][4] += e - s
    elif len(out) > 1 and base in ('k_lsd_spec_prefix', 'k_lsd_spec_validate') and out[-1][0] in ('k_lsd_spec_prefix', 'k_lsd_spec_validate', 'rounds'):
        out[-1][0] = 'rounds'; out[-1][2] = e; out[-1][3] += 1; out[-1][4] += e - s
    else: out.append([base, s, e, 1, e - s])
for n, s, e, c, busy in out:
    print("    %-28s x%-3d %8.1f -> %8.1f  (span %7.1f, kernel time %7.1f)" % (n[:28], c, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, busy / 1e3))
PY
done; done
cat $OUT/${tag}_latency_timeline.txt
