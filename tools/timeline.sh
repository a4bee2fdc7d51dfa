#!/bin/bash
# kernel timeline of one bench step (overlapped streams): tools/timeline.sh <tag> [bench args].  Run ON the GPU box.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/tl_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$tag -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --cpu-seconds 0 --no-extras "$@" > /tmp/tl_$tag.log 2>&1
python3 - <<PY
import csv, glob
fn = glob.glob('/tmp/tl_$tag/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(fn)) if r['Kernel_Name'].startswith('k_')]
for r in rows: r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
pre = [r for r in rows if r['Kernel_Name'].startswith('k_lsd_pre')]
t0 = pre[-2]["s"] if len(pre) > 1 else pre[-1]["s"]
out = []
for r in rows:
    if r['s'] < t0: continue
    name = r['Kernel_Name'].split('(')[0]; q = r.get('Queue_Id', r.get('Stream_Id', '?'))
    if out and out[-1][0] == name and out[-1][1] == q and r['s'] - out[-1][3] < 3e6: out[-1][3] = r['e']; out[-1][4] += 1
    else: out.append([name, q, r['s'], r['e'], 1])
print("timeline of the last two steps (ms after the k_lsd_pre of the first of them started):")
for name, q, s, e, n in sorted(out, key=lambda x: x[2]):
    print("  q%-3s %-26s x%-3d %8.2f -> %8.2f  (%.2f)" % (q, name[:26], n, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6))
PY
