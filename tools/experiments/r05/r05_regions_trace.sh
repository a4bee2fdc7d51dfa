#!/bin/bash
# every launch of k_lsd_regions2 of a full bench.py run (all extras), in order: is there a slow mode?
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/rt
rocprofv3 --kernel-trace --output-format csv -d /tmp/rt -- python $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 > /tmp/rt.json 2>/dev/null
python3 - <<'PY'
import csv, glob, json
fn = glob.glob('/tmp/rt/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(fn)), key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
prev_end = {}
for i, r in enumerate(rows):
    n = r['Kernel_Name'].split('(')[0]
    if n.startswith('k_lsd_regions2'):
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        # the kernel in front of it on the same queue
        q = r.get('Queue_Id')
        j = i - 1
        while j >= 0 and rows[j].get('Queue_Id') != q: j -= 1
        gap = (s - int(rows[j]['End_Timestamp'])) / 1e6 if j >= 0 else -1
        print("t=%9.1f ms  k_lsd_regions2 %8.2f ms  grid %s  gap after %s: %.3f ms" % ((s - t0) / 1e6, (e - s) / 1e6, r.get('Grid_Size', r.get('Grid_Size_X', '?')), rows[j]['Kernel_Name'].split('(')[0][:20] if j >= 0 else '-', gap))
d = json.loads(open('/tmp/rt.json').read().strip().splitlines()[-1])
print("value", d["value"], "natural", d.get("natural", {}).get("in_flight_8192", {}).get("value"), d.get("natural", {}).get("in_flight_8192", {}).get("region_kernel_ms"))
PY
