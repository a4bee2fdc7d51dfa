"""kernels of the LAST line-extractor call of a rocprofv3 --kernel-trace run, in order, with start and duration:
    rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python tools/latency_probe.py; python tools/kernel_sequence.py /tmp/kt"""
import csv, sys, glob
fn = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(fn)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last occurrence of k_lsd_pre starts the last frame
idx = max(i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_lsd_pre"))
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    print("%-28s start %8.1f us  dur %7.1f us" % (r["Kernel_Name"].split("(")[0][:28], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
