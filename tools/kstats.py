#!/usr/bin/env python3
"""print a compact table of a rocprofv3 *_kernel_stats.csv: kstats.py <csv> [passes] [min_total_ms]"""
import csv, sys
fn = sys.argv[1]; passes = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0; mn = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
tot = 0.0
for r in csv.DictReader(open(fn)):
    n = r["Name"].split("(")[0][:34]; c = int(r["Calls"]); t = float(r["TotalDurationNs"]) / 1e6
    tot += t
    if t >= mn:
        print("%-34s calls %4d avg %8.3f ms  per-pass %8.3f ms" % (n, c, t / c, t / passes))
print("total per pass %.2f ms" % (tot / passes))
