#!/usr/bin/env python3
"""Collect per-kernel HBM traffic of bench.py from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) and
write profiles/<tag>_pmc_traffic.json.  Run ON the GPU box:  python tools/pmc_traffic.py <tag> [bench args...]
Counter units: KB (rocprofv3 derived metrics).  See /opt/skills/guides/MI355X_MICROARCH.md, HBM section."""
import csv, glob, json, os, subprocess, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
bargs = sys.argv[2:] or ["--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--no-extras"]
out = {"_doc": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel-trace only) of `python bench.py %s` on one MI355X; "
               "KB per launch as reported by rocprofv3.  MI355X_MICROARCH.md: FETCH_SIZE under-counts wide (16 B/lane) streaming reads by 2x on "
               "gfx950; k_lsd_regions2 issues 4-16-byte gathers, so no correction is applied to it; WRITE_SIZE is uncalibrated." % " ".join(bargs),
       "counters": {}}
env = dict(os.environ, TMPDIR="/tmp")
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    d = "/tmp/pmc_%s" % ctr
    subprocess.run(["rm", "-rf", d])
    subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable,
                    os.path.join(ROOT, "bench.py")] + bargs, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
    acc = collections.defaultdict(lambda: [0, 0.0])
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if r.get("Counter_Name") != ctr:
                continue
            k = r["Kernel_Name"].split("(")[0]
            acc[k][0] += 1
            acc[k][1] += float(r["Counter_Value"])
    out["counters"][ctr] = {k: {"launches": v[0], "per_launch_KB": round(v[1] / v[0], 1)} for k, v in acc.items() if k.startswith("k_")}
B = 8192   # bench.py's default frames in flight (config 2)
for i, a in enumerate(bargs):
    if a == "--batch":
        B = int(bargs[i + 1])
out["frames_per_launch"] = B
out["width"], out["height"] = 640, 480   # (bench.py --config 2)
rk = [k for k in out["counters"]["FETCH_SIZE"] if k.startswith("k_lsd_regions")]
rk = rk[0] if rk else "k_lsd_regions2"
f = out["counters"]["FETCH_SIZE"].get(rk, {}).get("per_launch_KB", 0.0)
w = out["counters"]["WRITE_SIZE"].get(rk, {}).get("per_launch_KB", 0.0)
out["region_kernel"] = {"name": rk, "hbm_bytes_per_launch": int((f + w) * 1024), "fetch_KB": f, "write_KB": w}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "%s_pmc_traffic.json" % tag), "w"), indent=1)
print(json.dumps(out["region_kernel"]))
