#!/usr/bin/env python3
"""Per-kernel SQ instruction / busy counters of bench.py (separate rocprofv3 --pmc passes, kernel-trace only).
Run ON the GPU box:  python tools/pmc_sq.py [bench args...]   -> prints per-launch averages for the heavy kernels."""
import csv, glob, os, subprocess, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
bargs = sys.argv[1:] or ["--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--serial"]
env = dict(os.environ, TMPDIR="/tmp")
if os.environ.get("PMC_GROUPS"):
    groups_env = [g.split(",") for g in os.environ["PMC_GROUPS"].split(";")]
groups = [["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD"], ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES"],
          ["SQ_INST_CYCLES_SALU", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAVES"]]
if os.environ.get("PMC_GROUPS"):
    groups = groups_env
res = collections.defaultdict(dict)
for gi, g in enumerate(groups):
    d = "/tmp/pmc_sq_%d" % gi
    subprocess.run(["rm", "-rf", d])
    subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + g + ["--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "bench.py")] + bargs,
                   cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
    acc = collections.defaultdict(lambda: [0, 0.0])
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            k = (r["Kernel_Name"].split("(")[0], r["Counter_Name"])
            acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
    for (k, c), v in acc.items():
        res[k][c] = v[1] / v[0]
for k in sorted(res, key=lambda k: -res[k].get("SQ_BUSY_CYCLES", 0))[:16]:
    print(k, " ".join("%s=%.3g" % (c, v) for c, v in sorted(res[k].items())))
