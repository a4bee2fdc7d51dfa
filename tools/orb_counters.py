#!/usr/bin/env python3
"""SQ counters of k_orb_level for a list of libraries ("tree" = in-tree, else tools/scratch/libplf_<name>.so): bench.py serial, 1024 frames, one step, under
rocprofv3 --pmc (kernel-trace only), one pass per counter group.  Run ON the GPU box:  python tools/orb_counters.py tree r6base ...  [KERNEL=k_orb_level]"""
import collections, csv, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = os.environ.get("KERNEL", "k_orb_level")
GROUPS = [["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD"], ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES"],
          ["SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_LDS", "SQ_WAVES"], ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INST_CYCLES_SALU", "SQ_WAIT_INST_LDS"],
          ["SQ_ACTIVE_INST_ANY", "SQ_INSTS_VMEM_WR", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC"]]
bargs = ["--steps", "1", "--warmup", "1", "--cpu-seconds", "0", "--serial", "--no-extras", "--batch", os.environ.get("BATCH", "1024")]
if os.environ.get("NATURAL"):
    bargs += ["--family", "natural"]
for v in sys.argv[1:]:
    env = dict(os.environ, TMPDIR="/tmp")
    if v != "tree":
        env["PLF_LIB_PATH"] = os.path.join(ROOT, "tools", "scratch", "libplf_%s.so" % v)
    res = {}
    for gi, g in enumerate(GROUPS):
        d = "/tmp/oc_%s_%d" % (v, gi)
        subprocess.run(["rm", "-rf", d])
        subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + g + ["--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "bench.py")] + bargs,
                       cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
        acc = collections.defaultdict(lambda: [0, 0.0])
        for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(fn)):
                if r["Kernel_Name"].startswith(KERNEL):
                    acc[r["Counter_Name"]][0] += 1; acc[r["Counter_Name"]][1] += float(r["Counter_Value"])
        for k, a in acc.items():
            res[k] = a[1] / max(a[0], 1)
    print(v, KERNEL, "per launch:", " ".join("%s=%.4g" % kv for kv in sorted(res.items())), flush=True)
