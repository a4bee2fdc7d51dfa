#!/usr/bin/env python3
"""SQ instruction counts of k_orb_level per phase: the kernel is built with -DOF_STOP=N (it returns after phase N; outputs are wrong, counts are valid) into
scratch libraries and bench.py (serial, 1024 frames, one step) runs under rocprofv3 --pmc for each.  Cumulative VALU / SALU / LDS / VMEM instruction counts per
launch come out; the differences are the phases (1 level pixels / resize, 2 plane write, 3 blur, 4 FAST pre-test, 5 score, 6 NMS, full = + emission).

    tools/orb_phase_insts.py build      # here (no GPU): tools/scratch/libplf_ofs{1..6}.so
    python tools/orb_phase_insts.py     # ON the GPU box
"""
import collections, csv, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1:] == ["build"]:
    for n in range(1, 7):
        subprocess.check_call([os.path.join(ROOT, "tools", "variant_build.sh"), "ofs%d" % n, "orb_front.hip=-DOF_STOP=%d" % n])
    sys.exit(0)
env0 = dict(os.environ, TMPDIR="/tmp")
for v in ["ofs1", "ofs2", "ofs3", "ofs4", "ofs5", "ofs6", "full"]:
    env = dict(env0)
    if v != "full":
        env["PLF_LIB_PATH"] = os.path.join(ROOT, "tools", "scratch", "libplf_%s.so" % v)
    d = "/tmp/oi_%s" % v
    subprocess.run(["rm", "-rf", d])
    subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "--output-format", "csv", "-d", d, "--",
                    sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--cpu-seconds", "0", "--serial", "--no-extras", "--batch", "1024"],
                   cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
    acc = collections.defaultdict(lambda: [0, 0.0])
    per_level = collections.defaultdict(list)   # VALU per launch in dispatch order: the 8 levels of each extractor call
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(fn)) if r["Kernel_Name"].startswith("k_orb_level")]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        for r in rows:
            acc[r["Counter_Name"]][0] += 1; acc[r["Counter_Name"]][1] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_INSTS_VALU":
                per_level[int(r["Grid_Size"])].append(float(r["Counter_Value"]))
    print(v, {k: "%.3g per launch (%d launches)" % (a[1] / max(a[0], 1), a[0]) for k, a in sorted(acc.items())}, flush=True)
    print(v, "SQ_INSTS_VALU per level (by grid size, largest = level 0):", ["%.3g" % (sum(x) / len(x)) for g_, x in sorted(per_level.items(), reverse=True)], flush=True)
