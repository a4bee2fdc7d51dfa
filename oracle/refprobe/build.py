#!/usr/bin/env python3
"""Build and run the reference-binary probe (container only; needs /root/reference).

  python oracle/refprobe/build.py            # build stubs + probe into oracle/_ref/, write tests/golden/ref_*.json

The prebuilt reference library has 8 NEEDED sonames that this image lacks (OpenCV 3.3,
Pangolin, DBoW2, g2o).  They are satisfied by abort()-stub shared objects generated from the
library's own undefined-symbol list, so that the dynamic loader can map it.  None of the stubbed
functions is ever reached by the functions probe.cpp executes (they would abort()).
Outputs: oracle/_ref/ (git-ignored binaries) and tests/golden/ref_*.json (committed fixtures).
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference/lib/libORB_SLAM2.so"
OUT = os.path.join(ROOT, "oracle/_ref")
STUBS = os.path.join(OUT, "stubs")
SONAMES = ["libpangolin.so", "libDBoW2.so", "libg2o.so", "libopencv_calib3d3.so.3.3",
           "libopencv_features2d3.so.3.3", "libopencv_highgui3.so.3.3", "libopencv_imgproc3.so.3.3",
           "libopencv_core3.so.3.3"]

def sh(cmd):
    print("+", " ".join(cmd))
    subprocess.check_call(cmd)


def main():
    if not os.path.exists(REF):
        print("reference binary not present; nothing to do (fixtures are committed)")
        return 0
    os.makedirs(STUBS, exist_ok=True)
    here = os.path.dirname(os.path.abspath(__file__))
    # entry points probe.cpp defines itself (the OpenCV substitutes of tiers B and C) must not be shadowed by abort stubs
    sh(["g++", "-c", "-O1", "-std=c++14", "-fno-builtin-malloc", "-I", os.path.join(ROOT, "oracle"), os.path.join(here, "probe.cpp"),
        "-o", os.path.join(OUT, "probe.o")])
    own = set()
    for ln in subprocess.check_output(["nm", "--defined-only", os.path.join(OUT, "probe.o")], text=True).splitlines():
        f = ln.split()
        if len(f) == 3 and f[1] in "TW" and (f[2].startswith("_ZN2cv") or f[2].startswith("_ZNK2cv")):
            own.add(f[2])
    syms = subprocess.check_output(["readelf", "-Ws", "--dyn-syms", REF], text=True).splitlines()
    funcs, objs = set(), set()
    for ln in syms:
        f = ln.split()
        if len(f) < 8 or f[6] != "UND" or "@" in f[7]:
            continue
        if f[3] == "FUNC" and f[7] not in own:
            funcs.add(f[7])
        elif f[3] == "OBJECT":
            objs.add(f[7])
    with open(os.path.join(STUBS, "stub.c"), "w") as f:
        f.write("#include <stdlib.h>\n")
        for o in sorted(objs):
            f.write("void *%s[64];\n" % o)
        for fn in sorted(funcs):
            f.write("void %s(void){abort();}\n" % fn)
    with open(os.path.join(STUBS, "empty.c"), "w") as f:
        f.write("int plf_refprobe_stub_%d;\n" % 0)
    for i, so in enumerate(SONAMES):
        src = "stub.c" if i == 0 else "empty.c"
        sh(["gcc", "-shared", "-fPIC", "-w", "-Wl,-soname," + so, "-o", os.path.join(STUBS, so),
            os.path.join(STUBS, src)])
    probe = os.path.join(OUT, "probe")
    sh(["gcc", "-c", "-O2", "-ffp-contract=off", "-fPIC", "-I", os.path.join(ROOT, "oracle"),
        os.path.join(ROOT, "oracle/orb_oracle.c"), "-o", os.path.join(OUT, "orb_oracle_probe.o")])
    sh(["gcc", "-c", "-O2", "-ffp-contract=off", "-fPIC", "-I", os.path.join(ROOT, "oracle"),
        os.path.join(ROOT, "oracle/frame_oracle.c"), "-o", os.path.join(OUT, "frame_oracle_probe.o")])
    sh(["gcc", "-c", "-O2", "-fPIC", "-I", os.path.join(ROOT, "oracle"),
        os.path.join(ROOT, "oracle/timing.c"), "-o", os.path.join(OUT, "timing_probe.o")])
    sh(["g++", "-rdynamic", os.path.join(OUT, "probe.o"),
        os.path.join(OUT, "orb_oracle_probe.o"), os.path.join(OUT, "frame_oracle_probe.o"), os.path.join(OUT, "timing_probe.o"), "-o", probe, "-L/root/reference/lib", "-l:libORB_SLAM2.so",
        "-L" + STUBS, "-Wl,--allow-shlib-undefined", "-Wl,-rpath-link," + STUBS, "-ldl", "-lm"])
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = STUBS + ":/root/reference/lib:" + env.get("LD_LIBRARY_PATH", "")
    gold = os.path.join(ROOT, "tests/golden")
    os.makedirs(gold, exist_ok=True)
    subprocess.check_call([probe, gold], env=env)
    # the ctor's pattern copy must equal the .data table
    a = open(os.path.join(gold, "ref_pattern_from_ctor.bin"), "rb").read()
    b = open(os.path.join(gold, "bit_pattern_31.bin"), "rb").read()
    assert a == b, "pattern copied by the reference ctor differs from .data table"
    os.remove(os.path.join(gold, "ref_pattern_from_ctor.bin"))
    print("ok")
    return 0


if __name__ == "__main__":
    sys.exit(main())
