#!/usr/bin/env python3
"""Issue cost of the step's VALU instructions, by class: which share of the vector instructions of each heavy kernel issues at the fast
rate (one per ~2.2 cycles per SIMD) and which at the slow ones, and what that adds up to per step.

  python tools/classify_isa.py <sq_counters.txt from tools/pmc_sq.py> <frames per step> [library.so]  > profiles/<tag>_valu_classes.json

Method.  (1) The kernel's machine code is disassembled from the built library (llvm-objdump of the gfx950 code object).  (2) Every v_* instruction is put in
a cost class with the per-SIMD issue intervals MEASURED by tools/valu_issue.hip (profiles/r03_valu_issue.json, 8 waves per SIMD):
      2.2 cycles  v_add/sub_u32, v_and/or/xor_b32, v_mov_b32, v_add/sub/mul/fma/mac/fmac_f32 with VGPR or inline-constant sources
      4.2 cycles  everything else that is full rate on paper: shifts, min/max, mul24/mad24, add3, lshl_add, bfe/bfi, perm, alignbit/alignbyte, bcnt, sad, dot,
                  cvt, v_cmp, v_cndmask_e64, readlane/readfirstlane/writelane, DPP/SDWA forms, all packed (v_pk_*) forms, all fp64 add/mul/fma,
                  and ANY instruction of the 2.2 class that has an SGPR, VCC/EXEC or 32-bit literal source
      8   cycles  f32 / f16 transcendentals (rcp, rsq, sqrt, exp, log, sin, cos) and v_cndmask_b32_e32 (implicit VCC: 8.05 per cmp + cndmask pair and more when
                  several follow one compare)
      16  cycles  fp64 rcp / rsq / sqrt
(3) There are no per-instruction execution counts, so the static mix is weighted by loop depth: an instruction inside d nested loops (ranges closed by a
backward branch) counts 8^min(d, 3) times -- the hot loops decide the mix, straight-line set-up code hardly matters.  (4) The kernel's dynamic VALU count
(SQ_INSTS_VALU per launch x launches per step, tools/pmc_sq.py) times its weighted mean cost = the SIMD-cycles its vector instructions need to issue; the sum
over the kernels, divided by 1024 SIMDs x clock x step time, is `valu_issue_frac`: the share of the step during which an average SIMD's vector issue port is
taken.  bench.py divides by its own measured step time.  Listed per kernel: the share (and the most frequent mnemonics) of instructions that are in the
4.2-cycle class ONLY because of an SGPR / VCC / literal operand, and the share of packed forms."""
import collections, json, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin/"
FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32",
        "v_fma_f32", "v_mac_f32", "v_fmac_f32"}
TRANS32 = re.compile(r"^v_(rcp|rsq|sqrt|exp|log|sin|cos)(_iflag|_legacy)?_(f32|f16)")
TRANS64 = re.compile(r"^v_(rcp|rsq|sqrt)_f64")
LAUNCHES = {"k_orb_level": 8, "k_nfa_count1_w": 3, "k_nfa_count1": 3, "k_nfa_eval": 5, "k_nfa_count_w": 2, "k_nfa_count": 2, "k_nfa_math": 5, "k_build_grid": 2}


def disassemble(so):
    subprocess.check_call([LLVM + "llvm-objcopy", "--dump-section", ".hip_fatbin=/tmp/ci_fb", so])
    subprocess.check_call([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=/tmp/ci_fb", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=/tmp/ci_dev.co"])
    txt = subprocess.check_output([LLVM + "llvm-objdump", "-d", "--no-show-raw-insn", "/tmp/ci_dev.co"], text=True)
    kernels, cur = {}, None
    for ln in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(_Z\d+(k_[a-z0-9_]+)\w*)>:", ln)
        if m:
            cur = kernels.setdefault(m.group(2), [])
            continue
        if re.match(r"^[0-9a-f]+ <", ln):
            cur = None
            continue
        if cur is None:
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):(.*)$", ln)
        if m:
            t = re.search(r"<\S+\+0x([0-9a-f]+)>", m.group(4))
            cur.append((int(m.group(3), 16), m.group(1), m.group(2), int(t.group(1), 16) if t else None))
    return kernels


def classify(mn, ops):
    """-> (cycles, reason)"""
    base = re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", mn)
    if TRANS64.match(base):
        return 16.0, "fp64 transcendental"
    if TRANS32.match(base):
        return 8.0, "transcendental"
    if base == "v_cndmask_b32" and mn.endswith("_e32"):
        return 8.0, "cndmask on implicit VCC"
    if base.startswith("v_pk_"):
        return 4.2, "packed form"
    if base in FAST and not mn.endswith(("_dpp", "_sdwa")):
        srcs = ops.split(",")[1:]
        for s in srcs:
            s = s.strip()
            if re.match(r"^(s\d+|s\[|vcc|exec|ttmp|m0|src_)", s) or re.match(r"^0x[0-9a-f]+$", s) or re.match(r"^-?\d+\.\d+(e[-+]?\d+)?$", s) and s not in ("0.5", "1.0", "2.0", "4.0", "-0.5", "-1.0", "-2.0", "-4.0"):
                return 4.2, "SGPR / literal operand"
            if re.match(r"^-?\d+$", s) and not -16 <= int(s) <= 64:
                return 4.2, "SGPR / literal operand"
        return 2.2, "fast"
    if "_f64" in base:
        return 4.2, "fp64"
    return 4.2, "other 4.2-cycle opcode"


def analyse(insts):
    addr = [x[0] for x in insts]
    depth = [0] * len(insts)
    pos = {a: i for i, a in enumerate(addr)}
    for i, (a, mn, ops, toff) in enumerate(insts):
        if (mn.startswith("s_cbranch") or mn == "s_branch") and toff is not None:
            tgt = addr[0] + toff
            if tgt <= a and tgt in pos:
                for j in range(pos[tgt], i + 1):
                    depth[j] += 1
    w_tot = 0.0; cyc = 0.0
    by_reason = collections.Counter(); sg = collections.Counter(); pk = collections.Counter()
    for (a, mn, ops, _t), d in zip(insts, depth):
        if not mn.startswith("v_"):
            continue
        w = 8.0 ** min(d, 3)
        c, why = classify(mn, ops)
        w_tot += w; cyc += w * c; by_reason[why] += w
        if why == "SGPR / literal operand": sg[re.sub(r"_(e32|e64)$", "", mn)] += w
        if why == "packed form": pk[mn] += w
    if w_tot == 0:
        return None
    return {"mean_cycles_per_valu": round(cyc / w_tot, 3), "share_by_class": {k: round(v / w_tot, 4) for k, v in by_reason.most_common()},
            "slow_only_by_sgpr_or_literal_operand": {"share": round(sum(sg.values()) / w_tot, 4), "top": [k for k, _ in sg.most_common(6)]},
            "packed_forms": {"share": round(sum(pk.values()) / w_tot, 4), "top": [k for k, _ in pk.most_common(6)]},
            "static_valu_instructions": sum(1 for x in insts if x[1].startswith("v_")), "max_loop_depth": max(depth) if depth else 0}


def main():
    sq, frames = sys.argv[1], int(sys.argv[2])
    so = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "rgbd_pl_slam_amd", "libplf_hip.so")
    valu = {}
    for ln in open(sq):
        m = re.match(r"(k_\w+) .*SQ_INSTS_VALU=([0-9.e+]+)", ln)
        if m:
            valu[m.group(1)] = float(m.group(2))
    kern = disassemble(so)
    out, total_cycles, total_valu = {}, 0.0, 0.0
    for k, n in sorted(valu.items(), key=lambda kv: -kv[1] * LAUNCHES.get(kv[0], 1)):
        if k not in kern:
            continue
        a = analyse(kern[k])
        if a is None:
            continue
        per_step = n * LAUNCHES.get(k, 1)
        a["valu_per_launch"] = n; a["launches_per_step"] = LAUNCHES.get(k, 1)
        a["issue_simd_cycles_per_step"] = per_step * a["mean_cycles_per_valu"]
        total_cycles += a["issue_simd_cycles_per_step"]; total_valu += per_step
        out[k] = a
    print(json.dumps({"_doc": __doc__.split("\n\n")[0] + "  Method: tools/classify_isa.py docstring.",
                      "frames_per_step": frames, "simds": 1024, "clock_hz": 2.4e9,
                      "valu_wave_instructions_per_step": total_valu, "valu_issue_simd_cycles_per_step": total_cycles,
                      "mean_cycles_per_valu": round(total_cycles / total_valu, 3),
                      "valu_issue_ms_per_step_if_spread_evenly": round(total_cycles / 1024 / 2.4e9 * 1e3, 2),
                      "kernels": out}, indent=1))


if __name__ == "__main__":
    main()
