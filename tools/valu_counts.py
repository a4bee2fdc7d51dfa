#!/usr/bin/env python3
"""profiles/<tag>_valu_counts.json from the output of tools/pmc_sq.py (serial step): wave-level VALU instructions per step = sum over the heaviest kernels
of SQ_INSTS_VALU per launch x launches per step.  Usage: python tools/valu_counts.py <sq_counters.txt> <frames per step> > profiles/<tag>_valu_counts.json"""
import json, re, sys
LAUNCHES = {"k_orb_level": 8, "k_nfa_count1": 3, "k_nfa_eval": 5, "k_nfa_count": 2, "k_nfa_math": 5, "k_build_grid": 2}
per = {}
for ln in open(sys.argv[1]):
    m = re.match(r"(k_\w+) .*SQ_INSTS_VALU=([0-9.e+]+)", ln)
    if m:
        per[m.group(1)] = float(m.group(2))
tot = sum(v * LAUNCHES.get(k, 1) for k, v in per.items())
print(json.dumps({"_doc": "wave-level VALU instructions of one step of `python bench.py` (serial run, SQ_INSTS_VALU per launch from tools/pmc_sq.py, times the launches "
                          "per step; the 16 heaviest kernels, i.e. all but ~1 %), and the chip's measured issue rates (profiles/r03_valu_issue.json: 256 CUs, 8 waves per SIMD)",
                  "frames_per_step": int(sys.argv[2]), "valu_wave_instructions_per_step": tot, "per_kernel_per_launch": per, "launches_per_step": LAUNCHES,
                  "rate_full_wave_instr_per_s": 1.12e12, "rate_half_wave_instr_per_s": 5.85e11}, indent=1))
