#!/usr/bin/env python3
"""profiles/<tag>_per_kernel.json and profiles/<tag>_issue_counters.json from the artefacts of one round's profile run (tools/final_r06.sh):

  python tools/per_kernel_roofline.py <tag> <serial kernel_stats.csv> <steps of that run> <overlapped kernel_stats.csv> <steps of that run> <pmc_traffic.json> <sq_counters.txt>

per kernel (the ten heaviest by solo time): launches per step, ALGORITHMIC bytes per launch (the SURVEY 8d term the kernel implements, for the 8192-frame launch of
bench.py's default step: VGA, 1000 ORB features, 100 lines), solo and overlapped duration, frac = algorithmic bytes / solo duration / 8 TB/s, counter traffic (FETCH_SIZE +
WRITE_SIZE of separate --pmc passes) and traffic / algorithmic.  Kernels whose work is not a term of 8d (octree, NFA validation) carry algorithmic_bytes null.
Issue counters: sum over the step's kernels of SQ_ACTIVE_INST_VALU (SQ counters count in units of 4 cycles of one SIMD) and of SQ_INSTS_VALU per 8192-frame step."""
import csv, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag, ser_csv, ser_steps, ovl_csv, ovl_steps, pmc_json, sq_txt = sys.argv[1:8]
ser_steps, ovl_steps = int(ser_steps), int(ovl_steps)
B, W, H, K, NL = 8192, 640, 480, 1000, 100
P0 = W * H; Ps = int(0.64 * P0); LAM = 80 * NL
import numpy as np
sp, plast, sf = 0, 0, np.float32(1.0)
for l in range(8):
    if l: sf = np.float32(np.float64(sf) * np.float64(np.float32(1.2)))
    lw, lh = int(np.rint(np.float32(W) / sf)), int(np.rint(np.float32(H) / sf))
    sp += lw * lh; plast = lw * lh
M, C = 5000, 0
TERMS = {   # kernel -> (SURVEY 8d term, bytes per frame)
    "k_orb_level": ("(SP - P_last) resize reads + (SP - P_0) level writes + SP FAST read + 2 SP blur, over the 8 launches", (sp - plast) + (sp - P0) + sp + 2 * sp),
    "k_orient_brief": ("(749 + 512 + 32 + 28) K", (749 + 512 + 32 + 28) * K),
    "k_lsd_pre": ("(P_0 + P_s) blur / downscale + 9 P_s gradient", (P0 + Ps) + 9 * Ps),
    "k_lsd_regions2": ("6 P_s region growing", 6 * Ps),
    "k_blur5_sobel3": ("5 P_0 Sobel", 5 * P0),
    "k_lbd": ("252 Lambda", 252 * LAM),
    "k_lsd_finalize": ("124 N_L", 124 * NL),
    "k_mp_candidates": ("56 M + 64 N map-point search (with k_mp_rounds; the 32 C candidate term is counted by the kernel itself)", 56 * M + 64 * K),
    "k_mp_rounds": ("(part of the term of k_mp_candidates)", None),
}
def stats(path, steps):
    out = {}
    rows = list(csv.DictReader(open(path)))
    # (the pipeline object runs one step of its own when it is built: the steps of a run = the launches of k_lsd_regions2, one per step, whatever --steps / --warmup said)
    for r in rows:
        if r["Name"].startswith("k_lsd_regions2("):
            steps = int(r["Calls"])
    for r in rows:
        n = r["Name"].split("(")[0]
        if n.startswith("k_"):
            out[n] = {"calls_per_step": float(r["Calls"]) / steps, "avg_ms": float(r["AverageNs"]) / 1e6, "per_step_ms": float(r["TotalDurationNs"]) / 1e6 / steps}
    return out
ser, ovl = stats(ser_csv, ser_steps), stats(ovl_csv, ovl_steps)
pmc = json.load(open(pmc_json))
rows = []
for n, s in sorted(ser.items(), key=lambda kv: -kv[1]["per_step_ms"])[:10]:
    term, bpf = TERMS.get(n, ("not a term of SURVEY 8d", None))
    launches = max(1, round(s["calls_per_step"]))
    alg = None if bpf is None else int(bpf * B / (8 if n == "k_orb_level" else 1))   # per launch (k_orb_level: the term covers its 8 launches)
    f = pmc["counters"]["FETCH_SIZE"].get(n, {}).get("per_launch_KB"); w = pmc["counters"]["WRITE_SIZE"].get(n, {}).get("per_launch_KB")
    traffic = None if f is None or w is None else int((f + w) * 1024 * B / pmc["frames_per_launch"])
    row = {"name": n, "launches_per_step": launches, "term": term, "algorithmic_bytes_per_launch": alg, "solo_ms_per_launch": round(s["avg_ms"], 3),
           "overlapped_ms_per_launch": round(ovl[n]["avg_ms"], 3) if n in ovl else None,
           "frac": None if alg is None else round(alg / (s["avg_ms"] * 1e-3) / 8e12, 5), "traffic_bytes_per_launch": traffic,
           "traffic_over_algorithmic": None if alg is None or traffic is None else round(traffic / alg, 2)}
    rows.append(row)
json.dump({"_doc": __doc__.split("\n\n")[1] if "\n\n" in __doc__ else "", "width": W, "height": H, "frames_per_launch": B, "kernels": rows},
          open(os.path.join(ROOT, "profiles", "%s_per_kernel.json" % tag), "w"), indent=1)
# issue counters
act, ins, per = 0.0, 0.0, {}
for ln in open(sq_txt):
    m = re.match(r"^(k_[a-z0-9_]+) (.*)$", ln.strip())
    if not m:
        continue
    kv = dict(p.split("=") for p in m.group(2).split())
    n = m.group(1)
    if n not in ser:
        continue
    L = ser[n]["calls_per_step"]
    a, i = float(kv.get("SQ_ACTIVE_INST_VALU", 0)) * L, float(kv.get("SQ_INSTS_VALU", 0)) * L
    per[n] = {"launches_per_step": round(L, 2), "active_inst_valu_per_step": a, "insts_valu_per_step": i, "wait_inst_any_per_step": float(kv.get("SQ_WAIT_INST_ANY", 0)) * L,
              "wave_cycles_per_step": float(kv.get("SQ_WAVE_CYCLES", 0)) * L, "busy_cycles_per_step": float(kv.get("SQ_BUSY_CYCLES", 0)) * L}
    act += a; ins += i
json.dump({"_doc": "rocprofv3 --pmc SQ counters of `bench.py --steps 2 --warmup 1 --cpu-seconds 0 --serial --no-extras` (tools/pmc_sq.py; the profiler serialises the dispatches "
                   "of a counter pass, so the per-kernel counts are those of the kernels run one after the other), scaled by the launches per step.  SQ_ACTIVE_INST_VALU "
                   "counts in units of 4 cycles of one SIMD: valu_busy_frac = sum x 4 / (1024 SIMDs x 2.4 GHz x step seconds), evaluated by bench.py with the step time "
                   "it measures (the OVERLAPPED step).", "frames_per_step": B, "simds": 1024, "clock_hz": 2.4e9, "active_inst_valu_per_step": act,
           "insts_valu_per_step": ins, "per_kernel": per}, open(os.path.join(ROOT, "profiles", "%s_issue_counters.json" % tag), "w"), indent=1)
print("wrote profiles/%s_per_kernel.json (%d kernels), profiles/%s_issue_counters.json: sum ACTIVE_INST_VALU %.4g, INSTS_VALU %.4g per step" % (tag, len(rows), tag, act, ins))
