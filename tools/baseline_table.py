#!/usr/bin/env python3
"""BASELINE.md section 4 table, measured in ONE run on the GPU box: for BASELINE configs 1-2 / 3 / 4 the GPU frames/s on one MI355X (device-resident
step of bench.py, the config's own frames in flight and -- for comparison -- many frames in flight), the CPU figures (a) one frame at a time on one
thread and (b) one frame per thread on all cores (oracle port, -O3 -march=native), and for config 5 the matchers alone.
    python tools/baseline_table.py  ->  gpurun_out/<tag>_baseline_table.json (tag = argv[1], default r05) + the table on stdout"""
import json, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench

out = {}
TAG = sys.argv[1] if len(sys.argv) > 1 else "r05"
cores = len(os.sched_getaffinity(0))
for cfg, big in ((2, 8192), (3, 8192), (4, 2048)):   # (eight region chains per SIMD: 32 frames per CU x 256 CUs)
    W, H, NF, NL, B0, label = bench.CONFIGS[cfg]
    row = {"label": label}
    for B in sorted({B0, big} if cfg != 2 else {8, big}):
        p = bench.Pipeline(W, H, NF, NL, B, 0, 30_000 + cfg)
        steps = 5 if B >= 1024 else 30
        el, _, _ = bench.timed(p, steps, 2)
        row["gpu_fps_%d_in_flight" % B] = round(B * steps / el, 1)
        if cfg == 2 and B == big:   # config 5: the four matchers alone on resident features
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): p.match_step()
            torch.cuda.synchronize()
            out["config5_match_only"] = {"gpu_frames_per_s": round(5 * B / (time.perf_counter() - t0), 1), "frames_in_flight": B,
                                         "what": "SearchByProjection vs %d map points + last frame, line projection vs %d map lines + kNN vs last frame's lines" % (bench.M_POINTS, bench.M_LINES)}
        p.close(); del p
    cpu = bench.cpu_baseline(20.0 if cfg != 4 else 30.0, cores, W, H, NF, NL)
    row["cpu_a_single_thread_fps"] = cpu["single_thread"]["frames_per_s"]; row["cpu_a_ms_median"] = cpu["single_thread"]["ms_median"]
    row["cpu_a_ms_p95"] = cpu["single_thread"]["ms_p95"]; row["cpu_a_two_thread_ms_median"] = cpu["single_thread"]["orb_lsd_on_two_threads_ms_median"]
    row["cpu_b_all_cores_fps"] = cpu["value"]; row["cpu_b_threads"] = cpu["cores"]; row["cpu_stage_ms"] = cpu["single_thread"]["stage_ms_per_frame"]
    bo, bl, _ = bench.algorithmic_bytes(W, H, NF, NL)
    row["algorithmic_MB_per_frame"] = round((bo + bl) / 1e6, 2)
    out["config%d" % cfg] = row
    if cfg == 2:
        # the same config on natural-image-like frames: 8 in flight, many in flight, one frame at a time
        nat = {}
        for B in (8, big):
            p = bench.Pipeline(W, H, NF, NL, B, 0, 31_000, family="natural")
            steps = 5 if B >= 1024 else 30
            el, rms, rn = bench.timed(p, steps, 2)
            nat["gpu_fps_%d_in_flight" % B] = round(B * steps / el, 1)
            if B == big:
                nat["region_kernel_ms"] = round(rms / max(rn, 1), 2); nat["region_chain_length"] = {k: v for k, v in p.chain_stats().items() if k != "what"}
                nat["nfa_rectangles_per_frame"] = p.rect_stats()
            p.close(); del p
        nat["one_frame"] = bench.single_frame_latency(0, W, H, NF, NL, family="natural")
        row["natural_frames"] = nat
        row["one_frame"] = bench.single_frame_latency(0, W, H, NF, NL)
        row["tracking_call"] = bench.tracking_call_latency(0, 2)
    if cfg == 3:
        row["tracking_call_one_frame"] = bench.tracking_call_latency(0, 3)
    if cfg == 2:
        st = cpu["single_thread"]["stage_ms_per_frame"]
        out.setdefault("config5_match_only", {})["cpu_single_thread_frames_per_s"] = round(1e3 / (st["match_points"] + st["match_lines"]), 1)
out["host_cores"] = cores
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "%s_baseline_table.json" % TAG), "w"), indent=1)
print(json.dumps(out, indent=1))
