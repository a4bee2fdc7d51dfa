import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from rgbd_pl_slam_amd import LineSegment
from rgbd_pl_slam_amd.synth import synth_batch_parallel
B = 8192
base = synth_batch_parallel(50_000, 256, 640, 480, family="natural")
d = torch.from_numpy(base).cuda()[torch.arange(B).cuda() % 256].contiguous()
lines = torch.zeros((B, 100, 17), dtype=torch.float32, device="cuda"); ldesc = torch.zeros((B, 100, 32), dtype=torch.uint8, device="cuda")
leq = torch.zeros((B, 100, 3), dtype=torch.float64, device="cuda"); nl = torch.zeros(B, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for bal in (1, 0, 97, 96, 0, 1):
    ls = LineSegment(nlines=100, max_width=640, max_height=480, max_batch=B)
    ls.tune("balance", 1 if bal else 0)
    if bal > 1: os.environ["PLF_LSD_BALANCE_SCATTER"] = str(bal)
    else: os.environ.pop("PLF_LSD_BALANCE_SCATTER", None)
    ls.extract_batch_device(d, 640, 480, lines, ldesc, leq, nl, 100, s); torch.cuda.synchronize()
    os.environ.pop("PLF_LSD_BALANCE_DEBUG", None)
    ls.profile(enable=True, reset=True)
    for _ in range(3): ls.extract_batch_device(d, 640, 480, lines, ldesc, leq, nl, 100, s)
    torch.cuda.synchronize()
    ms, n = ls.profile(enable=False, reset=True)
    print("balance", bal, "region stage %.2f ms" % (ms / n), "chain", ls.chain_lengths(8)[:8])
    os.environ["PLF_LSD_BALANCE_DEBUG"] = "1"
    ls.close()
