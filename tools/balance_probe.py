"""Does the large-batch region kernel (one wave per frame, 8 frames per workgroup, 8 waves per SIMD) last as long as its most loaded SIMD?  The frames of a batch are
re-ordered on the HOST so that every workgroup gets one frame of each octile of the chain lengths (pairs (0,7) (1,6) (2,5) (3,4) on waves v, v + 4 = one SIMD), and the
kernel is timed again.    python tools/balance_probe.py [natural|polygons] [B=8192] [distinct=1024]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from rgbd_pl_slam_amd import LineSegment
from rgbd_pl_slam_amd.synth import synth_batch_parallel
fam = sys.argv[1] if len(sys.argv) > 1 else "natural"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
ND = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
base = synth_batch_parallel(50_000, ND, 640, 480, family=fam)
ls = LineSegment(nlines=100, max_width=640, max_height=480, max_batch=B)
d = torch.empty((B, 480, 640), dtype=torch.uint8, device="cuda")
lines = torch.zeros((B, 100, 17), dtype=torch.float32, device="cuda"); ldesc = torch.zeros((B, 100, 32), dtype=torch.uint8, device="cuda")
leq = torch.zeros((B, 100, 3), dtype=torch.float64, device="cuda"); nl = torch.zeros(B, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
def run(order, label):
    src = torch.from_numpy(base).cuda()
    d.copy_(src[torch.from_numpy(order).cuda()])
    torch.cuda.synchronize()
    ls.extract_batch_device(d, 640, 480, lines, ldesc, leq, nl, 100, s); torch.cuda.synchronize()
    ls.profile(enable=True, reset=True)
    t = time.perf_counter()
    for _ in range(4): ls.extract_batch_device(d, 640, 480, lines, ldesc, leq, nl, 100, s)
    torch.cuda.synchronize(); el = (time.perf_counter() - t) / 4
    ms, n = ls.profile(enable=False, reset=True)
    c = ls.chain_lengths(B)
    print("%-34s region kernel %.2f ms, LSD+LBD %.2f ms per %d frames | chain min %d median %d max %d, per-workgroup sums max / mean %.3f" %
          (label, ms / n, el * 1e3, B, c.min(), np.median(c), c.max(), c.reshape(-1, 8).sum(1).max() / c.reshape(-1, 8).sum(1).mean()))
    return c
tiled = np.arange(B) % ND
c = run(tiled, "as tiled")
cd = c[:ND]                                   # chain length of each distinct frame
G = B // 8
srt = np.argsort(-c, kind="stable")           # frame slots by descending chain length
oct_of_wave = [0, 1, 2, 3, 7, 6, 5, 4]        # waves v and v + 4 (one SIMD) get octiles k and 7 - k
order = np.zeros(B, np.int64)
for w in range(G):
    for v in range(8):
        k = oct_of_wave[v]
        order[w * 8 + v] = tiled[srt[k * G + (w if k % 2 == 0 else G - 1 - w)]]
run(order, "balanced workgroups (snake)")
run(tiled[np.argsort(-c, kind="stable")], "sorted descending")
# sorted blocks of 8 (one workgroup = 8 adjacent ranks), the blocks snaked over the four workgroups of a CU (workgroups c, c + 256, c + 512, c + 768 share a CU)
blk = np.arange(G)
q, cpos = blk // (G // 4), blk % (G // 4)
src_blk = q * (G // 4) + np.where(q % 2 == 0, cpos, G // 4 - 1 - cpos)
order2 = np.concatenate([tiled[srt[b * 8:(b + 1) * 8]] for b in src_blk])
run(order2, "sorted blocks, snaked over the CU")
rng = np.random.default_rng(1)
run(tiled[rng.permutation(B)], "random order")
ls.close()
