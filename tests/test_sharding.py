"""Multi-GPU path: frames are sharded over ranks with NO data-path collective; the only
communication is the barrier + MAX-reduction of the elapsed time in bench.py.  Exercised here with
world_size 2 on the gloo backend (CPU)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def shard(n_frames, rank, world):
    """contiguous block sharding used by the batch driver: frame f goes to rank f*world//n_frames"""
    lo = rank * n_frames // world
    hi = (rank + 1) * n_frames // world
    return lo, hi


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard(37, rank, world)
    elapsed = torch.tensor([0.25 + 0.5 * rank], dtype=torch.float64)   # pretend per-rank wall time
    dist.barrier()
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    counts = torch.tensor([hi - lo], dtype=torch.int64)
    dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    q.put((rank, lo, hi, float(elapsed[0]), int(counts[0])))
    dist.destroy_process_group()


def test_world2_sharding_and_timing_reduction():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(timeout=60)
    assert res[0][1:3] == (0, 18) and res[1][1:3] == (18, 37)      # disjoint, covering, balanced
    assert all(r[3] == 0.75 for r in res)                           # MAX over ranks
    assert all(r[4] == 37 for r in res)


def test_shard_covers_every_frame_once():
    for n in (1, 7, 64, 257):
        for world in (1, 2, 4, 8):
            seen = np.zeros(n, int)
            for r in range(world):
                lo, hi = shard(n, r, world)
                seen[lo:hi] += 1
            assert np.all(seen == 1)
