"""Multi-GPU path: frames are sharded over GPUs / ranks by contiguous blocks with NO data-path collective;
the only communication is the barrier + MAX-reduction of the elapsed time in bench.py.  The partition is PRODUCT code:
plf_batch_shard in libplf_hip.so (rgbd_pl_slam_amd/csrc/batch_host.hip), the same function the batch driver uses to cut a
host batch over its GPUs and bench.py uses to cut a job over its ranks.  Exercised here with world_size 2 on gloo (CPU); the
GPU side of the driver is covered by tests/test_gpu_batch.py."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rgbd_pl_slam_amd import _lib as L
from rgbd_pl_slam_amd.batch import shard


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard(37, world, rank)                                     # the product's partition (C ABI)
    owner = torch.zeros(37, dtype=torch.int64); owner[lo:hi] = 1        # frames this rank would extract
    elapsed = torch.tensor([0.25 + 0.5 * rank], dtype=torch.float64)   # pretend per-rank wall time
    dist.barrier()
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    dist.all_reduce(owner, op=dist.ReduceOp.SUM)                        # test-only reduction: every frame owned exactly once
    q.put((rank, lo, hi, float(elapsed[0]), owner.tolist()))
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
    assert all(r[4] == [1] * 37 for r in res)


def test_shard_covers_every_frame_once_and_is_balanced():
    for n in (0, 1, 7, 64, 257, 4096 * 8 + 3):
        for world in (1, 2, 3, 4, 8):
            seen = np.zeros(n, int)
            sizes = []
            prev_hi = 0
            for r in range(world):
                lo, hi = shard(n, world, r)
                assert lo == prev_hi and hi >= lo                   # contiguous blocks in rank order
                prev_hi = hi
                seen[lo:hi] += 1
                sizes.append(hi - lo)
            assert prev_hi == n and np.all(seen == 1)
            assert max(sizes) - min(sizes) <= 1
    assert shard(64, 8, 3) == (24, 32)                              # BASELINE config 4: 64 frames over 8 GPUs, 8 each


def test_shard_rejects_bad_arguments():
    lib = L.lib()
    a, b = C.c_int64(), C.c_int64()
    assert lib.plf_batch_shard(C.c_int64(10), 0, 0, C.byref(a), C.byref(b)) == L.PLF_E_BADARG
    assert lib.plf_batch_shard(C.c_int64(10), 2, 2, C.byref(a), C.byref(b)) == L.PLF_E_BADARG
    assert lib.plf_batch_shard(C.c_int64(-1), 2, 0, C.byref(a), C.byref(b)) == L.PLF_E_BADARG
    assert lib.plf_batch_shard(C.c_int64(10), 2, 0, None, C.byref(b)) == L.PLF_E_BADARG


def test_batch_driver_refuses_to_run_without_a_gpu():
    from conftest import gpu_available
    if gpu_available():
        pytest.skip("a GPU is visible")
    from rgbd_pl_slam_amd.batch import BatchExtractor
    with pytest.raises(L.PlfError) as e:
        BatchExtractor()
    assert e.value.status == L.PLF_E_HIP                            # no CPU path behind the driver either
