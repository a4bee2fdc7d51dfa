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


def _pcie_worker(rank, world, port, q):
    """`bench.py --gpus 2 --pcie` without a GPU: the rank logic (barrier, MAX over the ranks, JSON assembly on rank 0) runs unchanged on gloo; only the GPU leg
    (plf_batch_extract on pinned host frames) and the local map are stubs"""
    import argparse
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def stub_pcie(device, w, h, nfeat, nlines, n_frames, in_flight, mp, ml, reps=2, warm=1, dist=None):
        total = (1.0 + 1.0 * device) * reps                      # rank 1 is twice as slow: the job lasts as long as rank 1
        dist.barrier()
        return {"value": round(n_frames / (total / reps), 1), "unit": "frames/s", "frames": n_frames, "frames_in_flight": in_flight,
                "elapsed_max": bench.reduce_elapsed_max(dist, total, device="cpu"), "worker_numa_node": -1, "worker_cpus_bound": 0}

    args = argparse.Namespace(steps=3, warmup=1, config=2)
    W, H, NFEAT, NLINES, B0, label = bench.CONFIGS[2]
    line = bench.run_pcie(args, dist, rank, rank, world, B0, W, H, NFEAT, NLINES, label, pcie_fn=stub_pcie, local_map_fn=lambda: (None, None))
    q.put((rank, line))
    dist.destroy_process_group()


def test_world2_pcie_branch_json_and_elapsed_max():
    """the --pcie branch had never executed with world > 1"""
    import json
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_pcie_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(2))
    for p in ps:
        p.join(timeout=60)
    assert res[1] is None                                   # only rank 0 prints
    line = res[0]
    json.dumps(line)                                        # serialisable as it stands
    in_flight = 4096; n_frames = 4 * in_flight
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak" and line["higher_is_better"] is True
    # slowest rank: 2.0 s per call x 3 calls = 6.0 s for 2 ranks x 3 calls x n_frames frames
    assert line["ms_per_step"] == 2000.0 and line["value"] == round(2 * 3 * n_frames / 6.0, 2)
    assert line["config"]["frames_per_call_per_gpu"] == n_frames and line["config"]["frames_in_flight_per_gpu"] == in_flight
    assert "elapsed_max" not in line["pcie"]["rank0"] and line["pcie"]["host_read_GBps_per_gpu"] * 2 == pytest.approx(line["pcie"]["host_read_GBps_aggregate"], abs=0.02)


def test_pcie_line_and_reduction_single_rank():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    assert bench.reduce_elapsed_max(None, 1.25) == 1.25
    W, H, NFEAT, NLINES, B0, label = bench.CONFIGS[2]
    line = bench.pcie_line({"value": 1.0, "elapsed_max": 4.0}, 4.0, 1, 16384, 4096, 2, 1, W, H, label, 2)
    assert line["value"] == round(16384 * 2 / 4.0, 2) and line["n_gpus"] == 1 and "elapsed_max" not in line["pcie"]["rank0"]


class _StubPipe:
    """what bench.run_headline touches of bench.Pipeline -- the rank logic (partition, barrier, MAX over the ranks, JSON assembly) runs unchanged without a GPU"""
    def __init__(self, nb):
        self.nb = nb; self.ndist = nb
    def matches_frame0(self):
        return {"points_map": 0}
    def close(self):
        pass


def _headline_worker(rank, world, port, q):
    """`bench.py --gpus 8 --config 4` (BASELINE configs[3]: 64 frames of 1280x960 over 8 GPUs) on gloo: every rank must get shard(64, 8, r) = 8 frames, the job lasts as
    long as the slowest rank, rank 0 assembles the line"""
    import argparse
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    made = []

    def make_pipe(nb):
        made.append(nb)
        return _StubPipe(nb)

    def stub_timed(pipe, steps, warmup, d):
        d.barrier()
        return 0.5 + 0.25 * rank, 10.0 * steps, steps      # rank 7 is the slowest: 2.25 s for the 5 steps

    args = argparse.Namespace(steps=5, warmup=1, config=4, batch=0)
    W, H, NFEAT, NLINES, B0, label = bench.CONFIGS[4]
    out, pipe, fps, elapsed, B = bench.run_headline(args, dist, rank, rank, world, B0, W, H, NFEAT, NLINES, label, make_pipe, timed_fn=stub_timed, reduce_device="cpu")
    q.put((rank, out, made, B, elapsed, tuple(shard(world * B0, world, rank))))
    dist.destroy_process_group()


def test_world8_config4_headline_sharding_and_line():
    """configs[3] exactly as written -- batch 64 over 8 GPUs, 8 frames each -- through bench.py's own rank logic (world 8 on gloo)"""
    import json
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_headline_worker, args=(r, 8, port, q)) for r in range(8)]
    for p in ps:
        p.start()
    res = {r[0]: r for r in (q.get(timeout=300) for _ in range(8))}
    for p in ps:
        p.join(timeout=60)
    for r in range(8):
        rank, out, made, B, elapsed, blk = res[r]
        assert made == [8] and B == 8 and blk == (8 * r, 8 * r + 8)          # 64 frames, 8 contiguous per rank
        assert elapsed == 0.5 + 0.25 * 7                                      # MAX over the ranks
        assert (out is None) == (r != 0)                                      # only rank 0 assembles the line
    line = res[0][1]
    json.dumps(line)
    assert line["n_gpus"] == 8 and line["scaling"] == "weak" and line["steps"] == 5 and line["frames_of_this_rank"] == [0, 8]
    assert line["value"] == round(8 * 8 * 5 / 2.25, 2) and line["ms_per_step"] == 450.0
    assert line["config"]["baseline_config"] == 4 and line["config"]["frames_in_flight_per_gpu"] == 8 and "1280x960" in line["metric"]
    assert line["roofline"]["avg_launch_ms"] == 10.0 and 0 < line["roofline"]["pipeline_frac"] < 1
