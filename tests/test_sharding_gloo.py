"""CPU: the multi-GPU contract of bench.py on a world-size-2 gloo group.

Frames shard along the batch dimension with no data-path collective; the only
inter-rank traffic is the timing barrier and the max-reduce.  This test runs the control plane of
bench.py (partitioning, barrier, MAX all-reduce, aggregate throughput) with the device work
replaced by a sleep, on two processes.
"""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import time

    from bench_shard import aggregate_fps, shard_frames
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = shard_frames(total=8192, rank=rank, world=world)          # strong-style partition helper
    dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.05 * (rank + 1))                                      # rank 1 is the slow one
    elapsed = time.perf_counter() - t0
    fps, tmax = aggregate_fps(frames_per_rank=4096, steps=1, elapsed=elapsed, world=world)
    q.put((rank, frames, fps, tmax))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_timing():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, f0, fps0, t0), (r1, f1, fps1, t1) = got
    assert f0 == (0, 4096) and f1 == (4096, 8192)                      # disjoint, covering
    assert abs(t0 - t1) < 1e-9 and t0 >= 0.1                           # both see the MAX over ranks
    assert abs(fps0 - 8192 / t0) < 1e-6 * fps0 and fps0 == fps1        # whole-job aggregate


def test_shard_helper_edges():
    from bench_shard import shard_frames
    assert shard_frames(10, 0, 3) == (0, 4) and shard_frames(10, 1, 3) == (4, 7) and shard_frames(10, 2, 3) == (7, 10)
    assert shard_frames(0, 0, 2) == (0, 0)
    assert shard_frames(4096, 0, 1) == (0, 4096)
