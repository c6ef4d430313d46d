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


def _run_bench(args, env_extra, timeout=300):
    import json
    import subprocess
    env = dict(os.environ, SA_BENCH_STUB="1", SA_BENCH_CPU_SECONDS="0.3", **env_extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        if k not in env_extra:
            env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                       text=True, timeout=timeout)
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p.returncode, lines, p.stderr


def test_bench_gpus_flag_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no torchrun around it (BASELINE.json configs[4] launched from the
    bench itself): the launcher starts two fresh rank processes, rank 0 reports n_gpus == 2 and the
    aggregate is 2 x 4096 frames per step over the MAX of the ranks' times.  The device step is stubbed
    (SA_BENCH_STUB=1: no GPU here); rank set-up, gloo barrier, MAX-reduce and the JSON line are the real code."""
    rc, lines, err = _run_bench(["--gpus", "2", "--steps", "4", "--warmup", "1"], {})
    assert rc == 0, err
    assert len(lines) == 1
    line = lines[0]
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["scaling"] == "weak"
    assert line["config"]["seeds"] == [10, 11] and line["config"]["frames_per_gpu"] == 4096
    # the stub's rank 1 sleeps twice as long as rank 0: the timed region must be rank 1's (MAX over ranks)
    assert line["ms_per_step"] >= 4.0
    assert abs(line["value"] - 2 * 4096 / (line["ms_per_step"] * 1e-3)) <= 1e-3 * line["value"]
    assert "roofline" not in line and line["data"].startswith("stub")
    # the CPU baseline of the same run sits beside the N > 1 line too (measured by the launcher before the ranks
    # start, on the same host cores), and rank 0 reports every rank's kernel time (launch skew between GPUs)
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "frames/s" and cb["value"] > 0 and cb["cores"] >= 1
    assert line["per_rank_kernel_ms"] == [2.0, 4.0]
    assert line["launches_in_flight"] == 2
    # the host cost of a process call is the MAX over ranks; the measured blocks of a GPU line (the other BASELINE
    # configurations, the roofline, the ordered-mode figures) never appear in a stub line
    assert line["host_us_per_call"]["ordered"] == 20.0 and line["host_us_per_call"]["headline_mode"] == 20.0
    assert not {"configs", "roofline", "ordered", "extras"} & set(line)


def test_bench_under_torchrun_env_and_mismatch():
    """The torchrun contract keeps working (one rank per process, env-driven), and a --gpus / WORLD_SIZE
    mismatch is an error instead of a silent n_gpus = 1."""
    rc, lines, err = _run_bench(["--gpus", "1", "--steps", "2", "--warmup", "0"], {})
    assert rc == 0 and lines[0]["n_gpus"] == 1, err
    assert "configs" not in lines[0] and lines[0]["host_us_per_call"]["ordered"] == 10.0
    rc, lines, err = _run_bench(["--gpus", "2", "--steps", "2"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc == 2 and not lines and "WORLD_SIZE" in err
    port = _free_port()
    import subprocess
    envs = [dict(os.environ, SA_BENCH_STUB="1", SA_BENCH_CPU_SECONDS="0.3", RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2",
                 MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)) for r in range(2)]
    ps = [subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0"],
                           env=e, stdout=subprocess.PIPE, text=True) for e in envs]
    outs = [p.communicate(timeout=300)[0] for p in ps]
    assert all(p.returncode == 0 for p in ps)
    assert '"n_gpus": 2' in outs[0] and "{" not in outs[1]              # only rank 0 prints the JSON line
    assert '"cpu_baseline"' in outs[0] and '"per_rank_kernel_ms": [2.0, 4.0]' in outs[0]    # rank 0 measured it itself
    # ... and nothing else reaches stdout (gloo announces its connections there unless the bench keeps it away)
    assert len(outs[0].strip().splitlines()) == 1 and outs[1].strip() == ""


def test_bench_launcher_fails_when_a_rank_fails():
    """A rank that dies takes the launch down with a non-zero exit code instead of leaving the others in the barrier."""
    rc, lines, err = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "0"], {"SA_BENCH_STUB_FAIL_RANK": "1"})
    assert rc != 0 and not lines
