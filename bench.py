#!/usr/bin/env python3
"""Benchmark of the hot path: 16K-point frames/s through window + IIR + FFT (+ magnitude).

Contract: ``python bench.py --gpus N --steps K --warmup W`` (N>1 is launched by the driver with
torch.distributed.run, one rank per GPU).  One step = one pass of the fused float chain over one
batch of 4096 synthetic frames resident in HBM (BASELINE.json configs[2], the configuration the
metric is quoted on).  Frames are independent, so ranks shard the batch dimension with no
data-path collective (weak scaling: 4096 frames per GPU); the only inter-rank traffic is the
timing barrier / max-reduce, done over gloo on the host.

Prints ONE JSON line on rank 0 with the driver's keys plus ``roofline`` and ``cpu_baseline``.  At N = 1 the line also
carries ``configs``: the other single-GPU configurations of BASELINE.json (configs[1]: B = 256 fp32, IIR bypassed;
configs[3]: B = 4096 Q15), each after the same conditioning as the headline, each with its own algorithmic bytes
(131 072 / 98 304 B per frame) -- bounded to about three seconds; they are reported beside the headline and are never
its ``value``.  ``host_us_per_call`` is the host time of one process call (Python + ctypes + launch), the only cost a
batch-sharded run adds per rank.  Per-kernel times come from the launches' OWN start / stop events
(``sa_set_profiling``, include/specan.h): they ride on the dispatch packets, so a kernel time cannot exceed its step.

Headline mode: overlapped launches (``sa_set_overlap``, include/specan.h) -- ``--overlap D`` (default 2) keeps D
launches of the handle in flight, so the tail of one batch runs under the head of the next (frames are
independent: every frame starts from a zero filter state, SURVEY quirk Q5 / new/filter_iir_cust.vhd:142-146).  The line says so
(``launches_in_flight``), its ``roofline`` is computed from the WALL time per step of the timed region (kernels
overlap, so per-kernel event durations no longer add up), and the strictly stream-ordered figures -- wall and
per-kernel HIP events, the number a ``rocprofv3 --kernel-trace`` of ``--overlap 1`` reproduces -- are kept beside
it under ``ordered``.  ``--overlap 1`` makes the ordered mode the headline.

Two ways to get N ranks (one process per GPU, BASELINE.json configs[4]):
  * under ``torch.distributed.run`` (WORLD_SIZE in the environment): this process IS one rank;
  * plain ``python bench.py --gpus N``: this process becomes a launcher that starts N fresh rank
    processes (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set, seeds 10..10+N-1) before anything touches a GPU,
    relays rank 0's JSON line and exits non-zero unless that line reports ``n_gpus == N``.
A rank never re-executes itself; on failure it exits non-zero and the launcher stops the others.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N = 16384
BYTES_PER_FRAME_F32 = 2 * N * 4          # read 64 KiB + write 64 KiB magnitudes
BYTES_PER_FRAME_Q15 = N * 2 + N * 4      # read 32 KiB + write 64 KiB IQ
HBM_PEAK_GBS = 8000.0                    # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)


def synth_host(B: int, seed: int) -> np.ndarray:
    rng = np.random.default_rng(seed)
    n = np.arange(N)
    fb = rng.uniform(0.01, 0.45, size=B)
    return (0.8 * np.sin(2 * np.pi * fb[:, None] * n[None, :]) + 0.05 * rng.standard_normal((B, N))).astype(np.float32)


def _cpu_slice(args):
    x, sos, hann = args
    from oracle import oracle as orc
    t0 = time.perf_counter()
    orc.cpu_baseline_chain(x, sos, hann)
    return time.perf_counter() - t0


def usable_cores() -> tuple:
    """(cores this process may actually use, note): the affinity mask capped by the cgroup CPU quota -- a GPU box
    shows all of the host's cores in the mask while the job owns a share of them."""
    import math
    aff = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())    # cgroup v1
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is None:
        return aff, f"{aff} cores in the affinity mask, no cgroup CPU quota"
    n = max(1, min(aff, int(math.ceil(quota))))
    return n, f"{aff} cores in the affinity mask, cgroup CPU quota {quota:g}"


def cpu_baseline(sos: np.ndarray) -> dict:
    """scipy/numpy chain on a bounded sample, single thread and all cores.
    Runs before anything touches the GPU (it forks a worker pool).  SA_BENCH_CPU_SECONDS bounds the all-core
    sample (default 10 s of aggregate CPU work; the CPU tests of the launcher use a fraction of a second)."""
    from multiprocessing import get_context
    from oracle import oracle as orc
    cores, cores_note = usable_cores()
    budget = float(os.environ.get("SA_BENCH_CPU_SECONDS", "10"))
    hann = orc.hann_f64().astype(np.float32)
    n_single = 256 if budget >= 5 else 16
    x1 = synth_host(n_single, seed=1)
    orc.cpu_baseline_chain(x1[:8], sos, hann)                # warm-up
    t0 = time.perf_counter()
    orc.cpu_baseline_chain(x1, sos, hann)
    t_single = time.perf_counter() - t0
    fps_single = x1.shape[0] / t_single
    # all cores: about `budget` seconds of aggregate CPU work, one 64-frame slice per task
    per_worker = max(64, int(fps_single * budget / 64) * 64 // max(cores, 1) // 64 * 64)
    nslices = cores * max(1, per_worker // 64)
    xs = synth_host(64, seed=2)
    with get_context("fork").Pool(cores) as pool:
        pool.map(_cpu_slice, [(xs, sos, hann)] * cores)      # warm-up
        t0 = time.perf_counter()
        pool.map(_cpu_slice, [(xs, sos, hann)] * nslices)
        t_all = time.perf_counter() - t0
    fps_all = nslices * 64 / t_all
    return {"value": round(fps_all, 1), "unit": "frames/s", "cores": cores, "kind": "port",
            "single_thread_frames_per_s": round(fps_single, 1),
            "sample": f"np.abs(np.fft.rfft(scipy.signal.sosfilt(sos, x*hann))) float32 in / float64 inside; "
                      f"single thread on {n_single} frames, {cores}-process pool on {nslices * 64} frames "
                      f"(64-frame slices), same synthetic distribution as the GPU run; {cores_note}"}


def _free_port() -> int:
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_ranks(n: int, argv: list[str]) -> int:
    """Launcher mode: start ``n`` rank processes of this script (one per GPU), relay rank 0's line.
    Nothing here imports torch or touches a GPU.  Returns the exit code for the launcher."""
    import subprocess
    # the CPU baseline of the same run, on the same host cores, BEFORE the ranks start (they would compete for the
    # cores): handed to rank 0 through the environment, which puts it into its line
    cpu_json = ""
    if "--no-cpu-baseline" not in argv:
        sos = np.load(os.path.join(ROOT, "tests", "golden", "g2_config1.npz"))["sos"]
        cpu_json = json.dumps(cpu_baseline(sos))
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SA_BENCH_CPU_JSON=cpu_json if r == 0 else "")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, rc, deadline = "", 0, time.time() + float(os.environ.get("SA_BENCH_LAUNCH_TIMEOUT", "1500"))
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            if procs[r].poll() is not None:
                pending.discard(r)
                if r == 0:
                    out0 = procs[0].stdout.read()
                if procs[r].returncode != 0:
                    rc = rc or procs[r].returncode or 1
        if rc or time.time() > deadline:           # one rank failed (or hung): stop the others, by PID
            for r in pending:
                procs[r].terminate()
            for r in pending:
                try:
                    procs[r].wait(timeout=20)
                except subprocess.TimeoutExpired:
                    procs[r].kill()
            rc = rc or 124
            break
        if pending:
            time.sleep(0.05)
    if not out0 and procs[0].stdout and not procs[0].stdout.closed:
        try:
            out0 = procs[0].stdout.read()
        except ValueError:
            pass
    line = None
    for ln in out0.splitlines():
        if ln.startswith("{"):
            try:
                line = json.loads(ln)
            except ValueError:
                continue
    if rc:
        print(f"bench.py launcher: a rank failed (exit code {rc})", file=sys.stderr)
        return rc
    if line is None or line.get("n_gpus") != n:
        print(f"bench.py launcher: asked for {n} ranks, rank 0 reported {None if line is None else line.get('n_gpus')}",
              file=sys.stderr)
        return 3
    print(json.dumps(line), flush=True)
    return 0


class StubDevice:
    """Control-plane rehearsal without a GPU (tests/test_sharding_gloo.py sets SA_BENCH_STUB=1): the device
    step is a short sleep, everything else (rank set-up, barrier, MAX-reduce, JSON line) is the real code.
    Its line says data = "stub" and carries no roofline; it is never a measurement."""

    def __init__(self, rank):
        self.rank = rank
        if os.environ.get("SA_BENCH_STUB_FAIL_RANK") == str(rank):     # failure-path rehearsal for the launcher test
            sys.exit(7)

    def step(self):
        time.sleep(0.002 * (1 + self.rank))

    def sync(self):
        pass

    def set_overlap(self, depth):
        pass

    def kernel_ms(self, steps):
        return [2.0 * (1 + self.rank)] * steps

    def host_us_per_call(self, n=8):
        return 10.0 * (1 + self.rank)

    def close(self):
        pass


class GpuWorkload:
    """One rank = one GPU, one handle, one stream; R rotating buffer pairs of B frames resident in HBM."""
    kernel_name = "chain_f32_kernel<6 sections, unit numerators, MAG_FULL>"

    def __init__(self, a, rank, local_rank, sos):
        import torch
        from fpga_real_time_fft_analyzer_amd.chain import SpectrumChain
        self.torch = torch
        # one rank per GPU; wraps around only when rehearsing N ranks on a box with fewer GPUs
        dev_index = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(dev_index)
        self.dev = dev = torch.device("cuda", dev_index)
        self.ch = ch = SpectrumChain(dev_index)
        ch.load_sos(sos)
        ch.set_filter_mode(0xA1)
        self.B = B = a.batch
        self.gen = gen = torch.Generator(device=dev).manual_seed(10 + rank)
        n = torch.arange(N, device=dev, dtype=torch.float32)
        # R distinct batches and output buffers, step i works on pair i mod R.  With R = 1 the 256 MiB input of
        # a 4096-frame batch survives in the 256 MB Infinity Cache between steps (the outputs are streaming
        # stores and do not displace it) and the step runs ~10 % faster than HBM can feed it.
        self.R = R = max(1, a.buffers)
        self.xs, self.outs = [], []
        for _ in range(R):
            fb = torch.rand(B, 1, generator=gen, device=dev) * 0.44 + 0.01
            self.xs.append((0.8 * torch.sin(2 * np.pi * fb * n)
                            + 0.05 * torch.randn(B, N, generator=gen, device=dev)).contiguous())
            self.outs.append(torch.empty((B, N), dtype=torch.float32, device=dev))
        self.step_no = 0

    def step(self):
        i = self.step_no % self.R
        self.step_no += 1
        self.ch.process_f32(self.xs[i], out=self.outs[i])

    def set_overlap(self, depth):
        """Launches of the handle kept in flight (include/specan.h, sa_set_overlap).  The rotating buffer pairs
        satisfy the mode's contract: a pair is reused R steps later, R > depth."""
        if depth > 1 and self.R <= depth:
            raise SystemExit(f"bench.py: --overlap {depth} needs more than {depth} buffer pairs (--buffers)")
        self.ch.set_overlap(depth)

    def sync(self):
        if self.ch.overlap > 1:
            self.ch.flush()                           # the current stream waits for the internal streams
        self.torch.cuda.synchronize(self.dev)

    def kernel_ms(self, steps):
        """Device time of each of `steps` back-to-back stream-ordered launches, from the launch's own start / stop events
        (sa_set_profiling: hipExtLaunchKernel binds them to the dispatch packet -- no marker packets, the train runs as
        it does untimed; the launch stream is torch's current stream of this device, handed to the C ABI)."""
        self.ch.set_profiling(steps)
        for _ in range(steps):
            self.step()
        ms = self.ch.profile_read(steps)
        self.ch.set_profiling(0)
        return ms

    def host_us_per_call(self, n=64):
        """Host time of one process call (Python wrapper + ctypes + launch) while the GPU is still busy with the calls
        before it: n calls in a row without a wait in between (n x 128 us of device work against well under a
        millisecond of host work, so the queue never fills)."""
        self.step()                   # (the first overlapped call after a change of mode probes its streams once: ~1 ms)
        self.sync()
        t0 = time.perf_counter()
        for _ in range(n):
            self.step()
        dt = time.perf_counter() - t0
        self.sync()
        return dt / n * 1e6

    def _measure(self, fn, frames, bytes_per_frame, steps, depth, timed_calls=True):
        """One configuration: `depth` launches in flight, 0.25 s of untimed conditioning (as the headline gets), then
        `steps` steps between two synchronisations (the wall figure, launch timing OFF: a timed launch carries a start
        event, which costs a small batch several microseconds per launch); in the ordered mode a second, shorter train
        with launch timing on gives the device time of a call."""
        ch = self.ch
        ch.set_overlap(depth)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.25:
            for _ in range(10):
                fn()
            self.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        self.sync()
        dt = (time.perf_counter() - t0) / steps
        res = {"frames_per_s": round(frames / dt, 1), "ms_per_step": round(dt * 1e3, 4),
               "achieved_GBps": round(frames * bytes_per_frame / dt / 1e9, 1),
               "frac": round(frames * bytes_per_frame / dt / 1e9 / HBM_PEAK_GBS, 4)}
        if depth == 1 and timed_calls:
            n = max(8, steps // 4)
            ch.set_profiling(n)
            for _ in range(n):
                fn()
            ms = ch.profile_read(n)
            ch.set_profiling(0)
            res["call_ms_avg"] = round(float(np.mean(ms)), 4)
            res["call_frac"] = round(frames * bytes_per_frame / (float(np.mean(ms)) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        ch.set_overlap(1)
        return res

    def configs(self, with_overlap=True):
        """The other single-GPU configurations of BASELINE.json, in the default line (the driver passes no flag).
        configs[1]: B = 256 fp32 frames, Hann + FFT + magnitude, IIR bypassed -- the board's power-on mode
        (new/command_control.vhd:31); 64 rotating 256-frame slices of the headline's buffers (1 GiB in, 1 GiB out: a
        handful of slices would live in the 256 MB Infinity Cache).  configs[3]: B = 4096 int16 frames, 12-bit samples,
        FPGA-exact window + default ALPHA / BETA cascade + SA-FXFFT-1, ordered and with two launches in flight."""
        torch, ch, dev, B, R, gen = self.torch, self.ch, self.dev, self.B, self.R, self.gen
        out = {}
        nb = 256
        slices = [(x[j:j + nb], o[j:j + nb]) for x, o in zip(self.xs, self.outs) for j in range(0, B - nb + 1, nb)]
        k = [0]

        def step256():
            x, o = slices[k[0] % len(slices)]
            k[0] += 1
            ch.process_f32(x, out=o)
        ch.set_filter_mode(0xB1)
        if slices:
            r = self._measure(step256, nb, BYTES_PER_FRAME_F32, 200, 1)
            r["bytes_per_frame"] = BYTES_PER_FRAME_F32
            r["workload"] = "BASELINE.json configs[1]: batch=256 x 16K fp32, Hann + 16K FFT + magnitude, IIR bypassed, stream-ordered"
            # (two 256-frame launches in flight do not help from Python: 15.3 us per step against 10.5 -- an overlapped call
            #  costs the host 14-18 us, more than the 10 us kernel it launches; a 512-frame batch is the way to fill the CUs)
            out["config2_b256_bypass"] = r
        # the same bypassed chain on the headline's batch (not a BASELINE configuration; SURVEY hypothesis H6 names the
        # bypassed chain as the place where 0.70 of the HBM roofline is realistic first)
        r = self._measure(self.step, B, BYTES_PER_FRAME_F32, 40, 1)
        r["bytes_per_frame"] = BYTES_PER_FRAME_F32
        r["workload"] = f"batch={B} x 16K fp32, Hann + 16K FFT + magnitude, IIR bypassed (the headline's buffers)"
        if R > 2 and with_overlap:
            r["overlap2"] = self._measure(self.step, B, BYTES_PER_FRAME_F32, 40, 2)
        out["bypass_b4096"] = r
        ch.set_filter_mode(0xA1)
        xqs = [torch.randint(-2048, 2048, (B, N), generator=gen, device=dev, dtype=torch.int32).to(torch.int16)
               for _ in range(R)]
        oqs = [torch.empty((B, N, 2), dtype=torch.int16, device=dev) for _ in range(R)]
        ch.reserve(B)

        def qstep():
            ch.process_q15(xqs[k[0] % R], out=oqs[k[0] % R])
            k[0] += 1
        ch.set_filter_mode(0x00)
        r = self._measure(qstep, B, BYTES_PER_FRAME_Q15, 30, 1)
        r["bytes_per_frame"] = BYTES_PER_FRAME_Q15
        r["workload"] = (f"BASELINE.json configs[3]: batch={B} x 16K int16 (12-bit samples), FPGA-exact window + default "
                         f"ALPHA/BETA cascade + fixed-point FFT, IQ frames out; call = cascade kernel + FFT kernel")
        if R > 2 and with_overlap:
            r["overlap2"] = self._measure(qstep, B, BYTES_PER_FRAME_Q15, 40, 2)
            ch.reserve(B)
        out["config4_q15_default"] = r
        ch.set_filter_mode(0xA1)
        del xqs, oqs
        return out

    def extras(self, steps, with_overlap=True):
        """Bypass (config 2) and Q15 (config 4) figures next to the headline; not the bench line's value.
        with_overlap = False (`--overlap 1`): every launch of the process is stream-ordered, so that a kernel trace of
        the run holds ordered launches only."""
        torch, ch, dev, B, R, gen = self.torch, self.ch, self.dev, self.B, self.R, self.gen
        out = {}

        def time_it(fn, k):
            for _ in range(3):
                fn()
            self.sync()
            t = time.perf_counter()
            for _ in range(k):
                fn()
            self.sync()
            return (time.perf_counter() - t) / k
        ch.set_overlap(1)
        ch.set_filter_mode(0xB1)
        x256 = self.xs[0][:256].contiguous()
        o256 = self.outs[0][:256]
        dt = time_it(lambda: ch.process_f32(x256, out=o256), 50)
        out["config2_bypass_b256"] = {"frames_per_s": 256 / dt, "GBps": 256 * BYTES_PER_FRAME_F32 / dt / 1e9}
        dt = time_it(self.step, steps)
        out["bypass_b4096"] = {"frames_per_s": B / dt, "GBps": B * BYTES_PER_FRAME_F32 / dt / 1e9}
        if R > 3 and with_overlap:   # the same with three launches in flight (the bypassed chain is not at the power cap)
            ch.set_overlap(3)
            dt = time_it(self.step, steps)
            out["bypass_b4096_overlap3"] = {"frames_per_s": B / dt, "GBps": B * BYTES_PER_FRAME_F32 / dt / 1e9}
            ch.set_overlap(1)
        ch.reserve(B)
        g4 = np.load(os.path.join(ROOT, "tests", "golden", "g4_q15_frames.npz"))
        ch.load_sos_q14(g4["sos_q14"])           # the Q2.14 quantisation of the headline's 12th-order Butterworth
        ch.load_coeffs_q7(g4["c_gui"])           # the GUI's default upload (B1 != 0: the nine-instruction step)
        for tag, lo, hi in (("", -2048, 2048), ("_fullscale", -32768, 32768)):
            xqs = [torch.randint(lo, hi, (B, N), generator=gen, device=dev, dtype=torch.int32).to(torch.int16)
                   for _ in range(R)]
            oqs = [torch.empty((B, N, 2), dtype=torch.int16, device=dev) for _ in range(R)]
            k = [0]

            def qstep():
                ch.process_q15(xqs[k[0] % R], out=oqs[k[0] % R])
                k[0] += 1
            for name, cmd in (("config4_q15_default_iir", 0x00), ("q15_bypass", 0xB1), ("q15_wide_6sec", 0xA2),
                              ("q15_gui_upload_9instr", 0xA1)):
                ch.set_filter_mode(cmd)
                dt = time_it(qstep, 5)
                out[name + tag] = {"frames_per_s": B / dt, "GBps": B * BYTES_PER_FRAME_Q15 / dt / 1e9}
            if R > 2 and with_overlap:   # the wide cascade (0xA2, all six designed sections in Q2.14) with two launches in flight
                ch.set_overlap(2)
                ch.reserve(B)
                ch.set_filter_mode(0xA2)
                dt = time_it(qstep, 12)
                out["q15_wide_6sec" + tag + "_overlap2"] = {"frames_per_s": B / dt, "GBps": B * BYTES_PER_FRAME_Q15 / dt / 1e9}
                ch.set_overlap(1)
            for depth in (2, 3):   # config 4 with launches in flight: the FFT of batch k under the filter of batch k+1
                if R <= depth or not with_overlap:
                    continue
                ch.set_overlap(depth)
                ch.reserve(B)
                ch.set_filter_mode(0x00)
                dt = time_it(qstep, 12)
                out["config4_q15_default_iir" + tag + f"_overlap{depth}"] = {"frames_per_s": B / dt,
                                                                            "GBps": B * BYTES_PER_FRAME_Q15 / dt / 1e9}
                ch.set_overlap(1)
            del xqs, oqs
        ch.set_filter_mode(0xA1)
        # the float chain fed with the ADC's int16 samples (sa_process_f32_i16: 32 KiB in + 64 KiB out per frame; the
        # same results as the float32 frames give, half the input bytes) -- not the workload BASELINE.json names
        xis = [torch.randint(-2048, 2048, (B, N), generator=gen, device=dev, dtype=torch.int32).to(torch.int16)
               for _ in range(R)]
        k = [0]

        def istep():
            ch.process_f32(xis[k[0] % R], out=self.outs[k[0] % R])
            k[0] += 1
        dt = time_it(istep, steps)
        out["float_chain_from_int16"] = {"frames_per_s": B / dt, "GBps": B * (32768 + 65536) / dt / 1e9}
        if R > 2 and with_overlap:
            ch.set_overlap(2)
            dt = time_it(istep, steps)
            out["float_chain_from_int16_overlap2"] = {"frames_per_s": B / dt, "GBps": B * (32768 + 65536) / dt / 1e9}
            ch.set_overlap(1)
        del xis
        return out

    def close(self):
        self.ch.close()


def _sysfs_card(torch, dev_index):
    """sysfs device directory of the card behind HIP device `dev_index` (matched by PCI address), or None."""
    import glob
    bdf = None
    try:
        p = torch.cuda.get_device_properties(dev_index)
        bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
    except Exception:                                                  # noqa: BLE001  (older torch: no PCI fields)
        pass
    amd = []
    for d in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        try:
            if open(os.path.join(d, "vendor")).read().strip() != "0x1002":
                continue
        except OSError:
            continue
        if bdf and os.path.basename(os.path.realpath(d)).lower() == bdf:
            return d
        amd.append(d)
    return amd[0] if (bdf is None and len(amd) == 1) else None


def _read_power_sysfs(card):
    """(socket W, cap W, sclk MHz) from hwmon / pp_dpm_sclk of one card; raises when the layout is not there."""
    import glob
    hw = glob.glob(os.path.join(card, "hwmon", "hwmon*"))[0]
    try:
        w = float(open(os.path.join(hw, "power1_average")).read()) / 1e6
    except OSError:
        w = float(open(os.path.join(hw, "power1_input")).read()) / 1e6
    cap = float(open(os.path.join(hw, "power1_cap")).read()) / 1e6
    mhz = None
    for ln in open(os.path.join(card, "pp_dpm_sclk")).read().splitlines():
        if ln.strip().endswith("*"):
            mhz = float(ln.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
    if mhz is None:
        raise OSError("no current level in pp_dpm_sclk")
    return w, cap, mhz


def power_sample(wl, dev_index, seconds):
    """Socket power and shader clock of the card while wl.step() runs back to back for `seconds` (untimed).  Read in
    process from sysfs (hwmon power1_average, pp_dpm_sclk).  Only when sysfs does not have them a `rocm-smi` child is
    tried -- never under a profiler (ROCP* / LD_PRELOAD in the environment: rocm-smi is a `#!/usr/bin/env python3`
    script, i.e. an exec hop inside a GPU-initialised, profiler-preloaded process tree), and with those variables
    stripped from the child's environment."""
    import subprocess
    import threading
    shots, stop = [], threading.Event()
    card = _sysfs_card(wl.torch, dev_index)
    source = ["sysfs hwmon power1_average / pp_dpm_sclk"]
    profiled = any(k.startswith(("ROCP", "ROCPROF")) or k == "LD_PRELOAD" for k in os.environ)

    def one_shot():
        if card is not None:
            try:
                return _read_power_sysfs(card)
            except (OSError, ValueError, IndexError):
                pass
        if profiled:
            return None
        env = {k: v for k, v in os.environ.items() if not (k.startswith(("ROCP", "ROCPROF", "HSA_TOOLS")) or k == "LD_PRELOAD")}
        r = subprocess.run(["rocm-smi", "-d", str(dev_index), "--showpower", "--showclocks", "--showmaxpower", "--json"],
                           capture_output=True, text=True, timeout=10, env=env)
        c = next(iter(json.loads(r.stdout).values()))
        source[0] = "rocm-smi"
        return (float(c["Current Socket Graphics Package Power (W)"]), float(c["Max Graphics Package Power (W)"]),
                float(c["sclk clock speed:"].strip("()").lower().replace("mhz", "")))

    def watch():
        while not stop.is_set():
            try:
                v = one_shot()
            except Exception:                                          # noqa: BLE001  (tool missing, other layout)
                return
            if v is None:
                return
            shots.append(v)
            stop.wait(0.1)
    th = threading.Thread(target=watch, daemon=True)
    th.start()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            wl.step()
        wl.sync()
    stop.set()
    th.join(timeout=15)
    shots = shots[2:] if len(shots) > 4 else shots                     # the first readings still average the ramp-up
    if not shots:
        return None
    w = sorted(x[0] for x in shots)
    f = sorted(x[2] for x in shots)
    return {"socket_w": round(w[len(w) // 2], 1), "cap_w": shots[0][1], "sclk_mhz": f[len(f) // 2], "samples": len(shots),
            "source": source[0] + " during an untimed repeat of the headline step loop"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096, help="frames per GPU per step")
    ap.add_argument("--buffers", type=int, default=4,
                    help="input/output buffer pairs used round-robin (one pair of 256 MiB + 256 MiB would stay in "
                         "the 256 MB Infinity Cache from step to step: the figure would be a cache figure, not HBM)")
    ap.add_argument("--overlap", type=int, default=2,
                    help="launches of the handle kept in flight in the timed region (sa_set_overlap; 1 = strictly "
                         "stream-ordered, which is also measured and reported under 'ordered')")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-power", action="store_true", help="skip the power / clock sample")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the `configs` block (BASELINE.json configs[1] and configs[3] beside the headline; N = 1 only)")
    ap.add_argument("--extras", action="store_true", help="also time the bypass (config 2) and Q15 (config 4) paths")
    a = ap.parse_args()
    if a.gpus < 1:
        ap.error("--gpus must be >= 1")
    if not 1 <= a.overlap <= 4:
        ap.error("--overlap must be 1..4")

    if "WORLD_SIZE" not in os.environ:
        if a.gpus > 1:                               # launcher mode: no GPU call has happened in this process
            sys.exit(launch_ranks(a.gpus, sys.argv[1:]))
    elif int(os.environ["WORLD_SIZE"]) != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: start one rank per GPU "
              f"(--nproc-per-node {a.gpus}) or drop WORLD_SIZE and let bench.py start the ranks", file=sys.stderr)
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    stub = os.environ.get("SA_BENCH_STUB") == "1"
    sos = np.load(os.path.join(ROOT, "tests", "golden", "g2_config1.npz"))["sos"]    # 12th-order Butterworth, wn = 0.2

    # CPU baseline beside every line, N > 1 included (north_star: "next to the scipy/numpy CPU baseline timed on
    # the same host cores in the same run"): from the launcher when it started the ranks, else measured here by
    # rank 0 before anything initialises the GPU (it forks; the other ranks wait in the rendezvous meanwhile)
    cpu = None
    if rank == 0 and not a.no_cpu_baseline:
        if os.environ.get("SA_BENCH_CPU_JSON"):
            cpu = json.loads(os.environ["SA_BENCH_CPU_JSON"])
        else:
            cpu = cpu_baseline(sos)

    import torch
    import torch.distributed as dist

    # Rank 0 prints ONE line on stdout.  Libraries underneath do not know that (gloo announces its connections on
    # stdout): file descriptor 1 points at stderr for the duration of the run and is restored for the JSON line only.
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)

    if world > 1:
        import datetime
        dist.init_process_group(backend="gloo", rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=600))
    B = a.batch
    R = max(1, a.buffers)
    if stub:
        wl = StubDevice(rank)
    else:
        wl = GpuWorkload(a, rank, local_rank, sos)   # raises if the HIP extension is missing

    def barrier():
        wl.sync()
        if world > 1:
            dist.barrier()

    from bench_shard import aggregate_fps, gather_floats

    def timed_region():
        """W untimed warm-up steps, then exactly K steps between barrier + synchronize on both sides; MAX over ranks."""
        for _ in range(a.warmup):
            wl.step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            wl.step()
        wl.sync()
        t1 = time.perf_counter()
        res = aggregate_fps(B, a.steps, t1 - t0, world)                 # MAX over ranks (gloo, host side)
        if world > 1:
            dist.barrier()
        return res

    # Untimed pre-warm: the chip leaves its idle power state only after ~100 ms of sustained load (the
    # same launches measured 8-10 % slower in the first milliseconds).  Then the W contract warm-up steps.
    wl.set_overlap(a.overlap)
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < (0.0 if stub else 0.25):
        for _ in range(10):
            wl.step()
        wl.sync()
    fps_total, elapsed = timed_region()                                  # the headline

    # strictly stream-ordered mode: wall (same contract) and per-launch kernel time with HIP events on the launch
    # stream, same number of steps -- what rocprofv3 --kernel-trace shows for `--overlap 1`
    wl.set_overlap(1)
    if a.overlap > 1:
        fps_ord, elapsed_ord = timed_region()
    else:
        fps_ord, elapsed_ord = fps_total, elapsed
    k_ms = sorted(wl.kernel_ms(a.steps))
    k_avg_ms = float(np.mean(k_ms))
    k_med_ms = float(k_ms[len(k_ms) // 2])
    per_rank_kernel_ms = gather_floats(k_avg_ms, world)                  # launch skew between GPUs, if any
    # host cost of one process call, ordered and in the headline mode: with independent frames and no collective it
    # is the only thing a rank adds per step, i.e. what weak scaling over GPUs depends on (MAX over ranks)
    host_us_ord = max(gather_floats(wl.host_us_per_call(), world))
    wl.set_overlap(a.overlap)
    host_us_head = max(gather_floats(wl.host_us_per_call(), world)) if a.overlap > 1 else host_us_ord
    wl.set_overlap(1)

    # the other single-GPU configurations of BASELINE.json, beside the headline in the default line, BEFORE the power
    # sample (2.5 s at the cap leave the chip hotter than the headline found it)
    configs = wl.configs(a.overlap > 1) if (world == 1 and rank == 0 and not stub and not a.no_configs) else {}

    # Package power and shader clock while the headline mode runs (untimed repeat of the step loop, rank 0's card only):
    # the launch sits at the power cap and the clock is what gives (profiles/r3_power_clock.txt), which is what bounds
    # the roofline fraction of this arithmetic.  A side thread reads sysfs (hwmon); null when it is not there.
    power = None
    if rank == 0 and not stub and not a.no_power:
        wl.set_overlap(a.overlap)
        power = power_sample(wl, dev_index=wl.dev.index or 0, seconds=2.5)
        wl.set_overlap(1)
        if power:                        # energy of one step at that power: what a power-limited launch is made of
            power["mj_per_step"] = round(power["socket_w"] * elapsed / a.steps * 1e3, 1)

    extras = wl.extras(a.steps, a.overlap > 1) if (a.extras and rank == 0 and not stub) else {}

    if rank == 0:
        # HBM traffic per launch: FETCH_SIZE (doubled, gfx950) + WRITE_SIZE from the committed separate
        # --pmc passes over this same kernel and batch (profiles/, tools/pmc_profile.sh).  Counters cannot be
        # read inside the timed process, so this is a constant quoted from that profile -- `traffic_source`
        # says which -- and null when the profile was taken at another batch size.
        traffic, traffic_src, valu_busy = None, None, None
        for name in ("r4_pmc_traffic.json", "r3_pmc_traffic.json"):   # regenerated with the round's kernel (tools/pmc_profile.sh)
            try:
                with open(os.path.join(ROOT, "profiles", name)) as fh:
                    pj = json.load(fh)
                if pj.get("batch") == B:
                    traffic, traffic_src = pj["traffic_bytes_per_launch"], f"profiles/{name}"
                    valu_busy = pj.get("valu_busy_frac")
                    break
            except (OSError, ValueError, KeyError):
                continue
        ms_step = elapsed / a.steps * 1e3
        achieved_ord = B * BYTES_PER_FRAME_F32 / (k_avg_ms * 1e-3) / 1e9
        # overlap mode: kernels run beside each other, so bytes / wall time per step (per GPU) is the achieved rate
        achieved = B * BYTES_PER_FRAME_F32 / (ms_step * 1e-3) / 1e9 if a.overlap > 1 else achieved_ord
        line = {
            "metric": "16K-pt frames/sec (window+IIR+FFT), batch=4096",
            "value": round(fps_total, 1),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 4),
            "launches_in_flight": a.overlap,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "stub (no device work: control-plane rehearsal)" if stub else "synthetic",
            "config": {"workload": f"batch={B}x16K fp32 frames per GPU, Hann + 6-biquad IIR (12th-order Butterworth "
                                   f"wn=0.2) + 16K FFT + magnitude, all 16384 bins written (BASELINE.json configs[2]"
                                   f"{'; x' + str(world) + ' GPUs = configs[4]' if world > 1 else ''})",
                       "frames_per_gpu": B, "sharding": "batch, independent per-GPU streams, no collective",
                       "buffer_pairs": R, "seeds": [10 + r for r in range(world)],
                       "launch_mode": (f"sa_set_overlap({a.overlap}): results of a call visible after the next "
                                       f"{a.overlap - 1} call(s) or sa_flush" if a.overlap > 1 else "stream-ordered")},
            "per_rank_kernel_ms": [round(v, 4) for v in per_rank_kernel_ms],
            "host_us_per_call": {"ordered": round(host_us_ord, 2), "headline_mode": round(host_us_head, 2),
                                 "note": "host time of one process call (Python + ctypes + launch), MAX over ranks; the "
                                         "only per-step cost a rank adds: frames are independent, no collective"},
        }
        if not stub:
            line["roofline"] = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                                "traffic_source": traffic_src,
                                "basis": (f"wall ms_per_step of the timed region, {a.overlap} launches in flight "
                                          f"(per-kernel durations overlap)" if a.overlap > 1
                                          else "the launches' own start / stop events (sa_set_profiling)"),
                                "kernel": wl.kernel_name, "kernel_ms_avg": round(k_avg_ms, 4),
                                "kernel_ms_median": round(k_med_ms, 4),
                                "algorithmic_bytes_per_launch": B * BYTES_PER_FRAME_F32,
                                # the second ceiling: share of an ordered launch a SIMD's vector pipe is
                                # executing, from the same committed counter passes as `traffic` (quoted, not live)
                                "secondary": {"bound": "valu_fp32", "busy_frac": valu_busy, "source": traffic_src},
                                "power": power}
            line["ordered"] = {"value": round(fps_ord, 1), "ms_per_step": round(elapsed_ord / a.steps * 1e3, 4),
                               "kernel_ms_avg": round(k_avg_ms, 4), "achieved": round(achieved_ord, 1),
                               "frac": round(achieved_ord / HBM_PEAK_GBS, 4),
                               "basis": "strictly stream-ordered launches; kernel time = the launch's own start / stop "
                                        "events on its dispatch packet (sa_set_profiling), as rocprofv3 --kernel-trace sees it"}
            if configs:
                line["configs"] = configs
        if cpu is not None:
            line["cpu_baseline"] = cpu
        if extras:
            line["extras"] = extras
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    wl.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
