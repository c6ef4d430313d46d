#!/usr/bin/env python3
"""N3 measured: host int16 sample stream -> FrameCutter -> pinned double buffer (DeviceFeeder) -> Q15 path.

Reports frames/s end to end and the host->device rate against the PCIe bound DESIGN.md quotes (63 GB/s spec:
1.97 M int16 frames/s), for (a) numpy batches copied into the pinned staging buffers by the feeder (what a
socket / file reader delivers) and (b) the same with the host copy taken out (samples produced in pinned
memory).  With --events it also prints, per batch, the copy and kernel intervals measured with HIP events on
their own streams, which shows the copy of batch k+1 running under the kernels of batch k.
--float: the same int16 batches into the FLOAT chain (sa_process_f32_i16, mode 0xA1 with the headline cascade): no
conversion pass on the device, the PCIe volume of the Q15 path.
usage: ingest_bench.py [batch_frames] [n_batches] [mode] [--events] [--float]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fpga_real_time_fft_analyzer_amd.chain import SpectrumChain  # noqa: E402
from fpga_real_time_fft_analyzer_amd.ingest import DeviceFeeder  # noqa: E402

# the box gives this job 16 CPUs of a 256-core host: torch's default intra-op pool (one thread per visible core)
# stalls the staging copy for 50-100 ms every dozen batches
torch.set_num_threads(min(8, len(os.sched_getaffinity(0))))
N = 16384
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 32
mode = int(sys.argv[3], 0) if len(sys.argv) > 3 and not sys.argv[3].startswith("--") else 0xB1
FLOAT = "--float" in sys.argv

ch = SpectrumChain(0)
if FLOAT:
    ch.load_sos(np.load(os.path.join(ROOT, "tests", "golden", "g2_config1.npz"))["sos"])
ch.set_filter_mode(mode)
ch.reserve(B)
rng = np.random.default_rng(0)
host = [rng.integers(-2048, 2048, size=(B, N), dtype=np.int16) for _ in range(4)]       # 4 distinct batches, reused
out = [torch.empty((B, N) if FLOAT else (B, N, 2), dtype=torch.float32 if FLOAT else torch.int16, device="cuda")
       for _ in range(2)]
feeder = DeviceFeeder(0, max_batch=B)


def process(xd, o):
    if FLOAT:
        ch.process_f32(xd, out=o)          # int16 tensor in: sa_process_f32_i16
    else:
        ch.process_q15(xd, out=o)


def run(batches):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i, xd in enumerate(feeder.feed(batches)):
        process(xd, out[i & 1])
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def pure_h2d():
    pin = torch.empty((B, N), dtype=torch.int16).pin_memory()
    dev = torch.empty((B, N), dtype=torch.int16, device="cuda")
    for _ in range(3):
        dev.copy_(pin, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        dev.copy_(pin, non_blocking=True)
    torch.cuda.synchronize()
    return 20 * B * N * 2 / (time.perf_counter() - t0)


def pure_kernels():
    xd = torch.from_numpy(host[0]).cuda()
    for _ in range(3):
        process(xd, out[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        process(xd, out[0])
    torch.cuda.synchronize()
    return 20 * B / (time.perf_counter() - t0)


run(host[i & 3] for i in range(4))                        # warm-up
dt = run(host[i & 3] for i in range(NB))
h2d = pure_h2d()
kfps = pure_kernels()
print(f"batch {B} frames x {NB} batches, filter mode 0x{mode:02X}")
print(f"  pinned host -> device copy alone      : {h2d / 1e9:6.1f} GB/s = {h2d / (N * 2) / 1e6:5.2f} M frames/s   (PCIe Gen5 x16 spec 63 GB/s = 1.92 M frames/s)")
print(f"  {'float chain from int16' if FLOAT else 'Q15 kernels'} alone, inputs resident: {kfps / 1e6:5.2f} M frames/s")
print(f"  feeder end to end (numpy -> pinned -> device -> path): {NB * B / dt / 1e6:5.2f} M frames/s = {NB * B * N * 2 / dt / 1e9:5.1f} GB/s of samples")
# the host copy into the staging buffer is part of the feeder; how much of the time is it?
t0 = time.perf_counter()
for i in range(8):
    feeder._pinned[i & 1][:B].copy_(torch.from_numpy(host[i & 3]))
tcopy = (time.perf_counter() - t0) / 8
print(f"  host copy numpy -> pinned staging     : {tcopy * 1e3:.2f} ms per batch (torch copy_, multi-threaded)")

if "--events" in sys.argv:
    # per-batch intervals on the copy stream and on the compute stream
    cs = feeder._copy_stream
    ev = []
    torch.cuda.synchronize()
    base = torch.cuda.Event(enable_timing=True)
    base.record()
    for i in range(6):
        slot = i & 1
        feeder._pinned[slot][:B].copy_(torch.from_numpy(host[i & 3]))
        c0, c1, k0, k1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
        with torch.cuda.stream(cs):
            c0.record(cs)
            feeder._dev[slot][:B].copy_(feeder._pinned[slot][:B], non_blocking=True)
            c1.record(cs)
        torch.cuda.current_stream().wait_event(c1)
        k0.record()
        process(feeder._dev[slot][:B], out[slot])
        k1.record()
        ev.append((c0, c1, k0, k1))
    torch.cuda.synchronize()
    print("  batch   copy [ms from start]        kernels [ms from start]")
    for i, (c0, c1, k0, k1) in enumerate(ev):
        print(f"   {i}     {base.elapsed_time(c0):7.3f} .. {base.elapsed_time(c1):7.3f}      {base.elapsed_time(k0):7.3f} .. {base.elapsed_time(k1):7.3f}")
ch.close()
