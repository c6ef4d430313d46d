"""Sample ingest / framing front-end and UDP frame emitter (row N2).

Stands where the XADC + acquisition sequencer stand (imp/dsp_system_top.vhd:412-435,
imp/sequencer_dsp.vhd): a continuous int16 sample stream is cut into 16384-sample frames and moved
host -> device through two pinned staging buffers so the copy of batch k+1 overlaps the processing of
batch k.  This is where PCIe (63 GB/s, ~0.48 M frames/s of int16... 1.9 M) rather than HBM becomes the
limit; the benchmark keeps inputs resident in HBM and never includes this stage.
"""
from __future__ import annotations

import socket
from typing import Iterable, Iterator, Optional

import numpy as np
import torch

from . import frames

N = frames.FFT_SIZE


class FrameCutter:
    """Cut an arbitrary sequence of int16 sample blocks into frames of 16384 samples.

    ``hop`` < 16384 gives overlapping frames (hop = 16384 reproduces the FPGA: back-to-back
    acquisitions, no overlap).  Samples are kept as delivered (the XADC delivers 12-bit values
    sign-extended to int16, imp/dsp_system_top.vhd:435)."""

    def __init__(self, hop: int = N):
        if not 0 < hop <= N:
            raise ValueError("hop must be in 1..16384")
        self.hop = hop
        self._buf = np.empty(0, np.int16)

    def push(self, samples) -> np.ndarray:
        """Append samples; return the frames that became complete, shape [k, 16384] (k may be 0)."""
        s = np.asarray(samples)
        if s.dtype != np.int16:
            if np.any(s < -32768) or np.any(s > 32767):
                raise ValueError("samples do not fit int16")
            s = s.astype(np.int16)
        self._buf = np.concatenate([self._buf, s.reshape(-1)])
        n = self._buf.size
        if n < N:
            return np.empty((0, N), np.int16)
        k = (n - N) // self.hop + 1
        idx = np.arange(k)[:, None] * self.hop + np.arange(N)[None, :]
        out = self._buf[idx]
        self._buf = self._buf[k * self.hop:]
        return out

    @property
    def pending(self) -> int:
        return int(self._buf.size)


class DeviceFeeder:
    """Double-buffered host -> device mover: ``feed(batch_iter)`` yields device tensors [B,16384] int16
    while the next batch is already in flight on a side stream.

    Two pinned staging buffers, two device buffers, and FOUR events created once (a "copied" and a
    "consumed" event per slot).  Measured (profiles/r2_ingest.txt): 1.42 M frames/s = 46.5 GB/s end to end
    against 55 GB/s for the bare pinned copy, i.e. PCIe-bound, with the copy of batch k+1 under the kernels of
    batch k.  The staging copy numpy -> pinned runs on torch's intra-op thread pool: on a host that exposes
    more cores than the job may use (a 16-CPU share of a 256-core box) the default pool stalls it for 50-100 ms
    every dozen batches -- pass ``host_threads`` (or call torch.set_num_threads yourself)."""

    def __init__(self, device: torch.device | int = 0, max_batch: int = 256, host_threads: Optional[int] = None):
        if host_threads is not None:
            torch.set_num_threads(int(host_threads))
        self.device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        self.max_batch = max_batch
        self._pinned = [torch.empty((max_batch, N), dtype=torch.int16).pin_memory() for _ in range(2)]
        self._dev = [torch.empty((max_batch, N), dtype=torch.int16, device=self.device) for _ in range(2)]
        self._copy_stream = torch.cuda.Stream(self.device)
        self._copied = [torch.cuda.Event() for _ in range(2)]      # slot's host->device copy has run
        self._consumed = [torch.cuda.Event() for _ in range(2)]    # slot's consumer work has been enqueued and run
        self._used = [False, False]

    def feed(self, batches: Iterable[np.ndarray]) -> Iterator[torch.Tensor]:
        cur = torch.cuda.current_stream(self.device)
        pend = None                                             # (slot, n_frames) handed out next
        for i, b in enumerate(batches):
            b = np.ascontiguousarray(b, np.int16).reshape(-1, N)
            n = b.shape[0]
            if n > self.max_batch:
                raise ValueError("batch larger than max_batch")
            slot = i & 1
            if self._used[slot]:
                self._consumed[slot].synchronize()              # the slot's previous consumer is done with it
            self._pinned[slot][:n].copy_(torch.from_numpy(b))
            with torch.cuda.stream(self._copy_stream):
                self._dev[slot][:n].copy_(self._pinned[slot][:n], non_blocking=True)
                self._copied[slot].record(self._copy_stream)
            if pend is not None:
                ps, pn = pend
                yield self._dev[ps][:pn]                        # the consumer enqueues its work on `cur` here
                self._consumed[ps].record(cur)
                self._used[ps] = True
            cur.wait_event(self._copied[slot])
            pend = (slot, n)
        if pend is not None:
            ps, pn = pend
            yield self._dev[ps][:pn]
            self._consumed[ps].record(cur)
            self._used[ps] = True


def udp_emit(frame_bytes: bytes, addr: tuple[str, int], sock: Optional[socket.socket] = None,
             src_port: Optional[int] = None) -> int:
    """Send one 65536-byte frame as the FPGA MAC does: 64 datagrams of 1 index byte + 1024 data bytes
    (gui.py:48-50, 318-339; the board sends from port 5005 to port 6006, imp/head_data.mif:27-38).
    Returns the number of datagrams sent."""
    own = sock is None
    if own:
        sock = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
        if src_port is not None:
            sock.bind(("", src_port))
    try:
        n = 0
        for p in frames.frame_to_udp_payloads(frame_bytes):
            sock.sendto(p, addr)
            n += 1
        return n
    finally:
        if own:
            sock.close()
