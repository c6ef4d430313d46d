"""Virtual FPGA: the board as `scripts/fft_analyzer_gui.py` sees it, backed by the MI355X path.

Stands where the serial port / UDP socket of `UartReceiver` / `UdpReceiver` stand (gui.py:355-747):
the host writes command bytes, the device answers with 65536-byte frames (UART mode: raw, after a
0xA5 request, gui.py:24-29; Ethernet mode: 64 datagram payloads of 1 index byte + 1024 data bytes,
gui.py:48-50, imp/phy_rmii_if.vhd:173,322).  Sample acquisition (XADC, imp/dsp_system_top.vhd:412-435)
is replaced by a caller-supplied source of int16 frames.

This is SURVEY.md section 8(f) rows N1/N2: an edge adapter around the hot path, not part of it.
"""
from __future__ import annotations

from typing import Callable, Iterable, Optional

import numpy as np
import torch

from . import frames
from .chain import (ETHERNET_MODE_CMD, FPGA_RESET_CMD, START_COMMAND, UART_MODE_CMD, UART_REQUEST_CMD,
                    SpectrumChain)

SampleSource = Callable[[int], np.ndarray]     # n_frames -> [n_frames, 16384] int16 (12-bit values)


class VirtualFpga:
    """Byte-level stand-in for the board.

    ``write(data)`` takes what the GUI would send down the UART; ``read()`` returns what the board
    would send back (bytes for UART mode, a list of 1025-byte datagram payloads for Ethernet mode).
    Frames are produced in batches on the GPU and handed out one per request, like the FIFO of
    imp/fifo.vhd hands out one acquisition at a time.
    """

    def __init__(self, source: SampleSource, device: Optional[int] = 0, batch: int = 64,
                 chain: Optional[SpectrumChain] = None):
        self.chain = chain if chain is not None else SpectrumChain(device)
        self.source = source
        self.batch = int(batch)
        self.transport = "UART"            # sequ2 powers up in UART mode (imp/sequ2.vhd:88-91)
        self.started = False
        self._pending: list[bytes] = []
        self._out_uart = bytearray()
        self._out_udp: list[bytes] = []

    # ---- host -> board
    def write(self, data: bytes) -> int:
        data = bytes(data)
        # transport-select and start/reset bytes act outside the coefficient window only; let the ABI's
        # state machine decide which bytes were commands by replaying its busy logic per byte
        requests = 0
        for b in data:
            busy_before = self._coeff_bytes_left > 0
            requests += self.chain.feed_command_bytes(bytes([b]))
            self._track(b)
            if busy_before:
                continue
            if b == ETHERNET_MODE_CMD:
                self.transport = "ETHERNET"
            elif b == UART_MODE_CMD:
                self.transport = "UART"
            elif b == START_COMMAND:
                self.started = True
            elif b == FPGA_RESET_CMD:
                self.started = False
                self._pending.clear()
                self._out_uart.clear()
                self._out_udp.clear()
        for _ in range(requests):
            self._emit_frame()
        return len(data)

    # mirror of the RX state machine's busy flag (new/rx_filter_coeff.vhd:45-56), tracked host-side so the
    # adapter knows whether a byte was a command or a coefficient
    _coeff_bytes_left = 0

    def _track(self, b: int):
        if self._coeff_bytes_left > 0:
            self._coeff_bytes_left -= 1
        elif b == 0xF1:
            self._coeff_bytes_left = 12

    # ---- board -> host
    def _refill(self):
        x = np.ascontiguousarray(self.source(self.batch), dtype=np.int16).reshape(-1, frames.FFT_SIZE)
        xd = torch.from_numpy(x).to(self.chain.device)
        iq = self.chain.process_q15(xd)
        self._pending.extend(self.chain.frames_bytes(iq))

    def _emit_frame(self):
        if not self._pending:
            self._refill()
        frame = self._pending.pop(0)
        if self.transport == "UART":
            self._out_uart += frame
        else:
            self._out_udp.extend(frames.frame_to_udp_payloads(frame))

    def read(self, max_bytes: Optional[int] = None) -> bytes:
        """UART side: up to ``max_bytes`` of pending frame bytes."""
        n = len(self._out_uart) if max_bytes is None else min(max_bytes, len(self._out_uart))
        out = bytes(self._out_uart[:n])
        del self._out_uart[:n]
        return out

    def read_datagrams(self) -> list[bytes]:
        """Ethernet side: pending UDP payloads (1025 bytes each)."""
        out, self._out_udp = self._out_udp, []
        return out

    def stream(self, n_frames: int) -> Iterable[bytes]:
        """Free-running Ethernet-style streaming: yield ``n_frames`` frames without per-frame requests."""
        for _ in range(n_frames):
            if not self._pending:
                self._refill()
            yield self._pending.pop(0)

    def close(self):
        self.chain.close()
