"""Virtual FPGA: the board as `scripts/fft_analyzer_gui.py` sees it, backed by the MI355X path.

Stands where the serial port / UDP socket of `UartReceiver` / `UdpReceiver` stand (gui.py:355-747).
The host writes command bytes; the board answers with 65536-byte frames, raw on the UART or as 64
datagram payloads of 1 index byte + 1024 data bytes on Ethernet (gui.py:48-50, imp/phy_rmii_if.vhd:173,322).
Sample acquisition (XADC, imp/dsp_system_top.vhd:412-435) is replaced by a caller-supplied source of
int16 frames.

Sequencing follows imp/sequ2.vhd (command decode through the C ABI, which mirrors
new/command_control.vhd and new/rx_filter_coeff.vhd):

* reset / power-on: Ethernet transport (sequ2.vhd:85-86), both output state machines idle,
  filter NONE, coefficients cleared;
* 0xEF / 0xFE select the transport (sequ2.vhd:88-91); the state machine of the transport that is
  NOT selected falls back to idle (sequ2.vhd:182-184, 262-264), so a new 0x55 is needed after a switch;
* Ethernet: 0x55 starts free-running output, one frame per acquisition (S_IDLE1 -> S_FILL ..., :116-178);
* UART: 0x55 only arms (U_IDLE1 -> U_IDLE2, :209-213); the first 0xA5 starts the byte stream (:219-223),
  which then continues frame after frame without further requests (U_READ -> U_FILL -> U_READ, :225-259) --
  gui.py sends 0x55, then 0xA5 once, 100 ms later (:529-549), and slices the stream into 65536-byte frames;
* bytes of a 0xF1 coefficient upload are seen by neither decoder (uart_rx_valid and not busy,
  imp/dsp_system_top.vhd:644).

A change of filter mode, coefficients or a reset drops every frame that was computed ahead with the
old settings: the board's FIFO holds one acquisition (imp/fifo.vhd, 16 K x 32), not a batch.

This is SURVEY.md section 8(f) rows N1/N2: an edge adapter around the hot path, not part of it.
"""
from __future__ import annotations

import socket
import time
from typing import Callable, Iterable, Optional

import numpy as np
import torch

from . import frames
from .chain import ETHERNET_MODE_CMD, UART_MODE_CMD, SpectrumChain

SampleSource = Callable[[int], np.ndarray]     # n_frames -> [n_frames, 16384] int16 (12-bit values)

ETHERNET_FPS_LIMIT = 30.0                      # gui.py:53
FPGA_SRC_PORT, HOST_DST_PORT = 5005, 6006      # imp/head_data.mif:27-38, gui.py:20-22


class VirtualFpga:
    """Byte-level stand-in for the board.

    ``write(data)`` takes what the GUI sends down the UART.  ``read(n)`` / ``in_waiting`` are the UART
    return path; ``read_datagrams(max_frames)`` / ``serve_udp()`` the Ethernet one.  Frames are computed on
    the GPU ``batch`` acquisitions at a time and handed out one by one.
    """

    def __init__(self, source: SampleSource, device: Optional[int] = 0, batch: int = 64,
                 chain: Optional[SpectrumChain] = None):
        self.chain = chain if chain is not None else SpectrumChain(device)
        self.source = source
        self.batch = int(batch)
        self._pending: list[bytes] = []
        self._pending_generation = -1          # chain.control_generation the frames in _pending were computed under
        self._out_uart = bytearray()
        self._idle()

    def _idle(self):
        self.eth_streaming = False             # sequ_2 Ethernet FSM past S_IDLE1
        self.uart_state = "IDLE1"              # IDLE1 -> IDLE2 (0x55) -> STREAM (0xA5)

    @property
    def transport(self) -> str:
        return "ETHERNET" if self.chain.transport == ETHERNET_MODE_CMD else "UART"

    @property
    def started(self) -> bool:
        return self.eth_streaming or self.uart_state != "IDLE1"

    # ---- host -> board
    def write(self, data: bytes) -> int:
        data = bytes(data)
        for b in data:                                         # byte by byte: order matters
            before = self.chain.transport
            ev = self.chain.feed_command_bytes_ex(bytes([b]))
            if ev.control_changed:                             # frames computed ahead belong to the old settings
                self._pending.clear()
            if ev.n_reset:
                self._idle()
                self._out_uart.clear()
                continue
            if ev.transport != before:                         # the deselected FSM drops to idle
                if ev.transport == ETHERNET_MODE_CMD:
                    self.uart_state = "IDLE1"
                    self._out_uart.clear()
                else:
                    self.eth_streaming = False
            if ev.n_start:                                     # start_fill reaches only the selected FSM
                if ev.transport == ETHERNET_MODE_CMD:
                    self.eth_streaming = True
                elif self.uart_state == "IDLE1":
                    self.uart_state = "IDLE2"
            if ev.n_uart_request and ev.transport == UART_MODE_CMD and self.uart_state == "IDLE2":
                self.uart_state = "STREAM"
        return len(data)

    # ---- acquisitions
    def _next_frame(self) -> bytes:
        # frames computed ahead belong to the settings of their batch: ANY control change on the chain since then --
        # command bytes, or a window / coefficient / mode call made directly on `self.chain` -- drops them
        if self._pending_generation != self.chain.control_generation:
            self._pending.clear()
        if not self._pending:
            x = np.ascontiguousarray(self.source(self.batch), dtype=np.int16).reshape(-1, frames.FFT_SIZE)
            iq = self.chain.process_q15(torch.from_numpy(x).to(self.chain.device))
            self._pending.extend(self.chain.frames_bytes(iq))
            self._pending_generation = self.chain.control_generation
        return self._pending.pop(0)

    # ---- board -> host, UART
    @property
    def in_waiting(self) -> int:
        """Bytes ready on the UART.  While streaming there is always at least one frame's worth coming."""
        if self.uart_state == "STREAM" and not self._out_uart:
            self._out_uart += self._next_frame()
        return len(self._out_uart)

    def read(self, max_bytes: Optional[int] = None) -> bytes:
        """Up to ``max_bytes`` of the UART stream (everything buffered when None; one more frame is produced
        when the buffer is empty and the stream is running)."""
        if self.uart_state == "STREAM":
            want = frames.FRAME_SIZE_BYTES if max_bytes is None else max_bytes
            while len(self._out_uart) < want and (max_bytes is not None or not self._out_uart):
                self._out_uart += self._next_frame()
        n = len(self._out_uart) if max_bytes is None else min(max_bytes, len(self._out_uart))
        out = bytes(self._out_uart[:n])
        del self._out_uart[:n]
        return out

    def reset_input_buffer(self):
        self._out_uart.clear()

    # ---- board -> host, Ethernet
    def read_datagrams(self, max_frames: int = 1) -> list[bytes]:
        """UDP payloads (1025 bytes each, 64 per frame) of up to ``max_frames`` acquisitions; empty unless the
        Ethernet state machine is running."""
        out: list[bytes] = []
        if self.eth_streaming:
            for _ in range(max_frames):
                out.extend(frames.frame_to_udp_payloads(self._next_frame()))
        return out

    def serve_udp(self, addr: tuple[str, int] = ("127.0.0.1", HOST_DST_PORT), n_frames: Optional[int] = None,
                  fps_limit: float = ETHERNET_FPS_LIMIT, src_port: Optional[int] = None,
                  stop: Optional[Callable[[], bool]] = None, idle_sleep: float = 0.005) -> int:
        """Emit frames to ``addr`` the way the MAC does: 64 datagrams of 1 + 1024 bytes per frame, at most
        ``fps_limit`` frames per second (the board delivers 30, README.md:168 / gui.py:53), for ``n_frames``
        frames or until ``stop()`` is true.  Nothing is sent while the Ethernet state machine is idle (no 0x55
        yet, or UART selected).  Returns the number of frames sent."""
        sock = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
        if src_port is not None:
            sock.bind(("", src_port))
        sent = 0
        period = 1.0 / fps_limit if fps_limit and fps_limit > 0 else 0.0
        next_t = time.monotonic()
        try:
            while (n_frames is None or sent < n_frames) and not (stop and stop()):
                if not self.eth_streaming:
                    time.sleep(idle_sleep)
                    continue
                now = time.monotonic()
                if now < next_t:
                    time.sleep(next_t - now)
                for p in frames.frame_to_udp_payloads(self._next_frame()):
                    sock.sendto(p, addr)
                sent += 1
                next_t = max(next_t + period, time.monotonic())
        finally:
            sock.close()
        return sent

    def stream(self, n_frames: int) -> Iterable[bytes]:
        """``n_frames`` consecutive acquisitions as frame bytes, regardless of transport state (bench helper)."""
        for _ in range(n_frames):
            yield self._next_frame()

    def close(self):
        self.chain.close()


class VirtualSerial:
    """The object ``UartReceiver`` expects from ``serial.Serial(port, baud, timeout=..., ...)``
    (gui.py:464-480, 496-498, 529-651): ``write``, ``flush``, ``in_waiting``, ``read(n)``,
    ``reset_input_buffer``, ``reset_output_buffer``, ``is_open``, ``close`` -- over a VirtualFpga.
    No pyserial import anywhere: a maintainer points ``serial.Serial`` at ``VirtualSerial.factory(fpga)``
    (INTEGRATION.md section 2)."""

    def __init__(self, fpga: VirtualFpga, port: str = "VIRTUAL", baudrate: int = 230400, timeout: Optional[float] = None,
                 **_ignored):
        self._fpga = fpga
        self.port, self.baudrate, self.timeout = port, baudrate, timeout
        self.is_open = True

    @classmethod
    def factory(cls, fpga: VirtualFpga):
        """A callable with pyserial's ``Serial(...)`` signature bound to ``fpga``."""
        def make(port="VIRTUAL", baudrate=230400, **kw):
            return cls(fpga, port, baudrate, **kw)
        return make

    def _check(self):
        if not self.is_open:
            raise OSError("port is closed")            # pyserial raises PortNotOpenError (an OSError/IOError)

    def write(self, data) -> int:
        self._check()
        return self._fpga.write(bytes(data))

    def flush(self):
        self._check()

    @property
    def in_waiting(self) -> int:
        self._check()
        return self._fpga.in_waiting

    def read(self, size: int = 1) -> bytes:
        self._check()
        return self._fpga.read(size)

    def reset_input_buffer(self):
        self._check()
        self._fpga.reset_input_buffer()

    def reset_output_buffer(self):
        self._check()

    def close(self):
        self.is_open = False

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
