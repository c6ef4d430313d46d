"""Host-side mirror of the reference's signal-path contract, over the C ABI.

``SpectrumChain`` is the "virtual FPGA" seen from the host: it takes the same command bytes
(scripts/fft_analyzer_gui.py:28-37), the same ``0xF1`` + 12 x int8 coefficient upload
(gui.py:591-613 -> new/rx_filter_coeff.vhd:41-66) and produces the same 65536-byte frames
(imp/sequ2.vhd:153, gui.py:250-260) -- batched over thousands of frames resident in HBM.
PyTorch is used only for device memory and streams.
"""
from __future__ import annotations

import collections
import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch

from . import abi
from .abi import (SA_FILTER_CUSTOM, SA_FILTER_DEFAULT, SA_FILTER_NONE, SA_FILTER_WIDE, SA_N,
                  SA_OUT_MAG_FULL, SA_OUT_MAG_HALF, SA_OUT_SPEC_HALF, SA_OUT_TIME, SpecanError)

# command bytes, same names as gui.py:28-37
UART_REQUEST_CMD = 0xA5
FPGA_RESET_CMD = 0xFF
ETHERNET_MODE_CMD = 0xEF
UART_MODE_CMD = 0xFE
START_COMMAND = 0x55
FILTER_UPDATE_CMD = 0xF1
FILTER_DEFAULT_CMD = 0x00
FILTER_CUSTOM_CMD = 0xA1
FILTER_NONE_CMD = 0xB1
FILTER_WIDE_CMD = 0xA2          # build extension

FRAME_SIZE_BYTES = 65536        # gui.py:42
SAMPLES_PER_FRAME = 16384       # gui.py:43
FFT_SIZE = 16384                # gui.py:44
FS_HZ = 1_000_000.0             # gui.py:45

_OUT_KINDS = {"mag_full": SA_OUT_MAG_FULL, "mag_half": SA_OUT_MAG_HALF, "spec_half": SA_OUT_SPEC_HALF,
              "time": SA_OUT_TIME}


class SpectrumChain:
    """One handle = one (GPU, stream) instance of window -> IIR -> 16K FFT.

    Not thread-safe (same rule as the C ABI).  All process calls are asynchronous on the current
    torch stream of the handle's device.

    Overlap mode (:meth:`set_overlap` with depth d > 1): a call runs on an internal stream of the library that torch's
    caching allocator knows nothing about, and the C contract lends the call's input and output tensors to the library
    until d-1 further calls have been made or :meth:`flush` is called.  The wrapper therefore keeps a reference to the
    tensors of the last d-1 calls (a temporary passed as input, or a result that is dropped, would otherwise be
    recycled by the allocator under the running kernel) and releases them at the join, in :meth:`flush` and in
    :meth:`set_overlap`.  A result returned by a process call in this mode holds valid data on the current stream
    only after those d-1 further calls or after :meth:`flush`.
    """

    def __init__(self, device: Optional[int | torch.device | str] = None):
        self._lib = abi.lib()
        if device is None:
            dev = torch.cuda.current_device() if torch.cuda.is_available() else 0
        else:
            d = torch.device(device) if not isinstance(device, int) else torch.device("cuda", device)
            dev = d.index if d.index is not None else 0
        self.device = torch.device("cuda", dev)
        h = C.c_void_p()
        rc = self._lib.sa_create(dev, C.byref(h))
        if rc != abi.SA_OK:
            raise SpecanError(rc, self._lib.sa_last_error(None).decode())
        self._h = h
        self.control_generation = 0        # bumped by every call that changes what the path computes (virtual_fpga.py)
        self._depth = 1                    # launches in flight (sa_set_overlap)
        self._lent = collections.deque()   # (input, output) of the overlapped calls the caller's stream has not joined

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc: int):
        if rc != abi.SA_OK:
            raise SpecanError(rc, self._lib.sa_last_error(self._h).decode())

    def _ctl(self, rc: int):
        """_check for the calls that change what the path computes: virtual_fpga.py drops frames computed ahead."""
        self._check(rc)
        self.control_generation += 1

    def close(self):
        if getattr(self, "_h", None):
            self._lib.sa_destroy(self._h)        # waits on the host for the handle's own work
            self._h = None
            self._lent.clear()

    def _lend(self, x: torch.Tensor, out: torch.Tensor):
        """Overlap mode: the call just made owns (x, out) until depth-1 further calls have been made."""
        if self._depth > 1:
            self._lent.append((x, out))
            while len(self._lent) > self._depth - 1:   # this call enqueued the join of the call made depth-1 calls ago
                self._lent.popleft()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def _check_in(self, x: torch.Tensor, dtype: torch.dtype) -> int:
        if not isinstance(x, torch.Tensor) or x.dtype != dtype:
            raise SpecanError(abi.SA_EINVAL, f"input must be a {dtype} tensor")
        if x.device != self.device:
            raise SpecanError(abi.SA_EINVAL, f"input must live on {self.device}")
        if x.dim() != 2 or x.shape[1] != SA_N or not x.is_contiguous():
            raise SpecanError(abi.SA_ESHAPE, "input must be a contiguous [B, 16384] tensor")
        return x.shape[0]

    # ------------------------------------------------------------------ control plane
    def set_filter_mode(self, cmd: int):
        """0x00 default / 0xA1 custom / 0xB1 none (new/command_control.vhd:53-58); 0xA2 wide."""
        if not 0 <= int(cmd) <= 255:
            raise SpecanError(abi.SA_EINVAL, "filter command must be one byte")
        self._ctl(self._lib.sa_set_filter_mode(self._h, int(cmd)))

    @property
    def filter_mode(self) -> int:
        v = C.c_uint8()
        self._check(self._lib.sa_get_filter_mode(self._h, C.byref(v)))
        return v.value

    def load_coeffs_q7(self, coeffs: Sequence[int]):
        """12 int8 in wire order [b0,b1,b2,a0,a1,a2] x 2 (gui.py:598-605)."""
        a = np.asarray(coeffs).reshape(-1)
        if a.size != 12:
            raise SpecanError(abi.SA_EINVAL, "expected 12 coefficients (2 sections x 6)")
        a = a.astype(np.int64)
        if a.min() < -128 or a.max() > 127:
            raise SpecanError(abi.SA_EINVAL, "coefficients must fit int8")
        a8 = np.ascontiguousarray(a.astype(np.int8))
        self._ctl(self._lib.sa_load_coeffs_q7(self._h, a8.ctypes.data_as(C.POINTER(C.c_int8))))

    def coeffs_q7(self) -> np.ndarray:
        a = np.zeros(12, np.int8)
        self._check(self._lib.sa_get_coeffs_q7(self._h, a.ctypes.data_as(C.POINTER(C.c_int8))))
        return a

    def feed_command_bytes(self, data: bytes) -> int:
        """Push raw UART bytes through the RX state machine; returns the number of UART read requests
        (0xA5, imp/sequ2.vhd:216) seen outside coefficient uploads."""
        buf = (C.c_uint8 * len(data)).from_buffer_copy(bytes(data)) if len(data) else (C.c_uint8 * 1)()
        n = C.c_int(0)
        self._ctl(self._lib.sa_feed_command_bytes(self._h, buf, len(data), C.byref(n)))
        return n.value

    def feed_command_bytes_ex(self, data: bytes) -> "abi.CmdEvents":
        """The same, reporting everything a transport shim needs (sa_cmd_events of include/specan.h)."""
        buf = (C.c_uint8 * len(data)).from_buffer_copy(bytes(data)) if len(data) else (C.c_uint8 * 1)()
        ev = abi.CmdEvents()
        self._check(self._lib.sa_feed_command_bytes_ex(self._h, buf, len(data), C.byref(ev)))
        if ev.control_changed:
            self.control_generation += 1
        return ev

    @property
    def transport(self) -> int:
        """0xEF (Ethernet, the reset state: imp/sequ2.vhd:85-86) or 0xFE (UART)."""
        v = C.c_uint8()
        self._check(self._lib.sa_get_transport(self._h, C.byref(v)))
        return v.value

    def send_filter_coefficients(self, quantized_sections) -> bytes:
        """Same wire bytes as UartReceiver.send_filter_coefficients (gui.py:591-613): 0xF1 then the
        12 coefficients of exactly two sections, each ``int(c) & 0xFF``.  Returns the bytes sent."""
        secs = [list(s) for s in quantized_sections]
        if len(secs) != 2 or any(len(s) != 6 for s in secs):
            raise SpecanError(abi.SA_EINVAL, "exactly two sections of six coefficients (gui.py:1186-1192)")
        payload = bytes([FILTER_UPDATE_CMD]) + bytes(int(c) & 0xFF for s in secs for c in s)
        self.feed_command_bytes(payload)
        return payload

    def load_sos(self, sos):
        """Wide float format: up to 6 sections, scipy rows [b0,b1,b2,a0,a1,a2] (float64)."""
        s = np.ascontiguousarray(np.asarray(sos, np.float64).reshape(-1, 6))
        self._ctl(self._lib.sa_load_sos_f64(self._h, s.ctypes.data_as(C.POINTER(C.c_double)), s.shape[0]))

    def load_sos_f32(self, sos):
        s = np.ascontiguousarray(np.asarray(sos, np.float32).reshape(-1, 6))
        self._ctl(self._lib.sa_load_sos_f32(self._h, s.ctypes.data_as(C.POINTER(C.c_float)), s.shape[0]))

    def load_sos_q14(self, sos_q14):
        s = np.ascontiguousarray(np.asarray(sos_q14, np.int16).reshape(-1, 6))
        self._ctl(self._lib.sa_load_sos_q14(self._h, s.ctypes.data_as(C.POINTER(C.c_int16)), s.shape[0]))

    def set_window_q15(self, rom: Optional[np.ndarray]):
        if rom is None:
            self._ctl(self._lib.sa_set_window_q15(self._h, None))
            return
        r = np.ascontiguousarray(rom, np.int16)
        if r.shape != (SA_N,):
            raise SpecanError(abi.SA_ESHAPE, "window ROM must have 16384 entries")
        self._ctl(self._lib.sa_set_window_q15(self._h, r.ctypes.data_as(C.POINTER(C.c_int16))))

    def window_q15(self) -> np.ndarray:
        r = np.zeros(SA_N, np.int16)
        self._check(self._lib.sa_get_window_q15(self._h, r.ctypes.data_as(C.POINTER(C.c_int16))))
        return r

    def set_window_f32(self, w: Optional[np.ndarray]):
        if w is None:
            self._ctl(self._lib.sa_set_window_f32(self._h, None))
            return
        a = np.ascontiguousarray(w, np.float32)
        if a.shape != (SA_N,):
            raise SpecanError(abi.SA_ESHAPE, "window must have 16384 entries")
        self._ctl(self._lib.sa_set_window_f32(self._h, a.ctypes.data_as(C.POINTER(C.c_float))))

    def set_window_mode_q15(self, mode: int):
        self._ctl(self._lib.sa_set_window_mode_q15(self._h, int(mode)))

    def reserve(self, max_batch: int):
        self._check(self._lib.sa_reserve(self._h, int(max_batch)))

    def set_overlap(self, depth: int):
        """Opt-in overlapped launches (include/specan.h, sa_set_overlap): with depth d > 1 the results of a
        process call are visible on the current stream after d-1 further calls or after :meth:`flush`, and the
        call's input and output tensors belong to the library until then (the wrapper holds them: class docstring).
        1 = strictly stream-ordered."""
        self._check(self._lib.sa_set_overlap(self._h, int(depth)))
        self._depth = int(depth)                 # a change of depth waited on the host for everything in flight
        self._lent.clear()

    @property
    def overlap(self) -> int:
        v = C.c_int()
        self._check(self._lib.sa_get_overlap(self._h, C.byref(v)))
        return v.value

    def overlap_streams_side_by_side(self) -> bool:
        """Tests: re-run the library's probe on the handle's internal streams and the current stream
        (sa_debug_overlap_streams)."""
        v = C.c_int()
        self._check(self._lib.sa_debug_overlap_streams(self._h, self._stream(), C.byref(v)))
        return bool(v.value)

    def flush(self):
        """Make the current stream wait for every outstanding overlapped call (no host wait)."""
        self._check(self._lib.sa_flush(self._h, self._stream()))
        self._lent.clear()

    def set_profiling(self, ring: int):
        """Launch timing (include/specan.h, sa_set_profiling): keep the device times of the last ``ring`` stream-ordered
        process calls (0 = off).  The events ride on the dispatch packets; a train of calls runs as it does untimed."""
        self._check(self._lib.sa_set_profiling(self._h, int(ring)))

    def profile_read(self, n: int) -> list:
        """Device time in ms of up to ``n`` of the most recent timed calls, oldest first (waits for them)."""
        buf = (C.c_float * max(1, int(n)))()
        got = self._lib.sa_profile_read(self._h, buf, int(n))
        if got < 0:
            self._check(got)
        return [float(buf[i]) for i in range(got)]

    def iir_plan(self) -> np.ndarray:
        n = self._lib.sa_debug_iir_plan_f32(self._h, None, 0)
        out = np.zeros(n, np.float32)
        self._lib.sa_debug_iir_plan_f32(self._h, out.ctypes.data_as(C.POINTER(C.c_float)), n)
        return out

    # ------------------------------------------------------------------ data plane
    def process_f32(self, x: torch.Tensor, out: Optional[torch.Tensor] = None, out_kind: str = "mag_full",
                    scale: float = 1.0 / 2048.0):
        """[B,16384] float32 -> per ``out_kind``: 'mag_full' [B,16384] f32, 'mag_half' [B,8193] f32,
        'spec_half' [B,8193] complex64, 'time' [B,16384] f32.

        An int16 tensor (the ADC's samples, what the ingest front-end delivers) takes the same float path through
        sa_process_f32_i16: x = float(sample) * ``scale``, rounded once, no conversion pass; ``scale`` is ignored for
        float32 input."""
        if out_kind not in _OUT_KINDS:
            raise SpecanError(abi.SA_EINVAL, f"out_kind must be one of {sorted(_OUT_KINDS)}")
        from_i16 = x.dtype == torch.int16
        B = self._check_in(x, torch.int16 if from_i16 else torch.float32)
        if out_kind in ("mag_full", "time"):
            shape, dt = (B, SA_N), torch.float32
        elif out_kind == "mag_half":
            shape, dt = (B, SA_N // 2 + 1), torch.float32
        else:
            shape, dt = (B, SA_N // 2 + 1), torch.complex64
        if out is None:
            out = torch.empty(shape, dtype=dt, device=self.device)
        elif tuple(out.shape) != shape or out.dtype != dt or out.device != self.device or not out.is_contiguous():
            raise SpecanError(abi.SA_ESHAPE, f"out must be a contiguous {dt} tensor of shape {shape}")
        if from_i16:
            self._check(self._lib.sa_process_f32_i16(self._h, x.data_ptr(), float(scale), out.data_ptr(), B,
                                                     _OUT_KINDS[out_kind], self._stream()))
        else:
            self._check(self._lib.sa_process_f32(self._h, x.data_ptr(), out.data_ptr(), B, _OUT_KINDS[out_kind],
                                                 self._stream()))
        self._lend(x, out)
        return out

    def process_q15(self, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[B,16384] int16 -> [B,16384,2] int16 (re, im): B frames of 65536 bytes."""
        B = self._check_in(x, torch.int16)
        shape = (B, SA_N, 2)
        if out is None:
            out = torch.empty(shape, dtype=torch.int16, device=self.device)
        elif tuple(out.shape) != shape or out.dtype != torch.int16 or out.device != self.device or not out.is_contiguous():
            raise SpecanError(abi.SA_ESHAPE, f"out must be a contiguous int16 tensor of shape {shape}")
        self._check(self._lib.sa_process_q15(self._h, x.data_ptr(), out.data_ptr(), B, self._stream()))
        self._lend(x, out)
        return out

    def filter_q15(self, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Window (+ integer IIR) only: the FFT input stream, [B,16384] int16."""
        B = self._check_in(x, torch.int16)
        if out is None:
            out = torch.empty((B, SA_N), dtype=torch.int16, device=self.device)
        elif tuple(out.shape) != (B, SA_N) or out.dtype != torch.int16 or out.device != self.device or not out.is_contiguous():
            raise SpecanError(abi.SA_ESHAPE, "out must be a contiguous int16 [B,16384] tensor")
        self._check(self._lib.sa_filter_q15(self._h, x.data_ptr(), out.data_ptr(), B, self._stream()))
        self._lend(x, out)
        return out

    def frames_bytes(self, iq: torch.Tensor) -> list[bytes]:
        """Device IQ tensor -> list of 65536-byte frames exactly as sequ2 emits them.  In overlap mode the current
        stream first joins the outstanding calls (the tensor may be the result of one of them)."""
        if self._depth > 1:
            self.flush()
        host = iq.detach().to("cpu").contiguous().numpy().astype("<i2", copy=False)
        return [host[i].tobytes() for i in range(host.shape[0])]


def iir_plan_from_sos(sos) -> np.ndarray:
    """Host-only: the float IIR plan (no GPU needed); see include/specan.h sa_iir_plan_from_sos."""
    L = abi.lib()
    s = np.ascontiguousarray(np.asarray(sos, np.float64).reshape(-1, 6))
    n = L.sa_iir_plan_from_sos(s.ctypes.data_as(C.POINTER(C.c_double)), s.shape[0], None, 0)
    if n < 0:
        raise SpecanError(n, "sa_iir_plan_from_sos: bad SOS")
    out = np.zeros(n, np.float32)
    L.sa_iir_plan_from_sos(s.ctypes.data_as(C.POINTER(C.c_double)), s.shape[0],
                           out.ctypes.data_as(C.POINTER(C.c_float)), n)
    return out


def pack_frame(iq_host: np.ndarray) -> bytes:
    L = abi.lib()
    a = np.ascontiguousarray(iq_host, np.int16).reshape(SA_N, 2)
    buf = (C.c_uint8 * FRAME_SIZE_BYTES)()
    rc = L.sa_pack_frame(a.ctypes.data_as(C.POINTER(C.c_int16)), buf)
    if rc != abi.SA_OK:
        raise SpecanError(rc, "sa_pack_frame")
    return bytes(buf)
