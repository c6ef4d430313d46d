"""Frame layout helpers: the byte contract between the signal path and the host GUI.

Frame = 16384 bins x (int16 re LE, int16 im LE) = 65536 bytes (imp/sequ2.vhd:153,
scripts/fft_analyzer_gui.py:250-270).  UDP transport = 64 datagrams per frame, payload =
1 index byte + 1024 data bytes (gui.py:48-50, imp/phy_rmii_if.vhd:173,322).
"""
from __future__ import annotations

import numpy as np

FRAME_SIZE_BYTES = 65536
FFT_SIZE = 16384
PACKETS_PER_FRAME = 64
PACKET_DATA_SIZE = FRAME_SIZE_BYTES // PACKETS_PER_FRAME     # 1024
ETHERNET_PAYLOAD_SIZE = PACKET_DATA_SIZE + 1                 # 1025
FS_HZ = 1_000_000.0


def _iq(frame_bytes: bytes):
    if len(frame_bytes) != FRAME_SIZE_BYTES:
        raise ValueError(f"Invalid frame size: {len(frame_bytes)} (expected {FRAME_SIZE_BYTES})")
    a = np.frombuffer(frame_bytes, dtype="<i2").reshape(FFT_SIZE, 2)
    return a[:, 0], a[:, 1]


def decode_mag_16iq_le(frame_bytes: bytes) -> np.ndarray:
    """Same result as gui.py:250-260: float32 sqrt(re^2 + im^2) over all 16384 bins."""
    re, im = _iq(frame_bytes)
    return np.sqrt(re.astype(np.float32) ** 2 + im.astype(np.float32) ** 2)


def decode_iq_components(frame_bytes: bytes):
    """Same result as gui.py:262-270: (re, im) as float32 arrays."""
    re, im = _iq(frame_bytes)
    return re.astype(np.float32), im.astype(np.float32)


def frequency_axis_khz(n_bins: int = FFT_SIZE) -> np.ndarray:
    """Bin k -> k*FS/N in kHz over all N bins (gui.py:297)."""
    return np.arange(n_bins, dtype=np.float32) * (FS_HZ / FFT_SIZE) / 1e3


def frame_to_udp_payloads(frame_bytes: bytes) -> list[bytes]:
    """Cut a frame into the 64 datagram payloads the FPGA MAC sends: index byte 0..63 followed by
    1024 data bytes (consumed by MultiPacketAssembler.add, gui.py:318-339)."""
    if len(frame_bytes) != FRAME_SIZE_BYTES:
        raise ValueError("frame must be 65536 bytes")
    return [bytes([i]) + frame_bytes[i * PACKET_DATA_SIZE:(i + 1) * PACKET_DATA_SIZE]
            for i in range(PACKETS_PER_FRAME)]
