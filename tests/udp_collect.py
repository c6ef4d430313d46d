"""Test helper: collect the 64 datagrams of a frame (index byte + 1024 data bytes, new/phy_rmii_if.vhd:173,322) back into
the 65536 frame bytes.  Receiver-side code is out of scope for the package; the tests only need
to see that what the emitter sends is a complete, correctly indexed frame.  The reference's own reassembler pins the
format through fixture G6."""


class FrameCollector:
    def __init__(self, packets: int = 64, data_size: int = 1024):
        self.packets, self.data_size = packets, data_size
        self.parts: dict[int, bytes] = {}
        self.frames = 0

    def add(self, payload: bytes, now_ms: int = 0):
        """Returns the frame bytes when the last missing index arrives, else None; malformed payloads are ignored."""
        if len(payload) != self.data_size + 1 or payload[0] >= self.packets:
            return None
        self.parts[payload[0]] = payload[1:]
        if len(self.parts) < self.packets:
            return None
        frame = b"".join(self.parts[i] for i in range(self.packets))
        self.parts = {}
        self.frames += 1
        return frame
