"""MI355X-native spectrum-analyser signal path (Hann window -> 6-biquad IIR -> 16K FFT).

Drop-in for the hot path of mfkiwl/fpga-real-time-fft-analyzer; see DESIGN.md / INTEGRATION.md.
Importing the package does not load the HIP library; constructing ``SpectrumChain`` does, and
fails loudly when it is missing (no CPU fallback).
"""
from . import abi, designer, frames  # noqa: F401
from .abi import SpecanError  # noqa: F401

__all__ = ["abi", "designer", "frames", "SpecanError", "SpectrumChain"]


def __getattr__(name):
    if name == "SpectrumChain":
        import importlib
        return importlib.import_module(__name__ + ".chain").SpectrumChain
    raise AttributeError(name)
