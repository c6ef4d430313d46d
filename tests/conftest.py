import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
N = 16384


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) when no device is present and -m gpu was not requested.
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def hip_lib_built():
    """The in-tree HIP extension; built on demand here (hipcc cross-compiles without a GPU)."""
    from fpga_real_time_fft_analyzer_amd import abi
    if not os.path.exists(abi.LIB_PATH):
        abi.build()
    return abi.lib()


@pytest.fixture(scope="session")
def chain_cls(hip_lib_built):
    from fpga_real_time_fft_analyzer_amd.chain import SpectrumChain
    return SpectrumChain


def rel_maxnorm(a, b):
    """max|a-b| / max|b| per frame (rows), the norm SURVEY 8(d) prescribes."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    a = a.reshape(b.shape[0], -1) if b.ndim > 1 else a.reshape(1, -1)
    b2 = b.reshape(a.shape)
    num = np.abs(a - b2).max(axis=1)
    den = np.abs(b2).max(axis=1)
    return float((num / np.maximum(den, 1e-300)).max())
