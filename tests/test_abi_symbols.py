"""CPU: the C-ABI library loads and exports every symbol include/specan.h declares (no compute)."""
import ctypes
import os
import re

from conftest import ROOT


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "specan.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sa_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_the_boundary():
    names = declared_symbols()
    for must in ("sa_create", "sa_destroy", "sa_set_filter_mode", "sa_load_coeffs_q7", "sa_feed_command_bytes",
                 "sa_load_sos_f32", "sa_load_sos_q14", "sa_set_window_q15", "sa_set_window_f32",
                 "sa_process_q15", "sa_process_f32", "sa_pack_frame", "sa_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol(hip_lib_built):
    for name in declared_symbols():
        assert hasattr(hip_lib_built, name), f"{name} declared in include/specan.h but not exported"
    assert hip_lib_built.sa_abi_version() == 4


def test_no_torch_or_oracle_linkage(hip_lib_built):
    """The product library links HIP only: no torch, no oracle."""
    from fpga_real_time_fft_analyzer_amd import abi
    import subprocess
    out = subprocess.run(["ldd", abi.LIB_PATH], capture_output=True, text=True).stdout
    assert "torch" not in out and "specan_oracle" not in out
    assert "amdhip64" in out


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "fpga_real_time_fft_analyzer_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dp, fn)).read()
                for bad in ("import oracle", "from oracle", "libspecan_oracle", "oracle/specan_oracle.c\"", "oracle.oracle"):
                    assert bad not in txt, (fn, bad)


def test_no_environment_switches_in_the_product():
    """Which kernel the product library launches never depends on the environment: A/B variants are -D builds
    (tools/ab_libs.py takes two libraries).  No getenv anywhere under csrc/, nor in the built library's imports."""
    csrc = os.path.join(ROOT, "fpga_real_time_fft_analyzer_amd", "csrc")
    for fn in os.listdir(csrc):
        if fn.endswith((".cpp", ".hip", ".hpp", ".h")):
            assert "getenv" not in open(os.path.join(csrc, fn)).read(), fn
    from fpga_real_time_fft_analyzer_amd import abi
    import subprocess
    syms = subprocess.run(["nm", "-D", "--undefined-only", abi.LIB_PATH], capture_output=True, text=True).stdout
    assert "getenv" not in syms


def test_create_without_gpu_fails_loudly(hip_lib_built):
    import torch
    if torch.cuda.is_available():
        return
    h = ctypes.c_void_p()
    rc = hip_lib_built.sa_create(0, ctypes.byref(h))
    assert rc == -3 and not h.value
    assert b"no usable HIP device" in hip_lib_built.sa_last_error(None)
