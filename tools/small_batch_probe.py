import os, sys, ctypes as C
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
from fpga_real_time_fft_analyzer_amd import abi
if len(sys.argv) > 2: abi.LIB_PATH = os.path.join(os.path.dirname(abi.LIB_PATH), sys.argv[2])
from fpga_real_time_fft_analyzer_amd.chain import SpectrumChain
B = int(sys.argv[1])
ch = SpectrumChain(0)
x = torch.randn(4096, 16384, device="cuda"); o = torch.empty(4096, 16384, device="cuda")
ch.set_profiling(64)
for rep in range(30):
    for j in range(0, 4096 - B + 1, B):
        ch.process_f32(x[j:j+B], out=o[j:j+B])
        if (j // B) % 16 == 15: break
ms = ch.profile_read(64)
print(f"{sys.argv[2] if len(sys.argv) > 2 else 'product':24s} B={B:4d}: kernel median {np.median(ms)*1e3:7.2f} us  min {min(ms)*1e3:7.2f}")
