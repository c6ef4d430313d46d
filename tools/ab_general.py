#!/usr/bin/env python3
"""A/B of two builds on a cascade of six GENERAL-numerator sections (elliptic band-pass) for the three spectrum outputs --
the instantiations <6, false, OUT, *> that the headline filter (unit numerators) never reaches.  Used in round 3 to price
the 4-7 spilled registers of the half-spectrum variants (profiles/r3_resource_usage.txt).
usage: ab_general.py   (libraries: libspecan_hip.so and libspecan_ab_prebh.so in the package directory)"""
import ctypes as C, os, sys, time
import numpy as np, torch
from scipy import signal
ROOT='/root/repo' if os.path.isdir('/root/repo') else os.getcwd()
PKG=os.path.join(ROOT,'fpga_real_time_fft_analyzer_amd')
sos=np.ascontiguousarray(signal.ellip(6,0.5,40.0,[0.1,0.3],btype='bandpass',output='sos'),np.float64)   # 6 general-numerator sections
B=4096
xs=[torch.randn(B,16384,device='cuda') for _ in range(4)]
res={}
for name in ('libspecan_hip.so','libspecan_ab_prebh.so'):
    L=C.CDLL(os.path.join(PKG,name)); h=C.c_void_p(); assert L.sa_create(0,C.byref(h))==0
    L.sa_load_sos_f64.argtypes=[C.c_void_p,C.POINTER(C.c_double),C.c_int]
    L.sa_process_f32.argtypes=[C.c_void_p,C.c_void_p,C.c_void_p,C.c_int,C.c_int,C.c_void_p]
    L.sa_set_filter_mode.argtypes=[C.c_void_p,C.c_uint8]
    assert L.sa_load_sos_f64(h,sos.ctypes.data_as(C.POINTER(C.c_double)),6)==0
    L.sa_set_filter_mode(h,0xA1)
    for kind,shape,dt in ((0,(B,16384),torch.float32),(1,(B,8193),torch.float32),(2,(B,8193),torch.complex64)):
        outs=[torch.empty(shape,dtype=dt,device='cuda') for _ in range(4)]
        ts=[]
        for rnd in range(8):
            for i in range(3): L.sa_process_f32(h,xs[i%4].data_ptr(),outs[i%4].data_ptr(),B,kind,None)
            torch.cuda.synchronize(); t0=time.perf_counter()
            for i in range(40): L.sa_process_f32(h,xs[i%4].data_ptr(),outs[i%4].data_ptr(),B,kind,None)
            torch.cuda.synchronize(); ts.append((time.perf_counter()-t0)/40*1e6)
        res[(name,kind)]=np.median(ts)
        print(f'{name:26s} out_kind {kind}: {np.median(ts):7.1f} us per 4096 frames')
