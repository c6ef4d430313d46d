import ctypes as C, os, sys, time, torch
ROOT="/root/repo"; PKG=os.path.join(ROOT,"fpga_real_time_fft_analyzer_amd")
L=C.CDLL(os.path.join(PKG,"libspecan_hip.so")); h=C.c_void_p(); assert L.sa_create(0,C.byref(h))==0
L.sa_filter_q15.argtypes=[C.c_void_p,C.c_void_p,C.c_void_p,C.c_int,C.c_void_p]; L.sa_set_filter_mode.argtypes=[C.c_void_p,C.c_uint8]
L.sa_set_filter_mode(h,0)
st=torch.cuda.current_stream().cuda_stream
for B in (1024,2048,4096,8192,16384):
    x=torch.randint(-2048,2048,(B,16384),device="cuda",dtype=torch.int32).to(torch.int16); o=torch.empty_like(x)
    for _ in range(2): L.sa_filter_q15(h,x.data_ptr(),o.data_ptr(),B,st)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(4): L.sa_filter_q15(h,x.data_ptr(),o.data_ptr(),B,st)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/4
    print(B, round(dt*1e6,1),"us", round(B/dt/1e6,2),"M frames/s")
