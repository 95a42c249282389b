import time, torch, ctypes
hip = ctypes.CDLL("libamdhip64.so")
torch.cuda.init(); torch.zeros(1, device="cuda")
def t_alloc(nbytes, reps=50):
    ptrs=[]; t0=time.perf_counter()
    for _ in range(reps):
        p=ctypes.c_void_p(); hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(nbytes)); ptrs.append(p)
    t1=time.perf_counter()
    for p in ptrs: hip.hipFree(p)
    t2=time.perf_counter()
    return 1e6*(t1-t0)/reps, 1e6*(t2-t1)/reps
for nb in (4096, 1<<20, 16<<20, 128<<20, 1<<30):
    a,f=t_alloc(nb, 20 if nb>=(128<<20) else 50); print("%10d bytes: hipMalloc %.1f us, hipFree %.1f us" % (nb,a,f))
