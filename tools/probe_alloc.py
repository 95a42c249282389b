"""how long do large device allocations take on this box?  (hipMalloc through torch's allocator, then a fill)"""
import time
import torch
torch.cuda.init()
for gb in (1, 8, 24, 48, 96, 160):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x = torch.empty(gb << 30, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    x.zero_()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    x.zero_()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    del x
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print("%4d GB: alloc %.3f s, first fill %.3f s, second fill %.3f s, free %.3f s" % (gb, t1 - t0, t2 - t1, t3 - t2, t4 - t3), flush=True)
