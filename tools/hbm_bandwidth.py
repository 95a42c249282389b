import torch, time
x = torch.empty(1 << 30, dtype=torch.int32, device="cuda")
y = torch.empty(1 << 30, dtype=torch.int32, device="cuda")
def t(f, n=10):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ms = t(lambda: x.fill_(7)); print("fill 4 GiB: %.3f ms -> %.2f TB/s" % (ms, 4.295 / ms))
ms = t(lambda: y.copy_(x)); print("copy 4 GiB: %.3f ms -> %.2f TB/s (read+write)" % (ms, 2 * 4.295 / ms))
ms = t(lambda: x.sum()); print("sum 4 GiB: %.3f ms -> %.2f TB/s" % (ms, 4.295 / ms))
