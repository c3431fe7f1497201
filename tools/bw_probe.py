import torch, time
x = torch.empty(2**28, dtype=torch.float64, device="cuda")   # 2 GiB
for fn, name in ((lambda: x.zero_(), "memset 2GiB"), (lambda: x.fill_(1.5), "fill 2GiB")):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(name, "%.3f ms  %.2f TB/s" % (ms, 2**31 / ms / 1e9))
y = torch.empty_like(x)
for _ in range(3): y.copy_(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): y.copy_(x)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("copy 2GiB  %.3f ms  %.2f TB/s (r+w)" % (ms, 2 * 2**31 / ms / 1e9))
