"""Reference streaming rates of torch kernels on the device (sum / max / copy / fill of 1 GiB, sum of 256 MiB)."""
import torch, time
x = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device="cuda").normal_()
y = torch.empty_like(x)
def t(f, n=20):
    f(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
ms = t(lambda: x.sum()); print("sum 1 GiB: %.3f ms  %.2f TB/s" % (ms, 1.0737 / ms))
ms = t(lambda: torch.max(x)); print("max 1 GiB: %.3f ms  %.2f TB/s" % (ms, 1.0737 / ms))
ms = t(lambda: y.copy_(x)); print("copy 1 GiB: %.3f ms  %.2f TB/s (r+w)" % (ms, 2 * 1.0737 / ms))
ms = t(lambda: y.fill_(1.0)); print("fill 1 GiB: %.3f ms  %.2f TB/s" % (ms, 1.0737 / ms))
xs = x[: 64 * 1024 * 1024]
ms = t(lambda: xs.sum()); print("sum 256 MiB: %.3f ms  %.2f TB/s" % (ms, 0.2684 / ms))
