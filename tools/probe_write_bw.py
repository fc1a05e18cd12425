"""What does a plain device-wide write / copy reach on this box?  (calibrates the 'hbm' roofline of the
write-bound 1x1 layers: their output is 16x the bytes of their input)."""
import torch, time
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for mb in (103, 411, 822, 1644):
    n = mb * 1000 * 1000 // 4
    a = torch.empty(n, dtype=torch.float32, device=dev)
    b = torch.empty(n, dtype=torch.float32, device=dev)
    ms = t(lambda: a.fill_(1.0)); print("fill   %5d MB  %.4f ms  %.2f TB/s" % (mb, ms, n * 4 / ms / 1e9))
    ms = t(lambda: a.zero_()); print("zero   %5d MB  %.4f ms  %.2f TB/s" % (mb, ms, n * 4 / ms / 1e9))
    ms = t(lambda: b.copy_(a)); print("copy   %5d MB  %.4f ms  %.2f TB/s (r+w)" % (mb, ms, 2 * n * 4 / ms / 1e9))
    ms = t(lambda: a.sum()); print("sum    %5d MB  %.4f ms  %.2f TB/s (read)" % (mb, ms, n * 4 / ms / 1e9))
    i8 = torch.empty(n // 4, dtype=torch.int8, device=dev)
    ms = t(lambda: torch.Tensor.copy_(a[: n // 4], i8)); print("i8->f32 %4d MB out %.4f ms  %.2f TB/s" % (mb // 4, ms, (n // 4 * 5) / ms / 1e9))
    del a, b, i8
