import torch, time
dev = "cuda"
n = 256 * 1024 * 1024
a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
b = torch.empty_like(a)
def t(f, reps=20):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
s = t(lambda: b.copy_(a)); print("copy 1 GiB->1 GiB: %.2f TB/s (read+write)" % (2 * n * 4 / s / 1e12))
s = t(lambda: a.sum()); print("sum 1 GiB: %.2f TB/s (read)" % (n * 4 / s / 1e12))
s = t(lambda: b.fill_(1.0)); print("fill 1 GiB: %.2f TB/s (write)" % (n * 4 / s / 1e12))
c = torch.empty(n // 4, dtype=torch.float32, device=dev)
s = t(lambda: torch.add(a[: n // 4], b[: n // 4], out=c)); print("add 256M+256M->256M: %.2f TB/s" % (3 * (n // 4) * 4 / s / 1e12))
# small working set like one frame (345 MB)
m = 43 * 1024 * 1024
s = t(lambda: b[:m].copy_(a[:m]), 50); print("copy 172 MB->172 MB: %.2f TB/s" % (2 * m * 4 / s / 1e12))
