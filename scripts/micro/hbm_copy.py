"""Achievable HBM rates on the box: device copy, fill (write only) and sum (read only) of a 420 MB float32 stack."""
import torch, time
n = 50 * 2048 * 2048
a = torch.empty(n, dtype=torch.float32, device="cuda"); b = torch.empty_like(a); a.normal_()
def t(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
d = t(lambda: b.copy_(a)); print("copy  : %.3f ms  %.2f TB/s (read+write)" % (d * 1e3, 2 * 4 * n / d / 1e12))
d = t(lambda: b.fill_(1.0)); print("fill  : %.3f ms  %.2f TB/s (write)" % (d * 1e3, 4 * n / d / 1e12))
d = t(lambda: a.sum()); print("sum   : %.3f ms  %.2f TB/s (read)" % (d * 1e3, 4 * n / d / 1e12))
d = t(lambda: torch.add(a, 1.0, out=b)); print("add   : %.3f ms  %.2f TB/s (read+write)" % (d * 1e3, 2 * 4 * n / d / 1e12))
