"""What HBM rate does a plain streaming kernel reach for the hot path's access mix (two reads, one
write per element)?  torch elementwise kernels on 4096 x 65536 f32 (GPU box)."""
import torch

n = 4096 * 65536
a = torch.randn(n, device="cuda")
b = torch.randn(n, device="cuda")
c = torch.empty(n, device="cuda")


def timed(fn, bytes_moved, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return bytes_moved / ms / 1e9


print("add  (2 reads + 1 write): %.2f TB/s" % timed(lambda: torch.add(a, b, out=c), 12 * n))
print("copy (1 read  + 1 write): %.2f TB/s" % timed(lambda: c.copy_(a), 8 * n))
print("sum  (1 read)           : %.2f TB/s" % timed(lambda: a.sum(), 4 * n))
print("fill (1 write)          : %.2f TB/s" % timed(lambda: c.fill_(1.0), 4 * n))
