"""HBM yardstick on this box: pure read (torch.sum over 1 GiB fp32), copy (read + write) and fill (pure write), GB/s."""
import torch
dev = "cuda"
n = 256 * 1024 * 1024
x = torch.randn(n, device=dev)
y = torch.empty_like(x)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3
s = t(lambda: x.sum()); print(f"read  (sum, 1 GiB):        {n*4/s/1e12:.2f} TB/s")
s = t(lambda: x.max()); print(f"read  (max, 1 GiB):        {n*4/s/1e12:.2f} TB/s")
s = t(lambda: y.copy_(x)); print(f"copy  (1 GiB -> 1 GiB):    {2*n*4/s/1e12:.2f} TB/s (read + write)")
s = t(lambda: y.fill_(1.0)); print(f"write (fill, 1 GiB):       {n*4/s/1e12:.2f} TB/s")
s = t(lambda: torch.add(x, 1.0, out=y)); print(f"add   (1 GiB -> 1 GiB):    {2*n*4/s/1e12:.2f} TB/s (read + write)")
