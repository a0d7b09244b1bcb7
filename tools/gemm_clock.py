"""What clock does the chip hold under the tile256 GEMM?  Runs ONE shape at `mult` x the step's row count (so that a
dispatch lasts ~10 ms) back to back for >= 2.5 s and prints the event-timed average per launch.  Under
`rocprofv3 --pmc GRBM_GUI_ACTIVE` the effective clock is GRBM_GUI_ACTIVE / 8 / launch time (MI355X_MICROARCH.md, DVFS).
usage: gemm_clock.py {fc|dfc|qkv} [mult]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neat_amd import lib, ops

which = sys.argv[1] if len(sys.argv) > 1 else "fc"
mult = int(sys.argv[2]) if len(sys.argv) > 2 else 40
epi, M, N, K = {"fc": (lib.EPI_BIAS_GELU, 78848, 2048, 512), "dfc": (lib.EPI_BF16, 59136, 512, 2048),
                "qkv": (lib.EPI_BIAS_BF16, 78848, 1536, 512)}[which]
M *= mult
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
W = ops.pack_weight(torch.randn(N, K, generator=g, device=dev) * K ** -0.5)
bias = torch.randn(N, generator=g, device=dev)
A = torch.randn(M, K, generator=g, device=dev).bfloat16()
out0 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
out1 = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if epi == lib.EPI_BIAS_GELU else None
def run():
    ops.gemm(A, W, epi, bias=bias, out0=out0, out1=out1, kernel="t256")
run(); torch.cuda.synchronize()
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 2.5:
    for _ in range(10):
        run()
    torch.cuda.synchronize(); n += 10
ts = []
for _ in range(10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
ts.sort()
ms = ts[len(ts) // 2]
print(f"{which} x{mult}: M={M} N={N} K={K}  {ms:.3f} ms per launch  {2.0*M*N*K/ms/1e9:.1f} TFLOP/s  launch_ms={ms:.4f}", flush=True)
