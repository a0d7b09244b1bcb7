"""GPU probe: per-K-tile time of the tile256 GEMM as a function of where the activations come from."""
import sys, torch
sys.path.insert(0, ".")
from neat_amd import lib, ops
dev = "cuda:0"
def run(M, N, K, epi=lib.EPI_BF16, reps=20):
    g = torch.Generator(device=dev).manual_seed(1)
    A = torch.randn(M, K, generator=g, device=dev).bfloat16()
    W = ops.pack_weight(torch.randn(N, K, generator=g, device=dev) * K ** -0.5)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3): ops.gemm(A, W, epi, out0=out)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        ops.gemm(A, W, epi, out0=out); ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(reps))
    t = ts[len(ts) // 2]
    units = ((M + 255) // 256) * (N // 256); rounds = -(-units // 256)
    kt = rounds * (K // 64)
    print(f"M={M:6d} N={N:5d} K={K:5d} A={M*K*2/1e6:6.1f} MB  {t:7.1f} us  rounds={rounds}  {t/kt*1e3:6.0f} ns per K-tile  {2*M*N*K/t/1e6:7.1f} TF/s")
for M in (16384, 32768, 65536, 131072):
    run(M, 512, 2048)
for M in (8192, 16384, 32768, 65536):
    run(M, 2048, 2048)
run(32768, 512, 8192)
