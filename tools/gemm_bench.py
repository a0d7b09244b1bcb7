"""GPU micro-benchmark of neat_gemm_bf16 on the text-tower shapes (random data, HIP events, interleaved rounds)."""
import sys, os, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neat_amd import lib, ops

dev = "cuda"
B = int(os.environ.get("B", 256))
# M4 / M3: rows of the forward (4 caption sets) and backward (3) GEMMs; override for other regimes, e.g. the packed text
# batches (M4=14848 M3=11136: 14.5 tokens per caption) or a 250-image ViT-B/32 gallery batch (M4=12500 C=768)
M4 = int(os.environ.get("M4", 4 * B * 77))
M3 = int(os.environ.get("M3", M4 * 3 // 4))
C = int(os.environ.get("C", 512))                 # text width: 512 = ViT-B-32 / B-16, 768 = ViT-L-14
shapes = [("qkv  fwd BIAS_BF16", lib.EPI_BIAS_BF16, M4, 3 * C, C), ("out  fwd BIAS_RES", lib.EPI_BIAS_RES_F32, M4, C, C),
          ("fc   fwd BIAS_GELU", lib.EPI_BIAS_GELU, M4, 4 * C, C), ("proj fwd BIAS_RES", lib.EPI_BIAS_RES_F32, M4, C, 4 * C),
          ("dproj bwd GELU_BWD", lib.EPI_GELU_BWD, M3, 4 * C, C), ("dfc  bwd BF16", lib.EPI_BF16, M3, C, 4 * C),
          ("do   bwd BF16", lib.EPI_BF16, M3, C, C), ("dqkv bwd BF16", lib.EPI_BF16, M3, C, 3 * C)]
g = torch.Generator(device=dev).manual_seed(0)
cases = []
for name, epi, M, N, K in shapes:
    A = torch.randn(M, K, generator=g, device=dev).bfloat16()
    W = ops.pack_weight(torch.randn(N, K, generator=g, device=dev) * K ** -0.5)
    bias = torch.randn(N, generator=g, device=dev)
    res = torch.randn(M, N, generator=g, device=dev) if epi == lib.EPI_BIAS_RES_F32 else None
    aux = torch.randn(M, N, generator=g, device=dev).bfloat16() if epi == lib.EPI_GELU_BWD else None
    out0 = torch.empty(M, N, device=dev, dtype=torch.float32 if epi == lib.EPI_BIAS_RES_F32 else torch.bfloat16)
    out1 = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if epi == lib.EPI_BIAS_GELU else None
    cases.append((name, epi, M, N, K, A, W, bias, res, aux, out0, out1))
rounds = int(os.environ.get("ROUNDS", 10))
times = {c[0]: [] for c in cases}
for r in range(rounds + 2):
    for name, epi, M, N, K, A, W, bias, res, aux, out0, out1 in cases:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.gemm(A, W, epi, bias=bias, res=res, aux=aux, out0=out0, out1=out1)
        b.record()
        torch.cuda.synchronize()
        if r >= 2:
            times[name].append(a.elapsed_time(b) * 1e3)
tot = 0
for name, epi, M, N, K, *_ in cases:
    t = sorted(times[name])
    med = t[len(t) // 2]
    tot += med
    ob = 4 if epi == lib.EPI_BIAS_RES_F32 else (4 if epi == lib.EPI_BIAS_GELU else 2)
    extra = M * N * (4 if epi == lib.EPI_BIAS_RES_F32 else (2 if epi == lib.EPI_GELU_BWD else 0))
    byts = M * K * 2 + N * K * 2 + M * N * ob + extra
    print(f"{name:20s} M={M} N={N:5d} K={K:5d}  med {med:8.1f} us  min {t[0]:8.1f}  {2*M*N*K/med/1e6:7.1f} TF/s  {byts/med/1e6:6.2f} TB/s")
print(f"sum of medians {tot:.1f} us (one layer fwd+bwd GEMMs)")
