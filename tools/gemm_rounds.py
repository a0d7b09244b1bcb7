"""How much of a GEMM's time is round quantisation: time vs number of 256-row blocks for one (epilogue, N, K), tile256
kernel (one 256 x 256 unit per CU per round).  Prints us and us per unit-round."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neat_amd import lib, ops

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
kernel = os.environ.get("KERNEL", "t256")
for name, epi, N, K, rbs in [("proj BIAS_RES", lib.EPI_BIAS_RES_F32, 512, 2048, (128, 192, 256, 282, 308, 320, 384)),
                             ("dfc BF16", lib.EPI_BF16, 512, 2048, (128, 192, 231, 256, 308, 384)),
                             ("qkv BIAS_BF16", lib.EPI_BIAS_BF16, 1536, 512, (256, 299, 308, 320, 341)),
                             ("fc BIAS_GELU", lib.EPI_BIAS_GELU, 2048, 512, (288, 308, 320))]:
    W = ops.pack_weight(torch.randn(N, K, generator=g, device=dev) * K ** -0.5)
    bias = torch.randn(N, generator=g, device=dev)
    for rb in rbs:
        M = rb * 256
        A = torch.randn(M, K, generator=g, device=dev).bfloat16()
        res = torch.randn(M, N, generator=g, device=dev) if epi == lib.EPI_BIAS_RES_F32 else None
        out0 = torch.empty(M, N, device=dev, dtype=torch.float32 if epi == lib.EPI_BIAS_RES_F32 else torch.bfloat16)
        out1 = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if epi == lib.EPI_BIAS_GELU else None
        ts = []
        for r in range(12):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ops.gemm(A, W, epi, bias=bias, res=res, out0=out0, out1=out1, kernel=kernel)
            b.record()
            torch.cuda.synchronize()
            if r >= 2:
                ts.append(a.elapsed_time(b) * 1e3)
        ts.sort()
        units = rb * (N // 256)
        print(f"{name:14s} N={N} K={K} row blocks {rb:4d} units {units:5d} rounds {units/256:5.2f}  med {ts[len(ts)//2]:7.1f} us  "
              f"{ts[len(ts)//2]/(units/256):6.1f} us/round")
