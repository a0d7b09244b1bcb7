"""Run ONE GEMM shape a few times (for rocprofv3 --pmc runs).  usage: gemm_one.py EPI M N K [iters]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neat_amd import lib, ops
epi, M, N, K = (int(x) for x in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 3
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
A = torch.randn(M, K, generator=g, device=dev).bfloat16()
W = ops.pack_weight(torch.randn(N, K, generator=g, device=dev) * K ** -0.5)
bias = torch.randn(N, generator=g, device=dev)
res = torch.randn(M, N, generator=g, device=dev) if epi == lib.EPI_BIAS_RES_F32 else None
aux = torch.randn(M, N, generator=g, device=dev).bfloat16() if epi == lib.EPI_GELU_BWD else None
for _ in range(iters):
    ops.gemm(A, W, epi, bias=bias, res=res, aux=aux)
torch.cuda.synchronize()
print("done")
