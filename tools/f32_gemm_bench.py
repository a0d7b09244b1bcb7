"""Exact-fp32 MFMA similarity GEMM (neat_gemm_f32_nt) at the retrieval shapes: us per launch, TFLOP/s and the fraction of
the 157 TFLOP/s fp32 matrix peak.  Shapes: the full 25 014 x 5 000 x 512 score matrix and the chunk shapes neat_sim_topk uses."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neat_amd import ops

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
def med(fn, n=10):
    ts = []
    for r in range(n + 2):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        if r >= 2:
            ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]
shapes = [(25014, 5000, 512), (5000, 25014, 512), (3328, 5000, 512), (3200, 5000, 512), (640, 25014, 512), (25014, 5000, 768)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
for M, N, K in shapes:
    A = torch.nn.functional.normalize(torch.randn(M, K, generator=g, device=dev), dim=-1)
    B = torch.nn.functional.normalize(torch.randn(N, K, generator=g, device=dev), dim=-1)
    ops.gemm_f32_nt(A, B)
    t, tmin = med(lambda: ops.gemm_f32_nt(A, B))
    fl = 2.0 * M * N * K
    print(f"gemm_f32_nt {M:6d} x {N:6d} x {K:4d}: med {t:8.1f} us  min {tmin:8.1f}  {fl/t/1e6:6.1f} TFLOP/s = {fl/t/1e6/157:.3f} of 157", flush=True)
