import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neat_amd import lib, ops
dev = "cuda"
def run(M, N, K):
    g = torch.Generator().manual_seed(1)
    A = torch.randint(-1, 2, (M, K), generator=g).float()
    W = torch.randint(-1, 2, (N, K), generator=g).float()
    Ad, Wd = A.bfloat16().to(dev), ops.pack_weight(W.to(dev))
    ref = A.to(dev) @ W.to(dev).t()
    o, _ = ops.gemm(Ad, Wd, lib.EPI_F32, kernel="panel")
    bad = (o != ref)
    print(f"M={M} N={N} K={K}: wrong {int(bad.sum())} of {bad.numel()}")
    if bad.any():
        rows = bad.any(1).nonzero().flatten()
        cols = bad.any(0).nonzero().flatten()
        print(" rows:", rows[:10].tolist(), "...", rows[-5:].tolist(), "count", len(rows))
        print(" cols:", cols[:10].tolist(), "...", cols[-5:].tolist(), "count", len(cols))
        subs = torch.unique(rows // 128)
        print(" sub-panels:", subs.tolist()[:40])
        r0 = int(rows[0])
        print(" row", r0, "bad cols", bad[r0].nonzero().flatten().tolist()[:40])
        print(" got", o[r0, cols[:8]].tolist(), "ref", ref[r0, cols[:8]].tolist())
for shp in ((33280, 256, 768), (33280, 256, 128), (33280, 256, 256), (33280, 256, 384), (40000, 256, 512)):
    run(*shp)
