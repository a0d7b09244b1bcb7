"""GPU diagnostic: exact-integer GEMMs on small shapes; prints which 16x16 output tiles differ (tile256 bring-up)."""
import sys, torch
sys.path.insert(0, ".")
from neat_amd import lib, ops
dev = "cuda:0"
KERN = sys.argv[1] if len(sys.argv) > 1 else "t256"
def run(M, N, K, epi=lib.EPI_F32):
    g = torch.Generator().manual_seed(1)
    A = torch.randint(-2, 3, (M, K), generator=g).float()
    W = torch.randint(-2, 3, (N, K), generator=g).float()
    ref = A @ W.t()
    o, _ = ops.gemm(A.bfloat16().to(dev), ops.pack_weight(W.to(dev)), epi, kernel=KERN)
    o = o.float().cpu()
    bad = (o != ref)
    print(f"M={M} N={N} K={K} epi={epi}: wrong {int(bad.sum())}/{bad.numel()}")
    if bad.any():
        Mt, Nt = (M + 15) // 16, N // 16
        pad = torch.zeros(Mt * 16, N, dtype=torch.bool); pad[:M] = bad
        tiles = pad.view(Mt, 16, Nt, 16).any(3).any(1)
        for r in range(min(Mt, 32)):
            print("".join("X" if tiles[r, c] else "." for c in range(min(Nt, 64))))
        # try to explain: is the wrong value a partial sum over some K-tiles?
        r, c = [int(x[0]) for x in torch.nonzero(bad, as_tuple=True)]
        parts = [(A[r, k:k + 64] * W[c, k:k + 64]).sum().item() for k in range(0, K, 64)]
        print(f" first bad ({r},{c}): got {o[r, c].item()} want {ref[r, c].item()} per-K-tile partial sums {parts}")
for shp in [(128, 256, 512), (256, 256, 512), (384, 256, 512), (256, 512, 512), (1000, 512, 512), (70000, 512, 512)]:
    run(*shp)
run(1000, 512, 512, lib.EPI_BF16)
