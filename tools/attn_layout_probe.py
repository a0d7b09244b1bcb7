import os, sys, torch
sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from neat_amd import ops
dev = "cuda"; L = 77
g = torch.Generator(device=dev).manual_seed(0)
def med(fn, n=20):
    ts = []
    for r in range(n + 3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        if r >= 3: ts.append(a.elapsed_time(b) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
for heads, nf, nb in ((8, 1024, 768), (1, 8192, 6144), (2, 4096, 3072)):
    C = heads * 64
    qkv = (torch.randn(nf * L, 3 * C, generator=g, device=dev) * 0.5).bfloat16()
    do = (torch.randn(nb * L, C, generator=g, device=dev) * 0.1).bfloat16()
    o, lse = ops.attn_fwd(qkv, nf, L, heads, True)
    tf = med(lambda: ops.attn_fwd(qkv, nf, L, heads, True))
    tb = med(lambda: ops.attn_bwd(qkv, do, lse, nb, L, heads, True))
    print(f"heads={heads} nseq={nf}/{nb}: fwd {tf:.1f} us  bwd {tb:.1f} us", flush=True)
