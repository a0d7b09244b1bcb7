"""Forward attention at the step's sizes (text: 1 024 x 8 heads, L = 77 causal; packed; vision L = 50), written to a file or
compared with one: run once with NEAT_ATTN_WS=1 and once without, the outputs must be equal bit for bit."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neat_amd import ops
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(5)
res = {}
for name, nseq, L, heads, causal, hm in (("text", 1024, 77, 8, True, True), ("text_plain", 1030, 77, 8, True, False),
                                         ("vision", 520, 50, 12, False, True), ("text_L", 300, 77, 12, True, True)):
    C = heads * 64
    qkv = (torch.randn(nseq * L, 3 * C, generator=g, device=dev) * 0.5).bfloat16()
    o, lse = ops.attn_fwd(qkv, nseq, L, heads, causal, head_major=hm)
    res[name] = (o.cpu(), lse.cpu())
# packed batch: ragged lengths
lens = torch.randint(3, 78, (1100,), generator=torch.Generator().manual_seed(3))
cu = torch.zeros(1101, dtype=torch.int32); cu[1:] = torch.cumsum(lens, 0)
rows = int(cu[-1])
qkv = (torch.randn(rows, 3 * 512, generator=g, device=dev) * 0.5).bfloat16()
o, lse = ops.attn_fwd(qkv, 1100, 77, 8, True, cu=cu.to(dev), head_major=True)
valid = (torch.arange(77)[None, :] < lens[:, None])            # lse beyond a sequence's length is not written
res["packed"] = (o.cpu(), torch.where(valid[:, None, :], lse.cpu(), torch.zeros(())))
path = sys.argv[2]
if sys.argv[1] == "save":
    torch.save(res, path)
else:
    ref = torch.load(path, weights_only=True)
    for k in res:
        assert torch.equal(res[k][0], ref[k][0]) and torch.equal(res[k][1], ref[k][1]), k
    print("equal:", ", ".join(res))
