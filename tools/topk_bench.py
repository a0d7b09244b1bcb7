"""Retrieval scoring at BASELINE config-2 size (25 014 captions x 5 000 images): time per kernel launch (HIP events) and the
algorithmic HBM rate, both directions.  Run under `rocprofv3 --kernel-trace --stats` / `--pmc FETCH_SIZE` for the
profiles/ summaries (tools/profile_round.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neat_amd import ops  # noqa: E402
from neat_amd.evaluation import retrieval as R  # noqa: E402

dev = "cuda"
NT, NI, D = 25014, 5000, 512
g = torch.Generator(device=dev).manual_seed(0)
txt = torch.nn.functional.normalize(torch.randn(NT, D, generator=g, device=dev), dim=-1)
img = torch.nn.functional.normalize(torch.randn(NI, D, generator=g, device=dev), dim=-1)
owner = (torch.arange(NT, device=dev) // 5).clamp_max(NI - 1).to(torch.int32)


def timed(fn, n=10):
    for _ in range(2):
        out = fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        out = fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3, out


us, scores = timed(lambda: ops.gemm_f32_nt(txt, img), 5)
print(f"similarity GEMM fp32 exact (MFMA f32) {NT}x{NI}x{D}: {us:9.1f} us  {2.0 * NT * NI * D / us / 1e6:7.1f} TFLOP/s")
byts = NT * NI * 4
us, (_, it) = timed(lambda: ops.topk_rows(scores, 10))
print(f"top-10 rows  (t2i, {NT} rows x {NI}):      {us:9.1f} us  {byts / us / 1e6:6.2f} TB/s of 8 = {byts / us / 8e6:.3f}")
us, (_, ii) = timed(lambda: ops.topk_rows(scores.T, 10))
print(f"top-10 cols  (i2t, {NI} rows x {NT}, view): {us:9.1f} us  {byts / us / 1e6:6.2f} TB/s of 8 = {byts / us / 8e6:.3f}")
wide = torch.empty(1280, (NT + 3) // 4 * 4, device=dev)[:, :NT]      # one fused-i2t chunk: 1 280 images x 25 014 captions
wide.copy_(scores.T[:1280])
us, _ = timed(lambda: ops.topk_rows(wide, 10))
wb = 1280 * NT * 4
print(f"top-10 rows  (1280 contiguous rows x {NT}, 5 segments + merge): {us:9.1f} us  {wb / us / 1e6:6.2f} TB/s (chunk is cache-resident)")
us, _ = timed(lambda: ops.sim_topk(txt, img, 10), 3)
print(f"fused similarity + top-10 (t2i, scores never in HBM as a whole):        {us:9.1f} us")
us, _ = timed(lambda: ops.sim_topk(img, txt, 10), 3)
print(f"fused similarity + top-10 (i2t, gallery = the captions):                {us:9.1f} us")
per_image = torch.bincount(owner.long(), minlength=NI).to(torch.int32)
us, _ = timed(lambda: (ops.recall_owner_topk(it, (1, 5, 10), row_owner=owner),
                       ops.recall_owner_topk(ii, (1, 5, 10), col_owner=owner, npos=per_image)))
print(f"recall@1,5,10 both directions from the two lists (owner vectors): {us:9.1f} us")
us, m = timed(lambda: R.recall_metrics(scores, None, (1, 5, 10), texts_image_index=owner), 5)
print(f"recall_metrics (2 passes over the matrix, 6 numbers, host .item()s included): {us:9.1f} us")
pos = torch.zeros(NT, NI, dtype=torch.bool, device=dev)
pos[torch.arange(NT, device=dev), owner.long()] = True
us, m2 = timed(lambda: R.recall_metrics(scores, pos, (1, 5, 10)), 3)
print(f"recall_metrics, dense bool positives (compatibility path):                   {us:9.1f} us")
assert m == m2, (m, m2)
us, _ = timed(lambda: (ops.best_positive_rank_owner(scores, row_owner=owner), ops.best_positive_rank_owner(scores.T, col_owner=owner)), 3)
print(f"best-positive ranks both directions (medR support, owner vectors): {us:9.1f} us")
print(m)

# how many rows the fast tier could not prove (they are redone by the exact kernels): the flags are the first R int32 of the
# workspace
from neat_amd import lib  # noqa: E402
for name, view in (("t2i", scores), ("i2t", scores.T)):
    Rr, Nn = view.shape
    sr, sc = view.stride()
    ws = torch.zeros(lib.load().neat_topk_ws_bytes(Rr, Nn, 10), dtype=torch.uint8, device=dev)
    idx = torch.empty(Rr, 10, dtype=torch.int32, device=dev)
    lib.call("neat_topk_rows", view.data_ptr(), sr, sc, Rr, Nn, 10, idx.data_ptr(), None, ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    flags = ws[: Rr * 4].view(torch.int32)
    print(f"{name}: rows redone by the exact tier: {int((flags != 0).sum())} of {Rr}")
