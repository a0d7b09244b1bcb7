"""GPU: the allocation-free entry point neat_coarse_align_static (static device scratch) gives, bit for bit, the
results of neat_coarse_align (caller workspace) -- on two calls in a row, the second with a smaller input -- and
rejects sizes beyond its static scratch.  Collected last (file name): see the note in DESIGN.md §7."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from neat_amd import lib, ops  # noqa: E402

DEV = "cuda:0"


def _unit(n, d, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=-1)


def _both(B, N, D, seed):
    t, p, n, G = (_unit(r, D, seed + i).to(DEV) for i, r in enumerate((B, B, B, N)))
    c_ref, idx_ref, S_ref = ops.coarse_align(t, p, n, G)
    S = torch.full((B, N), float("nan"), device=DEV)
    idx = torch.full((B,), -7, dtype=torch.int32, device=DEV)
    c = torch.full((B, D), float("nan"), device=DEV)
    torch.cuda.synchronize()
    before = torch.cuda.memory_stats(0)["allocation.all.allocated"]
    ops.coarse_align_static(t, p, n, G, S, idx, c)
    after = torch.cuda.memory_stats(0)["allocation.all.allocated"]
    torch.cuda.synchronize()
    assert after == before, "the static entry point went through the allocator"
    assert torch.equal(idx, idx_ref)
    assert torch.equal(S, S_ref) and torch.equal(c, c_ref)


def test_static_scratch_entry_point_equals_workspace_entry_point():
    # B * ceil(N/64) = 256 * 79 slots, then a smaller problem that leaves most of the previous call's partials in place
    _both(256, 5000, 512, 11)
    _both(37, 130, 64, 23)
    # and the larger one again, after the smaller
    _both(182, 777, 768, 5)


def test_static_scratch_entry_point_rejects_oversize():
    L = lib.load()
    mx = L.neat_coarse_static_max_slots()
    assert mx == 1 << 18
    B, N, D = 4096, 8192, 16                      # 4096 * 128 slots = 2 x the limit
    assert B * ((N + 63) // 64) > mx
    rc = L.neat_coarse_align_static(1, 1, 1, 1, 1, 1, 1, B, N, D, None)     # validation precedes any device access
    assert rc == -1 and "static scratch" in lib.last_error()
    rc = L.neat_coarse_align_static(None, 1, 1, 1, 1, 1, 1, 8, 8, 16, None)
    assert rc == -2
