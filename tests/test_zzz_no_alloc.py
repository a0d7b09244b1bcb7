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
    lib.call("neat_coarse_static_init")                    # set-up for this device (idempotent)
    torch.cuda.synchronize()
    before = torch.cuda.memory_stats(0)["allocation.all.allocated"]
    ops.coarse_align_static(t, p, n, G, S, idx, c)
    after = torch.cuda.memory_stats(0)["allocation.all.allocated"]
    torch.cuda.synchronize()
    assert after == before, "the static entry point went through torch's allocator"
    assert torch.equal(idx, idx_ref)
    assert torch.equal(S, S_ref) and torch.equal(c, c_ref)
    # torch's counter sees only its own caching allocator: a hipMalloc / hipMallocAsync inside the library (or a lazy
    # module load) would show up in the DEVICE's free memory instead.  Compared around a second, warm call.
    free0 = torch.cuda.mem_get_info(0)[0]
    S.fill_(float("nan"))
    ops.coarse_align_static(t, p, n, G, S, idx, c)
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info(0)[0]
    assert free1 == free0, f"device free memory moved by {free0 - free1} bytes across a warm call"
    assert torch.equal(S, S_ref) and torch.equal(idx, idx_ref)


def test_static_scratch_entry_point_equals_workspace_entry_point():
    # B * ceil(N/64) = 256 * 79 slots, then a smaller problem that leaves most of the previous call's partials in place
    _both(256, 5000, 512, 11)
    _both(37, 130, 64, 23)
    # and the larger one again, after the smaller
    _both(182, 777, 768, 5)


def test_static_scratch_entry_point_under_hip_graph_capture():
    """What the entry point is for: captured into a HIP graph after neat_coarse_static_init() (no allocation, no
    synchronous lookup during the capture), replayed on new inputs, equal to the workspace entry point."""
    B, N, D = 64, 1000, 512
    t, p, n, G = (_unit(r, D, 40 + i).to(DEV) for i, r in enumerate((B, B, B, N)))
    S = torch.empty(B, N, device=DEV)
    idx = torch.empty(B, dtype=torch.int32, device=DEV)
    c = torch.empty(B, D, device=DEV)
    lib.call("neat_coarse_static_init")
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ops.coarse_align_static(t, p, n, G, S, idx, c)
    for seed in (50, 60):                                   # new contents in the captured input buffers
        for buf, rows, k in ((t, B, 0), (p, B, 1), (n, B, 2)):
            buf.copy_(_unit(rows, D, seed + k))
        graph.replay()
        torch.cuda.synchronize()
        c_ref, idx_ref, S_ref = ops.coarse_align(t, p, n, G)
        assert torch.equal(idx, idx_ref) and torch.equal(S, S_ref) and torch.equal(c, c_ref)


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
