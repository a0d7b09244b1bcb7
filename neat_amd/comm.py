"""The library's own RCCL communicator (C ABI ``neat_comm_*``, ``include/neat_hip.h``) from Python.

Opt-in: ``NEAT_COMM=rccl`` makes the data-parallel step (``train_TTA._gather_into``) issue its
collectives through ``libneat_hip.so`` instead of ``torch.distributed``; the default stays ``torch.distributed`` on the
``nccl`` backend, which IS RCCL on ROCm.  The communicator's id travels through the existing process group (any backend),
so the launcher (``torchrun``: one process per GPU) does not change.  Stands in for the reference's
``src/ddp.py:6-70`` for callers without torch.distributed: they exchange the 128-byte id themselves.
"""
from __future__ import annotations

import ctypes
import os

import torch

from . import lib

ENABLED = os.environ.get("NEAT_COMM", "") == "rccl"
_ready = False


def unique_id() -> bytes:
    buf = ctypes.create_string_buffer(128)
    lib.call("neat_comm_unique_id", ctypes.cast(buf, ctypes.c_void_p))
    return buf.raw


def init(id_bytes: bytes, world: int, rank: int):
    """Every rank, with its GPU current (torch.cuda.set_device), the same id."""
    global _ready
    if len(id_bytes) != 128:
        raise RuntimeError("comm.init: the id is 128 bytes (neat_comm_unique_id)")
    buf = ctypes.create_string_buffer(id_bytes, 128)
    lib.call("neat_comm_init", ctypes.cast(buf, ctypes.c_void_p), int(world), int(rank))
    _ready = True


def init_from_process_group():
    """Rank 0 draws the id, the existing torch.distributed group broadcasts it, every rank joins.

    Every rank first probes what it needs alone (the library loads, RCCL resolves, rank 0 can draw an id) and the ranks
    exchange an ok flag through the process group BEFORE the id broadcast: a rank that cannot go on makes ALL ranks raise
    together instead of leaving the others blocked in the broadcast."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    on_gpu = dist.get_backend() == "nccl"
    t = torch.zeros(128, dtype=torch.uint8)
    err = ""
    try:
        lib.load()
        ident = unique_id()                      # resolves RCCL with dlopen on every rank; only rank 0's id is used
        if rank == 0:
            t = torch.frombuffer(bytearray(ident), dtype=torch.uint8).clone()
    except Exception as e:                       # noqa: BLE001 -- reported below, on every rank
        err = f"{type(e).__name__}: {e}"
    ok = torch.tensor([0 if err else 1], dtype=torch.int64)
    oks = torch.empty(world, dtype=torch.int64)
    if on_gpu:
        ok, oks, t = ok.cuda(), oks.cuda(), t.cuda()
    dist.all_gather_into_tensor(oks, ok)
    bad = [r for r, v in enumerate(oks.tolist()) if not v]
    if bad:
        raise RuntimeError(f"comm.init_from_process_group: ranks {bad} cannot create the RCCL communicator"
                           + (f" (this rank: {err})" if err else ""))
    dist.broadcast(t, 0)
    init(bytes(t.cpu().tolist()), world, rank)


def ready() -> bool:
    return _ready


def world() -> int:
    return lib.load().neat_comm_world()


def rank() -> int:
    return lib.load().neat_comm_rank()


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def all_gather_into(out: torch.Tensor, x: torch.Tensor):
    """out (contiguous, world x the bytes of x) <- every rank's x in rank order, on the current stream."""
    x = x.contiguous()
    if not (x.is_cuda and out.is_cuda and out.is_contiguous()):
        raise RuntimeError("comm.all_gather_into: contiguous GPU tensors only")
    nbytes = x.numel() * x.element_size()
    if out.numel() * out.element_size() != nbytes * world():
        raise RuntimeError("comm.all_gather_into: out must hold world x the bytes of x")
    if nbytes:
        lib.call("neat_comm_allgather", ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(out.data_ptr()), nbytes, _stream())


def all_reduce_sum_(x: torch.Tensor):
    """x (float32, contiguous, GPU) <- sum over ranks, in place, on the current stream."""
    if not (x.is_cuda and x.is_contiguous() and x.dtype == torch.float32):
        raise RuntimeError("comm.all_reduce_sum_: contiguous float32 GPU tensors only")
    if x.numel():
        lib.call("neat_comm_allreduce_f32", ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(x.data_ptr()), x.numel(), _stream())


def destroy():
    global _ready
    lib.call("neat_comm_destroy")
    _ready = False
