"""GPU: the data-parallel call path of bench.py through RCCL, rehearsed at world size 1 (collected last).

The one-GPU box cannot hold two RCCL ranks, so NEAT_FORCE_COLLECTIVES=1 makes a single rank launched by
torch.distributed.run go through everything an N-GPU run does: process group on backend nccl with device_id, the gallery
all-gather, both all-gathers of every TTA step, the barrier and the MAX all-reduce around the timed region.  The run must
produce the bench line with the same losses as a run without collectives (world size 1: gathering is the identity)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--gpus", "1", "--steps", "1", "--warmup", "1", "--gallery", "500", "--no-cpu-baseline", "--no-packed"]


def _bench_line(cmd, env):
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.mark.timeout(600)
def test_bench_runs_through_rccl_at_world_size_one():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, NEAT_FORCE_COLLECTIVES="1")
    forced = _bench_line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py"] + ARGS, env)
    plain = _bench_line([sys.executable, "bench.py"] + ARGS, dict(os.environ))
    assert forced["n_gpus"] == 1 and forced["config"]["parallelism"] == "dp1"
    assert forced["last_losses"] == plain["last_losses"]
    assert forced["value"] > 0.5 * plain["value"]


@pytest.mark.timeout(600)
def test_bench_runs_through_the_librarys_own_communicator():
    """NEAT_COMM=rccl: the same rehearsal with every GPU collective of the step issued by libneat_hip.so's own RCCL
    communicator (neat_comm_* of the C ABI; id broadcast through the torch.distributed group)."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, NEAT_FORCE_COLLECTIVES="1", NEAT_COMM="rccl")
    own = _bench_line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                       "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py"] + ARGS, env)
    plain = _bench_line([sys.executable, "bench.py"] + ARGS, dict(os.environ))
    assert own["last_losses"] == plain["last_losses"]
    assert own["value"] > 0.5 * plain["value"]


def test_comm_c_abi_world_one():
    """neat_comm_* directly: id, init at world size 1, all-gather and all-reduce are the identity, clear errors before
    init and after destroy."""
    import torch
    from neat_amd import comm, lib
    if comm.ready():
        comm.destroy()
    x = torch.arange(1000, dtype=torch.float32, device="cuda")
    with pytest.raises(RuntimeError, match="no communicator"):
        comm.all_reduce_sum_(x.clone())
    uid = comm.unique_id()
    assert len(uid) == 128 and any(uid)
    comm.init(uid, 1, 0)
    try:
        assert comm.world() == 1 and comm.rank() == 0
        with pytest.raises(RuntimeError, match="already exists"):
            comm.init(uid, 1, 0)
        out = torch.empty_like(x)
        comm.all_gather_into(out, x)
        y = x.clone()
        comm.all_reduce_sum_(y)
        b = torch.randint(0, 255, (4097,), dtype=torch.uint8, device="cuda")      # odd byte count
        ob = torch.empty_like(b)
        comm.all_gather_into(ob, b)
        torch.cuda.synchronize()
        assert torch.equal(out, x) and torch.equal(y, x) and torch.equal(ob, b)
    finally:
        comm.destroy()
    assert lib.load().neat_comm_world() == 0
