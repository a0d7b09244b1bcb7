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
