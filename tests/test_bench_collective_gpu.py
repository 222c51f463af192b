"""The N > 1 path of bench.py on the hardware that exists: `--force-collective` makes a job of ONE rank initialise the
RCCL process group ("nccl" backend), pass the barriers, all-reduce the step time (MAX) and gather the 64-byte per-rank
records on the device — the calls that otherwise first run on the driver's 8-GPU node.  Started as a child process under
torch.distributed.run, the way the driver launches N > 1 (one process on the card: within the box's process guard)."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_single_rank_job_runs_the_collectives(gpu_ctx):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "16",
           "--cpu-sample", "0", "--h2d-steps", "0", "--overlap-pass", "0", "--force-collective"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["value"] > 0
    ranks = line["ranks"]
    assert isinstance(ranks, list) and len(ranks) == 1                      # gathered on the device through RCCL
    assert ranks[0]["n_images"] == 32 and ranks[0]["seconds"] > 0
    assert line["value"] == pytest.approx(32 / ranks[0]["seconds"], rel=1e-3)   # job throughput = all images / slowest rank
