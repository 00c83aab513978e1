"""The multi-GPU path on real devices: tests/dist_selftest.py under torch.distributed.run.

With one visible GPU (what `gpurun` offers) the same script runs at world size 1 -- the collective code path (RCCL through
torch.distributed, and RCCL through the C-ABI's hwy_comm_init / hwy_gather) is still executed; with two or more visible
GPUs it runs with two ranks and rank 0 checks every rank's block bit for bit.  The sharding / packing logic itself is
covered on CPU at world size 2 by tests/test_dist_gloo.py (gloo)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(world: int, comm: str, port: int):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    script = os.path.join(ROOT, "tests", "dist_selftest.py")
    if world == 1:
        cmd = [sys.executable, script, "--comm", comm]
        env.update(MASTER_PORT=str(port))
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), script, "--comm", comm]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "SELFTEST OK" in r.stdout, r.stdout[-2000:]
    return r.stdout


@pytest.mark.parametrize("comm", ["torch", "abi"])
def test_gather_path_single_rank(comm):
    out = _run(1, comm, 29551 if comm == "torch" else 29552)
    assert "world=1" in out


@pytest.mark.parametrize("comm", ["torch", "abi"])
def test_gather_path_two_ranks(comm):
    from highwayenv_amd.engine import device_count
    if device_count() < 2:
        pytest.skip("needs two visible GPUs (gpurun offers one; the driver's multi-GPU box runs this)")
    out = _run(2, comm, 29553 if comm == "torch" else 29554)
    assert "world=2" in out and "[501, 500]" in out
