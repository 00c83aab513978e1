"""bench.py's launcher path (CPU, no GPU needed): `python bench.py --gpus N` started WITHOUT a torch.distributed environment must
re-launch itself under torch.distributed.run with N ranks on 127.0.0.1 -- the form the driver uses for N = 1 must not fail at
the launcher for N = 2, 4, 8 -- and must NOT re-launch when it already runs as a rank."""
import subprocess
import sys

import pytest
import torch

import bench


def _run_main(monkeypatch, argv, device_count, env=None):
    calls = []
    monkeypatch.setattr(torch.cuda, "device_count", lambda: device_count)
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: calls.append((cmd, env)) or 0)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    for k, v in (env or {}).items():
        monkeypatch.setenv(k, v)
    monkeypatch.setattr(sys, "argv", ["bench.py", *argv])
    with pytest.raises(SystemExit) as ex:
        bench.main()
    return calls, ex.value.code


@pytest.mark.parametrize("n", [2, 4, 8])
def test_gpus_n_without_a_dist_environment_self_launches(monkeypatch, n):
    argv = ["--gpus", str(n), "--steps", "20", "--warmup", "5", "--scaling", "strong"]
    calls, code = _run_main(monkeypatch, argv, device_count=8)
    assert code == 0 and len(calls) == 1
    cmd, env = calls[0]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and f"--nproc-per-node={n}" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 0 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    k = cmd.index(bench.__file__ if bench.__file__ in cmd else [c for c in cmd if c.endswith("bench.py")][0])
    assert cmd[k + 1:] == argv            # every flag travels to the ranks unchanged
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"   # dmabuf IPC: RCCL across processes needs it on this driver


def test_too_few_gpus_is_a_clear_error_not_a_launch(monkeypatch):
    calls, code = _run_main(monkeypatch, ["--gpus", "8"], device_count=1)
    assert not calls and "only 1 GPU" in str(code)


def test_a_rank_does_not_relaunch(monkeypatch):
    """Inside torch.distributed.run (WORLD_SIZE set) --gpus N is the world size, not a request to launch."""
    calls, code = _run_main(monkeypatch, ["--gpus", "2"], device_count=0, env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert not calls and "MI355X" in str(code)     # falls through to "needs a GPU" here in the build container
    calls, code = _run_main(monkeypatch, ["--gpus", "4"], device_count=0, env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert not calls and "WORLD_SIZE=2" in str(code)


def test_single_gpu_default_never_launches(monkeypatch):
    calls, code = _run_main(monkeypatch, [], device_count=0)
    assert not calls and "MI355X" in str(code)


def test_strong_scaling_blocks_cover_the_total():
    from highwayenv_amd.dist import shard_range
    for world in (1, 2, 4, 8):
        blocks = [shard_range(4096, world, r) for r in range(world)]
        assert sum(len(b) for b in blocks) == 4096 and all(len(b) == 4096 // world for b in blocks)
        assert [b.start for b in blocks] == [r * (4096 // world) for r in range(world)]
