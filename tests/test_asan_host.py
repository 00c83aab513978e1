"""The host side of the engine library under AddressSanitizer: libhwy_engine_asan.so = hwy_engine.hip /
hwy_comm.hip instrumented, the kernels as shipped (highwayenv_amd.build.build_engine_asan).  The ABI tests -- and, on the GPU box,
a parity test, the rollout and the device-reset tests, which exercise the staging buffers, the state packing and the event
bookkeeping -- run in a subprocess with ROCm clang's ASan runtime preloaded; any report fails the test."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_under_asan(selection):
    from highwayenv_amd import build
    try:
        rt = build.asan_runtime()
    except RuntimeError as ex:
        pytest.skip(str(ex))
    lib = build.build_engine_asan()
    env = dict(os.environ, LD_PRELOAD=rt, HWY_ENGINE_LIB=lib,
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:exitcode=99:protect_shadow_gap=0")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", *selection], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1500)
    out = r.stdout + r.stderr
    assert "AddressSanitizer" not in out, out[-4000:]
    assert r.returncode == 0, out[-4000:]
    return out


def test_abi_argument_paths_under_asan():
    out = _run_under_asan(["tests/test_abi.py", "-m", "not gpu"])
    assert " passed" in out


@pytest.mark.gpu
def test_engine_paths_under_asan_on_the_gpu():
    out = _run_under_asan(["tests/test_abi.py", "tests/test_rollout.py", "tests/test_device_reset.py", "tests/test_block_order.py",
                           "tests/test_engine_parity.py", "-m", "gpu", "-k", "not emu"])
    assert " passed" in out
