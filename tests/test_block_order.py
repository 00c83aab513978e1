"""hwy_set_block_order: which workgroup of the one-wavefront launch steps which environment is a PLACEMENT (environments are
independent, the dispatcher puts workgroup b on the same SIMD launch after launch) -- it must never change a result."""
import numpy as np
import pytest

from highwayenv_amd import _abi
from tests.backends import BACKENDS, make_engine


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("fast", [True, False], ids=["highway-fast", "highway-v0"])
def test_any_block_order_gives_the_same_bits(backend, fast):
    cfg_d = _abi.highway_fast_default_config() if fast else _abi.highway_default_config()
    cfg_d.update({"vehicles_count": 20, "lanes_count": 3, "duration": 4})
    E, steps = (6, 6) if backend == "emu" else (512, 12)
    cfg = _abi.make_config(cfg_d, E, fast=fast)
    a, b = make_engine(backend, cfg), make_engine(backend, cfg)
    rng = np.random.default_rng(4)
    for eng in (a, b):
        eng.reset(base_seed=8)
        eng.set_autoreset(True, base_seed=9)
    for t in range(steps):
        if t % 3 == 0:
            b.set_block_order(rng.permutation(E))
        if t == steps - 2:
            b.set_block_order(None)  # back to the identity
        acts = rng.integers(0, 5, size=(E, 1)).astype(np.int32)
        ra, rb = a.step(acts), b.step(acts)
        for x, y in zip(ra[:4], rb[:4]):
            np.testing.assert_array_equal(x, y)
        for k in ("speed", "crashed"):
            np.testing.assert_array_equal(ra[4][k], rb[4][k])
    sa, sb = a.get_state(), b.get_state()
    for f in sa:
        np.testing.assert_array_equal(sa[f], sb[f], err_msg=f)
    for eng in (a, b):
        eng.close()


@pytest.mark.gpu
def test_block_order_must_be_a_permutation_of_the_one_wavefront_kernel():
    from highwayenv_amd.engine import Engine, EngineError
    cfg_d = _abi.highway_fast_default_config()
    eng = Engine(_abi.make_config(cfg_d, 8, fast=True))
    with pytest.raises(EngineError, match="permutation"):
        eng.set_block_order([0, 1, 2, 3, 4, 5, 6, 6])
    with pytest.raises(ValueError):
        eng.set_block_order([0, 1])
    eng.close()
    wide = _abi.highway_default_config()
    wide.update({"vehicles_count": 100})
    eng = Engine(_abi.make_config(wide, 4, fast=False))   # two wavefronts per environment: the workgroup kernel
    with pytest.raises(EngineError, match="one-wavefront"):
        eng.set_block_order([3, 2, 1, 0])
    eng.close()
