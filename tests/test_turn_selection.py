"""The issue-priority turn is scheduling only (hwy_wave.h: WaveTurn), and the engine picks its length itself (hwy_get_prio_turn):
whatever the turn -- off, a power of two, a multiple of 64 ticks, or the engine's own selection, which steps through five candidates
on its first launches -- every output and the final state are the same, bit for bit.  The reference has no counterpart (one CPU
thread, envs/common/abstract.py:287-317); the configurations are BASELINE's headline shape and a `configure()`d one
(envs/highway_env.py:25-53)."""
import numpy as np
import pytest

from highwayenv_amd import _abi

pytestmark = pytest.mark.gpu


def _run(cfg_dict, fast, E, steps, prio):
    from highwayenv_amd.engine import Engine
    cfg = _abi.make_config(dict(cfg_dict, tuning={"prio_shift": prio} if prio else {}), E, fast=fast)
    eng = Engine(cfg)
    kw = {"ego_spacing": cfg_dict["ego_spacing"], "vehicles_density": cfg_dict["vehicles_density"]}
    eng.reset(base_seed=4242, **kw)
    eng.set_autoreset(True, base_seed=777, **kw)
    rng = np.random.default_rng(3)
    outs = []
    for _ in range(steps):
        obs, rew, term, trunc, info = eng.step(rng.integers(0, 5, size=(E, cfg.num_agents)).astype(np.int32))
        outs.append((obs.copy(), rew.copy(), term.copy(), trunc.copy()))
    st = eng.get_state()
    turn = eng.prio_turn()
    eng.close()
    return outs, st, turn


@pytest.mark.parametrize("shape", ["headline", "configured"])
def test_results_do_not_depend_on_the_turn_and_the_engine_selects_one(shape):
    if shape == "headline":
        cfg, fast, E = _abi.highway_fast_default_config(), True, 4096
        cfg.update({"vehicles_count": 50, "lanes_count": 4})
    else:  # another (E, N, L, T) than any the defaults were swept on
        cfg, fast, E = _abi.highway_default_config(), False, 1536
        cfg.update({"vehicles_count": 33, "lanes_count": 3, "simulation_frequency": 10, "duration": 20})
    steps = 270  # three stages of 85 launches at most
    ref, st_ref, turn_ref = _run(cfg, fast, E, steps, -1)
    assert turn_ref == (0, 0)   # off: no turn, no selection
    for prio in (14, 320, 0):
        outs, st, (turn, state) = _run(cfg, fast, E, steps, prio)
        for t, (a, b) in enumerate(zip(outs, ref)):
            for x, y, name in zip(a, b, ("obs", "reward", "terminated", "truncated")):
                np.testing.assert_array_equal(x, y, err_msg=f"prio_shift {prio}, step {t}: {name}")
        for k in st_ref:
            np.testing.assert_array_equal(st[k], st_ref[k], err_msg=f"prio_shift {prio}: state {k}")
        if prio:
            assert (turn, state) == (prio, 0)   # an explicit value is never touched
        else:
            assert state == 2 and 64 <= turn <= (1 << 20), (turn, state)  # the engine chose, within the validated range
