"""Pin the road-network C oracle (oracle/hwy_oracle_net.c) against traces of the unmodified reference's
MergeEnv / MergeGenericEnv, and the product's own lane table + stream-identical reset against the
reference's RoadNetwork and reset(seed=s).

CPU only.  Golden fixtures: tests/golden/merge_*.npz (tests/golden/make_golden_merge.py).
Tolerances as in test_oracle_golden.py: f64 with glibc libm vs numpy's libm; flags / lane indices exact.
"""
import numpy as np
import pytest

from highwayenv_amd import _abi, merge
from oracle import oracle
from tests.golden_util import KNIFE, MERGE, MERGE_CRASH, MERGE_GRID, GoldenMerge, assert_net_state_close


@pytest.mark.parametrize("name", MERGE + MERGE_CRASH)
def test_lane_table_is_the_reference_network(name):
    """hwy_config.net built by highwayenv_amd.merge == the reference's RoadNetwork, bit for bit, in
    get_closest_lane_index order (road.py:55-71)."""
    g = GoldenMerge(name)
    c = g.hwy_config()
    tab = merge.table_from_config(c)
    for k in _abi.LANE_F64 + _abi.LANE_I32:
        np.testing.assert_array_equal(tab[k], g.z["lane_" + k], err_msg=k)
    assert c.num_vehicles == g.N
    assert c.merge_end_x == float(g.z["end_position"])


@pytest.mark.parametrize("name", MERGE + ["merge_crash_generic", "merge_crash_ma4"])
def test_reset_replays_the_reference_stream(name):
    """merge.spawn_reference_stream(seed) == the reference's reset(seed=seed), bit for bit (the reference
    appends vehicles compactly; the engine leaves HWY_F_ABSENT slots where the spawn gave up)."""
    g = GoldenMerge(name)
    c = g.hwy_config()
    st = merge.spawn_reference_stream(c, g.config, g.generic, g.seeds)
    want = g.state("init")
    for e in range(g.E):
        a = (st["flags"][e] & _abi.F_ABSENT) == 0
        b = (want["flags"][e] & _abi.F_ABSENT) == 0
        assert a.sum() == b.sum()
        for k in _abi.STATE_F64 + _abi.STATE_I32:
            np.testing.assert_array_equal(st[k][e][a], want[k][e][b], err_msg=f"{name} env {e}: {k}")


def check_teacher_forced_frames(g):
    """Every single frame, started from the reference's own state: Road.act + Road.step.  Impacts are compared SIGNED wherever
    the collision is well conditioned (oracle.impact_margins >= KNIFE), incl. the vehicle-vs-Obstacle branch."""
    name = g.name
    Ef = g.frames_for
    cfg = g.hwy_config(Ef)
    envs = slice(0, Ef)
    for step in range(g.steps):
        for fr in range(g.T):
            k = step * g.T + fr
            st = g.state("init", envs=envs) if k == 0 else g.state("frame", k - 1)
            acts = g.actions[step, :Ef] if fr == 0 else None
            with oracle.impact_margins(cfg) as m:
                oracle.frames(cfg, st, acts, 1)
            assert_net_state_close(st, g.state("frame", k), atol=1e-10, what=f"{name} step {step} frame {fr}",
                                   signed=m.margin >= KNIFE)


@pytest.mark.parametrize("name", MERGE + MERGE_CRASH)
def test_oracle_teacher_forced_frames(name):
    check_teacher_forced_frames(GoldenMerge(name))


def check_free_running_steps(g):
    """Whole episodes from the reset state, compared while the episode is live (up to and including the
    terminal step; see DESIGN.md section 4 on post-termination wrecks)."""
    name = g.name
    cfg = g.hwy_config()
    st = g.state("init")
    np.testing.assert_allclose(oracle.observe(cfg, st), g.z["obs0"], rtol=0, atol=1e-6)
    live = np.ones(g.E, bool)
    for t in range(g.steps):
        with oracle.impact_margins(cfg) as m:
            obs, reward, term, trunc, info = oracle.step(cfg, st, g.actions[t])
        what = f"{name} step {t}"
        np.testing.assert_array_equal(term[live], g.z["terminated"][t].astype(bool)[live], err_msg=what)
        np.testing.assert_array_equal(trunc[live], g.z["truncated"][t].astype(bool)[live], err_msg=what)
        # a wreck resting EXACTLY touching a third body (flag_margin < KNIFE: whether that pair "will intersect", and with its ~0
        # translation replaces a real pending impact, hinges on the last bit -- tests/test_net_parity.py): a free-running episode
        # leaves the comparison there (live-reference case 23; every frame of it is still pinned teacher-forced at 1e-10)
        live = live & ~(np.asarray(m.flag_margin) < KNIFE).any(1)
        np.testing.assert_allclose(obs[live], g.z["obs"][t][live], rtol=0, atol=1e-6, err_msg=what)
        # the reference evaluates `action in [0, 2]` on the joint action tuple (never true) when A > 1
        ok = live & ((g.A == 1) | ~np.isin(g.actions[t, :, 0], [0, 2]))
        np.testing.assert_allclose(reward[ok, 0], g.z["reward"][t][ok], rtol=0, atol=1e-9, err_msg=what)
        np.testing.assert_allclose(info["speed"][live, 0], g.z["info_speed"][t][live], rtol=0, atol=1e-9, err_msg=what)
        np.testing.assert_array_equal(info["crashed"][live, 0], g.z["info_crashed"][t].astype(bool)[live], err_msg=what)
        want = g.state("step", t)
        sub = lambda d: {k: v[live] for k, v in d.items()}  # noqa: E731
        assert_net_state_close(sub(st), sub(want), atol=1e-8, what=what, signed=(m.margin >= KNIFE)[live])
        live &= ~g.z["terminated"][t].astype(bool)
    assert not live.all() or g.steps < 12  # the fixtures do reach termination
    return int(live.sum())


@pytest.mark.parametrize("name", MERGE + MERGE_GRID + MERGE_CRASH)
def test_oracle_free_running_steps(name):
    check_free_running_steps(GoldenMerge(name))


@pytest.mark.parametrize("name", ["merge_v1", "merge_generic_v1"])
def test_connected_lanes_flag_is_load_bearing(name):
    """merge-v1 / merge-generic-v1 (neighbour_vehicles_connected_lanes, road.py:508-529): without the flag the
    restatement must NOT reproduce the reference's v1 frames -- i.e. the fixtures do exercise the connected search."""
    g = GoldenMerge(name)
    cfg = g.hwy_config(g.frames_for)
    assert cfg.flags & _abi.C_CONNECTED_LANES
    cfg.flags &= ~_abi.C_CONNECTED_LANES
    worst = 0.0
    for k in range(g.steps * g.T):
        st = g.state("init", envs=slice(0, g.frames_for)) if k == 0 else g.state("frame", k - 1)
        oracle.frames(cfg, st, g.actions[k // g.T, :g.frames_for] if k % g.T == 0 else None, 1)
        want = g.state("frame", k)
        pres = (want["flags"] & _abi.F_ABSENT) == 0
        worst = max(worst, float(np.abs(st["speed"] - want["speed"])[pres].max()))
    assert worst > 1e-3


def test_connected_lane_masks():
    """hwy_lane.connected on the merge-v0 table [ab0 ab1 | bc0 bc1 lbc | cd0 cd1 | jk | kb]: lane id (else 0) of the road
    leaving `_to` and of every road arriving at `_from`."""
    from highwayenv_amd import merge
    masks = merge.connected_masks(merge.lane_table(merge.merge_default_config(), generic=False))
    bit = lambda *ks: sum(1 << k for k in ks)  # noqa: E731
    assert masks == [bit(0, 2), bit(1, 3),                       # a->b: next b->c same id
                     bit(2, 5, 0, 8), bit(3, 6, 1, 8),           # b->c: next c->d, prev a->b same id, prev k->b lane 0
                     bit(4, 5, 0, 8),                            # lbc (id 2): c->d has 2 lanes -> lane 0; a->b lane 0; k->b
                     bit(5, 2), bit(6, 3),                       # c->d: prev b->c same id
                     bit(7, 8),                                  # j->k: next k->b
                     bit(8, 2, 7)]                               # k->b (id 0): next b->c LANE 0, prev j->k
