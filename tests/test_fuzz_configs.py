"""Randomised configurations of the straight-road scenarios against the oracle (MI355X only: 40 configurations x 8
envs x 8 steps take seconds there and minutes on the CPU emulation).  Lanes, traffic size, agents, frequencies, speed
ladders, observation variants, reward / termination switches are drawn from a seeded generator; the comparison is the
one of tests/test_edge_cases.py (flags exact, obs 1e-6, reward 1e-9, state 1e-7 on live, wreck-free episodes)."""
import os

import numpy as np
import pytest

from highwayenv_amd import _abi
from tests.test_edge_cases import rollout

pytestmark = pytest.mark.gpu
# HWY_FUZZ_BACKEND=emu replays a chunk on the CPU emulator of the kernel source (tests/emu): `pytest -m gpu` with that variable set
# needs no GPU -- how a failing chunk of a large GPU run (HWY_FUZZ_CHUNKS) is taken apart
BACKEND = os.environ.get("HWY_FUZZ_BACKEND", "hip")
# HWY_FUZZ_CHUNKS chunks starting at HWY_FUZZ_FIRST (a chunk's seed is its number: large runs continue where the last one ended).
# The default -- what the round driver's `pytest -m gpu` runs -- is 100 chunks per family: ~0.23 s each on the MI355X.
CHUNKS = range(int(os.environ.get("HWY_FUZZ_FIRST", "0")), int(os.environ.get("HWY_FUZZ_FIRST", "0")) + int(os.environ.get("HWY_FUZZ_CHUNKS", "100")))

# whole-step coverage of the intersection fuzz (printed per chunk): the rest are env-steps in which some car is below 1 m/s.
# A per-chunk statistic over 6 random configurations: of 150 chunks on the GPU one fell to 38.8 %, the others stay above 40 %.
INTERSECTION_WHOLE_STEP_FLOOR = 0.33
# per-chunk ceilings of the counted knife-edge cases of the intersection fuzz (4 configurations x 12 envs x 10 steps x 15 frames =
# ~7 000 teacher-forced frames and ~450 env-steps per chunk).  Measured over chunks 0..199 on the MI355X (profiles/r04_history.md):
# whole steps diverging on a touching pair 0; touching-pair slots 16 in all, 12 of them ONE resting pair of chunk 11 seen in 12
# consecutive frames; last-bit lane indices in a frame 0; lane flips in a whole step 0; queue-order cuts 3 in all, at most 2 in
# a chunk.  The ceilings are those maxima plus a small margin (a resting pair may last a whole 15-frame step):
EDGE_MAX, TOUCH_MAX, LANE_FRAMES_MAX, FLIP_MAX, CUT_MAX = 1, 15, 2, 2, 4
# queue-order cuts among the end-of-step products of EVERY live env-step (queues included: that is where the cuts are)
CUT_P_MAX = 2  # (0 in 144 chunks on the emulator)


def random_config(rng):
    fast = bool(rng.integers(2))
    cfg = _abi.highway_fast_default_config() if fast else _abi.highway_default_config()
    lanes = int(rng.integers(1, 7))
    agents = int(rng.integers(1, 4))
    sim = int(rng.choice([5, 10, 15]))
    feats = [["presence", "x", "y", "vx", "vy"], ["presence", "x", "y", "vx", "vy", "cos_h", "sin_h"],
             ["x", "y", "heading", "vx"]][int(rng.integers(3))]
    obs = {"type": "Kinematics", "vehicles_count": int(rng.integers(2, 9)), "features": feats,
           "absolute": bool(rng.integers(2)), "see_behind": bool(rng.integers(2)), "normalize": bool(rng.integers(2)),
           "clip": bool(rng.integers(2)),
           "order": "shuffled" if rng.integers(4) == 0 else "sorted"}  # (the engine's part: list-order selection)
    if rng.integers(4) == 0:  # OccupancyGrid (borders kept off the waypoint lattice, see the intersection test below)
        gfeats = [["presence", "vx", "vy", "on_road"], ["presence", "x", "y", "vx", "vy", "cos_h", "sin_h", "on_road"]][int(rng.integers(2))]
        obs = {"type": "OccupancyGrid", "features": gfeats, "grid_size": [[-31.3, 28.7], [-13.3, 11.7]],
               "grid_step": [float(rng.choice([2.5, 5]))] * 2, "align_to_vehicle_axes": bool(rng.integers(2)),
               "clip": bool(rng.integers(2)), "as_image": bool(rng.integers(2))}
    cfg.update({"lanes_count": lanes, "vehicles_count": int(rng.integers(0, 70)), "controlled_vehicles": agents,
                "simulation_frequency": sim, "policy_frequency": int(rng.choice([1, 5]) if sim % 5 == 0 else 1),
                "duration": int(rng.integers(4, 12)), "vehicles_density": float(rng.uniform(0.8, 2.0)),
                "ego_spacing": float(rng.uniform(1.0, 2.5)), "normalize_reward": bool(rng.integers(2)),
                "offroad_terminal": bool(rng.integers(2)), "collision_reward": float(rng.uniform(-2, 0)),
                "right_lane_reward": float(rng.uniform(0, 0.5)), "high_speed_reward": float(rng.uniform(0.1, 1)),
                "observation": obs if agents == 1 else {"type": "MultiAgentObservation", "observation_config": obs}})
    if agents > 1:
        cfg["action"] = {"type": "MultiAgentAction", "action_config": {"type": "DiscreteMetaAction"}}
    if rng.integers(3) == 0:
        tgt = {"type": "DiscreteMetaAction", "target_speeds": sorted(rng.uniform(10, 35, size=int(rng.integers(2, 6))).tolist())}
        table = int(rng.integers(4))  # ACTIONS_ALL twice as often as the lateral-only / longitudinal-only tables
        if table == 2:
            tgt["longitudinal"] = False
        elif table == 3:
            tgt["lateral"] = False
        cfg["action"] = tgt if agents == 1 else {"type": "MultiAgentAction", "action_config": tgt}
    return cfg, fast


@pytest.mark.parametrize("chunk", CHUNKS)  # 10 configurations each
def test_random_configurations_vs_oracle(chunk):
    rng = np.random.default_rng(9000 + chunk)
    for k in range(10):
        cfg, fast = random_config(rng)
        try:
            rollout(BACKEND, cfg, fast, E=8, steps=8, seed=chunk * 100 + k)
        except AssertionError as ex:  # name the configuration in the failure
            raise AssertionError(f"chunk {chunk} config {k}: {cfg}\n{ex}") from ex


@pytest.mark.parametrize("chunk", CHUNKS)  # 4 configurations each
def test_random_wide_configurations_vs_oracle(chunk):
    """The same family with 63 .. 127 other vehicles -- the traffic sizes of the two-vehicles-per-thread kernel (csrc/hwy_wave2.h; the
    OccupancyGrid draws of the family run the workgroup kernel there) -- and, fourth configuration of every chunk, 128 .. 255: three
    and four vehicles per thread, the N > 128 path since round 5."""
    rng = np.random.default_rng(19000 + chunk)
    for k in range(4):
        cfg, fast = random_config(rng)
        agents = cfg["controlled_vehicles"]
        cfg["vehicles_count"] = int(rng.integers(64 - agents, 129 - agents))
        if k == 3:   # (the engine's own choice beyond N = 128 is the workgroup kernel: alternate between the two)
            cfg["vehicles_count"] = int(np.random.default_rng(23000 + chunk).integers(129 - agents, 257 - agents))
            cfg["tuning"] = {"block_kernel": 2 if chunk % 2 == 0 else 0}
        try:
            rollout(BACKEND, cfg, fast, E=4 if k < 3 else 2, steps=6 if k < 3 else 4, seed=chunk * 100 + k)
        except AssertionError as ex:  # name the configuration in the failure
            raise AssertionError(f"chunk {chunk} config {k}: {cfg}\n{ex}") from ex


def random_merge_config(rng):
    from highwayenv_amd import merge
    cfg = merge.merge_generic_default_config()
    lanes = int(rng.integers(1, 5))
    n = int(rng.integers(0, 50))
    agents = int(rng.integers(1, min(4, n + 1) + 1))
    cfg.update({"lanes_count": lanes, "vehicles_count": n, "controlled_vehicles": agents,
                "before_merge_length": int(rng.integers(50, 200)), "converge_merge_length": int(rng.integers(40, 120)),
                "parallel_merge_length": int(rng.integers(40, 120)), "after_merge_length": int(rng.integers(90, 200)),
                "simulation_frequency": int(rng.choice([5, 15])), "collision_reward": float(rng.uniform(-2, -0.1)),
                "merging_speed_reward": float(rng.uniform(-1, -0.1)), "lane_change_reward": float(rng.uniform(-0.2, 0)),
                "neighbour_vehicles_connected_lanes": bool(rng.integers(2))})   # merge-generic-v0 / -v1
    kin = {"type": "Kinematics", "include_obstacles": bool(rng.integers(3)), "order": "shuffled" if rng.integers(4) == 0 else "sorted"}
    obs = kin
    if rng.integers(4) == 0:  # OccupancyGrid on the merge network (borders off the waypoint lattice)
        obs = {"type": "OccupancyGrid", "grid_size": [[-31.3, 28.7], [-13.3, 11.7]], "grid_step": [float(rng.choice([2.5, 5]))] * 2,
               "align_to_vehicle_axes": bool(rng.integers(2)), "clip": bool(rng.integers(2)), "as_image": bool(rng.integers(2))}
    cfg["observation"] = obs
    if agents > 1:
        cfg["action"] = {"type": "MultiAgentAction", "action_config": {"type": "DiscreteMetaAction"}}
        cfg["observation"] = {"type": "MultiAgentObservation", "observation_config": obs}
    return cfg


@pytest.mark.parametrize("chunk", CHUNKS)
def test_random_merge_configurations_vs_oracle(chunk):
    from tests.test_net_parity import _rollout_vs_oracle
    rng = np.random.default_rng(7000 + chunk)
    for k in range(6):
        cfg = random_merge_config(rng)
        try:
            _rollout_vs_oracle(BACKEND, cfg, "merge-generic", E=6, steps=10, seed=chunk * 100 + k + 1)
        except AssertionError as ex:
            raise AssertionError(f"chunk {chunk} config {k}: {cfg}\n{ex}") from ex


def random_intersection_config(rng):
    from highwayenv_amd import intersection as hix
    cfg = hix.intersection_default_config()
    kind = int(rng.integers(3))
    if kind == 1:
        cfg["observation"] = {"type": "OccupancyGrid", "as_image": bool(rng.integers(2))}
    elif kind == 2:
        cfg["observation"] = {"type": "OccupancyGrid", "align_to_vehicle_axes": bool(rng.integers(2)),
                              # (cell borders that are whole multiples of the waypoint spacing away from the observer put
                              #  every waypoint of an axis-aligned lane through the observer ON a border, where the last
                              #  bit of np.dot decides the cell: the offsets below keep the borders off that lattice)
                              "grid_size": [[-31.3, 28.7], [-21.3, 18.7]], "grid_step": [float(rng.choice([2.5, 4, 5]))] * 2,
                              "features": ["presence", "x", "y", "vx", "vy", "cos_h", "sin_h", "on_road"],
                              "features_range": {"x": [-50, 50], "y": [-50, 50], "vx": [-20, 20], "vy": [-20, 20]}}
    else:
        extra = [[], ["cos_d", "sin_d"], ["long_off", "lat_off", "ang_off"], ["heading", "cos_d", "sin_d", "lat_off"]][int(rng.integers(4))]
        cfg["observation"] = dict(cfg["observation"], vehicles_count=int(rng.integers(3, 16)), see_behind=bool(rng.integers(2)),
                                  absolute=bool(rng.integers(2)), observe_intentions=bool(rng.integers(2)),
                                  features=cfg["observation"]["features"] + extra)
    cfg.update({"initial_vehicle_count": int(rng.integers(2, 15)), "spawn_probability": float(rng.uniform(0, 1)),
                "duration": int(rng.integers(5, 20)), "max_vehicles": int(rng.choice([20, 28, 32, 40])),
                "destination": f"o{int(rng.integers(0, 4))}", "normalize_reward": bool(rng.integers(2)),
                "offroad_terminal": bool(rng.integers(2)), "collision_reward": float(rng.uniform(-6, -1)),
                "arrived_reward": float(rng.uniform(0.5, 2)),
                "neighbour_vehicles_connected_lanes": bool(rng.integers(2)),    # intersection-v0 / -v2
                "controlled_vehicles": int(rng.choice([1, 1, 2, 3, 4]))})       # > 1: MultiAgentIntersectionEnv
    if rng.integers(3) == 0:
        cfg["destination"] = None  # "o" + str(np_random.integers(1, 4)) per controlled vehicle
    return cfg


def _unmatched_rows(h, o, tol=1e-6, rows=False):
    """Rows of `h` [V, F] with no partner in `o` within `tol` (each row of `o` used once): the two observations as SETS.
    rows=True: (unmatched rows of h, unmatched rows of o) instead of the count."""
    free = list(range(len(o)))
    miss = []
    for i, r in enumerate(h):
        j = next((j for j in free if np.abs(r - o[j]).max() <= tol), None)
        if j is None:
            miss.append(i)
        else:
            free.remove(j)
    return (h[miss], o[free]) if rows else len(miss)


@pytest.mark.parametrize("chunk", CHUNKS)
def test_random_intersection_configurations_vs_oracle(chunk):
    """Device-traffic engine drives the episodes (reset, clear, spawn, auto-reset on Philox); every step is replayed from its
    own state on a host-traffic engine and on the oracle.  Two comparisons, and the fraction each covers is PRINTED and asserted:

    * EVERY FRAME of every live env-step, teacher-forced (meta-action on the first, Road.act, RegulatedRoad.step incl. collisions):
      frame k of engine and oracle both start from the engine's state after frame k - 1, compared at 1e-9 -- every environment,
      whatever its speeds and wrecks; only a collision push on the knife edge (|d.normal| < 1e-9) takes an environment's frame out;
    * the WHOLE policy step free-running (15 frames + observation + reward + termination) wherever 15 frames of one trajectory
      are comparable between two arithmetics: no vehicle below 1 m/s (steering_control divides by not_zero(speed) twice, so
      last-bit differences grow 1e2..1e4 x per frame there -- at an intersection cars yield and queue, so this is the bulk of
      the exclusions), no lane-index knife edge, and -- steps WITH a collision are compared like any other -- no push on the
      knife edge.  What this leaves out is covered by the first comparison, frame by frame;
    * the END-OF-STEP PRODUCTS (observation, reward, terminated, truncated: observation.py:234-276 / :354-413, intersection_env.py:
      70-134) of EVERY live env-step, slow and queueing ones included: the LAST frame of the teacher-forced sequence is run as one
      policy step of a second engine / oracle pair configured with policy_frequency = simulation_frequency (one frame per step,
      IDLE meta-action == no meta-action: controller.py:295-315) from the engine's state after the frame before it -- both sides
      observe, reward and terminate on states that differ by one frame's rounding (1e-9), not by fifteen."""
    from oracle import oracle, oracle_ix
    from tests.backends import make_engine
    from tests.golden_util import KNIFE, assert_obs_close, ix_oracle_config, ix_oracle_state
    rng = np.random.default_rng(5000 + chunk)
    tot_steps = tot_checked = tot_frames = tot_col = tot_col_full = tot_img_cells = tot_edge = tot_touch = 0
    tot_all_frames = tot_live_frames = tot_lane_frames = tot_flip = tot_cut = tot_prod = tot_cut_p = 0
    for k in range(4):
        cfg = random_intersection_config(rng)
        E = 12
        try:
            c = _abi.make_config(dict(cfg, host_traffic=False), E, scenario="intersection")
            ch = _abi.make_config(dict(cfg, host_traffic=True), E, scenario="intersection")
            dev, host = make_engine(BACKEND, c), make_engine(BACKEND, ch)
            oc = ix_oracle_config(dict(cfg, host_traffic=True), ch, E)
            T_frames = int(c.frames_per_step)
            # one frame per policy step: the last frame of a step + its products from one state (docstring, third comparison)
            cfg1 = dict(cfg, host_traffic=True, policy_frequency=cfg["simulation_frequency"])
            c1 = _abi.make_config(cfg1, E, scenario="intersection")
            assert int(c1.frames_per_step) == 1
            host1, oc1 = make_engine(BACKEND, c1), ix_oracle_config(cfg1, c1, E)
            idle = np.ones((E, c.num_agents), np.int32)
            dev.reset(base_seed=chunk * 1000 + k)
            dev.set_autoreset(True, base_seed=chunk * 1000 + k)
            checked = n_flip = n_cut = n_live = n_edge = 0
            feats = list(cfg["observation"].get("features") or [])
            done_prev = np.zeros(E, bool)
            for t in range(10):
                st = dev.get_state()
                acts = rng.integers(0, 3, size=(E, c.num_agents)).astype(np.int32)
                pres = (st["flags"] & _abi.F_ABSENT) == 0
                # -- every frame of the step, every live environment, teacher-forced ------------------------------------
                # frame k starts from the ENGINE's state after frame k - 1 (read back through hwy_get_state), handed to the oracle in
                # its own layout: both advance ONE frame from identical inputs, so no difference is carried from frame to frame and
                # the slow, queueing, knife-edge env-steps that the whole-step comparison below has to leave out are compared too,
                # frame by frame, at 1e-9 (positions, headings, speeds, signed impacts; pending-impact and yielding bits exact)
                host.set_state(st)
                st_k = st
                for fr in range(T_frames):
                    ost1 = ix_oracle_state(st_k, ch)
                    host.step_frames(acts if fr == 0 else None, 1)
                    with oracle.impact_margins(oc) as m1:
                        oracle_ix.frames(oc, ost1, acts if fr == 0 else None, 1)
                    g1 = host.get_state()
                    fine = ~done_prev & (m1.margin.min(1) >= 1e-9)
                    if fr == T_frames - 1:
                        # -- end-of-step products of EVERY live env-step: this last frame as a one-frame policy step ------------
                        host1.set_state(st_k)
                        p_obs, p_rew, p_term, p_trunc, _ = host1.step(idle)
                        ost_p = ix_oracle_state(st_k, c1)
                        q_obs, _, q_term, q_trunc, q_info = oracle_ix.step(oc1, ost_p, idle)
                        g_p = host1.get_state()
                        # (a last-bit lane index changes the ego's lane coordinate the observation is ordered by, a knife-edge push
                        #  the pose itself: the frame comparison below counts / excludes those, and so does this one)
                        pres_p = (g_p["flags"] & _abi.F_ABSENT) == 0
                        ok_p = fine & ~(pres_p & (g_p["lane"] != ost_p["lane"])).any(1)
                        np.testing.assert_array_equal(p_term[ok_p], q_term[ok_p], err_msg=f"step {t}: terminated (products of every step)")
                        np.testing.assert_array_equal(p_trunc[ok_p], q_trunc[ok_p], err_msg=f"step {t}: truncated (products of every step)")
                        np.testing.assert_allclose(p_rew[ok_p], q_info["agents_rewards"][ok_p], rtol=0, atol=1e-9,
                                                   err_msg=f"step {t}: reward (products of every step)")
                        rows_p = lambda o: o.reshape(-1, *o.shape[-2:])  # noqa: E731
                        if c.obs_type == _abi.OBS_KINEMATICS:
                            ix_x, ix_y = feats.index("x"), feats.index("y")
                            for hq, oq in zip(rows_p(p_obs[ok_p]).astype(np.float64), rows_p(q_obs[ok_p]).astype(np.float64)):
                                if np.abs(hq - oq).max() <= 1e-6 or _unmatched_rows(hq, oq) == 0:
                                    continue
                                # (the queue-order knife edge of the whole-step comparison below, same rule, its own counter)
                                seen = oq[oq[:, 0] > 0]
                                queued = any((np.abs(seen[:, col][:, None] - seen[:, col][None, :]) < 1e-4).sum() > len(seen)
                                             for col in (ix_x, ix_y))
                                if not queued:
                                    only_h, only_o = _unmatched_rows(hq, oq, rows=True)
                                    queued = len(only_h) == len(only_o) and all(
                                        any(min(abs(a[ix_x] - b[ix_x]), abs(a[ix_y] - b[ix_y])) < 1e-6 for b in only_o) for a in only_h)
                                assert queued, f"step {t} (products of every step): {hq} != {oq}"
                                tot_cut_p += 1
                            np.testing.assert_allclose(rows_p(p_obs[ok_p])[:, 0], rows_p(q_obs[ok_p])[:, 0], rtol=0, atol=1e-6,
                                                       err_msg=f"step {t}: ego row (products of every step)")
                        else:
                            image = bool(c.flags & _abi.C_GRID_IMAGE)
                            assert_obs_close(p_obs[ok_p].reshape(q_obs[ok_p].shape), q_obs[ok_p], image, f"step {t} (products of every step)")
                        tot_prod += int(ok_p.sum())
                    for f in ("x", "y", "heading", "speed"):
                        np.testing.assert_allclose(g1[f][fine], ost1[f][fine], rtol=0, atol=1e-9, err_msg=f"step {t} frame {fr}: {f}")
                    # two wrecks resting EXACTLY touching (flag_margin < KNIFE: the `will_intersect` of that pair hinges on a distance
                    # of ~1e-16, emulator fuzz chunk 30103): whether the pair gets its ~0 translation as a pending impact -- in place of
                    # the translation another pair gave that slot earlier in the loop, "last pair wins" -- is decided by the last bit
                    # (fuzz chunks 1, 11, 23 from frame 1 on: resting pile-ups).  Bit and translation are compared on every other slot,
                    # such slots are counted
                    touching = np.asarray(m1.flag_margin) < KNIFE
                    sure = fine[:, None] & ~touching
                    # signed; 1e-9 + 1e-7 of the translation: when two bodies are parallel to ~1e-8 rad (a rear-end collision on one
                    # lane centre) the projections on their two lateral normals tie in the last bit, "first minimum wins" picks either
                    # body's normal, and the translation turns by that angle -- 4.3e-9 m of a 0.22 m push in GPU fuzz chunk 354, the one
                    # case in 3.5 M frames; a wrong pair or a wrong sign is off by the whole translation
                    tol = 1e-9 + 1e-7 * np.hypot(ost1["impact_x"], ost1["impact_y"])
                    for f in ("impact_x", "impact_y"):
                        bad_imp = sure & ~(np.abs(g1[f] - ost1[f]) <= tol)
                        assert not bad_imp.any(), (f"step {t} frame {fr}: {f}: engine {g1[f][bad_imp]} oracle {ost1[f][bad_imp]} "
                                                   f"(|translation| {np.hypot(ost1['impact_x'], ost1['impact_y'])[bad_imp]})")
                    hi_g, hi_o = (g1["flags"] & _abi.F_HAS_IMPACT) != 0, ost1["has_impact"] != 0
                    np.testing.assert_array_equal((hi_g | touching)[fine], (hi_o | touching)[fine], err_msg=f"step {t} frame {fr}: pending impacts")
                    tot_touch += int((touching & ((hi_g != hi_o) | (np.abs(g1["impact_x"] - ost1["impact_x"]) > 1e-9)
                                                  | (np.abs(g1["impact_y"] - ost1["impact_y"]) > 1e-9)))[fine].sum())
                    np.testing.assert_array_equal(((g1["flags"] & _abi.F_YIELDING) != 0)[fine], (ost1["is_yielding"] != 0)[fine],
                                                  err_msg=f"step {t} frame {fr}: is_yielding")
                    np.testing.assert_array_equal(((g1["flags"] & _abi.F_CRASHED) != 0)[fine], (ost1["crashed"] != 0)[fine],
                                                  err_msg=f"step {t} frame {fr}: crashed")
                    # lane indices: exact wherever the oracle's own closest-lane decision is not a tie of the last bit (three lanes
                    # leave an "ir" node together: see `flip` below) -- counted, the next frame starts from the engine's index
                    pres_k = (g1["flags"] & _abi.F_ABSENT) == 0
                    lane_diff = (pres_k & (g1["lane"] != ost1["lane"]))[fine]
                    tot_lane_frames += int(lane_diff.any(1).sum())
                    if fr == 0:
                        tot_frames += int(fine.sum())
                    tot_all_frames += int(fine.sum())
                    tot_live_frames += int((~done_prev).sum())
                    st_k = g1
                # -- the whole policy step -----------------------------------------------------------------------------
                ost = ix_oracle_state(st, ch)
                host.set_state(st)
                h_obs, h_rew, h_term, h_trunc, _ = host.step(acts)
                with oracle.impact_margins(oc) as m:
                    o_obs, _, o_term, o_trunc, o_info = oracle_ix.step(oc, ost, acts)
                o_rew = o_info["agents_rewards"]
                rows = lambda o: o.reshape(-1, *o.shape[-2:])  # noqa: E731  ([n, A, V, F] or [n, V, F] -> agent rows)
                # (steering_control divides by not_zero(speed) twice: below ~1 m/s last-bit differences grow fast)
                # (and a car that IDM has pushed into REVERSE behind a leader a few centimetres away -- seen with the connected-lane
                #  search, which finds leaders across segment ends -- multiplies differences ~8x per frame through 1 / d**2)
                bad = (pres & (st["speed"] < 1.0)).any(1) | ((ost["present"] != 0) & (ost["speed"] < 1.0)).any(1)
                wreck = ((ost["present"] != 0) & ((ost["crashed"] != 0) | (ost["has_impact"] != 0))).any(1)
                tot_col += int((wreck & ~bad & ~done_prev).sum())
                bad |= m.margin.min(1) < 1e-9      # a push direction on the knife edge (utils.py:232-236)
                tot_col_full += int((wreck & ~bad & ~done_prev).sum())
                ok = ~bad & ~done_prev  # (a finished episode is re-spawned by the device engine in this very step)
                n_live += int((~done_prev).sum())
                got = host.get_state()
                # Knife edge of the reference itself: where three lanes leave an "ir" node together (same start point, same
                # heading) a vehicle is equally close to all of them, and get_closest_lane_index is decided by the last bit
                # of the projections -- two libms may index the vehicle on different lanes there (and, for the ego, order the
                # observation by another lane's coordinate).  Such env-steps are skipped and counted.
                flip = (pres & (got["lane"] != ost["lane"])).any(1)
                n_flip += int((flip & ok).sum())
                ok &= ~flip
                # A wreck resting EXACTLY touching a third body while another pair pushes it (flag_margin < KNIFE): whether the
                # touching pair "will intersect" is decided by the last bit, and if it does its ~0 translation replaces the pending
                # impact of the real collision -- a finite difference within the step, either being the reference's answer for one
                # rounding (tests/test_net_parity.py: _rollout_vs_oracle).  Intersection pile-ups persist for many steps, so such
                # env-steps are compared like the others, one environment at a time, and a mismatch is counted instead of raised.
                edge = ok & wreck & (np.asarray(m.flag_margin) < KNIFE).any(1)
                ok_all = ok
                counts = [n_cut, tot_img_cells]

                def compare(ok):
                    n_cut, tot_img_cells = 0, 0
                    np.testing.assert_array_equal(h_term[ok], o_term[ok], err_msg=f"step {t}")
                    np.testing.assert_array_equal(h_trunc[ok], o_trunc[ok], err_msg=f"step {t}")
                    if c.obs_type == _abi.OBS_KINEMATICS:
                        # second knife edge of the reference: cars queued on a road PERPENDICULAR to the observer's lane all have
                        # the same longitudinal coordinate on that lane up to rounding, and close_objects_to sorts by it
                        # (road.py:446) -- the row order among them is noise: first the rows in order at 1e-6, and only where
                        # that fails as a SET at 1e-6 (no rounding) ...
                        # ... and when more vehicles are eligible than the observation has rows, the same tie decides WHICH of the
                        # queued cars make the cut: an agent's observation may differ in rows that share their x or their y
                        # (the queue's coordinate) with another observed row -- tolerated, and counted
                        ix_x, ix_y = feats.index("x"), feats.index("y")
                        for hq, oq in zip(rows(h_obs[ok]).astype(np.float64), rows(o_obs[ok]).astype(np.float64)):
                            if np.abs(hq - oq).max() <= 1e-6 or _unmatched_rows(hq, oq) == 0:
                                continue
                            seen = oq[oq[:, 0] > 0]
                            queued = any((np.abs(seen[:, col][:, None] - seen[:, col][None, :]) < 1e-4).sum() > len(seen)
                                         for col in (ix_x, ix_y))
                            if not queued:  # ... or with the car that did NOT make the cut: the rows only one side has, pairwise
                                only_h, only_o = _unmatched_rows(hq, oq, rows=True)
                                queued = len(only_h) == len(only_o) and all(
                                    any(min(abs(a[ix_x] - b[ix_x]), abs(a[ix_y] - b[ix_y])) < 1e-6 for b in only_o) for a in only_h)
                            assert queued, f"step {t}: {hq} != {oq}"
                            n_cut += 1
                        np.testing.assert_allclose(rows(h_obs[ok])[:, 0], rows(o_obs[ok])[:, 0], rtol=0, atol=1e-6, err_msg=f"step {t}: ego row")
                    else:
                        image = bool(c.flags & _abi.C_GRID_IMAGE)
                        assert_obs_close(h_obs[ok].reshape(o_obs[ok].shape), o_obs[ok], image, f"step {t}")
                        if image:
                            tot_img_cells += int((h_obs[ok].reshape(o_obs[ok].shape) != o_obs[ok]).sum())
                    np.testing.assert_allclose(h_rew[ok], o_rew[ok], rtol=0, atol=1e-9, err_msg=f"step {t}")
                    # 1e-7 after 15 free-running frames -- for vehicles that stay above 2 m/s: between 1 and 2 m/s the two divisions
                    # by the speed in steering_control still amplify a last-bit difference to ~1e-7 of LATERAL offset within a step
                    # (emulator fuzz chunk 20547: 1.36e-7 on a car slowing from 1.8 to 1.2 m/s); those are held to 1e-6 below
                    brisk = (ok & ~wreck)[:, None] & ~(pres & ((st["speed"] < 2.0) | (got["speed"] < 2.0)))
                    np.testing.assert_allclose(got["x"][brisk], ost["x"][brisk], rtol=0, atol=1e-7, err_msg=f"step {t}")
                    np.testing.assert_allclose(got["x"][ok], ost["x"][ok], rtol=0, atol=1e-6, err_msg=f"step {t}")
                    counts[0] += n_cut
                    counts[1] += tot_img_cells

                compare(ok_all & ~edge)
                for e_ in np.flatnonzero(edge):
                    try:
                        compare(np.arange(E) == e_)
                    except AssertionError:
                        n_edge += 1
                        ok_all = ok_all & (np.arange(E) != e_)
                n_cut, tot_img_cells = counts
                ok = ok_all
                checked += int(ok.sum())
                d_obs, d_rew, d_term, d_trunc, _ = dev.step(acts)
                np.testing.assert_array_equal(d_term[ok], h_term[ok])  # same dynamics with device traffic switched on
                done_prev = d_term | d_trunc
            tot_edge += n_edge
            tot_flip += n_flip
            tot_cut += n_cut
            tot_col_full -= n_edge
            tot_steps += n_live
            tot_checked += checked
            for e_ in (dev, host, host1):
                e_.close()
        except AssertionError as ex:
            raise AssertionError(f"chunk {chunk} config {k}: {cfg}\n{ex}") from ex
    frac = tot_checked / max(tot_steps, 1)
    print(f"\nintersection fuzz chunk {chunk}: {tot_steps} live env-steps = {tot_live_frames} frames; teacher-forced frames compared at "
          f"1e-9: {tot_all_frames} ({100.0 * tot_all_frames / max(tot_live_frames, 1):.2f} %; first frames {tot_frames}); end-of-step "
          f"products (obs, reward, flags) on {tot_prod} env-steps ({100.0 * tot_prod / max(tot_steps, 1):.2f} %, {tot_cut_p} queue-order "
          f"cuts); whole step on "
          f"{tot_checked} ({100.0 * frac:.1f} %); {tot_col} fast-enough steps with a wreck, {tot_col_full} of them in full; "
          f"as_image cells off by one: {tot_img_cells}; tolerated and counted: {tot_edge} env-steps diverged on a touching pair's knife "
          f"edge, {tot_touch} pending-impact bits of exactly touching wrecks, {tot_lane_frames} frames with a last-bit lane index, "
          f"{tot_flip} lane-index flips and {tot_cut} queue-order cuts in whole steps")
    # the tolerated knife-edge cases, bounded by what 200 chunks measured on the GPU (above) -- a defect in the pile-up path FAILS
    # outright (tests/test_mutations.py: the seeded "first pair wins" bug does not even reach these counters)
    if os.environ.get("HWY_FUZZ_CALIBRATE") != "1":  # (calibration runs only print the counts: tools/gpu_fuzz.sh)
        assert tot_edge <= EDGE_MAX and tot_touch <= TOUCH_MAX and tot_lane_frames <= LANE_FRAMES_MAX, (tot_edge, tot_touch, tot_lane_frames)
        assert tot_flip <= FLIP_MAX and tot_cut <= CUT_MAX, (tot_flip, tot_cut)
    assert tot_all_frames >= 0.985 * tot_live_frames, "the teacher-forced comparison must cover (nearly) every live frame"
    assert tot_prod >= 0.97 * tot_steps, "the end-of-step products must be compared on (nearly) every live env-step"
    if os.environ.get("HWY_FUZZ_CALIBRATE") != "1":
        assert tot_cut_p <= CUT_P_MAX, tot_cut_p
    assert frac >= INTERSECTION_WHOLE_STEP_FLOOR, f"only {100 * frac:.1f} % of the env-steps were compared as whole steps"
    assert tot_col_full >= 0.9 * tot_col - 1
