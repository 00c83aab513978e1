"""Host-side mirror of the reference's environment API for the accelerated path.

``BatchedHighwayEnv`` keeps the shape of ``AbstractEnv`` (highway_env/envs/common/abstract.py):
``default_config()`` / ``configure()`` / ``reset(seed=, options=)`` / ``step(action)`` / ``close()``,
same config dict keys as ``HighwayEnv.default_config`` (envs/highway_env.py:25-53) and the same
exceptions for what the reference rejects -- but steps E environments per call, and the body of
``_simulate`` + ``observe`` + ``_reward`` + ``_is_terminated`` + ``_is_truncated``
(abstract.py:259-317) is ONE C-ABI call into the HIP engine.

``HighwayEnv`` / ``HighwayEnvFast`` are the E == 1 drop-ins with the reference's unbatched return
types (obs ``(5,5) float32``, ``float`` reward, ``bool`` flags, ``info`` dict with
``speed/crashed/action/rewards``).
"""
from __future__ import annotations

import copy

import numpy as np

from . import _abi, spawn
from . import merge as _merge
from . import intersection as _ix
from .engine import Engine

try:  # optional: only to expose real spaces when gymnasium is installed
    import gymnasium as _gym
except Exception:  # pragma: no cover - gymnasium is absent in the build image
    _gym = None


class _Discrete:
    def __init__(self, n):
        self.n, self.shape, self.dtype = n, (), np.int64

    def contains(self, x):
        return 0 <= int(x) < self.n

    def sample(self):
        return int(np.random.randint(self.n))


class _Tuple:
    """gymnasium.spaces.Tuple stand-in (MultiAgentAction.space, action.py:336-340): one sub-space per agent."""

    def __init__(self, spaces):
        self.spaces = tuple(spaces)

    def __len__(self):
        return len(self.spaces)

    def contains(self, x):
        return len(x) == len(self.spaces) and all(sp.contains(v) for sp, v in zip(self.spaces, x))

    def sample(self):
        return tuple(sp.sample() for sp in self.spaces)


class _Box:
    def __init__(self, shape, dtype=np.float32):
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)
        self.low, self.high = (0, 255) if self.dtype == np.uint8 else (-np.inf, np.inf)

    def contains(self, x):
        return tuple(np.shape(x)) == self.shape


class VehicleView:
    """Read-only view of one vehicle of one env (reference: Vehicle / ControlledVehicle attributes)."""

    def __init__(self, st, e, i):
        self.position = np.array([st["x"][e, i], st["y"][e, i]])
        self.heading = float(st["heading"][e, i])
        self.speed = float(st["speed"][e, i])
        self.lane_index = ("0", "1", int(st["lane"][e, i]))
        self.target_lane_index = ("0", "1", int(st["target_lane"][e, i]))
        self.target_speed = float(st["target_speed"][e, i])
        f = int(st["flags"][e, i])
        self.crashed = bool(f & _abi.F_CRASHED)
        self.check_collisions = bool(f & _abi.F_CHECK_COLLISIONS)
        self.controlled = bool(f & _abi.F_CONTROLLED)
        self.speed_index = int(st["speed_index"][e, i]) if self.controlled else None
        self.timer = None if self.controlled else float(st["timer"][e, i])
        self.DELTA = None if self.controlled else float(st["delta"][e, i])

    @property
    def velocity(self):
        return self.speed * np.array([np.cos(self.heading), np.sin(self.heading)])


class RoadView:
    def __init__(self, st, e):
        self.vehicles = [VehicleView(st, e, i) for i in range(st["x"].shape[1])]
        self.objects = []


class BatchedHighwayEnv:
    """E parallel ``highway-v0``-family environments on one MI355X."""

    #: ``HighwayEnvFast`` semantics (ego-only collision checks, highway_env.py:177-182)
    FAST = False
    #: "highway" | "merge" | "merge-generic" (``_abi.make_config``)
    SCENARIO = "highway"
    PERCEPTION_DISTANCE = 5.0 * 40.0
    #: the HIP engine; tests substitute the CPU emulation of the same kernel source
    _engine_factory = staticmethod(lambda cfg, device, stream: Engine(cfg, device=device, stream=stream))

    def __init__(self, config: dict | None = None, num_envs: int = 1, device: int = 0, render_mode=None,
                 spawn_mode: str = "reference", autoreset: bool = False, stream: int | None = None):
        if render_mode is not None:
            raise NotImplementedError("rendering is outside the MI355X hot-path scope")
        if spawn_mode not in ("reference", "device"):
            raise ValueError("spawn_mode must be 'reference' (numpy PCG64 stream) or 'device' (Philox)")
        self.num_envs = int(num_envs)
        self.device = device
        self.spawn_mode = spawn_mode
        self.autoreset = bool(autoreset)
        self._stream = stream
        self.config = self.default_config()
        self.configure(config)
        self._engine = None
        self._engine_key = None
        self._np_randoms = [None] * self.num_envs  # per-env Generator == reference env.np_random
        self.time = np.zeros(self.num_envs)
        self.steps = 0
        self._define_spaces()

    @property
    def np_random(self):
        """Batched classes: the list of per-environment Generators (env e's == the reference env's ``np_random`` after
        ``reset(seed=seeds[e])``).  The single-environment drop-ins return THE Generator (``_SingleEnvMixin``)."""
        return self._np_randoms

    @np_random.setter
    def np_random(self, value):
        self._np_randoms = list(value) if isinstance(value, (list, tuple)) else [value] * self.num_envs

    # ---- config (abstract.py:101-144) ------------------------------------------------------
    @classmethod
    def default_config(cls) -> dict:
        return _abi.highway_fast_default_config() if cls.FAST else _abi.highway_default_config()

    def configure(self, config: dict | None) -> None:
        if config:
            self.config.update(config)

    def _define_spaces(self):
        hc = _abi.make_config(self.config, self.num_envs, fast=self.FAST, scenario=self.SCENARIO)
        self._hcfg = hc
        A = hc.num_agents
        self.single_observation_shape = _abi.obs_shape(hc) if A == 1 else (A, *_abi.obs_shape(hc))
        if _gym is not None:
            self.single_action_space = _gym.spaces.Discrete(_abi.num_actions(hc))  # len(self.actions), action.py:252-253
            image = bool(hc.flags & _abi.C_GRID_IMAGE)  # observation.py:330-331: Box(0, 255, uint8)
            self.single_observation_space = (_gym.spaces.Box(0, 255, self.single_observation_shape, np.uint8) if image else
                                             _gym.spaces.Box(-np.inf, np.inf, self.single_observation_shape, np.float32))
        else:
            self.single_action_space = _Discrete(_abi.num_actions(hc))
            self.single_observation_space = _Box(self.single_observation_shape, np.uint8 if hc.flags & _abi.C_GRID_IMAGE else np.float32)
        self.action_space, self.observation_space = self.single_action_space, self.single_observation_space

    def _ensure_engine(self):
        key = bytes(self._hcfg)
        if self._engine is None or key != self._engine_key:
            if self._engine is not None:
                self._engine.close()
            self._engine = self._engine_factory(self._hcfg, self.device, self._stream)  # raises without a GPU
            self._engine_key = key
        return self._engine

    # ---- reset (abstract.py:219-249) -----------------------------------------------------------
    def reset(self, *, seed=None, options: dict | None = None):
        if options and "config" in options:
            self.configure(options["config"])
        self._define_spaces()
        eng = self._ensure_engine()
        E = self.num_envs
        if seed is None:
            seeds = [None] * E
        elif np.ndim(seed) == 0:
            seeds = [int(seed) + e for e in range(E)]  # gymnasium vector-env convention
        else:
            seeds = list(seed)
            if len(seeds) != E:
                raise ValueError("one seed per env expected")
        for e, s in enumerate(seeds):
            if s is not None or self._np_randoms[e] is None:
                # gymnasium.utils.seeding.np_random(seed): Generator(PCG64(SeedSequence(seed)))
                self._np_randoms[e] = np.random.Generator(np.random.PCG64(np.random.SeedSequence(s)))
        if self.spawn_mode == "reference":
            eng.set_state(self._spawn_reference(self._np_randoms))
            obs = eng.observe()
        else:
            sd = np.array([np.random.SeedSequence(s).generate_state(1, np.uint64)[0] if s is None else s
                           for s in seeds], np.uint64)
            obs = eng.reset(seeds=sd, **self._device_spawn_args())
        # auto-reset episodes: derived from the given seed (reproducible), or as unseeded as the first episode when
        # seed is None (fresh OS entropy -- otherwise every seed=None worker would replay the same later episodes)
        if seeds[0] is None:
            base_seed = int(np.random.SeedSequence(None).generate_state(1, np.uint64)[0] >> np.uint64(1))
        else:
            base_seed = int(seeds[0]) + 0x9E3779B9
        eng.set_autoreset(self.autoreset, base_seed=base_seed, **self._device_spawn_args())
        # SameStep re-spawns (reset_done): their own stream, keyed by the user's seed (reproducible; fresh entropy for seed=None)
        # and restarted by every reset() -- a SeedSequence child, so that no (seed, env, episode) meets a first-episode seed s + e
        self._same_step_entropy = int(base_seed)
        self._episodes = np.zeros(E, np.uint64)
        self.time[:] = 0
        self.steps = 0
        st = eng.get_state()
        rows, ego = np.arange(E), self._ego_slots(st)
        info = {"speed": st["speed"][rows, ego].copy(), "crashed": (st["flags"][rows, ego] & _abi.F_CRASHED) != 0,
                "action": None}
        return self._shape_obs(obs), info

    def reset_done(self, mask) -> np.ndarray:
        """Re-spawn the environments of ``mask`` [E] on the device NOW (the "SameStep" autoreset of a vector env: the step that
        ended an episode also returns the next episode's first observation) and return their observations [k, ...].  Seeds:
        hashed from (the seed given to reset(), env, episode number) -- reproducible per user seed, different between user
        seeds, restarted by reset() (``spawn_mode="device"`` only)."""
        if self.spawn_mode != "device":
            raise NotImplementedError("reset_done (SameStep autoreset) needs spawn_mode='device'")
        mask = np.asarray(mask, bool)
        if self._engine is None or getattr(self, "_episodes", None) is None or len(self._episodes) != self.num_envs:
            raise RuntimeError("reset_done before reset()")
        self._episodes[mask] += np.uint64(1)
        # seed of (env e, episode k) = SeedSequence(entropy of this reset(), spawn_key=(0x5A3E, e, k)): hashed, so two user seeds
        # never replay each other's later episodes and no later episode repeats a first one (seed + e)
        seeds = np.zeros(self.num_envs, np.uint64)
        for e in np.flatnonzero(mask):
            seeds[e] = np.random.SeedSequence(self._same_step_entropy, spawn_key=(0x5A3E, int(e), int(self._episodes[e]))
                                              ).generate_state(1, np.uint64)[0]
        obs = self._engine.reset(seeds=seeds, mask=mask.astype(np.uint8), **self._device_spawn_args())
        self.time[mask] = 0
        return self._shape_obs(obs)[mask]

    def _ego_slots(self, st) -> np.ndarray:
        """Slot of the (first) controlled vehicle of every env."""
        return np.full(self.num_envs, self._hcfg.agent_index[0])

    def _spawn_reference(self, generators) -> dict:
        """``_create_vehicles`` replayed on each env's numpy Generator (highway_env.py:72-98)."""
        cfg = self.config
        return spawn.spawn_reference_stream(self._hcfg, generators, cfg["ego_spacing"], cfg["vehicles_density"],
                                            cfg["initial_lane_id"])

    def _device_spawn_args(self) -> dict:
        cfg = self.config
        lane_id = cfg["initial_lane_id"]
        return {"ego_spacing": cfg["ego_spacing"], "vehicles_density": cfg["vehicles_density"],
                "initial_lane_id": -1 if lane_id is None else lane_id}

    # ---- step (abstract.py:259-285) ---------------------------------------------------------------
    def step(self, action):
        if self._engine is None:
            # the reference raises NotImplementedError when road/vehicle are unset (abstract.py:269-272)
            raise NotImplementedError("The road and vehicle must be initialized in the environment implementation")
        E, A = self.num_envs, self._hcfg.num_agents
        acts = np.asarray(action)
        # accepted shapes: scalar (every env, every agent), (E,) = one action per env (single-agent envs; with A > 1 it
        # drives every agent of env e, never "agent a of every env"), (E, A) / (E*A,) as is
        if acts.ndim == 0:
            acts = np.broadcast_to(acts, (E, A))
        elif acts.shape == (E,) and not (A > 1 and E == 1):
            acts = np.broadcast_to(acts[:, None], (E, A))
        elif acts.shape == (E, A) or (acts.ndim == 1 and acts.size == E * A):
            acts = acts.reshape(E, A)
        else:
            raise ValueError(f"action must be a scalar, shape ({E},) or shape ({E}, {A}); got {acts.shape}")
        obs, reward, term, trunc, info = self._engine.step(acts.astype(np.int32))  # KeyError on bad action id
        self.time += 1 / self.config["policy_frequency"]
        self.steps += self._hcfg.frames_per_step
        out_info = {"speed": info["speed"][:, 0], "crashed": info["crashed"][:, 0], "action": acts if A > 1 else acts[:, 0]}
        self._agents_step = (reward, info)  # per-agent [E, A] results of this step (scenario subclasses build their info from them)
        if A > 1:
            out_info["agents_rewards"] = reward
            out_info["agents_speed"], out_info["agents_crashed"] = info["speed"], info["crashed"]
        return self._shape_obs(obs), reward[:, 0], term, trunc, out_info

    def _shape_obs(self, obs):
        # KinematicObservation(order="shuffled") (observation.py:273-274): `self.env.np_random.shuffle(obs[1:])` after every
        # observe(), agent by agent -- the engine returned the rows in list order (HWY_C_OBS_UNSORTED), the shuffle draws
        # from each environment's own generator like the reference's (same stream as its spawn in spawn_mode="reference")
        if self._hcfg.flags & _abi.C_OBS_UNSORTED:
            obs = np.array(obs, copy=True)
            for e in range(self.num_envs):
                if self._np_randoms[e] is None:
                    self._np_randoms[e] = np.random.Generator(np.random.PCG64(np.random.SeedSequence(None)))
                for a in range(self._hcfg.num_agents):
                    self._np_randoms[e].shuffle(obs[e, a, 1:])
        if self._hcfg.flags & _abi.C_GRID_IMAGE:  # OccupancyGrid(as_image=True): the engine wrote the uint8 values as f32
            obs = obs.astype(np.uint8)
        return obs[:, 0] if self._hcfg.num_agents == 1 else obs

    # ---- inspection --------------------------------------------------------------------------------
    def get_state(self) -> dict:
        return self._engine.get_state()

    def set_state(self, st: dict) -> None:
        self._ensure_engine().set_state(st)

    def road(self, env_index: int = 0) -> RoadView:
        return RoadView(self._engine.get_state(), env_index)

    def rewards(self, env_index: int = 0) -> dict:
        """HighwayEnv._rewards (highway_env.py:122-139) of one env, from the device state."""
        st = self._engine.get_state()
        c, i = self._hcfg, self._hcfg.agent_index[0]
        x, y, h, v = (st[k][env_index, i] for k in ("x", "y", "heading", "speed"))
        lane, tgt = int(st["lane"][env_index, i]), int(st["target_lane"][env_index, i])
        fs = v * np.cos(h)
        r0, r1 = self.config["reward_speed_range"]
        scaled = 0 + (fs - r0) * (1 - 0) / (r1 - r0)
        on_road = abs(y - lane * c.lane_width) <= c.lane_width / 2 and -5 <= x < c.road_length + 5
        return {"collision_reward": float(bool(st["flags"][env_index, i] & _abi.F_CRASHED)),
                "right_lane_reward": tgt / max(c.lanes_count - 1, 1),
                "high_speed_reward": float(np.clip(scaled, 0, 1)),
                "on_road_reward": float(on_road)}

    def close(self) -> None:
        if self._engine is not None:
            self._engine.close()
            self._engine = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BatchedHighwayEnvFast(BatchedHighwayEnv):
    """E parallel ``highway-fast-v0`` environments (HighwayEnvFast, highway_env.py:154-182)."""
    FAST = True


class BatchedMergeEnv(BatchedHighwayEnv):
    """E parallel ``merge-v0`` environments (MergeEnv, highway_env/envs/merge_env.py:15-190): a two-lane highway,
    an access ramp (``SineLane``) converging onto a forbidden acceleration lane that ends in an ``Obstacle``."""
    SCENARIO = "merge"
    GENERIC = False

    @classmethod
    def default_config(cls) -> dict:
        return _merge.merge_generic_default_config() if cls.GENERIC else _merge.merge_default_config()

    def _spawn_reference(self, generators) -> dict:
        return _merge.spawn_reference_stream(self._hcfg, self.config, self.GENERIC, generators)

    def _device_spawn_args(self) -> dict:
        return {}  # _make_vehicles has no spacing / density / lane parameters (merge_env.py:162-187, 320-363)

    def rewards(self, env_index: int = 0, action=None) -> dict:
        """MergeEnv._rewards (merge_env.py:62-77) of one env, from the device state.  ``action``: the ego's last
        meta-action (the reference evaluates ``action in [0, 2]``); None -> lane_change_reward False."""
        st = self._engine.get_state()
        c = self._hcfg
        e = env_index
        v = st["speed"][e, 0]
        r0, r1 = self.config["reward_speed_range"]
        tab = _merge.table_from_config(c)
        present = (st["flags"][e] & (_abi.F_ABSENT | _abi.F_OBSTACLE)) == 0
        on_merge = present & (st["lane"][e] == c.merge_lane) if c.merge_lane >= 0 else np.zeros_like(present)
        ts, sp = st["target_speed"][e][on_merge], st["speed"][e][on_merge]
        return {"collision_reward": float(bool(st["flags"][e, 0] & _abi.F_CRASHED)),
                "right_lane_reward": float(tab["id"][st["lane"][e, 0]]) / 1,
                "high_speed_reward": float(0 + (v - r0) * (1 - 0) / (r1 - r0)),
                "lane_change_reward": bool(action is not None and np.ndim(action) == 0 and int(action) in (0, 2)),
                "merging_speed_reward": float(np.sum((ts - sp) / ts))}


class BatchedMergeGenericEnv(BatchedMergeEnv):
    """E parallel ``merge-generic-v0`` environments (MergeGenericEnv, merge_env.py:193-371): configurable lane
    count, traffic count and section lengths; ``controlled_vehicles`` > 1 gives BASELINE config 5's multi-agent
    variant (see highwayenv_amd/merge.py)."""
    SCENARIO = "merge-generic"
    GENERIC = True


class _ConnectedLaneNeighboursMixin:
    """ConnectedLaneNeighboursMixin (envs/common/abstract.py:26-37): ``neighbour_vehicles_connected_lanes`` on --
    Road.neighbour_vehicles also searches the lane segments connected to the queried one (road.py:508-529)."""

    @classmethod
    def default_config(cls) -> dict:
        cfg = super().default_config()
        cfg["neighbour_vehicles_connected_lanes"] = True
        return cfg


class BatchedConnectedLaneMergeEnv(_ConnectedLaneNeighboursMixin, BatchedMergeEnv):
    """E parallel ``merge-v1`` environments (ConnectedLaneMergeEnv, merge_env.py:189)."""


class BatchedConnectedLaneMergeGenericEnv(_ConnectedLaneNeighboursMixin, BatchedMergeGenericEnv):
    """E parallel ``merge-generic-v1`` environments (ConnectedLaneMergeGenericEnv, merge_env.py:378)."""


class BatchedIntersectionEnv(BatchedHighwayEnv):
    """E parallel ``intersection-v0`` environments (IntersectionEnv, highway_env/envs/intersection_env.py): a 4-way
    junction of straight and circular lanes, planned routes, ``RegulatedRoad`` priorities, traffic that is cleared and
    spawned while the episode runs, 3 longitudinal meta-actions, Kinematics ``15 x 7`` absolute observation.

    ``spawn_mode="reference"``: traffic management replays the reference's numpy stream on the host
    (``highwayenv_amd/intersection.py``; ``reset(seed=s)`` and every later spawn are the reference's);
    ``spawn_mode="device"``: the step kernel clears / spawns / resets on Philox draws (no host round trip).
    ``config["max_vehicles"]`` (default 32) is the number of vehicle slots per environment."""
    SCENARIO = "intersection"

    @classmethod
    def default_config(cls) -> dict:
        return _ix.intersection_default_config()

    def _define_spaces(self):
        self.config["host_traffic"] = self.spawn_mode == "reference"
        super()._define_spaces()
        A = self._hcfg.num_agents
        if _gym is not None:
            one = _gym.spaces.Discrete(3)
            # MultiAgentAction.space / MultiAgentObservation.space: a Tuple with one entry per agent (action.py:336-340,
            # observation.py:727-731); the observations come stacked as ONE [A, 15, 7] array instead of a tuple of A
            self.single_action_space = one if A == 1 else _gym.spaces.Tuple([one] * A)
        else:
            self.single_action_space = _Discrete(3) if A == 1 else _Tuple([_Discrete(3)] * A)
        self.action_space = self.single_action_space

    def _device_spawn_args(self) -> dict:
        return {}

    def _ego_slots(self, st) -> np.ndarray:  # the ego's slot moves when the vehicle list is re-compacted
        return np.argmax((st["flags"] & _abi.F_CONTROLLED) != 0, axis=1)

    def reset(self, *, seed=None, options: dict | None = None):
        if self.spawn_mode != "reference":
            return super().reset(seed=seed, options=options)
        if options and "config" in options:
            self.configure(options["config"])
        self._define_spaces()
        eng = self._ensure_engine()
        E = self.num_envs
        seeds = [None] * E if seed is None else ([int(seed) + e for e in range(E)] if np.ndim(seed) == 0 else list(seed))
        for e, s in enumerate(seeds):
            if s is not None or self._np_randoms[e] is None:
                self._np_randoms[e] = np.random.Generator(np.random.PCG64(np.random.SeedSequence(s)))
        eng.set_autoreset(False)
        st = _ix.reset_reference_stream(eng, self._hcfg, self.config, self._np_randoms)
        obs = eng.observe()
        self.time[:] = 0
        self.steps = 0
        rows, ego = np.arange(E), self._ego_slots(st)
        info = {"speed": st["speed"][rows, ego].copy(), "crashed": (st["flags"][rows, ego] & _abi.F_CRASHED) != 0, "action": None}
        return self._shape_obs(obs), info

    def step(self, action):
        obs, reward, term, trunc, info = super().step(action)
        # IntersectionEnv._reward: the mean of the agents' rewards, summed left to right (:62-66); _info adds the per-agent
        # rewards and "crashed or arrived" flags (:114-122) -- what MultiAgentWrapper hands out as reward / terminated
        agents_rewards, agents = self._agents_step
        A = self._hcfg.num_agents
        if A > 1:
            total = np.zeros(self.num_envs)
            for a in range(A):
                total = total + agents_rewards[:, a]
            reward = total / A
        info["agents_rewards"] = agents_rewards
        info["agents_terminated"] = agents["crashed"] | agents["arrived"]
        if self.spawn_mode == "reference":  # IntersectionEnv.step: _clear_vehicles, _spawn_vehicle (:136-140)
            _ix.clear_and_spawn_reference_stream(self._engine, self._hcfg, self.config, self._np_randoms)
        return obs, reward, term, trunc, info

    def rewards(self, env_index: int = 0) -> dict:
        """IntersectionEnv._rewards (intersection_env.py:68-77): the agents' _agent_rewards (:96-105) averaged, of one env,
        from the device state."""
        st = self._engine.get_state()
        e = env_index
        tab = _ix.table_from_config(self._hcfg)
        r0, r1 = self.config["reward_speed_range"]
        per_agent = []
        for i in np.nonzero(st["flags"][e] & _abi.F_CONTROLLED)[0][:self._hcfg.num_agents]:
            lane = int(st["lane"][e, i])
            s, lat = _ix.lane_local(tab, lane, (st["x"][e, i], st["y"][e, i]))
            scaled = 0 + (st["speed"][e, i] - r0) * (1 - 0) / (r1 - r0)
            on_road = abs(lat) <= tab["width"][lane] / 2 and -5.0 <= s < tab["length"][lane] + 5.0
            per_agent.append({"collision_reward": float(bool(st["flags"][e, i] & _abi.F_CRASHED)),
                              "high_speed_reward": float(np.clip(scaled, 0, 1)),
                              "arrived_reward": float(bool(tab["exit_lane"][lane]) and s >= 25),
                              "on_road_reward": float(on_road)})
        return {name: sum(r[name] for r in per_agent) / len(per_agent) for name in per_agent[0]}


class BatchedMultiAgentIntersectionEnv(BatchedIntersectionEnv):
    """E parallel ``intersection-multi-agent-v0`` environments (MultiAgentIntersectionEnv, intersection_env.py:348-399):
    ``controlled_vehicles`` (default 2, at most 4) MDP vehicles, agent k entering from road ``o{k % 4}``; actions [E, A],
    observations [E, A, 15, 7], ``reward`` the mean of ``info["agents_rewards"]`` [E, A], ``info["agents_terminated"]``."""

    @classmethod
    def default_config(cls) -> dict:
        cfg = _ix.intersection_default_config()
        cfg["action"] = {"type": "MultiAgentAction", "action_config": cfg["action"]}
        cfg["observation"] = {"type": "MultiAgentObservation", "observation_config": cfg["observation"]}
        cfg["controlled_vehicles"] = 2
        return cfg


# with gymnasium installed the single-environment drop-ins ARE gymnasium.Env's (gym.make / wrappers / checkers accept them)
_GYM_BASES = (_gym.Env,) if _gym is not None else ()


class _SingleEnvMixin:
    """E == 1 with the reference's unbatched signature."""

    def __init__(self, config: dict | None = None, render_mode=None, device: int = 0):
        super().__init__(config, num_envs=1, device=device, render_mode=render_mode)
        self.reset()  # AbstractEnv.__init__ resets (abstract.py:89)

    @property
    def np_random(self):
        """gymnasium.Env.np_random: ONE Generator (wrappers and check_env call .integers / .bit_generator on it)."""
        if self._np_randoms[0] is None:
            self._np_randoms[0] = np.random.Generator(np.random.PCG64(np.random.SeedSequence(None)))
        return self._np_randoms[0]

    @np_random.setter
    def np_random(self, value):
        self._np_randoms = [value]

    def reset(self, *, seed=None, options=None):
        obs, info = super().reset(seed=seed, options=options)
        return obs[0], {"speed": float(info["speed"][0]), "crashed": bool(info["crashed"][0]),
                        "action": self.single_action_space.sample(), "rewards": self.rewards(0)}

    def step(self, action):
        obs, reward, term, trunc, info = super().step(np.asarray([action]).reshape(1, -1))
        return (obs[0], float(reward[0]), bool(term[0]), bool(trunc[0]),
                {"speed": float(info["speed"][0]), "crashed": bool(info["crashed"][0]), "action": action,
                 "rewards": self.rewards(0)})

    @property
    def vehicle(self) -> VehicleView:
        return self.road().vehicles[self._hcfg.agent_index[0]]

    @property
    def controlled_vehicles(self):
        vs = self.road().vehicles
        return [vs[self._hcfg.agent_index[a]] for a in range(self._hcfg.num_agents)]


class HighwayEnv(_SingleEnvMixin, BatchedHighwayEnv, *_GYM_BASES):
    """Drop-in for ``highway_env.envs.highway_env.HighwayEnv`` (``highway-v0``)."""


class HighwayEnvFast(_SingleEnvMixin, BatchedHighwayEnvFast, *_GYM_BASES):
    """Drop-in for ``highway_env.envs.highway_env.HighwayEnvFast`` (``highway-fast-v0``)."""


class _SingleMergeMixin(_SingleEnvMixin):
    def step(self, action):
        obs, reward, term, trunc, info = BatchedHighwayEnv.step(self, np.asarray([action]).reshape(1, -1))
        return (obs[0], float(reward[0]), bool(term[0]), bool(trunc[0]),
                {"speed": float(info["speed"][0]), "crashed": bool(info["crashed"][0]), "action": action,
                 "rewards": self.rewards(0, action)})


class _SingleIntersectionMixin(_SingleEnvMixin):
    def step(self, action):
        obs, reward, term, trunc, info = super().step(action)
        batched = self._agents_step
        info["agents_rewards"] = tuple(float(r) for r in batched[0][0])
        info["agents_terminated"] = tuple(bool(c or a) for c, a in zip(batched[1]["crashed"][0], batched[1]["arrived"][0]))
        return obs, reward, term, trunc, info

    @property
    def vehicle(self) -> VehicleView:
        return self.controlled_vehicles[0]

    @property
    def controlled_vehicles(self):
        st = self._engine.get_state()
        return [VehicleView(st, 0, int(i)) for i in np.nonzero(st["flags"][0] & _abi.F_CONTROLLED)[0]]


class IntersectionEnv(_SingleIntersectionMixin, BatchedIntersectionEnv, *_GYM_BASES):
    """Drop-in for ``highway_env.envs.intersection_env.IntersectionEnv`` (``intersection-v0``)."""


class MultiAgentIntersectionEnv(_SingleIntersectionMixin, BatchedMultiAgentIntersectionEnv, *_GYM_BASES):
    """Drop-in for ``highway_env.envs.intersection_env.MultiAgentIntersectionEnv`` (``intersection-multi-agent-v0``): ``step``
    takes a tuple of A meta-actions and returns the A observations stacked in one [A, 15, 7] array."""


class BatchedConnectedLaneMultiAgentIntersectionEnv(_ConnectedLaneNeighboursMixin, BatchedMultiAgentIntersectionEnv):
    """E parallel ConnectedLaneMultiAgentIntersectionEnv (intersection_env.py:405-408)."""


class ConnectedLaneMultiAgentIntersectionEnv(_SingleIntersectionMixin, BatchedConnectedLaneMultiAgentIntersectionEnv, *_GYM_BASES):
    """Drop-in for ``highway_env.envs.intersection_env.ConnectedLaneMultiAgentIntersectionEnv``."""


class _MultiAgentWrapperMixin:
    """``MultiAgentWrapper`` (envs/common/abstract.py:468-477), which the reference registers on top of the
    ``intersection-multi-agent-v1`` / ``-v2`` ids: ``reward`` and ``terminated`` become the per-agent tuples of the info dict."""

    def step(self, action):
        obs, _, _, truncated, info = super().step(action)
        return obs, info["agents_rewards"], info["agents_terminated"], truncated, info


class MultiAgentIntersectionEnvV1(_MultiAgentWrapperMixin, MultiAgentIntersectionEnv):
    """Drop-in for ``gym.make("intersection-multi-agent-v1")``: MultiAgentIntersectionEnv under MultiAgentWrapper."""


class MultiAgentIntersectionEnvV2(_MultiAgentWrapperMixin, ConnectedLaneMultiAgentIntersectionEnv):
    """Drop-in for ``gym.make("intersection-multi-agent-v2")``: ConnectedLaneMultiAgentIntersectionEnv under MultiAgentWrapper."""


class _BatchedMultiAgentWrapperMixin:
    def step(self, action):
        obs, _, _, truncated, info = super().step(action)
        return obs, info["agents_rewards"], info["agents_terminated"], truncated, info


class BatchedMultiAgentIntersectionEnvV1(_BatchedMultiAgentWrapperMixin, BatchedMultiAgentIntersectionEnv):
    """E parallel ``intersection-multi-agent-v1``: ``reward`` [E, A] and ``terminated`` [E, A] per agent (the auto-reset of the
    engine still follows the ENVIRONMENT's terminated | truncated)."""


class BatchedMultiAgentIntersectionEnvV2(_BatchedMultiAgentWrapperMixin, BatchedConnectedLaneMultiAgentIntersectionEnv):
    """E parallel ``intersection-multi-agent-v2``."""


class BatchedConnectedLaneIntersectionEnv(_ConnectedLaneNeighboursMixin, BatchedIntersectionEnv):
    """E parallel ``intersection-v2`` environments (ConnectedLaneIntersectionEnv, intersection_env.py:423)."""


class ConnectedLaneIntersectionEnv(_SingleIntersectionMixin, BatchedConnectedLaneIntersectionEnv, *_GYM_BASES):
    """Drop-in for ``highway_env.envs.intersection_env.ConnectedLaneIntersectionEnv`` (``intersection-v2``)."""


class MergeEnv(_SingleMergeMixin, BatchedMergeEnv, *_GYM_BASES):
    """Drop-in for ``highway_env.envs.merge_env.MergeEnv`` (``merge-v0``)."""


class MergeGenericEnv(_SingleMergeMixin, BatchedMergeGenericEnv, *_GYM_BASES):
    """Drop-in for ``highway_env.envs.merge_env.MergeGenericEnv`` (``merge-generic-v0``)."""


class ConnectedLaneMergeEnv(_SingleMergeMixin, BatchedConnectedLaneMergeEnv, *_GYM_BASES):
    """Drop-in for ``highway_env.envs.merge_env.ConnectedLaneMergeEnv`` (``merge-v1``)."""


class ConnectedLaneMergeGenericEnv(_SingleMergeMixin, BatchedConnectedLaneMergeGenericEnv, *_GYM_BASES):
    """Drop-in for ``highway_env.envs.merge_env.ConnectedLaneMergeGenericEnv`` (``merge-generic-v1``)."""


# The ids `highway_env/__init__.py:30-190` registers for this path (gym.make(id) -> the entry-point class): single
# environment and batched class (the -v1 / -v2 multi-agent intersection ids are the reference's classes under its
# MultiAgentWrapper: the drop-ins fold the wrapper in).  Ids of scenarios outside the path (parking, racetrack, roundabout,
# ...) and the continuous-action intersection id raise KeyError.
REGISTRY = {
    "highway-v0": (HighwayEnv, BatchedHighwayEnv),
    "highway-fast-v0": (HighwayEnvFast, BatchedHighwayEnvFast),
    "merge-v0": (MergeEnv, BatchedMergeEnv),
    "merge-v1": (ConnectedLaneMergeEnv, BatchedConnectedLaneMergeEnv),
    "merge-generic-v0": (MergeGenericEnv, BatchedMergeGenericEnv),
    "merge-generic-v1": (ConnectedLaneMergeGenericEnv, BatchedConnectedLaneMergeGenericEnv),
    "intersection-v0": (IntersectionEnv, BatchedIntersectionEnv),
    "intersection-v2": (ConnectedLaneIntersectionEnv, BatchedConnectedLaneIntersectionEnv),
    "intersection-multi-agent-v0": (MultiAgentIntersectionEnv, BatchedMultiAgentIntersectionEnv),
    "intersection-multi-agent-v1": (MultiAgentIntersectionEnvV1, BatchedMultiAgentIntersectionEnvV1),
    "intersection-multi-agent-v2": (MultiAgentIntersectionEnvV2, BatchedMultiAgentIntersectionEnvV2),
}


GYM_NAMESPACE = "highwayenv_amd"


def register_envs(namespace: str | None = GYM_NAMESPACE) -> list:
    """Register the single-environment drop-ins of REGISTRY with gymnasium, like ``highway_env/__init__.py:22-187`` does for
    the reference at import time: ``gym.make("highwayenv_amd/highway-fast-v0", config={...})``.  The ids live in their own
    namespace so that this package and ``highway_env`` can be installed side by side; ``namespace=None`` registers the
    reference's bare ids (for a box without the reference).  Idempotent; returns the ids it registered; does nothing
    (returns []) when gymnasium is not importable.  Importing ``highwayenv_amd`` calls it."""
    if _gym is None:
        return []
    from gymnasium.envs.registration import register, registry
    done = []
    for env_id, (single, _batched) in REGISTRY.items():
        full = f"{namespace}/{env_id}" if namespace else env_id
        if full in registry:
            continue
        register(id=full, entry_point=f"{__name__}:{single.__name__}")
        done.append(full)
    return done


def batched_class(env_id: str):
    """The ``Batched*Env`` class of an id of REGISTRY (accepts the "highwayenv_amd/" namespace prefix too)."""
    return REGISTRY[env_id.split("/", 1)[-1]][1]


def make(env_id: str, config: dict = None, **kwargs):
    """``gym.make(env_id, config=config)`` for the ids of REGISTRY: the drop-in single environment."""
    return REGISTRY[env_id][0](config, **kwargs)


def make_vec(env_id: str, num_envs: int, config: dict = None, **kwargs):
    """``num_envs`` environments of ``env_id`` stepped by one kernel launch (gymnasium "next-step" auto-reset)."""
    return REGISTRY[env_id][1](config, num_envs=num_envs, **kwargs)


def _copy_config(cfg):
    return copy.deepcopy(cfg)
