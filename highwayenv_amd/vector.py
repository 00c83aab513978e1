"""A ``gymnasium.vector.VectorEnv``-shaped front end of the batched environments.

What the reference's users plug into their training loops is a vector env -- ``gym.vector.SyncVectorEnv([...],
autoreset_mode="SameStep")`` in ``/root/reference/tests/envs/test_gym.py:158-165``, ``make_vec_env(..., SubprocVecEnv)`` in
``scripts/sb3_highway_ppo.py:16-18`` -- i.e. E Python processes stepping E copies of ``AbstractEnv.step``.  Here the E
environments are ONE engine (one kernel launch per step), and this class gives them that interface:

* ``num_envs``, ``single_observation_space`` / ``single_action_space`` and the batched ``observation_space`` / ``action_space``;
* ``reset(seed=..., options=...) -> (obs [E, ...], infos)``, ``step(actions) -> (obs, rewards f64 [E], terminations, truncations,
  infos)``, ``step_async`` / ``step_wait``, ``close``;
* a DECLARED autoreset mode (``metadata["autoreset_mode"]``, gymnasium >= 1.0 vocabulary):

  - ``"NextStep"`` (default; gymnasium's default too): an environment that ends at step t returns its terminal observation with
    ``terminated`` / ``truncated`` set; the call of step t + 1 re-spawns it INSTEAD of stepping (its action is ignored) and
    returns the first observation of the new episode with reward 0 and both flags False.  That is exactly what the step kernels
    do on the device (``hwy_set_autoreset``), so this mode costs nothing;
  - ``"SameStep"`` (what the reference's vectorisation test asks for): the step that ends an episode already returns the
    first observation of the next one; the terminal observation and info travel in ``infos["final_obs"]`` / ``infos["final_info"]``
    (object arrays, ``None`` where the episode goes on) with the ``_final_obs`` / ``_final_info`` masks.  One masked device reset
    per step in which some episode ended;
  - ``"Disabled"``: finished environments keep stepping until the caller resets.

* ``output="torch"``: observations, rewards and flags are ``torch`` tensors that ALIAS the buffers the kernel writes (no host
  round trip), actions may be a device tensor, and everything is ordered on torch's current stream -- a policy network can read
  ``obs`` and write ``actions`` without leaving the GPU ("keeping the policy on-GPU").  ``NextStep`` /
  ``Disabled`` only.  ``rollout(actions[K])`` exposes ``hwy_rollout_device`` (K steps per launch) in this mode.

With gymnasium installed the class IS a ``gymnasium.vector.VectorEnv``; without it (this image) the same attributes exist on a
plain object.
"""
from __future__ import annotations

import numpy as np

from . import _abi, envs as _envs

try:  # optional
    import gymnasium as _gym
    _Base = _gym.vector.VectorEnv
except Exception:  # pragma: no cover - gymnasium is absent in the build image
    _gym = None
    _Base = object

AUTORESET_MODES = ("NextStep", "SameStep", "Disabled")


class _Batched:
    """Batched view of a single-environment space (stand-in for gymnasium.vector.utils.batch_space without gymnasium)."""

    def __init__(self, single, n):
        self.single, self.n = single, n
        self.shape = (n, *getattr(single, "shape", ()))
        self.dtype = getattr(single, "dtype", None)

    def sample(self):
        return np.stack([np.asarray(self.single.sample()) for _ in range(self.n)])

    def contains(self, x):
        return len(x) == self.n and all(self.single.contains(v) for v in x)


class HighwayVectorEnv(_Base):
    """E environments of one scenario behind the vector-env interface (see the module docstring).

    ``env``: a ``Batched*Env`` class or instance, or an id of ``highwayenv_amd.envs.REGISTRY`` ("highway-fast-v0", "merge-v0",
    "intersection-v0", ...)."""

    def __init__(self, env="highway-fast-v0", num_envs: int = 1, config: dict | None = None, autoreset_mode: str = "NextStep",
                 output: str = "numpy", device: int = 0, spawn_mode: str = "device", stream=None, **kwargs):
        mode = getattr(autoreset_mode, "value", autoreset_mode)  # gymnasium.vector.AutoresetMode or its string
        mode = {"next_step": "NextStep", "same_step": "SameStep", "disabled": "Disabled"}.get(str(mode).lower(), str(mode))
        if mode not in AUTORESET_MODES:
            raise ValueError(f"autoreset_mode must be one of {AUTORESET_MODES}")
        if output not in ("numpy", "torch"):
            raise ValueError("output must be 'numpy' or 'torch'")
        if output == "torch" and mode == "SameStep":
            raise ValueError("output='torch' supports the NextStep and Disabled autoreset modes (SameStep merges a masked reset on the host)")
        self.autoreset_mode = mode
        self.output = output
        if output == "torch":
            import torch  # the device-pointer path: torch owns the output tensors and the stream
            self._torch = torch
            torch.cuda.set_device(device)
            # the engine gets a stream of its own (the handle of torch's default stream is NULL, which hwy_create reads as
            # "create one"); every call orders it after the caller's current stream and the caller's stream after it.
            # ``stream=`` (a torch.cuda.Stream, not the default one): the engine runs ON that stream, and a step() issued while
            # it is the current stream needs no cross-stream ordering at all (two event record / wait pairs less per step).
            if stream is None:  # built under `with torch.cuda.stream(s)`: that stream; under the default stream: a new one
                cur = torch.cuda.current_stream(device)
                stream = cur if cur.cuda_stream else torch.cuda.Stream(device=device)
            self._stream = stream
            stream = self._stream.cuda_stream
            if not stream:
                raise ValueError("stream= must be a real torch.cuda.Stream (the default stream's handle is NULL)")
        elif stream is not None:
            raise ValueError("stream= needs output='torch'")
        if isinstance(env, str):
            env = _envs.batched_class(env)
        if isinstance(env, type):
            env = env(config, num_envs=num_envs, device=device, spawn_mode=spawn_mode, autoreset=(mode == "NextStep"),
                      stream=stream, **kwargs)
        elif config:
            env.configure(config)
        self.env = env
        env.autoreset = mode == "NextStep"
        self.num_envs = env.num_envs
        self.single_observation_space, self.single_action_space = env.single_observation_space, env.single_action_space
        if _gym is not None:
            self.observation_space = _gym.vector.utils.batch_space(self.single_observation_space, self.num_envs)
            self.action_space = _gym.vector.utils.batch_space(self.single_action_space, self.num_envs)
        else:
            self.observation_space = _Batched(self.single_observation_space, self.num_envs)
            self.action_space = _Batched(self.single_action_space, self.num_envs)
        self.metadata = {"autoreset_mode": (getattr(_gym.vector, "AutoresetMode", None)(_snake(mode))
                                            if _gym is not None and hasattr(_gym.vector, "AutoresetMode") else mode),
                         "render_modes": []}
        self.render_mode = None
        self.closed = False
        self._pending = None
        self._needs_reset = np.zeros(self.num_envs, bool)  # SameStep bookkeeping is immediate; NextStep lives in the engine
        self._dev = None

    # ---- reset / step ------------------------------------------------------------------------------------------------------
    def reset(self, *, seed=None, options=None):
        obs, info = self.env.reset(seed=seed, options=options)
        self._needs_reset[:] = False
        infos = {"speed": np.asarray(info["speed"], np.float64), "crashed": np.asarray(info["crashed"], bool)}
        if self.output == "torch":
            self._bind_device_buffers()
            t = self._torch
            return t.from_numpy(np.ascontiguousarray(obs)).to(self._dev["obs"].device), infos
        return obs, infos

    def step_async(self, actions) -> None:
        self._pending = actions

    def step_wait(self):
        actions, self._pending = self._pending, None
        if actions is None:
            raise RuntimeError("step_wait() without step_async()")
        return self.step(actions)

    def step(self, actions):
        if self.output == "torch":
            return self._step_torch(actions)
        obs, reward, term, trunc, info = self.env.step(actions)
        infos = {"speed": np.asarray(info["speed"], np.float64), "crashed": np.asarray(info["crashed"], bool)}
        for k in ("agents_rewards", "agents_terminated", "rewards"):
            if k in info:
                infos[k] = info[k]
        done = term | trunc
        if self.autoreset_mode == "SameStep" and done.any():
            obs = np.array(obs, copy=True)
            final_obs = np.full(self.num_envs, None, dtype=object)
            final_info = np.full(self.num_envs, None, dtype=object)
            for e in np.nonzero(done)[0]:
                final_obs[e] = obs[e].copy()
                final_info[e] = {"speed": float(infos["speed"][e]), "crashed": bool(infos["crashed"][e])}
            obs[done] = self.env.reset_done(done)
            infos.update({"final_obs": final_obs, "_final_obs": done.copy(), "final_info": final_info, "_final_info": done.copy()})
        return obs, np.asarray(reward, np.float64), term, trunc, infos

    # ---- device-resident outputs ------------------------------------------------------------------------------------------------
    def _bind_device_buffers(self):
        t, E, A = self._torch, self.num_envs, self.env._hcfg.num_agents
        dev = t.device("cuda", self.env.device)
        shape = _abi.obs_shape(self.env._hcfg)
        self._dev = {"obs": t.empty((E, A, *shape), dtype=t.float32, device=dev), "reward": t.empty((E, A), dtype=t.float64, device=dev),
                     "terminated": t.empty(E, dtype=t.uint8, device=dev), "truncated": t.empty(E, dtype=t.uint8, device=dev),
                     "speed": t.empty((E, A), dtype=t.float64, device=dev), "crashed": t.empty((E, A), dtype=t.uint8, device=dev)}
        d = self._dev
        self._n_actions = E * A
        self._out_ptrs = tuple(d[k].data_ptr() for k in ("obs", "reward", "terminated", "truncated", "speed", "crashed"))
        # what step() hands out: views of the planes the kernel writes (bool views of the 0 / 1 flag bytes; the intersection's
        # info_crashed carries has_arrived in bit 1, include/hwy_engine.h, so its `crashed` is computed per step)
        plain_crashed = self.env._hcfg.scenario != _abi.SCENARIO_INTERSECTION
        self._views = {"obs": d["obs"][:, 0] if A == 1 else d["obs"], "reward": d["reward"][:, 0],
                       "terminated": d["terminated"].view(t.bool), "truncated": d["truncated"].view(t.bool),
                       "speed": d["speed"][:, 0], "crashed": d["crashed"][:, 0].view(t.bool) if plain_crashed else None}

    def _device_actions(self, actions, lead=()):
        t, E, A = self._torch, self.num_envs, self.env._hcfg.num_agents
        a = actions if isinstance(actions, t.Tensor) else t.as_tensor(np.asarray(actions))
        a = a.to(device=self._dev["obs"].device, dtype=t.int32)
        if a.dim() == len(lead) + 1 and A > 1:
            a = a.unsqueeze(-1).expand(*lead, E, A)
        return a.reshape(*lead, E, A).contiguous()

    def _step_torch(self, actions):
        """One launch on torch's stream; the returned tensors alias the engine's output buffers (valid until the next step).
        Action ids are NOT validated on this path (include/hwy_engine.h: hwy_step_device); ids outside the table act as IDLE.
        The host side of a step is kept to the launch and two stream waits: no elementwise kernel is launched for the outputs
        (the flag planes hold 0 / 1 bytes and are returned as zero-copy bool VIEWS), and an int32 device tensor of the right
        shape goes to the kernel as it is."""
        t, d = self._torch, self._dev
        if not (isinstance(actions, t.Tensor) and actions.dtype == t.int32 and actions.is_cuda and actions.is_contiguous()
                and actions.numel() == self._n_actions):
            actions = self._device_actions(actions)
        cur = t.cuda.current_stream()
        same = cur.cuda_stream == self._stream.cuda_stream
        if not same:
            self._stream.wait_stream(cur)      # the actions (and the consumer of the previous outputs) come first
        self.env._engine.step_device(actions.data_ptr(), *self._out_ptrs)
        if not same:
            cur.wait_stream(self._stream)      # whatever the caller enqueues next sees this step's outputs
            actions.record_stream(self._stream)
        v = self._views
        infos = {"speed": v["speed"], "crashed": v["crashed"] if v["crashed"] is not None else (d["crashed"][:, 0] & 1).bool()}
        return v["obs"], v["reward"], v["terminated"], v["truncated"], infos

    def rollout(self, actions):
        """``actions`` [K, E(, A)] -> (obs [K, E, ...], rewards [K, E], terminations [K, E], truncations [K, E]) as device tensors:
        K policy steps in one call of ``hwy_rollout_device`` (one LAUNCH on the one-wavefront kernel), NextStep autoreset between
        the steps.  ``output="torch"`` only."""
        if self.output != "torch":
            raise ValueError("rollout() needs output='torch'")
        t, E, A = self._torch, self.num_envs, self.env._hcfg.num_agents
        K = int(actions.shape[0])
        a = self._device_actions(actions, lead=(K,))
        dev = self._dev["obs"].device
        obs = t.empty((K, E, A, *_abi.obs_shape(self.env._hcfg)), dtype=t.float32, device=dev)
        rew = t.empty((K, E, A), dtype=t.float64, device=dev)
        term, trunc = t.empty((K, E), dtype=t.uint8, device=dev), t.empty((K, E), dtype=t.uint8, device=dev)
        cur = t.cuda.current_stream()
        self._stream.wait_stream(cur)
        self.env._engine.rollout_device(K, a.data_ptr(), obs.data_ptr(), rew.data_ptr(), term.data_ptr(), trunc.data_ptr())
        cur.wait_stream(self._stream)
        for x in (a, obs, rew, term, trunc):
            x.record_stream(self._stream)
        return (obs[:, :, 0] if A == 1 else obs), rew[:, :, 0], term.bool(), trunc.bool()

    # ---- the rest of the interface ----------------------------------------------------------------------------------------------
    @property
    def stream(self):
        """output='torch': the torch.cuda.Stream the engine launches on (the ``stream=`` argument, or the one created for it).
        A step() issued while it is the CURRENT stream (``with torch.cuda.stream(venv.stream): ...``) needs no cross-stream
        ordering: 44.6 us per step at 4096 x 51 against 68 with an event wait on either side of the launch."""
        return getattr(self, "_stream", None)

    def close(self, **kwargs):
        if not self.closed:
            self.env.close()
            self.closed = True

    def close_extras(self, **kwargs):
        self.close()

    @property
    def unwrapped(self):
        return self

    def call(self, name, *args, **kwargs):
        """gymnasium.vector.VectorEnv.call: the batched environment is ONE object; its answer is returned once per environment."""
        attr = getattr(self.env, name)
        out = attr(*args, **kwargs) if callable(attr) else attr
        return tuple(out for _ in range(self.num_envs))

    def get_attr(self, name):
        return self.call(name)


def _snake(mode: str) -> str:
    return {"NextStep": "NextStep", "SameStep": "SameStep", "Disabled": "Disabled"}[mode]
