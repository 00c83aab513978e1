"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

In-memory stand-ins for ``gymnasium`` and ``pygame`` so that the *unmodified*
reference package under /root/reference can be imported in the build
container (where neither is installed, and there is no network).  Used only
by ``tests/golden/make_golden.py`` (fixture generation) and by tests that are
skipped when /root/reference is absent (i.e. on the GPU box).

The stub copies no reference code.  The only behaviour it has to reproduce is
the part of ``gymnasium.Env.reset(seed=...)`` the reference relies on:
``self.np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))``
(gymnasium.utils.seeding.np_random, gymnasium 1.x as pinned by the
reference's uv.lock).
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("HWY_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "highway_env"))


class _Space:
    def __init__(self, shape=None, dtype=None, seed=None):
        self.shape = shape
        self.dtype = dtype

    def sample(self):
        return None

    def contains(self, x):
        return True


class _Box(_Space):
    def __init__(self, low=None, high=None, shape=None, dtype=np.float32, seed=None):
        if shape is None and low is not None and np.ndim(low) > 0:
            shape = np.shape(low)
        super().__init__(shape, np.dtype(dtype))
        self.low, self.high = low, high

    def sample(self):
        return np.zeros(self.shape, dtype=self.dtype)


class _Discrete(_Space):
    def __init__(self, n, seed=None, start=0):
        super().__init__((), np.dtype(np.int64))
        self.n = int(n)
        self.start = start

    def sample(self):
        # gymnasium samples from the space's *own* RNG, never from env.np_random,
        # so returning a constant leaves the env's random stream untouched.
        return self.start


class _Tuple(_Space):
    def __init__(self, spaces=(), seed=None):
        super().__init__()
        self.spaces = tuple(spaces)

    def sample(self):
        return tuple(s.sample() for s in self.spaces)


class _Dict(_Space):
    def __init__(self, spaces=None, seed=None, **kw):
        super().__init__()
        self.spaces = dict(spaces or {}, **kw)

    def sample(self):
        return {k: s.sample() for k, s in self.spaces.items()}


class _Env:
    metadata: dict = {}
    spec = None
    render_mode = None
    _np_random = None

    @property
    def np_random(self):
        if self._np_random is None:
            self._np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence()))
        return self._np_random

    @np_random.setter
    def np_random(self, value):
        self._np_random = value

    @property
    def unwrapped(self):
        return self

    def reset(self, *, seed=None, options=None):
        if seed is not None:
            self._np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))

    def close(self):
        pass


class _Wrapper(_Env):
    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        return getattr(self.env, name)

    def step(self, action):
        return self.env.step(action)

    def reset(self, **kw):
        return self.env.reset(**kw)


class _RecordConstructorArgs:
    def __init__(self, **kw):
        pass


_CLASS_DEFAULTS = {}


def install() -> None:
    """Put the stubs in sys.modules and the reference on sys.path (idempotent)."""
    sys.dont_write_bytecode = True  # /root/reference is read-only for us: importing it must not leave __pycache__ behind
    if "gymnasium" not in sys.modules:
        gym = types.ModuleType("gymnasium")
        gym.Env = _Env
        gym.Wrapper = _Wrapper
        gym.logger = types.SimpleNamespace(warn=lambda *a, **k: None)

        spaces = types.ModuleType("gymnasium.spaces")
        spaces.Space, spaces.Box, spaces.Discrete = _Space, _Box, _Discrete
        spaces.Tuple, spaces.Dict = _Tuple, _Dict
        gym.spaces = spaces

        utils = types.ModuleType("gymnasium.utils")
        utils.RecordConstructorArgs = _RecordConstructorArgs
        gym.utils = utils

        wrappers = types.ModuleType("gymnasium.wrappers")
        wrappers.RecordVideo = type("RecordVideo", (_Wrapper,), {})
        gym.wrappers = wrappers

        envs = types.ModuleType("gymnasium.envs")
        registration = types.ModuleType("gymnasium.envs.registration")
        registration.register = lambda *a, **k: None
        # "highway-v0" present => highway_env/__init__.py's idempotency guard returns
        # before it imports (and registers) every scenario.
        registration.registry = {"highway-v0": None}
        envs.registration = registration
        gym.envs = envs

        sys.modules.update({
            "gymnasium": gym,
            "gymnasium.spaces": spaces,
            "gymnasium.utils": utils,
            "gymnasium.wrappers": wrappers,
            "gymnasium.envs": envs,
            "gymnasium.envs.registration": registration,
        })
    if "pygame" not in sys.modules:
        pg = types.ModuleType("pygame")
        pg.Surface = type("Surface", (), {})
        pg.Rect = type("Rect", (), {})
        pg.SRCALPHA = 0
        sys.modules["pygame"] = pg
    if REFERENCE_ROOT not in sys.path:
        sys.path.append(REFERENCE_ROOT)   # (at the END: the reference has a `tests` package of its own, ours must keep winning)
    if reference_available() and not _CLASS_DEFAULTS:
        _CLASS_DEFAULTS[None] = True   # (re-entrancy guard: restore_class_defaults calls install)
        restore_class_defaults()       # snapshot the class attributes as imported, before any environment is created


def restore_class_defaults() -> None:
    """IntersectionEnv._make_vehicles writes its IDM parameters onto the vehicle CLASS (intersection_env.py:243-247:
    DISTANCE_WANTED 7, COMFORT_ACC_MAX 6, COMFORT_ACC_MIN -3), where they stay for every environment the process creates later --
    a highway env stepped after an intersection env in one process drives with the intersection's parameters.  The fixture
    generators and the live-reference tests call this before every scenario so that a trace does not depend on which scenarios
    ran before it in the same process (the values restored are the ones the classes were IMPORTED with)."""
    if not reference_available():
        return
    install()
    from highway_env.vehicle import behavior
    for cls in (behavior.IDMVehicle, behavior.AggressiveVehicle, behavior.DefensiveVehicle, behavior.LinearVehicle):
        saved = _CLASS_DEFAULTS.setdefault(cls.__name__, {k: v for k, v in vars(cls).items() if k.isupper()})
        for k, v in saved.items():
            if getattr(cls, k) is not v:
                setattr(cls, k, v)
