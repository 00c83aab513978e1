"""highwayenv_amd -- MI355X-native batched step engine for HighwayEnv's Road.act()/Road.step()
hot path (gfx950 HIP kernels behind the C-ABI of include/hwy_engine.h).

Importing the package does not load the native library; creating an engine / environment does,
and fails loudly when libhwy_engine.so or a GPU is missing (there is no CPU fallback).
"""
from ._abi import (highway_default_config, highway_fast_default_config, make_config)  # noqa: F401

__version__ = "0.1.0"

__all__ = ["highway_default_config", "highway_fast_default_config", "make_config", "register_envs", "__version__"]


def register_envs(namespace="highwayenv_amd"):
    """gymnasium registration of the drop-in environments (highwayenv_amd.envs.register_envs); a no-op without gymnasium."""
    from .envs import register_envs as _register
    return _register(namespace)


def _register_if_gymnasium_is_installed() -> None:
    # the reference registers its ids on `import highway_env` (highway_env/__init__.py:190); same here, but only when
    # gymnasium exists -- the package never needs it otherwise, and importing it must not load the native library
    import importlib.util
    try:
        if importlib.util.find_spec("gymnasium") is not None:
            register_envs()
    except Exception:  # a broken / partial gymnasium must not make the engine unusable
        pass


_register_if_gymnasium_is_installed()
