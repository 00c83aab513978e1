"""highwayenv_amd -- MI355X-native batched step engine for HighwayEnv's Road.act()/Road.step()
hot path (gfx950 HIP kernels behind the C-ABI of include/hwy_engine.h).

Importing the package does not load the native library; creating an engine / environment does,
and fails loudly when libhwy_engine.so or a GPU is missing (there is no CPU fallback).
"""
from ._abi import (highway_default_config, highway_fast_default_config, make_config)  # noqa: F401

__version__ = "0.1.0"

__all__ = ["highway_default_config", "highway_fast_default_config", "make_config", "__version__"]
