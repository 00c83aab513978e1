"""Backends the parity tests run against.

* ``emu``  -- the product's kernel SOURCE executed on the CPU by tests/emu (no GPU needed);
* ``hip``  -- the product itself: libhwy_engine.so on a real MI355X through the C-ABI.
"""
import pytest


def make_engine(backend: str, cfg):
    if backend == "emu":
        from tests.emu.emu import EmuEngine
        return EmuEngine(cfg)
    from highwayenv_amd.engine import Engine
    return Engine(cfg)


BACKENDS = [pytest.param("emu", id="emu"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]
