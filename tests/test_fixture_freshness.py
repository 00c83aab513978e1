"""The committed fixtures under tests/golden are what the committed generators produce from the UNMODIFIED reference: env 0 of
every fixture is regenerated here (build container only: needs /root/reference) and compared with the committed .npz, array by
array, bit for bit.  A generator edit that is not followed by a regeneration, or a fixture edited by hand, fails here."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import ref_stub

pytestmark = [pytest.mark.reference,
              pytest.mark.skipif(not ref_stub.reference_available(), reason="needs the reference package (build container)")]
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_mods = {}


def _generator(fname):
    if fname not in _mods:
        spec = importlib.util.spec_from_file_location(fname[:-3], os.path.join(GOLDEN, fname))
        _mods[fname] = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(_mods[fname])
    return _mods[fname]


def _scenarios():
    if not ref_stub.reference_available():
        return []
    out = []
    for fname in ("make_golden.py", "make_golden_merge.py", "make_golden_intersection.py"):
        out += [pytest.param(fname, sc["name"], id=sc["name"]) for sc in _generator(fname).SCENARIOS]
    return out


@pytest.mark.parametrize("fname,name", _scenarios())
def test_env0_of_the_committed_fixture_is_what_the_generator_makes(fname, name):
    gen = _generator(fname)
    sc = next(s for s in gen.SCENARIOS if s["name"] == name)
    fresh = gen.run_scenario(sc, only_envs=[0])
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    assert set(fresh) == set(z.files), sorted(set(fresh) ^ set(z.files))
    n_arrays = 0
    for k, a in fresh.items():
        a, b = np.asarray(a), z[k]
        if k == "meta":
            a, b = a[1:], b[1:]   # (meta[0] = number of envs)
        elif a.shape != b.shape:  # the env axis: the committed array holds all envs (or the first `frames_for`), the fresh one env 0
            diff = [ax for ax in range(a.ndim) if a.shape[ax] != b.shape[ax]]
            assert a.ndim == b.ndim and len(diff) == 1 and a.shape[diff[0]] == 1, (k, a.shape, b.shape)
            b = np.take(b, [0], axis=diff[0])
        assert a.dtype == b.dtype, (k, a.dtype, b.dtype)
        if a.dtype.kind == "f":
            np.testing.assert_array_equal(a.view(np.uint64 if a.dtype == np.float64 else np.uint32),
                                          b.view(np.uint64 if b.dtype == np.float64 else np.uint32), err_msg=k)  # bits, NaNs included
        else:
            np.testing.assert_array_equal(a, b, err_msg=k)
        n_arrays += 1
    assert n_arrays > 40
