"""Host reset (highwayenv_amd/spawn.py) against the reference's reset(seed=s) states."""
import numpy as np
import pytest

from highwayenv_amd import _abi, spawn
from tests.golden_util import ALL, Golden


@pytest.mark.parametrize("name", ALL)
def test_spawn_matches_reference_reset(name):
    g = Golden(name)
    cfg = _abi.make_config(g.config, g.E, fast=g.fast)
    st = spawn.spawn_reference_stream(cfg, g.seeds, g.config["ego_spacing"], g.config["vehicles_density"],
                                      g.config["initial_lane_id"])
    want = g.state("init")
    for k in ["lane", "target_lane", "flags", "speed_index"]:
        np.testing.assert_array_equal(st[k], want[k], err_msg=k)
    for k in ["x", "y", "heading", "speed", "target_speed", "timer", "delta"]:
        np.testing.assert_allclose(st[k], want[k], rtol=0, atol=1e-12, err_msg=k)


def test_agent_indices_multi_agent():
    # HighwayEnv._create_vehicles with near_split(10, 3) = [4, 3, 3]
    assert _abi.agent_indices(10, 3) == [0, 5, 9]
    assert _abi.agent_indices(50, 1) == [0]
