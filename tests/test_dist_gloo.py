"""world_size-2 gloo test (CPU) of the env sharding + packed gather used on N>1 GPUs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from highwayenv_amd import _abi
from highwayenv_amd.dist import PackedStepOutputs, scatter_actions, shard_range


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _config(kind, E):
    """The three output shapes the engine produces: [E,1,5,5] (highway-fast), [E,4,5,5] (merge, 4 agents), [E,1,4,11,11]
    (intersection with the OccupancyGrid observation)."""
    if kind == "merge_ma4":
        from highwayenv_amd import merge
        c = merge.merge_generic_default_config()
        c.update({"lanes_count": 4, "vehicles_count": 40, "controlled_vehicles": 4,
                  "action": {"type": "MultiAgentAction", "action_config": {"type": "DiscreteMetaAction"}},
                  "observation": {"type": "MultiAgentObservation", "observation_config": {"type": "Kinematics"}}})
        return _abi.make_config(c, E, scenario="merge-generic")
    if kind == "intersection_grid":
        from highwayenv_amd import intersection
        c = intersection.intersection_default_config()
        c.update({"observation": {"type": "OccupancyGrid"}})
        return _abi.make_config(c, E, scenario="intersection")
    return _abi.make_config(_abi.highway_fast_default_config(), E, fast=True)


def _worker(rank, world, port, E, q, kind="fast"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = _config(kind, E)
        out = PackedStepOutputs(cfg, "cpu", world, rank)
        v = out.views()
        env_ids = torch.tensor(list(shard_range(E * world, world, rank)))
        # fake "engine outputs": every field encodes the global env id
        v["reward"][:, 0] = env_ids.double() + 0.25
        v["info_speed"][:, 0] = env_ids.double() * 2
        v["obs"][:] = env_ids.float().view(-1, *([1] * (v["obs"].dim() - 1)))
        v["terminated"][:] = (env_ids % 2).to(torch.uint8)
        v["truncated"][:] = (env_ids % 3 == 0).to(torch.uint8)
        v["info_crashed"][:, 0] = (env_ids % 5 == 0).to(torch.uint8)
        got = out.gather_to_rank0()
        work = out.gather_async()  # the overlapped form used by bench.py
        work.wait()
        per_rank = out.rank0_views()
        if rank == 0:
            assert len(per_rank) == world and bool((per_rank[1]["reward"][:, 0] == torch.arange(E, 2 * E).double() + 0.25).all())
        else:
            assert per_rank is None
        # K steps per collective (bench.py --gather-every): slot s of a depth-K buffer carries step s
        deep = PackedStepOutputs(cfg, "cpu", world, rank, depth=3)
        for slot in range(3):
            deep.views(slot=slot)["reward"][:, 0] = env_ids.double() + 100.0 * slot
        assert len(set(deep.pointers(s)[1] for s in range(3))) == 3 and deep.pointers(1)[1] - deep.pointers(0)[1] == deep.nbytes
        deep.gather_async().wait()
        if rank == 0:
            for slot in range(3):
                got_r = deep.rank0_views(slot)[1]["reward"][:, 0]
                assert bool((got_r == torch.arange(E, 2 * E).double() + 100.0 * slot).all())
        acts_global = (torch.arange(E * world, dtype=torch.int32).view(-1, 1) % 5) if rank == 0 else None
        mine = scatter_actions(acts_global, world, rank, E, "cpu")
        ok = bool((mine[:, 0] == (env_ids % 5).int()).all())
        if rank == 0:
            ids = torch.arange(E * world)
            ok &= bool((got["reward"][:, 0] == ids.double() + 0.25).all())
            ok &= bool((got["info_speed"][:, 0] == ids.double() * 2).all())
            ok &= bool((got["obs"].reshape(E * world, -1)[:, -1] == ids.float()).all())
            ok &= bool((got["terminated"] == (ids % 2).to(torch.uint8)).all())
            ok &= bool((got["truncated"] == (ids % 3 == 0).to(torch.uint8)).all())
            ok &= bool((got["info_crashed"][:, 0] == (ids % 5 == 0).to(torch.uint8)).all())
            ok &= got["obs"].shape == (E * world, cfg.num_agents, *_abi.obs_shape(cfg))
        else:
            ok &= got is None
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_shard_range_partitions_everything():
    for total, world in [(4096, 8), (10, 3), (7, 8)]:
        ids = [i for r in range(world) for i in shard_range(total, world, r)]
        assert ids == list(range(total))


@pytest.mark.parametrize("kind", ["fast", "merge_ma4", "intersection_grid"])
def test_packed_gather_world2_gloo(kind):
    world, E = 2, 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, E, q, kind)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))  # (a cold `import torch` in a fresh container can take minutes)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]


def test_packed_layout_is_aligned():
    cfg = _abi.make_config(_abi.highway_fast_default_config(), 5, fast=True)
    out = PackedStepOutputs(cfg, "cpu")
    for name, (off, _) in out.offsets.items():
        assert off % 8 == 0, name
    assert len(out.pointers()) == 6
