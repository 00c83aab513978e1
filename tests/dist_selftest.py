#!/usr/bin/env python3
"""Multi-GPU self-test of the env-sharded step + gather-to-rank-0 path, runnable the day a multi-GPU box exists:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 \
        tests/dist_selftest.py --comm torch        # torch.distributed gather (nccl == RCCL), the bench.py path
    ... tests/dist_selftest.py --comm abi          # hwy_comm_init / hwy_gather of the C-ABI (RCCL, no torch collective)
    python tests/dist_selftest.py --comm abi       # world size 1 (a single GPU): the same code path, gather to self

Every rank steps its own block of environments (different seeds), packs (reward | speed | obs | flags) into one buffer
per step (highwayenv_amd.dist.PackedStepOutputs, `--depth` steps per buffer) and the buffers travel to rank 0 in ONE
collective.  Rank 0 then checks, rank by rank, that what arrived is what that rank computed: every rank sends a sha256
of each of its local blocks over a gloo side channel.  Prints SELFTEST OK (rank 0) and exits 0.
"""
import argparse
import hashlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from highwayenv_amd import _abi  # noqa: E402
from highwayenv_amd.dist import PackedStepOutputs, shard_range  # noqa: E402
from highwayenv_amd.engine import Engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--comm", choices=["torch", "abi"], default="torch")
    ap.add_argument("--envs", type=int, default=1001, help="TOTAL environments (not a multiple of the world size on purpose)")
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--depth", type=int, default=4)
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29544")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    side = None
    if args.comm == "torch":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        side = dist.new_group(backend="gloo")
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)  # control plane only (ships the RCCL id, the hashes)
    mine = shard_range(args.envs, world, rank)
    E = len(mine)
    cfg_d = _abi.highway_fast_default_config()
    cfg_d.update({"vehicles_count": 20, "lanes_count": 3})
    cfg = _abi.make_config(cfg_d, E, fast=True)
    # one real stream for the engine, torch and the event torch.distributed records for RCCL (the default stream's handle is
    # NULL, which hwy_create reads as "create your own": an engine-owned stream would not be ordered with torch's)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    eng = Engine(cfg, device=local, stream=stream.cuda_stream)
    eng.reset(seeds=np.asarray(list(mine), np.uint64) + 17, ego_spacing=1.5, vehicles_density=1.0)
    eng.set_autoreset(True, base_seed=1234 + mine.start, ego_spacing=1.5, vehicles_density=1.0)
    # shards differ in size when world does not divide the total: every rank pads its block to the largest shard
    E_max = len(shard_range(args.envs, world, 0))
    pad_cfg = _abi.make_config(cfg_d, E_max, fast=True)
    out = PackedStepOutputs(pad_cfg, dev, world, rank, force_collective=(args.comm == "torch"), depth=args.depth)
    if args.comm == "abi":
        ids = [Engine.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        eng.comm_init(ids[0], rank, world)
        recv = torch.zeros(out.nbytes * args.depth * world, dtype=torch.uint8, device=dev) if rank == 0 else None
    g = torch.Generator(device=dev)
    g.manual_seed(99 + rank)
    hashes = []
    n_gathers = 0
    for t in range(args.steps):
        slot = t % args.depth
        acts = torch.randint(0, 5, (E, 1), generator=g, device=dev, dtype=torch.int32)
        eng.step_device(acts.data_ptr(), *out.pointers(slot))
        if slot == args.depth - 1 or t == args.steps - 1:
            torch.cuda.synchronize(dev)
            local_bytes = out.buf.cpu().numpy().tobytes()
            hashes.append(hashlib.sha256(local_bytes).hexdigest())
            if args.comm == "torch":
                out.gather_async().wait()
                torch.cuda.synchronize(dev)
                got = [b.cpu().numpy().tobytes() for b in out.gathered] if rank == 0 else None
            else:
                eng.gather(out.buf.data_ptr(), recv.data_ptr() if rank == 0 else 0, out.buf.numel(), root=0)
                eng.sync()
                n = out.buf.numel()
                got = [recv[r * n:(r + 1) * n].cpu().numpy().tobytes() for r in range(world)] if rank == 0 else None
            n_gathers += 1
            all_hashes = [None] * world
            dist.all_gather_object(all_hashes, hashes[-1], group=side)
            if rank == 0:
                for r in range(world):
                    assert hashlib.sha256(got[r]).hexdigest() == all_hashes[r], f"gather {n_gathers}: block of rank {r} differs"
                # and the views decode: rank r's first env reward / obs are finite, flags are 0/1
                for r in range(world):
                    v = out.views(torch.frombuffer(bytearray(got[r]), dtype=torch.uint8), slot=slot)
                    Er = len(shard_range(args.envs, world, r))
                    assert torch.isfinite(v["reward"][:Er]).all() and torch.isfinite(v["obs"][:Er]).all()
                    assert (v["terminated"][:Er] <= 1).all()
    if args.comm == "abi":
        eng.comm_destroy()
    eng.close()
    dist.barrier(group=side)
    if rank == 0:
        print(f"SELFTEST OK: comm={args.comm} world={world} (reported by the process group: {dist.get_world_size()}), "
              f"{args.envs} envs in shards {[len(shard_range(args.envs, world, r)) for r in range(world)]}, "
              f"{args.steps} steps, {n_gathers} gathers of {out.buf.numel()} bytes per rank", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
