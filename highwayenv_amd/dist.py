"""Multi-GPU: environments are independent, so they shard across ranks with NO data-path
collective while stepping.  The only exchange is one gather per batched
step of each rank's packed (reward | info_speed | obs | terminated | truncated | info_crashed)
block to rank 0 -- ``torch.distributed.gather`` on the ``nccl`` backend == RCCL over xGMI; every
peer -> rank-0 transfer rides its own direct xGMI link, so the step is latency-bound, which is
why everything is packed into ONE buffer (one collective launch per step, not six).

One process per GPU (torch.distributed.run).  The same code runs on ``gloo``/CPU tensors for
the world_size-2 tests.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from . import _abi


def shard_range(total_envs: int, world: int, rank: int) -> range:
    """Contiguous block partition of env ids; remainders go to the lowest ranks."""
    q, r = divmod(total_envs, world)
    start = rank * q + min(rank, r)
    return range(start, start + q + (1 if rank < r else 0))


class PackedStepOutputs:
    """Per-rank output block of ``hwy_step_device`` laid out in one contiguous byte buffer.

    Layout (8-byte aligned f64 first): reward f64[E,A] | info_speed f64[E,A] | obs f32[E,A,V,F]
    | terminated u8[E] | truncated u8[E] | info_crashed u8[E,A].
    """

    def __init__(self, cfg: _abi.HwyConfig, device, world: int = 1, rank: int = 0, force_collective: bool = False,
                 depth: int = 1):
        """``depth`` > 1: the buffer holds the blocks of ``depth`` consecutive steps (slot = step % depth) and ONE gather
        moves them all -- the collective's fixed cost (launch, stream hand-over; ~15 us measured even with a single
        rank) is then paid once per ``depth`` steps instead of once per 53 us step."""
        E, A = cfg.num_envs, cfg.num_agents
        self.depth = int(depth)
        self.obs_shape = _abi.obs_shape(cfg)
        self.E, self.A = E, A
        obs_len = int(torch.tensor(self.obs_shape).prod())
        self.world, self.rank = world, rank
        self.collective = world > 1 or force_collective  # force: exercise the collective with a single rank
        sizes = [("reward", E * A * 8), ("info_speed", E * A * 8), ("obs", E * A * obs_len * 4),
                 ("terminated", E), ("truncated", E), ("info_crashed", E * A)]
        self.offsets, off = {}, 0
        for name, nbytes in sizes:
            self.offsets[name] = (off, nbytes)
            off += (nbytes + 7) & ~7
        self.nbytes = off
        self.buf = torch.zeros(self.nbytes * self.depth, dtype=torch.uint8, device=device)
        self.gathered = ([torch.zeros(self.nbytes * self.depth, dtype=torch.uint8, device=device) for _ in range(world)]
                         if (self.collective and rank == 0) else None)

    def _view(self, buf, name, dtype, shape):
        off, nbytes = self.offsets[name]
        return buf[off:off + nbytes].view(dtype).view(*shape)

    def views(self, buf=None, slot: int = 0) -> dict:
        b = self.buf if buf is None else buf
        b = b[slot * self.nbytes:(slot + 1) * self.nbytes]
        E, A = self.E, self.A
        return {
            "reward": self._view(b, "reward", torch.float64, (E, A)),
            "info_speed": self._view(b, "info_speed", torch.float64, (E, A)),
            "obs": self._view(b, "obs", torch.float32, (E, A, *self.obs_shape)),
            "terminated": self._view(b, "terminated", torch.uint8, (E,)),
            "truncated": self._view(b, "truncated", torch.uint8, (E,)),
            "info_crashed": self._view(b, "info_crashed", torch.uint8, (E, A)),
        }

    def pointers(self, slot: int = 0):
        """(d_obs, d_reward, d_terminated, d_truncated, d_info_speed, d_info_crashed) for Engine.step_device."""
        base = self.buf.data_ptr() + slot * self.nbytes
        o = self.offsets
        return (base + o["obs"][0], base + o["reward"][0], base + o["terminated"][0], base + o["truncated"][0],
                base + o["info_speed"][0], base + o["info_crashed"][0])

    def terminated(self, slot: int = 0):
        return self.views(slot=slot)["terminated"]

    def gather_async(self):
        """Start the gather of this block to rank 0 and return the ``Work`` handle (None for world 1).
        ORDERING: torch.distributed orders the collective after torch's CURRENT stream, so the engine that fills this block
        must run on that stream -- create it with ``stream=torch.cuda.Stream().cuda_stream`` made current
        (``torch.cuda.set_stream``), as bench.py does; the default stream's handle is NULL, which ``hwy_create`` reads as
        "engine-owned stream".
        With two alternating ``PackedStepOutputs`` the collective of step t overlaps the kernel of step t+1;
        call ``work.wait()`` before the engine writes into this block again and before reading
        ``rank0_views()``."""
        if not self.collective:
            return None
        return dist.gather(self.buf, self.gathered if self.rank == 0 else None, dst=0, async_op=True)

    def rank0_views(self, slot: int = 0):
        """Per-rank view dicts of the last completed gather (rank 0), zero-copy."""
        if not self.collective:
            return [self.views(slot=slot)]
        return [self.views(b, slot) for b in self.gathered] if self.rank == 0 else None

    def gather_to_rank0(self, assemble: bool = True):
        """One collective per batched step.  On rank 0 returns the dict of global arrays (env-major
        concatenation over ranks) -- or, with ``assemble=False``, the list of per-rank view dicts (zero-copy:
        rank r's envs are ``shard_range(total, world, r)``); None on the other ranks."""
        if not self.collective:
            return self.views() if assemble else [self.views()]
        dist.gather(self.buf, self.gathered if self.rank == 0 else None, dst=0)
        if self.rank != 0:
            return None
        per_rank = [self.views(b) for b in self.gathered]
        if not assemble:
            return per_rank
        return {k: torch.cat([v[k] for v in per_rank], dim=0) for k in per_rank[0]}


def scatter_actions(actions_global, world: int, rank: int, envs_per_rank: int, device, total_envs: int | None = None):
    """Rank 0 holds int32 [total_envs, A] actions; rank r receives the rows of ``shard_range(total_envs, world, r)``.

    ``dist.scatter`` needs equal-sized pieces, so every piece is padded to the largest shard (``ceil(total / world)``
    rows) and the receiver drops the padding: the split follows ``shard_range`` exactly, also when ``total_envs`` is
    not a multiple of ``world``.  ``envs_per_rank`` is this rank's own shard size."""
    if world == 1:
        return actions_global
    if total_envs is None:
        total_envs = envs_per_rank * world  # equal shards
    mine = shard_range(total_envs, world, rank)
    if len(mine) != envs_per_rank:
        raise ValueError(f"rank {rank} owns {len(mine)} of {total_envs} envs, not {envs_per_rank}")
    A = actions_global.shape[-1] if rank == 0 else None
    shape = torch.tensor([A or 0], device=device)
    dist.broadcast(shape, src=0)
    A = int(shape.item())
    rows = -(-total_envs // world)
    out = torch.empty((rows, A), dtype=torch.int32, device=device)
    chunks = None
    if rank == 0:
        if actions_global.shape[0] != total_envs:
            raise ValueError(f"actions_global has {actions_global.shape[0]} rows, expected {total_envs}")
        chunks = []
        for r in range(world):
            rr = shard_range(total_envs, world, r)
            piece = torch.zeros((rows, A), dtype=torch.int32, device=device)
            piece[:len(rr)] = actions_global[rr.start:rr.stop]
            chunks.append(piece)
    dist.scatter(out, chunks, src=0)
    return out[:envs_per_rank]
