#!/usr/bin/env python3
"""Developer tool (GPU): the headline batch as S independent sub-batches, one Engine and one HIP stream each, stepped
round-robin with pre-staged device actions -- the tail of one sub-batch's launch overlaps the body of the next one's.
    python tools/split_batch_probe.py [total_envs] [S ...]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from highwayenv_amd import _abi  # noqa: E402
from highwayenv_amd.dist import PackedStepOutputs  # noqa: E402
from highwayenv_amd.engine import Engine  # noqa: E402

TOTAL = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
SPLITS = [int(a) for a in sys.argv[2:]] or [1, 2, 4]
dev = torch.device("cuda", 0)
cfg_d = _abi.highway_fast_default_config()
cfg_d.update({"vehicles_count": 50, "lanes_count": 4})
STEPS, WARM = 600, 50
for S in SPLITS:
    E = TOTAL // S
    cfg = _abi.make_config(cfg_d, E, fast=True)
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    engs, outs, acts = [], [], []
    for s in range(S):
        eng = Engine(cfg, device=0, stream=streams[s].cuda_stream)
        eng.reset(base_seed=1_000_003 * (s + 1), ego_spacing=1.5, vehicles_density=1.0)
        eng.set_autoreset(True, base_seed=77_000_001 * (s + 1), ego_spacing=1.5, vehicles_density=1.0)
        engs.append(eng)
        outs.append(PackedStepOutputs(cfg, dev, 1, 0, force_collective=False, depth=1))
        g = torch.Generator(device=dev)
        g.manual_seed(1234 + s)
        acts.append(torch.randint(0, 5, (STEPS + WARM, E, 1), generator=g, device=dev, dtype=torch.int32))
    torch.cuda.synchronize(dev)
    res = []
    for rep in range(4):
        t0 = None
        for t in range(STEPS + WARM):
            if t == WARM:
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
            for s in range(S):
                engs[s].step_device(acts[s][t].data_ptr(), *outs[s].pointers(0))
        torch.cuda.synchronize(dev)
        res.append((time.perf_counter() - t0) / STEPS * 1e6)
    print(f"{TOTAL} envs as {S} x {E}: {np.median(res):.2f} us per step of the whole batch ({TOTAL / np.median(res):.1f} M env-steps/s)  {np.round(res, 2)}")
    for e_ in engs:
        e_.close()
