#!/usr/bin/env python3
"""Developer tool (GPU): the two-half-batches actor loop of INTEGRATION.md section 3 on HighwayVectorEnv(output="torch"), timed
against one vector env of the whole batch (random device actions stand in for the policy)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from highwayenv_amd.vector import HighwayVectorEnv  # noqa: E402

cfg = {"vehicles_count": 50, "lanes_count": 4}
STEPS = 500


def loop(envs, lanes):
    obs = [h.reset(seed=s)[0] for s, h in enumerate(envs)]
    acts = [torch.randint(0, 5, (STEPS + 50, h.num_envs), device="cuda", dtype=torch.int32) for h in envs]
    torch.cuda.synchronize()
    for t in range(STEPS + 50):
        if t == 50:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        for k, h in enumerate(envs):
            with torch.cuda.stream(lanes[k]):
                obs[k], rew, term, trunc, info = h.step(acts[k][t])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / STEPS * 1e6


lanes = [torch.cuda.Stream() for _ in range(2)]
one = [HighwayVectorEnv("highway-fast-v0", num_envs=4096, output="torch", config=cfg, stream=lanes[0])]
print(f"one vector env of 4096: {loop(one, lanes[:1]):.1f} us per step")
one[0].close()
two = [HighwayVectorEnv("highway-fast-v0", num_envs=2048, output="torch", config=cfg, stream=lanes[k]) for k in range(2)]
print(f"two vector envs of 2048, alternating: {loop(two, lanes):.1f} us per step of the whole batch")
for h in two:
    h.close()
torch.cuda.synchronize()
