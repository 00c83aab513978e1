import sys, numpy as np
sys.path.insert(0, '/root/repo')
import bench
from highwayenv_amd import _abi
from highwayenv_amd.engine import Engine
E = 2048
cfg_d, fast, scenario = bench.workload_config("intersection")
tuning = {k: int(v) for k, v in (a.split("=") for a in sys.argv[1:])}
eng = Engine(_abi.make_config(cfg_d, E, fast=fast, scenario=scenario, tuning=tuning))
eng.reset(base_seed=5); eng.set_autoreset(True, base_seed=99)
rng = np.random.default_rng(0)
rows = []
for t in range(100):
    obs = eng.step(rng.integers(0, 3, size=(E, 1)))[0].reshape(E, -1)
    if t >= 40:
        rows.append(obs[:, :16].astype(np.float64))
X = np.concatenate(rows)
R = X[X[:, 14] == 1]; S = X[X[:, 14] == 0]
tot = lambda A: A[:, :14].sum(1)
print("tuning", tuning, "respawn waves: %.1f %% of launches' step blocks" % (100 * len(R) / len(X)))
fr = R[:, 15]
print("inline frames histogram:", {int(k): int((fr == k).sum()) for k in np.unique(fr)})
print("respawn wave ticks: mean %.0f p90 %.0f p99 %.0f max %.0f | step wave ticks: mean %.0f p99 %.0f max %.0f" % (
    tot(R).mean(), *np.percentile(tot(R), [90, 99]), tot(R).max(), tot(S).mean(), np.percentile(tot(S), 99), tot(S).max()))
per_launch_max = [max(tot(x[x[:, 14] == 1]).max(initial=0), tot(x[x[:, 14] == 0]).max()) for x in rows]
print("slowest wave per launch: mean %.0f (respawn is the slowest in %d of %d launches)" % (
    np.mean(per_launch_max), sum(tot(x[x[:, 14] == 1]).max(initial=0) > tot(x[x[:, 14] == 0]).max() for x in rows), len(rows)))
