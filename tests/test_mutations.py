"""Mutation check of the parity rules (round-3 verdict): the randomised engine-vs-oracle comparisons tolerate -- and count -- a
few knife-edge cases of the reference itself (a push direction, a touching pair, a lane index or a queue order decided by the
last bit).  Could those rules hide a real defect?  Four bugs are seeded into the kernel SOURCE, the CPU emulator is built from
each mutated copy (tests/emu: the same headers the GPU build compiles), and the comparison that covers the mutated path must
FAIL on it -- while passing on the unmutated build:

* `ix_first_pair_wins`   -- the intersection kernel keeps the impact of the LOWEST partner slot (the reference's loop
  overwrites: the last pair wins, objects.py:104-112): only visible in pile-ups, where the exemption rules live;
* `net_pair_list_carry`  -- the road-network kernel's pair list carries the overflow of a pass (> 64 close pairs) to the next
  pass shifted by one entry: only visible with more close pairs than one pass holds;
* `mobil_sides_swapped`  -- the one-wavefront kernel weighs the LEFT lane change with the right lane's gap and vice versa
  (behavior.py:265-324);
* `reach_ignores_motion` -- the forward collision walk of the three highway kernels stops at the pre-check radius of bodies at rest,
  forgetting what the frame moved them (road.py:477-481 tests ALL pairs): only visible with bodies that close a gap within one frame.

Each case runs the real test functions in a subprocess with HWY_EMU_LIB pointing at the mutant."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "emu", "_build")

MUTANTS = {
    "ix_first_pair_wins": [
        ("hwy_ix.h", "__hip_atomic_fetch_max(&sh.jmax[a], b, __ATOMIC_RELAXED", "__hip_atomic_fetch_min(&sh.jmax[a], b, __ATOMIC_RELAXED"),
        ("hwy_ix.h", "__hip_atomic_fetch_max(&sh.jmax[b], a, __ATOMIC_RELAXED", "__hip_atomic_fetch_min(&sh.jmax[b], a, __ATOMIC_RELAXED"),
        ("hwy_ix.h", "if (i < SH::kCap) { sh.jmax[i] = -1; sh.flag[i] = 0; }", "if (i < SH::kCap) { sh.jmax[i] = 0x7fffffff; sh.flag[i] = 0; }"),
        ("hwy_ix.h", "const int j_imp = i < SH::kCap ? sh.jmax[i] : -1;",
         "const int j_imp = (i < SH::kCap && sh.jmax[i] != 0x7fffffff) ? sh.jmax[i] : -1;")],
    "net_pair_list_carry": [
        ("hwy_net.h", "const int carry = i < left ? (int)plist[count + i] : 0,", "const int carry = i < left ? (int)plist[count + i + 1] : 0,")],
    "reach_ignores_motion": [   # (round 6: the forward collision walk's reach from the frame's maxima, hwy_device.h reach_from_keys)
        ("hwy_device.h", "return ((5.5 + S * dt) + 2.0 * D) + 1e-6;", "return ((5.5 + 0.0 * S * dt) + 0.0 * D) + 1e-6;")],
    "mobil_sides_swapped": [   # (the compacted MOBIL tasks of round 6: side 0 = left reads row lane, side 1 = right row lane + 2)
        ("hwy_wave.h", "const u64 m = sh.lane_mask[ln + (side ? 2 : 0)];", "const u64 m = sh.lane_mask[ln + (side ? 0 : 2)];")],
}
# (mutant, pytest selection, extra environment, must the selection pass?)  The selections are the suite's own tests: the fuzz families
# at their default chunk numbers (what `pytest -m gpu` runs on the GPU box, here on the emulator) and the directed pile-up test.
FUZZ = {"HWY_FUZZ_BACKEND": "emu", "HWY_FUZZ_FIRST": "0", "HWY_FUZZ_CHUNKS": "1"}
CASES = [
    # (chunk 0 holds no frame in which one vehicle is hit by two others; chunk 1 does -- one chunk is 30 s on the emulator)
    ("ix_first_pair_wins", ["tests/test_fuzz_configs.py", "-m", "gpu", "-k", "intersection"], dict(FUZZ, HWY_FUZZ_FIRST="1")),
    ("net_pair_list_carry", ["tests/test_pileup.py", "-m", "not gpu", "-k", "merge"], {}),
    ("mobil_sides_swapped", ["tests/test_fuzz_configs.py", "-m", "gpu", "-k", "test_random_configurations_vs_oracle"], FUZZ),
    ("reach_ignores_motion", ["tests/test_collision_steps.py", "-m", "not gpu", "-k", "fast_bodies_are_not_missed"], {}),
]


def build_mutant(name: str) -> str:
    from tests.emu import emu
    src = os.path.join(BUILD, f"mut_{name}_{os.getpid()}")
    shutil.rmtree(src, ignore_errors=True)
    for d in ("tests/emu", "highwayenv_amd/csrc", "include"):
        os.makedirs(os.path.join(src, d))
        for f in os.listdir(os.path.join(ROOT, d)):
            if f.endswith((".h", ".cpp")):
                shutil.copy(os.path.join(ROOT, d, f), os.path.join(src, d, f))
    for fname, old, new in MUTANTS[name]:
        path = os.path.join(src, "highwayenv_amd", "csrc", fname)
        text = open(path).read()
        assert text.count(old) == 1, f"mutation site of {name} not found exactly once in {fname}: {old}"
        open(path, "w").write(text.replace(old, new))
    lib = os.path.join(BUILD, f"libhwy_emu_mut_{name}.so")
    emu.compile_emulator(os.path.join(src, "tests", "emu", "emu_engine.cpp"), lib)
    shutil.rmtree(src)
    return lib


def run_selection(lib: str, selection, env_extra) -> subprocess.CompletedProcess:
    env = dict(os.environ, **env_extra)
    env.pop("HWY_EMU_LIB", None)
    if lib:
        env["HWY_EMU_LIB"] = lib
    return subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", *selection], cwd=ROOT, env=env,
                          capture_output=True, text=True, timeout=1500)


@pytest.mark.parametrize("mutant,selection,env_extra", CASES, ids=[c[0] for c in CASES])
def test_seeded_bug_fails_the_comparison_that_covers_it(mutant, selection, env_extra):
    from concurrent.futures import ThreadPoolExecutor
    from tests.emu import emu
    emu.build()   # (the suite's own emulator build, before two processes could both start building it)
    with ThreadPoolExecutor(2) as pool:   # the control and the mutant side by side (two subprocesses)
        f_good = pool.submit(run_selection, None, selection, env_extra)
        f_bad = pool.submit(lambda: run_selection(build_mutant(mutant), selection, env_extra))
        good, bad = f_good.result(), f_bad.result()
    assert good.returncode == 0, f"the selection must pass on the unmutated kernel source:\n{good.stdout[-3000:]}"
    assert bad.returncode == 1 and "AssertionError" in bad.stdout, \
        f"mutant {mutant} SURVIVED {selection} (rc {bad.returncode}):\n{bad.stdout[-3000:]}"
