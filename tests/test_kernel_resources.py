"""The register / LDS allocation of the step kernels is a property of the BUILD: DESIGN.md section 3 quotes an occupancy for each
of them (waves per SIMD by registers, workgroups per CU by LDS), and a compiler or source change that silently costs a resident
wavefront or brings spills back would only show up as a slower bench line a round later.  Checked here from the code object's own
metadata (highwayenv_amd.build.kernel_resources: no GPU needed), with bounds, not exact counts.

gfx950: 512 VGPRs per SIMD lane (allocation granule 8), 160 KB of LDS per CU, 4 SIMDs per CU."""
import pytest

from highwayenv_amd import build

LDS_PER_CU = 160 * 1024


@pytest.fixture(scope="module")
def res():
    pytest.importorskip("msgpack")  # (the metadata note is msgpack; present in the build image and on the GPU box)
    if build.is_stale():
        build.build_engine()
    return build.kernel_resources()


def waves_per_simd(vgpr: int) -> int:
    return min(8, 512 // (((vgpr + 7) // 8) * 8))


def test_every_step_family_is_in_the_code_object(res):
    names = " ".join(res)
    for fam in ("hwy_step_wave_kernel", "hwy_rollout_wave_kernel", "hwy_step_kernel", "hwy_rollout_kernel", "hwy_reset_kernel",
                "hwy_observe_kernel", "hwy_net_step_kernel", "hwy_net_rollout_kernel", "hwy_net_reset_kernel",
                "hwy_net_observe_kernel", "hwy_ix_step_kernel", "hwy_ix_rollout_kernel", "hwy_ix_reset_kernel",
                "hwy_ix_observe_kernel"):
        assert f"hwy::{fam}<" in names, fam
    for k, r in res.items():
        assert r["sgpr"] <= 106, (k, r)  # (102 + VCC / flat scratch: nothing asks for more than the hardware has)


@pytest.mark.parametrize("full_scan", ["false", "true"])
def test_one_wavefront_kernel_four_waves_per_simd_no_vgpr_spills(res, full_scan):
    """hwy_step_wave_kernel<WPE, FULL_SCAN>: the headline launch (4096 envs = 4 wavefronts on each of the 1024 SIMDs) needs 4 resident
    wavefronts per SIMD by registers and 16 one-wavefront workgroups per CU by LDS; no VGPR is spilled."""
    for wpe in (1, 2, 3, 4):
        for fam in ("hwy_step_wave_kernel", "hwy_rollout_wave_kernel"):
            r = res[f"hwy::{fam}<{wpe}, {full_scan}>"]
            # (SGPRs spill to VGPR lanes, outside the frame loop: 73 before round 6's compacted MOBIL tasks, 83 .. 85 with them)
            assert r["vgpr_spill"] == 0 and r["scratch"] <= 36 and r["sgpr_spill"] <= 88, r
            # round 6: the SAT evaluates its four axis directions one after the other (hwy_device.h: HWY_SAT_FENCE) -- 99 / 117
            # registers where the interleaved form held 111 / 129 (and one spilled VGPR in the <4, true> build)
            assert waves_per_simd(r["vgpr"]) >= 4 and r["vgpr"] <= 120, r
            assert 16 * r["lds"] <= LDS_PER_CU, r
            assert r["workgroup"] == 64


def test_merge_kernel_four_waves_per_simd(res):
    """hwy_net_step_kernel<4, false> (BASELINE config 5): 128 VGPRs at 4 waves/SIMD -- the round-2 build spilled 54 there."""
    r = res["hwy::hwy_net_step_kernel<4, false>"]
    assert waves_per_simd(r["vgpr"]) >= 4 and r["vgpr_spill"] == 0, r  # (3 spilled before the sequential SAT of round 6)
    assert 16 * r["lds"] <= LDS_PER_CU, r
    r = res["hwy::hwy_net_rollout_kernel<4, false>"]
    assert waves_per_simd(r["vgpr"]) >= 4 and r["vgpr_spill"] <= 4, r


def test_intersection_kernel_allocation(res):
    """hwy_ix_step_kernel<2, 32, 64> (BASELINE config 4: 32 slots + 32 helper lanes): no spills, registers for three wavefronts
    per SIMD, LDS for eight workgroups per CU (the two per SIMD the launch is tuned for, profiles/r03_history.md)."""
    r = res["hwy::hwy_ix_step_kernel<2, 32, 64>"]
    # (scratch: 36 B is the frame the compiler reserves next to SGPRs spilled to VGPR LANES -- the kernel holds no scratch_* /
    #  buffer_* instruction; a VGPR spill would add to it)
    assert r["vgpr_spill"] == 0 and r["scratch"] <= 36, r
    assert waves_per_simd(r["vgpr"]) >= 3, r
    assert 8 * r["lds"] <= LDS_PER_CU, r
    r = res["hwy::hwy_ix_rollout_kernel<2, 32, 64>"]
    assert r["vgpr_spill"] == 0 and waves_per_simd(r["vgpr"]) >= 2 and 8 * r["lds"] <= LDS_PER_CU, r


def test_workgroup_kernel_allocation(res):
    """hwy_step_kernel<W, WPE> (N > 64: W wavefronts per environment).  The 3-wave builds hold no spills; the 4-wave builds
    (batches beyond 3 resident wavefronts per SIMD) carried 16 .. 34 spilled VGPRs through round 5 -- the SAT's interleaved axis
    directions on top of the frame loop's state, and the sparse-checker loop unrolled over the workgroup's wavefronts with one
    inlined SAT each -- and hold at most 2 since round 6 (none before the abort rule moved to the per-thread rank-space chain:
    the post-allocation liveness of the frame loop peaks at 125 registers, inside the SAT, and the allocator needs 130 for it; the
    chain is worth 12.5 us of 215 at 1024 x 201 with those two in scratch, profiles/r06_history.md section 7)."""
    for w in (1, 2, 3, 4):
        r3, r4 = res[f"hwy::hwy_step_kernel<{w}, 3>"], res[f"hwy::hwy_step_kernel<{w}, 4>"]
        assert r3["vgpr_spill"] == 0 and waves_per_simd(r3["vgpr"]) >= 3, r3
        assert waves_per_simd(r4["vgpr"]) >= 4 and r4["vgpr_spill"] <= 2, r4
        assert r3["workgroup"] == 64 * w and r3["lds"] == r4["lds"]
        # LDS never limits below what the registers allow: (waves/SIMD x 4 SIMDs) / W workgroups per CU
        assert (12 // w) * r3["lds"] <= LDS_PER_CU and (16 // w) * r4["lds"] <= LDS_PER_CU, (r3, r4)


def test_wide_kernel_allocation(res):
    """hwy_step_wide_kernel<2, 2> (64 < N <= 128: BASELINE config 3, one wavefront per environment with two vehicles per thread):
    no spills, registers for two wavefronts per SIMD (1024 environments hold one, 2048 two) and LDS for the eight one-wavefront
    workgroups per CU that go with them."""
    for fam in ("hwy_step_wide_kernel", "hwy_rollout_wide_kernel"):
        r = res[f"hwy::{fam}<2, 2>"]
        assert r["vgpr_spill"] == 0 and r["scratch"] <= 36, r
        assert waves_per_simd(r["vgpr"]) >= 2, r
        assert 8 * r["lds"] <= LDS_PER_CU, r
        assert r["workgroup"] == 64
        # three / four vehicles per thread (128 < N <= 192 / 256, round 5): the N > 128 path holds NO spilled VGPR either (the
        # workgroup kernel it replaced there carried 16-32 in its 4-wave builds), one wavefront per SIMD, LDS for four per CU
        for k in (3, 4):
            r = res[f"hwy::{fam}<{k}, 1>"]
            assert r["vgpr_spill"] == 0 and r["vgpr"] <= 512 and r["workgroup"] == 64, r
            assert 4 * r["lds"] <= LDS_PER_CU, r
