"""The register allocator's verdict on the 256x256 GEMM kernels, recorded by the build (lib/kernel_resources.json).

hipcc 7.2 flips between two allocations of vlm_gemm_big_kernel on edits that do not change a single value (the staging
offsets written directly instead of "swizzled, then un-swizzled"): in the bad one 40 accumulator registers live in VGPRs
and are shuffled through a[52:55] inside the K loop -- 1 832 instead of 1 488 cycles per 32-deep step, -11 % on the
forward / dgrad GEMMs, with every numerics test still green.  This test is the tripwire."""
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
RES = os.path.join(HERE, "..", "vl-merging_amd", "lib", "kernel_resources.json")


CSRC = os.path.join(HERE, "..", "vl-merging_amd", "csrc")


def _stale():
    """No record, or a kernel source / header newer than the record (an edit that was never rebuilt)."""
    if not os.path.exists(RES):
        return True
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    return max(os.path.getmtime(f) for f in srcs) > os.path.getmtime(RES)


@pytest.fixture(scope="module")
def resources():
    """lib/kernel_resources.json is a BUILD product (git-ignored): rebuilt here whenever it is missing or older than a
    source, so the tripwire can never pass on stale numbers."""
    if _stale():
        import importlib.util
        spec = importlib.util.spec_from_file_location("graft_entry", os.path.join(HERE, "..", "__graft_entry__.py"))
        ge = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ge)
        ge.build()
    assert os.path.exists(RES), "the build did not record kernel resources"
    with open(RES) as f:
        return json.load(f)


def test_big_gemm_accumulators_stay_in_agprs(resources):
    big = {k: v for k, v in resources.items() if "vlm_gemm_big_kernel" in k}
    assert len(big) == 11, sorted(big)  # six plain + the five GROUPED variants a block uses (round 5: + residual without aux)
    for name, r in big.items():
        assert r["AGPRs"] == 256, (name, r)            # the 4 x 64 accumulator registers of a 128x128 wave tile
        assert r["VGPRs Spill"] == 0 and r["ScratchSize"] == 0, (name, r)
        assert r["Occupancy"] == 1 and r["LDS Size"] <= 160 * 1024, (name, r)
        residual = "ILb1ELb1E" in name                 # <OUT_F32, RES, AUX>: the residual variants hold 36-register input sets
        assert r["VGPRs"] <= (248 if residual else 232), (name, r)


def test_wgrad_kernel_resources(resources):
    both = [(k, v) for k, v in resources.items() if "vlm_gemm_bigT_kernel" in k]
    assert len(both) == 2, sorted(k for k, _ in both)  # plain + GROUPED
    for name, r in both:
        assert r["AGPRs"] == 256 and r["Occupancy"] == 1, (name, r)
        assert r["VGPRs Spill"] <= 32, (name, r)       # the tail's epilogue setup spills a few; the K loop does not


def test_attention_kernels_keep_their_occupancy(resources):
    occ = {k: v["Occupancy"] for k, v in resources.items() if "attn_" in k and "kernel" in k}
    assert occ, "no attention kernels recorded"
    for name, o in occ.items():
        if "attn_bwd_dkvb_kernelILi4E" in name:
            continue  # the four-wave form of the fused dK/dV + bias-gradient kernel (480^2 panels): one wave per SIMD by design
        if "attn_fwd_kernel" in name or "attn_bwd_dq" in name or "attn_bwd_dkv" in name or "attn_bwd_dbias" in name:
            assert o >= 2, (name, o)


def test_hand_placed_forward_fits_two_waves_per_simd(resources):
    """attn_fwd2_kernel's stream names its registers (v32..v191, a0..a63): 192 + 64 = the 256 a wave may have at two waves
    per SIMD.  One register more and the second workgroup of a CU -- the one that covers a workgroup's prologue -- is gone."""
    r = [v for k, v in resources.items() if "attn_fwd2_kernel" in k]
    assert len(r) == 1
    r = r[0]
    assert r["Occupancy"] == 2 and r["VGPRs"] <= 192 and r["AGPRs"] == 64, r
    assert r["VGPRs Spill"] == 0 and r["ScratchSize"] == 0 and 2 * r["LDS Size"] <= 160 * 1024, r


def test_generated_stream_is_current():
    """csrc/attention_fwd2_body.inc is what csrc/gen/attn_fwd2_gen.py writes."""
    import subprocess
    import sys
    gen = os.path.join(CSRC, "gen", "attn_fwd2_gen.py")
    assert subprocess.call([sys.executable, gen, "--check"]) == 0, "run python vl-merging_amd/csrc/gen/attn_fwd2_gen.py"


def test_hand_placed_dq_fits_two_waves_per_simd(resources):
    """attn_bwd_dq2_kernel: v32..v201 + 32 accumulator registers, two workgroups of 34 KiB per CU."""
    r = [v for k, v in resources.items() if "attn_bwd_dq2_kernel" in k]
    assert len(r) == 1
    r = r[0]
    assert r["Occupancy"] == 2 and r["VGPRs"] + r["AGPRs"] <= 256 and r["AGPRs"] == 32, r
    assert r["VGPRs Spill"] == 0 and r["ScratchSize"] == 0 and 2 * r["LDS Size"] <= 160 * 1024, r


def test_generated_dq_stream_is_current():
    import subprocess
    import sys
    gen = os.path.join(CSRC, "gen", "attn_dq2_gen.py")
    assert subprocess.call([sys.executable, gen, "--check"]) == 0, "run python vl-merging_amd/csrc/gen/attn_dq2_gen.py"
