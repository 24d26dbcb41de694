"""Register budgets of the hot kernels, checked at build time (hipcc cross-compiles without a GPU): the persistent
convolver transforms run ONE 512-thread workgroup per CU, i.e. two wavefronts per SIMD and 256 registers per thread; a
spill there is not a slowdown of a few percent but of 1.5x (measured: 51 spilled registers in the filter-stage kernel,
3.05 -> 4.8 ms on T1), and it appears or disappears with innocent-looking source changes.  This test compiles the files
to ISA and reads the kernel descriptors."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "web-audio-api-rs_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fgpu-flush-denormals-to-zero",
         "--cuda-device-only", "-S"]


def kernel_resources(src, tmp_path):
    out = str(tmp_path / (src + ".s"))
    subprocess.check_call([HIPCC] + FLAGS + [os.path.join(CSRC, src), "-o", out], stderr=subprocess.DEVNULL)
    text = open(out).read()
    res = {}
    for m in re.finditer(r"\.name:\s+(\S+)(.*?)\.wavefront_size", text, re.S):
        body = m.group(2)
        get = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, body).group(1))  # noqa: E731
        res[m.group(1)] = {"vgpr": get("vgpr_count"), "spill": get("vgpr_spill_count"), "sgpr_spill": get("sgpr_spill_count"),
                           "scratch": get("private_segment_fixed_size")}
    return res


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_convolver_transform_kernels_do_not_spill(tmp_path):
    res = kernel_resources("waa_conv3.hip", tmp_path)
    hot = {k: v for k, v in res.items() if "conv_fft3" in k}
    assert len(hot) == 4, sorted(res)  # forward, forward + filter stage, IR spectra, inverse
    for name, r in hot.items():
        # (a couple of spilled SCALAR registers — the filter-stage kernel has two — cost a few scalar moves per block)
        # ... and no scratch at all: a local array indexed by a rolled loop silently moves to scratch memory (the filter stage's
        # redo path did: +3.96 GB of stores per T1 render, measured with WRITE_SIZE)
        assert r["spill"] == 0 and r["sgpr_spill"] <= 4 and r["scratch"] == 0, (name, r)
        assert r["vgpr"] <= 256, (name, r)  # two wavefronts per SIMD


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_convolver_product_kernel_keeps_three_waves_per_simd(tmp_path):
    res = kernel_resources("waa_conv.hip", tmp_path)
    k = [v for n, v in res.items() if "conv_mac_win_kernelILi16ELi22ELb1" in n]
    assert len(k) == 1 and k[0]["spill"] == 0 and k[0]["vgpr"] <= 168, k  # (176+ registers = two waves per SIMD)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_lane_per_stream_biquad_passes_keep_four_waves_per_simd(tmp_path):
    """waa_biquad_lanes.hip: both passes of the fast instantiation fit 128 registers (four wavefronts per SIMD; the exact-order
    pass without a spill — the zero-state pass spills outside its chunk loop only)"""
    res = kernel_resources("waa_biquad_lanes.hip", tmp_path)
    fast = {n: v for n, v in res.items() if "biquad_lanes_kernelILi" in n and "ELb1E" in n}
    assert len(fast) == 2, sorted(res)
    for name, r in fast.items():
        assert r["vgpr"] <= 128, (name, r)
        if "ILi1E" in name:
            assert r["spill"] == 0 and r["scratch"] == 0, (name, r)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_element_wise_chain_kernels_keep_their_occupancy(tmp_path):
    """waa_kernels.hip: the tile-parallel chain kernel is latency-bound — waves per SIMD are its lever.  The stereo
    single-input form stays at 64 registers (eight waves), the stereo summing form at 80 (six); round 3's persistent-loop
    bookkeeping cost the former 17 registers until it got its own instantiation (C4's pan stage 0.84 -> 1.07 ms)."""
    res = kernel_resources("waa_kernels.hip", tmp_path)
    single = [v for n, v in res.items() if "chain_kernelILi2ELi4ELb0ELb0ELb0ELb0E" in n]
    summing = [v for n, v in res.items() if "chain_kernelILi2ELi4ELb0ELb1ELb0ELb0E" in n]
    nospill = [v for n, v in res.items() if "chain_kernelILi2ELi4ELb0ELb1ELb0ELb1E" in n]
    assert len(single) == 1 and single[0]["vgpr"] <= 64 and single[0]["spill"] == 0, single
    # (the summing form is held at six waves by its occupancy attribute and pays four spilled registers for it, since round 2)
    assert len(summing) == 1 and summing[0]["vgpr"] <= 80 and summing[0]["spill"] <= 4, summing
    # (the five-wave form that chains with per-frame panning take: no scratch memory at all)
    assert len(nospill) == 1 and nospill[0]["vgpr"] <= 96 and nospill[0]["spill"] == 0 and nospill[0]["scratch"] == 0, nospill


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_echo_ring_kernel_fits_one_workgroup_of_sixteen_waves(tmp_path):
    """waa_echo.hip: the workgroup is 1024 threads (four wavefronts per SIMD) — more than 128 registers and it does not launch;
    and nothing of the walk may live in scratch memory (one operand array indexed by an edge's selector did: 47 scratch loads
    and a hundred waits per chunk)"""
    res = kernel_resources("waa_echo.hip", tmp_path)
    k = {n: v for n, v in res.items() if "echo_ring_kernel" in n and "Lb1EEEvN" not in n}   # (BQ = false)
    assert len(k) == 8, sorted(res)
    for name, r in k.items():
        assert r["vgpr"] <= 128 and r["spill"] == 0 and r["scratch"] == 0, (name, r)
    # the BQ form (a Biquad between the ring and the sum, round 4): eight waves, 256 registers each, nothing in scratch
    bq = {n: v for n, v in res.items() if "echo_ring_kernel" in n and "Lb1EEEvN" in n}
    assert len(bq) == 10, sorted(res)
    for name, r in bq.items():
        # (one instantiation reserves a 36-byte frame for its scalar-register spills without a single scratch instruction in the listing)
        assert r["vgpr"] <= 256 and r["spill"] == 0 and r["scratch"] <= 64, (name, r)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_oversampler_transform_kernels_keep_two_waves_per_simd(tmp_path):
    """waa_osfft.hip: a group's whole state (two stages' overlaps, two spectra, the transform in flight) stays within 256
    registers — two wavefronts per SIMD hide each other's LDS exchanges.  It was 304 (2x) / 426 (4x) with a divergent branch
    around the state reset, an early exit from the loop and branchy curve lookups (copies of the carried state), which is how
    the kernel came to be written the way it is.  The 4x instantiations sit exactly at the limit and may park one pointer in
    scratch (reloaded once per quantum); nothing else may spill."""
    res = kernel_resources("waa_osfft.hip", tmp_path)
    x2 = {n: v for n, v in res.items() if "osfft_kernelILi2E" in n}
    x4 = {n: v for n, v in res.items() if "osfft_kernelILi4E" in n}
    assert len(x2) == 4 and len(x4) == 4, sorted(res)
    for name, r in x2.items():
        assert r["vgpr"] <= 256 and r["spill"] == 0 and r["scratch"] == 0, (name, r)
    for name, r in x4.items():
        assert r["vgpr"] <= 256 and r["spill"] <= 2 and r["scratch"] <= 16, (name, r)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_hrtf_transform_kernel_holds_its_state_in_registers(tmp_path):
    """waa_hrtf_fft.hip: three spectra of history, the carry and two transforms in flight per lane = one wavefront per SIMD on the
    unified register file (arch + accumulation registers); nothing in scratch memory"""
    res = kernel_resources("waa_hrtf_fft.hip", tmp_path)
    k = [v for n, v in res.items() if "hrtf_fft_kernel" in n]  # (the plain form and the exact-zeros form)
    # (the exact-zeros form parks two values in accumulation registers — counted as spills, not scratch: nothing goes to memory)
    assert len(k) == 2 and all(r["scratch"] == 0 and r["spill"] <= 4 for r in k) and min(r["spill"] for r in k) == 0, k
