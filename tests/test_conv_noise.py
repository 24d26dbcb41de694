"""WHERE the reference's ConvolverNode leaves FFT roundoff noise instead of exact zeros (csrc/waa_conv_noise.hpp): the
automaton the device's code kernel runs, compiled with g++ and held against the oracle's restated FFTConvolver
(oracle/waa_oracle.c::conv_process, fft-convolver 0.3 as convolver.rs:284-306 / :384-466 calls it) on the CPU — for every quantum,
"the automaton says noise" == "the convolver's output holds a non-zero sample".  The GPU side of it (the silence decisions of
DelayNodes behind a convolver) is tests/test_conv_noise_gpu.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FP = C.POINTER(C.c_float)
RQ = 128


@pytest.fixture(scope="module")
def cn(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("conv_noise") / "conv_noise_check.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "conv_noise_check.cpp")])
    lib = C.CDLL(so)
    lib.cn_describe.argtypes = [FP, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
    lib.cn_predict.argtypes = [C.c_uint32, C.c_uint64, FP, C.c_uint32, C.POINTER(C.c_uint8)]
    lib.cn_floor.restype = C.c_float
    return lib


@pytest.fixture(scope="module")
def ref(orc_lib):  # (tests/conftest.py: the raw ctypes handle of oracle/liboracle.so)
    orc_lib.orc_fftconvolver_run.argtypes = [FP, C.c_uint64, FP, C.c_uint64, FP]
    return orc_lib


def reference_output(orc, h, x):
    y = np.zeros_like(x)
    orc.orc_fftconvolver_run(h.ctypes.data_as(FP), len(h), x.ctypes.data_as(FP), len(x), y.ctypes.data_as(FP))
    return y


def predicted(cn, h, x):
    segc, mask = C.c_uint32(), C.c_uint64()
    cn.cn_describe(h.ctypes.data_as(FP), len(h), C.byref(segc), C.byref(mask))
    out = np.zeros(len(x) // RQ, np.uint8)
    cn.cn_predict(segc.value, mask.value, x.ctypes.data_as(FP), len(out), out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out.astype(bool), segc.value, mask.value


def check(cn, ref, h, x, what):
    h = np.ascontiguousarray(h, np.float32)
    x = np.ascontiguousarray(x, np.float32)
    y = reference_output(ref, h, x)
    got = np.any(y.reshape(-1, RQ) != 0, axis=1)
    want, segc, mask = predicted(cn, h, x)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, f"{what}: quanta {bad[:8]} (of {len(got)}; {segc} segment(s), mask {mask:#x}): reference {got[bad[:8]]}"
    return got, y


def bursts(rng, nq):
    x = np.zeros(nq * RQ, np.float32)
    for _ in range(int(rng.integers(1, 4))):
        a = int(rng.integers(0, nq * RQ))
        b = min(nq * RQ, a + int(rng.integers(1, 3000)))
        kind = int(rng.integers(0, 3))
        x[a:b] = rng.standard_normal(b - a) if kind == 0 else (0.5 if kind == 1 else np.sin(np.arange(b - a) * 0.05))
    return x


@pytest.mark.parametrize("seed", range(6))
def test_noise_quanta_of_random_bursts(cn, ref, seed):
    """bursts of noise / DC / a sine separated by exact zeros, through responses of 1 ... 5000 taps: some with an all-zero first
    segment (a pre-delay of a block: the block in progress then contributes nothing), a unit impulse, a zero second half"""
    rng = np.random.default_rng(9000 + seed)
    for trial in range(60):
        nq = int(rng.integers(40, 200))
        x = bursts(rng, nq)
        n_taps = int(rng.choice([1, 16, 100, 1024, 1025, 1500, 2048, 3000, 5000]))
        h = rng.standard_normal(n_taps).astype(np.float32) * 0.1
        if rng.random() < 0.3 and n_taps > 1100:
            h[:1024] = 0
        if rng.random() < 0.2:
            h = np.zeros(n_taps, np.float32)
            h[int(rng.integers(0, n_taps))] = 1.0
        if rng.random() < 0.2:
            h[n_taps // 2:] = 0
        check(cn, ref, h, x, f"seed {seed} trial {trial} ({n_taps} taps)")


def test_noise_outlives_the_input_by_up_to_two_blocks_and_precedes_an_onset(cn, ref):
    """the two shapes of DESIGN 5 2b: (i) a source that starts at frame 187 of quantum 1 — the reference's output of quantum 1 is
    noise from frame 128 on (fuzz seed 42294); (ii) after the input has gone to zero the noise lasts to the end of the NEXT block
    behind the response's last segment, not to the end of the response"""
    rng = np.random.default_rng(4)
    h = rng.standard_normal(16).astype(np.float32) * 0.2
    x = np.zeros(40 * RQ, np.float32)
    x[187:187 + 300] = rng.standard_normal(300)
    got, y = check(cn, ref, h, x, "onset")
    assert not got[0] and got[1] and np.count_nonzero(y[128:187]) > 40  # noise in front of the onset, none in the quantum before
    assert 0 < np.abs(y[128:187]).max() < 1e-6
    # input zero from frame 487 on (quantum 3): block 0 = quanta 0-7 carries it, its overlap block 1 = quanta 8-15, then exact zeros
    assert got[1:16].all() and not got[16:].any()
    assert 0 < np.abs(y[8 * RQ:16 * RQ]).max() < 1e-5  # (only the overlap's roundoff is left there: 1e-7 of the burst)


def test_more_than_64_segments_fall_back_to_the_age_form(cn, ref):
    rng = np.random.default_rng(5)
    h = rng.standard_normal(70 * 1024 + 17).astype(np.float32) * 0.01
    x = np.zeros(700 * RQ, np.float32)
    x[5000:5200] = rng.standard_normal(200)
    got, _ = check(cn, ref, h, x, "71 segments")
    assert got.any() and not got[-1]


def test_the_floor_is_far_from_both_ends(cn):
    f = cn.cn_floor()
    assert 1e-30 < f < 1e-12  # normal after a long chain of small gains, nothing a parity tolerance (>= 1e-7) could see
