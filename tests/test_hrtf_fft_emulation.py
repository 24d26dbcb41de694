"""Host replay of the transform form of the HRTF panner's FIR (waa_hrtf_fft.hip): the header the kernel is built from
(web-audio-api-rs_amd/csrc/waa_hrtf_fft.hpp + the table builder) compiled for the HOST, tools/hrtf_fft_emulate.cpp walking the
kernel's choreography — groups of 16 lanes, the LDS exchange, uniform partitioned overlap-add with the three previous spectra and
the carry as the only state, runs whose state is recomputed from the four processed quanta in front of them, LINK_SKIP /
LINK_FRESH — against the float64 direct form of the definition (out_q[i] = sum_j h[j] x[i - j], x continued into the
previously PROCESSED quanta).  No GPU needed; tests/test_hrtf.py then holds the kernel itself against the oracle."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
RQ = 128
SKIP, FRESH = -2, -1


@pytest.fixture(scope="module")
def emulator(tmp_path_factory):
    if not os.path.exists(CLANG):
        pytest.skip("no clang++ (the header uses ext_vector_type)")
    out = tmp_path_factory.mktemp("hrtffft") / "hrtf_fft_emulate"
    subprocess.check_call([CLANG, "-O2", "-std=c++17", "-ffp-contract=off", os.path.join(ROOT, "tools", "hrtf_fft_emulate.cpp"),
                           "-o", str(out)])
    return str(out)


def run_emulator(emulator, tmp_path, x, pair, prev, seg_len):
    nq = x.size // RQ
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        np.int32([pair.shape[0], nq, seg_len]).tofile(f)
        np.asarray(prev, np.int32).tofile(f)
        np.ascontiguousarray(pair, np.float32).tofile(f)
        np.ascontiguousarray(x, np.float32).tofile(f)
    subprocess.check_call([emulator, fin, fout])
    return np.fromfile(fout, np.float32).reshape(2, nq * RQ)


def definition(x, pair, prev):
    """the direct form over the stream of PROCESSED quanta (a skipped quantum is not part of the FIR's history; FRESH starts a
    new stream), float64"""
    nq = x.size // RQ
    out = np.zeros((2, nq * RQ))
    h = pair.astype(np.float64)
    stream = np.zeros(0)
    for q, link in enumerate(prev):
        if link == SKIP:
            continue
        if link == FRESH:
            stream = np.zeros(0)
        stream = np.concatenate([stream, x[q * RQ:(q + 1) * RQ].astype(np.float64)])
        win = stream[-(RQ + h.shape[0] - 1):]
        win = np.concatenate([np.zeros(RQ + h.shape[0] - 1 - win.size), win])
        for ear in range(2):
            out[ear, q * RQ:(q + 1) * RQ] = np.convolve(win, h[:, ear])[h.shape[0] - 1:h.shape[0] - 1 + RQ]
    return out


def links(pattern):
    """'P' processed (prev = the last processed quantum, FRESH for the first), '.' skipped, 'F' processed and fresh"""
    prev, last = [], None
    for c in pattern:
        if c == ".":
            prev.append(SKIP)
        else:
            prev.append(FRESH if (c == "F" or last is None) else last)
            last = len(prev) - 1
    return prev


def hrir(taps, seed):
    rng = np.random.default_rng(seed)
    env = np.exp(-np.arange(taps) / (taps / 5.0))
    return (rng.standard_normal((taps, 2)) * env[:, None] * 0.2).astype(np.float32)


@pytest.mark.parametrize("taps", [415, 512, 300, 128, 37])
@pytest.mark.parametrize("seg_len", [5, 64])
def test_all_processed(emulator, tmp_path, taps, seg_len):
    nq = 40
    x = np.random.default_rng(taps).uniform(-1, 1, nq * RQ).astype(np.float32)
    pair = hrir(taps, taps + 1)
    prev = links("P" * nq)
    got = run_emulator(emulator, tmp_path, x, pair, prev, seg_len)
    want = definition(x, pair, prev)
    scale = np.abs(want).max()
    assert np.sqrt(np.mean((got - want) ** 2)) <= 3e-7 * scale and np.abs(got - want).max() <= 2e-6 * scale


@pytest.mark.parametrize("pattern", ["PPPP....PPPP..P.P.PPPPPPPP....PPPP", "....PPPPPPPPPPPP........", "PPPPPPFPPPP..PPPFPPPPPPP", "P" + "." * 30 + "P",
                                     "PP.PP.PP.PP.PP.PP.PP.PP.PP.PP.PP"])
@pytest.mark.parametrize("seg_len", [3, 7, 100])
def test_skipped_and_fresh_quanta(emulator, tmp_path, pattern, seg_len):
    nq = len(pattern)
    x = np.random.default_rng(len(pattern)).uniform(-1, 1, nq * RQ).astype(np.float32)
    pair = hrir(415, 9)
    prev = links(pattern)
    got = run_emulator(emulator, tmp_path, x, pair, prev, seg_len)
    want = definition(x, pair, prev)
    scale = np.abs(want).max()
    assert np.abs(got - want).max() <= 2e-6 * scale
    for q, link in enumerate(prev):
        if link == SKIP:
            assert not got[:, q * RQ:(q + 1) * RQ].any()
