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
    lib.cn_node.argtypes = [C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.c_uint64, C.c_uint32, C.POINTER(C.c_uint8),
                            C.POINTER(C.c_uint8), C.POINTER(C.c_uint8)]
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


# ---- the whole node: tail counter + routing (waa_conv_noise.hpp::conv_noise_node_step) against the oracle's ConvolverNode -------------
NODE_SR = 32768.0  # (a quantum is 1 / 256 s: start times on quantum boundaries are exact in binary, the sources start ON them)


def _node_case(cn, orc, rng, ir_nch, taps, pad_to):
    """a stereo and a mono BufferSource (quantum-aligned start, whole quanta long, runs of digital silence inside) into one
    ConvolverNode; the response: `taps` taps, zero-padded to `pad_to` frames (the tail counter runs to the PADDED length)"""
    import web_audio_api_rs_amd as waa
    nq = 150
    h = np.zeros((ir_nch, pad_to), np.float32)
    h[:, :taps] = rng.standard_normal((ir_nch, taps)) * 0.1
    srcs = []
    for nch in (2, 1):
        n_q = int(rng.integers(4, 40))
        x = rng.standard_normal((nch, n_q * RQ)).astype(np.float32) * 0.3
        for _ in range(int(rng.integers(1, 4))):   # digital silence inside the buffer: active quanta that hold zeros
            a = int(rng.integers(0, n_q * RQ))
            x[:, a:a + int(rng.integers(100, 1500))] = 0
        srcs.append((int(rng.integers(0, 70)), x))
    c = waa.OfflineAudioContext(2, nq * RQ, NODE_SR, n_instances=1, binding=orc)
    conv = c.create_convolver(buffer=waa.AudioBuffer(h, NODE_SR), disable_normalization=True)
    mixed = np.zeros((2, nq * RQ), np.float32)
    count = np.zeros(nq, np.int32)
    for start, x in srcs:
        s = c.create_buffer_source()
        s.set_buffer_batch(x[None], NODE_SR)
        s.start_at(start * RQ / NODE_SR)
        s.connect(conv)
        n = min(x.shape[1], (nq - start) * RQ)
        mixed[0, start * RQ:start * RQ + n] += x[0, :n]
        mixed[1, start * RQ:start * RQ + n] += x[-1, :n]   # (a mono source is up-mixed to both channels when the input is stereo)
        count[start:start + n // RQ] = np.maximum(count[start:start + n // RQ], x.shape[0])
    conv.connect(c.destination())
    o = c.start_rendering_sync().data[0]
    c.close()
    code = np.where(count == 0, 0x81, count).astype(np.uint8)
    nzq = np.any(mixed.reshape(2, nq, RQ) != 0, axis=2)
    nz = (nzq[0].astype(np.uint8) | (np.where(count == 2, nzq[1], False).astype(np.uint8) << 1)).astype(np.uint8)
    segc = (C.c_uint32 * 4)()
    mask = (C.c_uint64 * 4)()
    for k in range(max(ir_nch, 2)):
        hk = np.ascontiguousarray(h[min(k, ir_nch - 1)])
        sc, mk = C.c_uint32(), C.c_uint64()
        cn.cn_describe(hk.ctypes.data_as(FP), len(hk), C.byref(sc), C.byref(mk))
        segc[k], mask[k] = sc.value, mk.value
    out = np.zeros(nq, np.uint8)
    U8 = C.POINTER(C.c_uint8)
    cn.cn_node(ir_nch, segc, mask, pad_to, nq, code.ctypes.data_as(U8), nz.ctypes.data_as(U8), out.ctypes.data_as(U8))
    got = np.any(o.reshape(2, nq, RQ) != 0, axis=2)
    cut = out == 0x80
    outn = out >> 4
    want0 = ~cut & ((out & 1) != 0)
    want1 = ~cut & np.where(outn == 1, (out & 1) != 0, (out & 2) != 0)   # (a mono output is up-mixed to both destination channels)
    bad = np.nonzero((got[0] != want0) | (got[1] != want1))[0]
    assert bad.size == 0, (f"ir {ir_nch}ch {taps}/{pad_to}: quanta {bad[:8]}: reference {got[:, bad[:8]].astype(int)}, "
                           f"automaton {out[bad[:8]]}, codes {code[bad[:8]]}, nz {nz[bad[:8]]}; sources {[(s, x.shape) for s, x in srcs]}")
    return cut.sum(), (~cut & (outn == 1)).sum()


@pytest.mark.parametrize("ir_nch", [1, 2, 4])
def test_node_routing_and_tail_counter_against_the_oracles_convolver_node(cn, orc, ir_nch):
    """mono / stereo / silent input quanta in every order, the tail counter's standstill (the blocks of the FFTConvolvers do not
    advance while the node puts out silence) and the second convolver of a mono response that only hears stereo quanta"""
    rng = np.random.default_rng(77 + ir_nch)
    cuts = monos = 0
    for trial in range(40):
        taps = int(rng.choice([8, 200, 1024, 1300, 2500]))
        pad_to = taps + int(rng.choice([0, 0, 700, 3000]))
        a, b = _node_case(cn, orc, rng, ir_nch, taps, pad_to)
        cuts += a
        monos += b
    assert cuts > 100 and (ir_nch != 1 or monos > 50)  # (the cases did reach the standstill and the one-convolver route)
