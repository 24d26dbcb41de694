"""CPU checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/waa_hip.h declares; the oracle exports the same set with the orc_ prefix; argument
validation that needs no GPU behaves like the reference's panics."""
import ctypes
import os
import re

import numpy as np
import pytest

import web_audio_api_rs_amd as waa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "waa_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(waa_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_exported_by_product_library(hip):
    lib = ctypes.CDLL(waa.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 29
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/waa_hip.h but not exported by libwaa_hip.so"


def test_binding_table_matches_header():
    assert sorted("waa_" + k for k in waa.api.ABI) == declared_symbols()


def test_oracle_exports_same_entry_points(orc_lib):
    for s in declared_symbols():
        assert hasattr(orc_lib, "orc_" + s[len("waa_"):])


def test_product_never_references_oracle():
    """The product path must not import, link or load anything under oracle/."""
    pkg = os.path.join(ROOT, "web-audio-api-rs_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f)).read()
                assert "liboracle" not in text and not re.search(r"(import|from)\s+oracle|oracle/", text), (dirpath, f)
    out = os.popen(f"ldd {waa.LIB_PATH}").read()
    assert "oracle" not in out


@pytest.fixture(params=["orc", "hip-plan-only"])
def be_plan(request, orc, hip):
    """(binding, context kwargs): the oracle, or the product on a plan-only batch (no GPU needed)"""
    return (orc, {}) if request.param == "orc" else (hip, {"device": waa.PLAN_ONLY})


def test_validation_without_gpu(hip):
    """Graph validation happens before the device is touched and mirrors the reference's panics."""
    c = waa.OfflineAudioContext(2, 128, 44100.0, binding=hip)
    c.create_stereo_panner(channel_count=2, channel_count_mode="max")
    with pytest.raises(waa.WaaError, match="NotSupportedError"):
        c.start_rendering_sync()
    c = waa.OfflineAudioContext(2, 128, 44100.0, binding=hip)
    c.create_analyser(fft_size=100)
    with pytest.raises(waa.WaaError, match="IndexSizeError"):
        c.start_rendering_sync()
    c = waa.OfflineAudioContext(2, 128, 1000.0, binding=hip)
    with pytest.raises(waa.WaaError, match="NotSupportedError"):
        c.start_rendering_sync()


def test_context_construction_limits(be_plan):
    """OfflineAudioContext::new (offline.rs:78-81): assert_valid_number_of_channels / _buffer_length / _sample_rate
    (src/lib.rs:165-228), same messages, on both libraries; a destination wider than the device path renders is
    refused with status 4 by the product (never truncated)."""
    be, kw = be_plan

    def attempt(ch, length, sr, n_inst=1):
        c = waa.OfflineAudioContext(ch, length, sr, n_instances=n_inst, binding=be, **kw)
        s = c.create_buffer_source()
        s.set_buffer(waa.AudioBuffer(np.ones((1, 4), np.float32), 48000.0))
        s.connect(c.destination())
        s.start()
        c.prepare()
        return c

    for ch in (0, 33):
        with pytest.raises(waa.WaaError, match="NotSupportedError - Invalid number of channels") as e:
            attempt(ch, 128, 48000.0)
        assert e.value.status == 2
    with pytest.raises(waa.WaaError, match="NotSupportedError - Invalid length: 0") as e:
        attempt(2, 0, 48000.0)
    assert e.value.status == 2
    for sr in (2999.0, 768001.0, float("nan")):
        with pytest.raises(waa.WaaError, match="NotSupportedError - Invalid sample rate"):
            attempt(2, 128, sr)
    with pytest.raises(waa.WaaError) as e:
        attempt(2, 128, 48000.0, n_inst=0)
    assert e.value.status == 1
    attempt(2, 1, 3000.0).close()  # the smallest legal context
    attempt(6, 129, 768000.0).close()


def test_host_helpers_match_oracle(hip, orc):
    """Control-side helpers of the product (resample, frequency response) against the oracle."""
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, (2, 1000)).astype(np.float32)
    assert np.array_equal(waa.resample(hip, x, 44100.0, 48000.0), waa.resample(orc, x, 44100.0, 48000.0))
    hz = np.linspace(0, 24000, 64).astype(np.float32)
    for t in waa.BIQUAD_TYPE:
        outs = []
        for b in (hip, orc):
            mag, ph = np.empty_like(hz), np.empty_like(hz)
            FP = ctypes.POINTER(ctypes.c_float)
            b.check(b.biquad_frequency_response(waa.BIQUAD_TYPE[t], 48000.0, 1234.0, 100.0, 2.0, 4.0,
                                                hz.ctypes.data_as(FP), mag.ctypes.data_as(FP), ph.ctypes.data_as(FP),
                                                hz.size))
            outs.append((mag, ph))
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def test_panner_option_limits(be_plan):
    """PannerOptions validation (panner.rs:408-428, assert_valid_cone_outer_gain :20-33): same messages and error
    classes on both libraries, before anything is rendered."""
    be, kw = be_plan

    def attempt(**opts):
        c = waa.OfflineAudioContext(2, 128, 48000.0, binding=be, **kw)
        s = c.create_constant_source()
        s.connect(c.create_panner(**opts)).connect(c.destination())
        s.start()
        c.prepare()
        c.close()

    attempt()  # defaults
    attempt(ref_distance=0.0, rolloff_factor=0.0, cone_outer_gain=1.0)  # the boundaries are legal
    for opts, msg, status in ((dict(ref_distance=-0.1), "RangeError - refDistance cannot be negative", 1),
                              (dict(max_distance=0.0), "RangeError - maxDistance must be strictly positive", 1),
                              (dict(rolloff_factor=-1.0), "RangeError - rolloffFactor cannot be negative", 1),
                              (dict(cone_outer_gain=1.5), "InvalidStateError - coneOuterGain must be in the range", 3),
                              (dict(cone_outer_gain=-0.1), "InvalidStateError - coneOuterGain must be in the range", 3)):
        with pytest.raises(waa.WaaError, match=msg) as e:
            attempt(**opts)
        assert e.value.status == status
