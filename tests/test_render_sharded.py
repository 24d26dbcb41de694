"""sharding.render_sharded — the N-device render component: host buffers of all contexts ->
contiguous ranges per device -> pipelined sub-batches (upload || render || download) -> host buffers.  The logic is
backend independent, so the partition / ordering / error paths are exercised here with the CPU oracle standing in for the
device library; the GPU tests run it on the HIP library (several slots on one device on a 1-GPU box, two devices where
there are two) and require the union of the shards to equal the unsharded render bit for bit."""
import numpy as np
import pytest

import web_audio_api_rs_amd as waa
from graphs import c2, rms_err, white_noise
from web_audio_api_rs_amd.sharding import plan_shards, render_sharded

RQ = 128


def _build(be, with_analyser=False):
    def build(n, device):
        ctx = waa.OfflineAudioContext(2, RQ * 30 + 5, 48000.0, n_instances=n, binding=be, device=device)
        src = ctx.create_buffer_source()
        node = src.connect(ctx.create_biquad_filter(type_="lowpass", frequency=200.0, q=1.0)).connect(ctx.create_gain(gain=0.5))
        if with_analyser:
            node = node.connect(ctx.create_analyser(fft_size=256))
        node.connect(ctx.destination())
        src.start()
        return ctx, src
    return build


def test_plan_shards_covers_everything_in_order():
    for n in (1, 5, 8, 64, 513):
        for devices in ([0], [0, 1], [0, 0, 1], list(range(8))):
            for parts in (1, 3, 8):
                sh = plan_shards(n, devices, parts)
                cover = sorted((lo, hi) for _, _, _, lo, hi in sh)
                assert cover[0][0] == 0 and cover[-1][1] == n
                assert all(cover[i][1] == cover[i + 1][0] for i in range(len(cover) - 1))
                for slot in range(len(devices)):
                    ks = [k for s, _, k, _, _ in sh if s == slot]
                    assert ks == list(range(len(ks)))  # the turn-taking order of a slot has no holes


@pytest.mark.parametrize("devices,parts", [([-1], 1), ([-1], 3), ([-1, -1], 2), ([-1, -1, -1], 8)])
def test_sharded_render_equals_single_batch_oracle(orc, devices, parts):
    n, frames = 7, RQ * 30 + 5
    noise = white_noise(n, 2, frames)
    out = np.zeros((n, 2, frames), np.float32)
    info = render_sharded(_build(orc), noise, out, devices=devices, sub_batches=parts)
    assert sorted((lo, hi) for _, lo, hi in info["shards"])[0][0] == 0 and info["seconds"] > 0
    ctx, src = _build(orc)(n, -1)
    src.set_buffer_batch(noise, 48000.0)
    ref = ctx.start_rendering_sync().data
    ctx.close()
    assert np.array_equal(out, ref)


def test_sharded_render_pcm16_and_pull(orc):
    n, frames = 5, RQ * 30 + 5
    rng = np.random.default_rng(3)
    pcm = rng.integers(-32768, 32767, (n, frames, 2), dtype=np.int16)
    out = np.zeros((n, 2, frames), np.float32)
    bins = np.zeros((n, 128), np.float32)

    def pull(ctx, lo, hi):
        an = next(nd for nd in ctx._nodes if isinstance(nd, waa.AnalyserNode))
        an.get_float_frequency_data_all(out=bins[lo:hi])

    render_sharded(_build(orc, with_analyser=True), pcm, out, devices=[-1, -1], sub_batches=2, pcm16=True, pull=pull)
    ctx, src = _build(orc, with_analyser=True)(n, -1)
    src.set_buffer_batch(np.ascontiguousarray(pcm.transpose(0, 2, 1).astype(np.float32) / np.float32(32768.0)), 48000.0)
    ref = ctx.start_rendering_sync().data
    an = next(nd for nd in ctx._nodes if isinstance(nd, waa.AnalyserNode))
    ref_bins = an.get_float_frequency_data_all()
    ctx.close()
    assert np.array_equal(out, ref) and np.array_equal(bins, ref_bins)


def test_a_failing_sub_batch_raises_and_does_not_hang(orc):
    def build(n, device):
        ctx, src = _build(orc)(n, device)
        if n == 2:  # (the second of the two sub-batches: 5 contexts = 3 + 2)
            ctx.create_stereo_panner(channel_count=2, channel_count_mode="max")  # NotSupportedError at batch creation
        return ctx, src
    noise = white_noise(5, 2, RQ * 30 + 5)
    with pytest.raises(waa.WaaError, match="NotSupportedError"):
        render_sharded(build, noise, np.zeros_like(noise), devices=[-1], sub_batches=2)


@pytest.mark.gpu
@pytest.mark.parametrize("slots,parts", [(1, 4), (2, 2), (3, 3)])
def test_sharded_render_equals_single_batch_hip(hip, orc, slots, parts):
    n, frames = 13, RQ * 30 + 5
    noise = white_noise(n, 2, frames)
    out = np.zeros((n, 2, frames), np.float32)
    ndev = hip.device_count()
    devices = [s % ndev for s in range(slots)]  # (distinct devices where the box has them)
    render_sharded(_build(hip), noise, out, devices=devices, sub_batches=parts)
    ctx, src = _build(hip)(n, 0)
    src.set_buffer_batch(noise, 48000.0)
    whole = ctx.start_rendering_sync().data
    ctx.close()
    assert np.array_equal(out, whole)  # sharding changes nothing, bit for bit
    ctx, src = _build(orc)(n, -1)
    src.set_buffer_batch(noise, 48000.0)
    ref = ctx.start_rendering_sync().data
    ctx.close()
    assert rms_err(out, ref).max() <= 1e-6


@pytest.mark.gpu
def test_sharded_render_two_devices_hip(hip):
    if hip.device_count() < 2:
        pytest.skip("needs a node with at least two GPUs (the driver's scaling run covers it otherwise)")
    n, frames = 16, RQ * 30 + 5
    noise = white_noise(n, 2, frames)
    out = np.zeros((n, 2, frames), np.float32)
    info = render_sharded(_build(hip), noise, out, devices=[0, 1], sub_batches=2)
    assert {d for d, _, _ in info["shards"]} == {0, 1}
    ctx, src = _build(hip)(n, 0)
    src.set_buffer_batch(noise, 48000.0)
    whole = ctx.start_rendering_sync().data
    ctx.close()
    assert np.array_equal(out, whole)
