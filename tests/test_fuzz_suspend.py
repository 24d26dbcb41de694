"""Random graphs (tests/test_fuzz_graphs.py's generator) MUTATED at random suspend points: connections cut and made again,
AudioParam values set, automation scheduled from inside callbacks, sources stopped — the device (which compiles the history into
its plan: gated connections, late events, clamped times) against the oracle (which applies every control message in its quantum
loop).  OfflineAudioContext::suspend_sync, src/context/offline.rs:359-397."""
import os

import numpy as np
import pytest

import web_audio_api_rs_amd as waa
from test_fuzz_graphs import FRAMES, SR, build_random_graph

RQ = 128


def rms_err(a, b):
    return np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2, axis=-1))


def mutate(ctx, seed):
    """the same random edits for both back-ends (drawn from the graph's node list, which is the same on both)"""
    rng = np.random.default_rng(seed + 424242)
    nq = (FRAMES + RQ - 1) // RQ
    points = sorted(set(int(q) for q in rng.integers(1, nq - 1, size=int(rng.integers(1, 5)))))
    edges = [e for e in ctx._edges if not (e[3] & 0x80000000)]
    params = [(nd, p) for nd in ctx._nodes for p in nd.params if type(nd).__name__ in ("GainNode", "BiquadFilterNode", "StereoPannerNode", "ConstantSourceNode")]
    sources = [nd for nd in ctx._nodes if type(nd).__name__ in ("ConstantSourceNode", "OscillatorNode")]
    descr = []
    cut = []
    stopped = set(nd.id for nd in sources if nd._stops)
    for q in points:
        kind = rng.choice(["cut", "set", "ramp", "stop", "rejoin"])
        t = q * RQ / SR
        if kind == "cut" and edges:
            e = edges[int(rng.integers(0, len(edges)))]
            if e not in cut:
                cut.append(e)
                frm, to = ctx._nodes[e[0]], ctx._nodes[e[2]]
                ctx.suspend_sync(t, lambda c, frm=frm, to=to: frm.disconnect(to))
                descr.append(f"q{q}:cut{e[0]}->{e[2]}")
                continue
        if kind == "rejoin" and cut:
            e = cut.pop(int(rng.integers(0, len(cut))))
            frm, to = ctx._nodes[e[0]], ctx._nodes[e[2]]
            ctx.suspend_sync(t, lambda c, frm=frm, to=to: frm.connect(to))
            descr.append(f"q{q}:join{e[0]}->{e[2]}")
            continue
        if kind == "stop" and sources:
            s = sources[int(rng.integers(0, len(sources)))]
            if s.id not in stopped:  # (one stop per source: a second stop message at a later suspend point is status 4 on the device)
                stopped.add(s.id)
                ctx.suspend_sync(t, lambda c, s=s, t=t: s.stop_at(t * float(0.5)))  # (in the past: stops at the block)
                descr.append(f"q{q}:stop{s.id}")
                continue
        if params:
            nd, p = params[int(rng.integers(0, len(params)))]
            name = type(nd).__name__
            lo, hi = {"GainNode": (0.0, 1.2), "StereoPannerNode": (-1.0, 1.0), "ConstantSourceNode": (-0.5, 0.5)}.get(name, (0.2, 3.0))
            if name == "BiquadFilterNode":
                p = nd.q  # (Q: any positive value is a legal filter)
            v = float(np.float32(rng.uniform(lo, hi)))
            if kind == "ramp":
                end = float(min(t + float(rng.uniform(0.005, 0.05)), FRAMES / SR))
                ctx.suspend_sync(t, lambda c, p=p, v=v, end=end: p.linear_ramp_to_value_at_time(v, end))
                descr.append(f"q{q}:ramp{nd.id}")
            else:
                ctx.suspend_sync(t, lambda c, p=p, v=v: p.set_value(v))
                descr.append(f"q{q}:set{nd.id}")
    return "+".join(descr)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(os.environ.get("FUZZ_FIRST", "0")), int(os.environ.get("FUZZ_FIRST", "0")) + int(os.environ.get("FUZZ_SEEDS_SUSPEND", "40"))))
def test_random_graph_mutated_at_suspend_points(hip, orc, seed):
    ch, descr = build_random_graph(hip, seed)
    try:
        edits = mutate(ch, seed)
        g = ch.start_rendering_sync().data
    except waa.WaaError as e:
        if e.status == 4:
            pytest.skip(f"out of scope on the device path: {e} [{descr}]")
        if "cannot suspend multiple times" in str(e):
            pytest.skip("two edits drew the same quantum")
        if "NotSupportedError - scheduling" in str(e):
            pytest.skip("an edit fell inside a value curve: the reference panics there too")
        raise
    ch.close()
    co, _ = build_random_graph(orc, seed)
    mutate(co, seed)
    o = co.start_rendering_sync().data
    co.close()
    assert np.isfinite(o).all() and np.isfinite(g).all(), (descr, edits)
    scale = max(1.0, float(np.abs(o).max()))
    assert rms_err(g, o).max() <= 1e-6 * scale, f"{descr} | {edits}: rms {rms_err(g, o).max():.3g}"
    assert np.abs(g - o).max() <= 2e-5 * scale, f"{descr} | {edits}: max |d| {np.abs(g - o).max():.3g}"


def test_mutations_run_on_the_oracle(orc):
    """(CPU) the generator + the edits + the oracle's quantum loop: finite output, and the edits change it"""
    changed = 0
    for seed in range(12):
        a, _ = build_random_graph(orc, seed)
        plain = a.start_rendering_sync().data
        a.close()
        b2, _ = build_random_graph(orc, seed)
        try:
            edits = mutate(b2, seed)
        except waa.WaaError:
            continue
        out = b2.start_rendering_sync().data
        b2.close()
        assert np.isfinite(out).all(), edits
        changed += int(not np.array_equal(out, plain))
    assert changed >= 6


@pytest.mark.gpu
def test_seed_16381_behind_six_other_graphs_on_poisoned_memory():
    """the one failure of the 12 000-graph campaign (profiles/r06ac_fuzz_suspend.txt): rendered alone the graph was equal, behind
    other graphs of the same process its last quantum held 1e23 in one channel.  Both AudioBuffers end inside the last render
    quantum (the render's length is no multiple of 128); a source whose quanta all lie in its buffer back to back was handed to its
    consumers as one linear run of n_quanta * 128 frames, and they read up to 127 frames BEHIND the buffer: stale memory of the
    batches before.  Harmless to causal consumers (those frames lie behind the render's end), not to the ConvolverNode behind
    them: it transforms the block, stale tail and all, and 1e30 there is 1e23 of roundoff in the block's valid frames
    (waa_plan_sources.cpp: such a source is no longer `linear_all`).  WAA_POISON_ALLOC is read once per process: a subprocess."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WAA_POISON_ALLOC="1")
    out = subprocess.check_output([sys.executable, os.path.join(root, "tools", "suspend_seq_probe.py"), "16375", "16381"], env=env).decode()
    head = out.splitlines()[0]
    assert head.startswith("16381 ") and " max " in head, out[:500]
    assert float(head.rsplit(" max ", 1)[1]) <= 1e-6, out[:1500]
