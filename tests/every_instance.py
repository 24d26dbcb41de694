"""Helpers of the every-instance parity tests (tests/test_full_size_all_instances.py): the oracle renders ALL contexts of a
batch on every usable CPU and every (instance, channel) row is compared.  Kept apart from the -m gpu module so that the
negative tests (tests/test_nan_proof_helpers.py: a non-finite sample MUST fail the helper) run on a box without a GPU."""
import ctypes
import os

import numpy as np

from graphs import assert_all_finite, assert_le, rms_err, strict_max

TOL = 1e-6  # the north star's tolerance: RMS per channel


def usable_cpus():
    """CPUs this process may really use: the affinity mask capped by the cgroup quota (the GPU box shows 256, allows 16)."""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            aff = max(1, min(aff, int(float(q) / float(per) + 0.999)))
    except (OSError, ValueError):
        pass
    return aff


def _oracle_chunks(orc, build, n_inst, chunk):
    """yields (lo, hi, ctx, nodes) for the oracle context of instances lo..hi, rendered on all usable CPUs"""
    orc.lib.orc_set_threads.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    threads = usable_cpus()
    for lo in range(0, n_inst, chunk):
        hi = min(n_inst, lo + chunk)
        ctx, nodes = build(orc, lo, hi)
        ctx.prepare()
        orc.lib.orc_set_threads(ctx._handle, threads)
        yield lo, hi, ctx, nodes


def _compare_all(orc, out, build, chunk, max_abs):
    """out: the device's [n_inst, 2, frames]; build(binding, lo, hi) -> (ctx, nodes) of instances lo..hi.
    NaN-proof (round-4 review, weak item 1): the device render and the oracle render must be finite everywhere, every
    accumulation goes through strict_max (which fails on a NaN instead of dropping it) and every bound through assert_le."""
    n_inst = out.shape[0]
    assert_all_finite(out, "device render")
    worst_rms, worst_abs, where, peak = 0.0, 0.0, None, 0.0
    for lo, hi, ctx, _ in _oracle_chunks(orc, build, n_inst, chunk):
        ref = ctx.start_rendering_sync().data
        ctx.close()
        assert ref.shape == out[lo:hi].shape, (ref.shape, out[lo:hi].shape)
        assert_all_finite(ref, f"oracle render of instances {lo}..{hi}")
        err = rms_err(out[lo:hi], ref)
        mab = np.abs(out[lo:hi] - ref).max(axis=-1)
        assert_all_finite(err, "per-channel RMS errors")
        k = np.unravel_index(int(np.argmax(err)), err.shape)
        if float(err[k]) > worst_rms:
            where = (lo + int(k[0]), int(k[1]))
        worst_rms = strict_max(worst_rms, err[k])
        worst_abs = strict_max(worst_abs, mab.max())
        peak = strict_max(peak, np.abs(ref).max())
        del ref
    print(f"all {n_inst} instances: worst per-channel RMS error {worst_rms:.3e} at (instance, channel) {where}, "
          f"max |diff| {worst_abs:.3e}, peak {peak:.3f}")
    assert_le(worst_rms, TOL, where)
    assert_le(worst_abs, max_abs, "max |diff|")
    assert peak > 1e-3
    return worst_rms, worst_abs


