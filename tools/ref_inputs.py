#!/usr/bin/env python
"""tools/ref_inputs.py <dir> — the fixed-seed inputs of the reference-side parity pin (oracle/build_ref.sh,
oracle/ref_harness, tests/test_reference_dumps.py): raw little-endian f32 files that the Rust harness, the oracle and the
HIP library all read, so that the three render the same samples."""
import os
import sys

import numpy as np

FRAMES = 96000


def write(directory):
    os.makedirs(directory, exist_ok=True)
    rng = np.random.default_rng(0xA0D10)
    rng.uniform(-1.0, 1.0, (2, FRAMES)).astype("<f4").tofile(os.path.join(directory, "noise_stereo.f32"))
    rng.uniform(-1.0, 1.0, FRAMES).astype("<f4").tofile(os.path.join(directory, "noise_mono.f32"))
    np.tanh(np.linspace(-3.0, 3.0, 2049)).astype("<f4").tofile(os.path.join(directory, "curve_tanh.f32"))


if __name__ == "__main__":
    write(sys.argv[1])
