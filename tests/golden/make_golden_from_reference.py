"""Extracts the reference's own known-answer vectors into small JSON fixtures.

Run in the authoring container only (it reads /root/reference, which does not exist on
the GPU box).  The numbers are the expected values of the reference's unit tests
(browser-derived biquad responses: src/node/biquad_filter.rs:1000-1412); they are data
the oracle must reproduce, not code.

    python tests/golden/make_golden_from_reference.py
"""
import json
import os
import re

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def floats(block):
    return [float(x.replace("_", "")) for x in re.findall(r"-?\d[\d_]*\.?\d*(?:e-?\d+)?", block)]


def biquad_kats():
    src = open(os.path.join(REF, "src/node/biquad_filter.rs")).read()
    out = {}
    for m in re.finditer(r"fn test_frequency_responses_(\w+)\(\)\s*\{(.*?)\n    \}\n", src, re.S):
        name, body = m.group(1), m.group(2)
        sr = float(re.search(r"OfflineAudioContext::new\(\d+, \d+, ([\d_\.]+)\)", body).group(1).replace("_", ""))
        g = lambda key: float(re.search(r"let %s = (-?[\d\.]+);" % key, body).group(1))
        arr = lambda key: floats(re.search(r"let %s = \[(.*?)\];" % key, body, re.S).group(1))
        out[name] = {
            "sample_rate": sr, "frequency": g("frequency"), "q": g("q"), "gain": g("gain"),
            "freqs": arr("freqs"), "expected_mags": arr("expected_mags"), "expected_phases": arr("expected_phases"),
            "source": "src/node/biquad_filter.rs test_frequency_responses_%s (abs_all <= 1e-6)" % name,
        }
    return out


if __name__ == "__main__":
    kats = biquad_kats()
    assert len(kats) == 8, sorted(kats)
    with open(os.path.join(HERE, "biquad_frequency_response.json"), "w") as f:
        json.dump(kats, f, indent=1)
    print("wrote", len(kats), "biquad KATs")
