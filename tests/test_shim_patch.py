"""shim/ — the reference-side binding as files.  Nothing here can compile it (no Rust toolchain in the
authoring container or on the GPU boxes); what CAN be checked without one: the patch still applies to the reference tree it was
cut from, the module files under shim/src/ are the ones the patch adds, and every library entry point the Rust side declares
exists in include/waa_hip.h with the same parameter count.  Skipped where /root/reference is absent (the GPU box)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
PATCH = os.path.join(ROOT, "shim", "reference.patch")


@pytest.mark.skipif(not os.path.isdir(REF) or shutil.which("patch") is None, reason="no reference tree / patch(1) here")
def test_patch_applies_and_carries_the_module_files(tmp_path):
    tree = tmp_path / "crate"
    shutil.copytree(REF, tree, ignore=shutil.ignore_patterns("target", ".git"))
    r = subprocess.run(["patch", "-p1", "-s", "-i", PATCH], cwd=tree, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    for f in ("mod.rs", "ffi.rs"):
        assert open(tree / "src" / "gpu" / f).read() == open(os.path.join(ROOT, "shim", "src", "gpu", f)).read(), f
    # every processor SURVEY section 8 names reports itself to the shim
    for node, proc in (("gain", "GainRenderer"), ("biquad_filter", "BiquadFilterRenderer"), ("convolver", "ConvolverRenderer"),
                       ("stereo_panner", "StereoPannerRenderer"), ("panner", "PannerRenderer"), ("analyser", "AnalyserRenderer"),
                       ("waveshaper", "WaveShaperRenderer"), ("audio_buffer_source", "AudioBufferSourceRenderer"),
                       ("iir_filter", "IirFilterRenderer"), ("oscillator", "OscillatorRenderer"), ("constant_source", "ConstantSourceRenderer"),
                       ("delay", "DelayWriter"), ("delay", "DelayReader"), ("destination", "DestinationRenderer")):
        src = open(tree / "src" / "node" / f"{node}.rs").read()
        body = src[src.index(f"impl AudioProcessor for {proc}"):]
        assert "fn gpu_desc(&self)" in body[:body.index("fn process(")], (node, proc)


def test_rust_declarations_match_the_header():
    ffi = open(os.path.join(ROOT, "shim", "src", "gpu", "ffi.rs")).read()
    header = open(os.path.join(ROOT, "include", "waa_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    decls = re.findall(r"pub fn (waa_\w+)\s*\((.*?)\)\s*(?:->\s*[\w:*\s]+)?;", ffi, flags=re.S)
    assert len(decls) >= 18
    for name, params in decls:
        m = re.search(r"\b%s\s*\((.*?)\)\s*;" % name, header, flags=re.S)
        assert m, f"{name} is declared in shim/src/gpu/ffi.rs but not in include/waa_hip.h"
        n_rust = len([p for p in params.split(",") if p.strip()])
        c_params = m.group(1).strip()
        n_c = 0 if c_params in ("", "void") else len([p for p in c_params.split(",") if p.strip()])
        assert n_rust == n_c, (name, n_rust, n_c)
