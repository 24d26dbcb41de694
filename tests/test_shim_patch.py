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


# ---- declarations: count AND type of every parameter, return type, struct layouts, constants ------------------------------------
_C_SCALARS = {"uint32_t": "u32", "int32_t": "i32", "uint64_t": "u64", "int64_t": "i64", "uint8_t": "u8", "int16_t": "i16", "float": "f32",
              "double": "f64", "size_t": "usize", "char": "c_char", "void": "c_void", "waa_status": "i32", "int": "i32"}


def _c_type(decl, named=True):
    """'const float* const* channels' -> ('ptr_const', 'ptr_const', 'f32'); the parameter name (if any) is dropped"""
    d = decl.strip()
    d = re.sub(r"/\*.*?\*/", "", d, flags=re.S).strip()
    toks = re.findall(r"[A-Za-z_]\w*|\*", d)
    if named and toks and toks[-1] != "*" and toks[-1] not in _C_SCALARS and not toks[-1].startswith("waa_") and len(toks) > 1:
        toks = toks[:-1]
    elif named and len([t for t in toks if t not in ("const", "*")]) > 1:
        toks = toks[:-1]
    base, rest = None, []
    pending_const = False
    i = 0
    # base type with its own const
    base_const = False
    while i < len(toks) and toks[i] != "*":
        if toks[i] == "const":
            base_const = True
        else:
            base = toks[i]
        i += 1
    out = []
    const_of_pointee = base_const
    while i < len(toks):
        assert toks[i] == "*", decl
        i += 1
        out.append("ptr_const" if const_of_pointee else "ptr_mut")
        const_of_pointee = False
        if i < len(toks) and toks[i] == "const":
            const_of_pointee = True
            i += 1
    out.reverse()
    scalar = _C_SCALARS.get(base, base)
    return tuple(out) + (scalar,)


def _rust_type(t):
    """'*const *const f32' -> ('ptr_const', 'ptr_const', 'f32')"""
    t = t.strip()
    out = []
    while t.startswith("*"):
        m = re.match(r"\*(const|mut)\s+", t)
        assert m, t
        out.append("ptr_" + m.group(1))
        t = t[m.end():]
    t = t.replace("std::ffi::", "").replace("std::os::raw::", "")
    return tuple(out) + (t,)


def _header():
    header = open(os.path.join(ROOT, "include", "waa_hip.h")).read()
    return re.sub(r"/\*.*?\*/", "", header, flags=re.S)


def _ffi():
    return open(os.path.join(ROOT, "shim", "src", "gpu", "ffi.rs")).read()


def _split_params(text):
    return [p for p in (q.strip() for q in text.split(",")) if p and p != "void"]


def test_type_translators():
    assert _c_type("const float* const* channels") == ("ptr_const", "ptr_const", "f32")
    assert _c_type("waa_batch** out") == ("ptr_mut", "ptr_mut", "waa_batch")
    assert _c_type("const waa_graph_desc* graph") == ("ptr_const", "waa_graph_desc")
    assert _c_type("uint32_t n") == ("u32",)
    assert _c_type("float* dst") == ("ptr_mut", "f32")
    assert _c_type("const void* data") == ("ptr_const", "c_void")
    assert _rust_type("*const *const f32") == ("ptr_const", "ptr_const", "f32")
    assert _rust_type("*mut *mut waa_batch") == ("ptr_mut", "ptr_mut", "waa_batch")
    assert _rust_type("*const std::ffi::c_void") == ("ptr_const", "c_void")
    assert _rust_type("u64") == ("u64",)


def test_rust_declarations_match_the_header():
    """every `pub fn waa_*` of ffi.rs against include/waa_hip.h: parameter count, every parameter's type (width, signedness,
    float / integer, pointer depth and constness per level) and the return type"""
    ffi, header = _ffi(), _header()
    decls = re.findall(r"pub fn (waa_\w+)\s*\((.*?)\)\s*(?:->\s*([\w:*\s]+?))?;", ffi, flags=re.S)
    assert len(decls) >= 19
    for name, params, ret in decls:
        m = re.search(r"([\w\s\*]+?)\b%s\s*\((.*?)\)\s*;" % name, header, flags=re.S)
        assert m, f"{name} is declared in shim/src/gpu/ffi.rs but not in include/waa_hip.h"
        rust = [p.split(":", 1) for p in _split_params(params)]
        c = _split_params(m.group(2))
        assert len(rust) == len(c), (name, len(rust), len(c))
        for (rname, rtype), cdecl in zip(rust, c):
            assert _rust_type(rtype) == _c_type(cdecl), (name, rname.strip(), rtype.strip(), cdecl)
        c_ret = m.group(1).strip().split("\n")[-1].strip()
        c_ret_t = _c_type(c_ret, named=False)
        rust_ret_t = _rust_type(ret) if ret else ("c_void",)
        assert rust_ret_t == c_ret_t, (name, ret, c_ret)


def test_repr_c_structs_match_the_header():
    """waa_node_desc / waa_edge_desc / waa_graph_desc: same fields in the same order with the same types (arrays included)"""
    ffi, header = _ffi(), _header()
    for struct in ("waa_node_desc", "waa_edge_desc", "waa_graph_desc"):
        mc = re.search(r"typedef struct\s*\{([^{}]*)\}\s*%s\s*;" % struct, header, flags=re.S)
        mr = re.search(r"#\[repr\(C\)\][^{]*pub struct %s\s*\{(.*?)\}" % struct, ffi, flags=re.S)
        assert mc and mr, struct
        c_fields = []
        for f in (x.strip() for x in mc.group(1).split(";")):
            if not f:
                continue
            arr = re.search(r"\[(\d+)\]$", f)
            f = re.sub(r"\[\d+\]$", "", f)
            name = re.findall(r"[A-Za-z_]\w*", f)[-1]
            c_fields.append((name, _c_type(f), int(arr.group(1)) if arr else None))
        r_fields = []
        for f in (x.strip() for x in mr.group(1).split(",")):
            if not f:
                continue
            name, t = f.replace("pub ", "").split(":", 1)
            arr = re.match(r"\s*\[(\w+);\s*(\d+)\]", t)
            if arr:
                r_fields.append((name.strip(), _rust_type(arr.group(1)), int(arr.group(2))))
            else:
                r_fields.append((name.strip(), _rust_type(t), None))
        # (`from` is a keyword in neither language; the header and the Rust side use the same names)
        assert c_fields == r_fields, (struct, c_fields, r_fields)


def test_rust_constants_match_the_header():
    ffi, header = _ffi(), open(os.path.join(ROOT, "include", "waa_hip.h")).read()
    consts = dict(re.findall(r"pub const (WAA_\w+): [ui]\d+ = ([^;]+);", ffi))
    assert len(consts) >= 20
    enums = {}
    for body in re.findall(r"enum\s*\{(.*?)\}", header, flags=re.S):
        for k, v in re.findall(r"(WAA_\w+)\s*=\s*(-?\w+)", body):
            enums[k] = int(v, 0)
    for k, v in re.findall(r"#define (WAA_\w+) \(?(-?(?:0x)?[0-9A-Fa-f]+)u?\)?\s", header):
        enums.setdefault(k, int(v, 0))
    for k, v in consts.items():
        rust_v = int(v.replace("_", ""), 0)
        assert k in enums, f"{k} is not a constant of include/waa_hip.h"
        assert enums[k] & 0xFFFFFFFFFFFFFFFF == rust_v & 0xFFFFFFFFFFFFFFFF, (k, v, enums[k])


def test_every_library_call_of_the_shim_is_declared():
    """every ffi::waa_* (function) and ffi::WAA_* (constant) that mod.rs uses exists in ffi.rs"""
    ffi = _ffi()
    mod = open(os.path.join(ROOT, "shim", "src", "gpu", "mod.rs")).read()
    fns = set(re.findall(r"pub fn (\w+)", ffi))
    consts = set(re.findall(r"pub const(?: fn)? (\w+)", ffi))
    types = set(re.findall(r"pub struct (\w+)", ffi))
    used = set(re.findall(r"ffi::(\w+)", mod)) - {"waa_", "WAA_"}  # (prose of the module header)
    assert len(used) >= 30
    missing = sorted(u for u in used if u not in fns | consts | types)
    assert not missing, f"mod.rs uses ffi::{missing} which shim/src/gpu/ffi.rs does not declare"
    # the custom PeriodicWave path of round 5
    assert "waa_oscillator_set_wavetable" in used and "waa_oscillator_set_wavetable" in fns


def test_module_header_is_not_stale():
    mod = open(os.path.join(ROOT, "shim", "src", "gpu", "mod.rs")).read()
    head = mod[:mod.index("mod ffi;")]
    assert "UNCOMPILED SKETCH" in head
    assert "not forwarded by this first version" not in head
    for kind in ("Panner", "Delay", "Oscillator"):
        assert f"GpuNode::{kind}" in mod and kind in head


def _strip_rust(src):
    """comments, string / char literals and lifetimes out of Rust source (enough for a bracket check)"""
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if src.startswith("//", i):
            j = src.find("\n", i)
            i = n if j < 0 else j
        elif src.startswith("/*", i):
            depth, i = 1, i + 2
            while i < n and depth:
                if src.startswith("/*", i):
                    depth, i = depth + 1, i + 2
                elif src.startswith("*/", i):
                    depth, i = depth - 1, i + 2
                else:
                    i += 1
        elif c == '"':
            i += 1
            while i < n and src[i] != '"':
                i += 2 if src[i] == "\\" else 1
            i += 1
        elif c == "r" and re.match(r'r#*"', src[i:]):
            m = re.match(r'r(#*)"', src[i:])
            end = src.find('"' + m.group(1), i + len(m.group(0)))
            i = n if end < 0 else end + 1 + len(m.group(1))
        elif c == "'":
            m = re.match(r"'(\\.|[^\\'])'", src[i:])      # a char literal ...
            if m:
                i += len(m.group(0))
            else:                                           # ... or a lifetime
                i += 1
        else:
            out.append(c)
            i += 1
    return "".join(out)


@pytest.mark.parametrize("path", ["shim/src/gpu/mod.rs", "shim/src/gpu/ffi.rs", "shim/harness/src/main.rs", "oracle/ref_harness/src/main.rs"])
def test_rust_sources_have_balanced_brackets(path):
    """the cheapest thing rustc would say first: every (, [, { closes in the right order (outside comments, strings, chars)"""
    full = os.path.join(ROOT, path)
    if not os.path.exists(full):
        pytest.skip(path)
    code = _strip_rust(open(full).read())
    stack, pairs = [], {")": "(", "]": "[", "}": "{"}
    line = 1
    for ch in code:
        if ch == "\n":
            line += 1
        elif ch in "([{":
            stack.append((ch, line))
        elif ch in ")]}":
            assert stack and stack[-1][0] == pairs[ch], f"{path}:{line}: unmatched {ch!r} (open: {stack[-1] if stack else None})"
            stack.pop()
    assert not stack, f"{path}: unclosed {stack[-1]}"


def test_patch_hunks_keep_brackets_balanced_per_added_function():
    """every `fn gpu_desc` the patch adds is a complete item: its added lines balance their own brackets"""
    patch = open(PATCH).read()
    added = "\n".join(l[1:] for l in patch.splitlines() if l.startswith("+") and not l.startswith("+++"))
    code = _strip_rust(added)
    assert code.count("{") == code.count("}") and code.count("(") == code.count(")") and code.count("[") == code.count("]")
    assert code.count("fn gpu_desc") >= 14
