"""tools/isa_waits.py <file.hip> [kernel-name-substring] — where a kernel waits for memory inside its loops.

Compiles the file for gfx950 into a temporary directory (hipcc -S, device only) and lists, per kernel and per loop (the
compiler's "=>This Inner Loop" / "in Loop:" block comments), the global / LDS accesses and every `s_waitcnt vmcnt(N)` with its
N.  vmcnt is ONE in-order counter for loads and stores: a `vmcnt(0)` inside a loop that also issues loads for a later
iteration waits for those too — the software prefetch is then none.  That is how the echo ring kernel's four hidden drains
were found; run it on a kernel before believing its pipeline.  No GPU needed."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fgpu-flush-denormals-to-zero",
         "--cuda-device-only", "-S"]


def main(argv):
    src = argv[0]
    want = argv[1] if len(argv) > 1 else ""
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-I" + os.path.join(ROOT, "web-audio-api-rs_amd", "csrc"),
                                                                  "-I" + os.path.join(ROOT, "include"), src, "-o", out],
                              stderr=subprocess.DEVNULL)
        text = open(out).read().splitlines()
    kernel, depth_of, cur = None, {}, None
    stats = {}
    for line in text:
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kernel = m.group(1)
            continue
        if kernel is None or (want and want not in kernel):
            continue
        if line.startswith("\t.end_amdhsa_kernel") or "s_endpgm" in line:
            pass
        m = re.match(r"^(\.LBB\d+_\d+):\s*;\s*(.*)$", line)
        if m:
            label, note = m.groups()
            d = re.search(r"Depth=(\d+)", note)
            h = re.search(r"Header=(BB\d+_\d+)", note)  # a block inside a loop names the loop's header block
            cur = ((".L" + h.group(1)) if h else label, int(d.group(1))) if ("Loop" in note and d) else None
            continue
        if re.match(r"^\.LBB\d+_\d+:", line):
            cur = None
            continue
        ins = line.strip().split()
        if not ins or ins[0].startswith((";", ".")):
            continue
        key = (kernel, cur[0] if cur else "(straight-line)", cur[1] if cur else 0)
        st = stats.setdefault(key, {"n": 0, "gload": 0, "gstore": 0, "ds": 0, "waits": []})
        st["n"] += 1
        op = ins[0]
        if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
            st["gload"] += 1
        elif op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic")):
            st["gstore"] += 1
        elif op.startswith("ds_"):
            st["ds"] += 1
        elif op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", line)
            if m:
                st["waits"].append(int(m.group(1)))
    last = None
    for (k, label, depth), st in stats.items():
        if depth == 0 and not st["waits"]:
            continue
        if k != last:
            print(k)
            last = k
        drains = sum(1 for w in st["waits"] if w == 0)
        flag = "  <-- vmcnt(0) in a loop that loads" if depth > 0 and drains and st["gload"] else ""
        print(f"  {label:16s} depth {depth}: {st['n']:5d} instr, {st['gload']:3d} global loads, {st['gstore']:3d} stores, {st['ds']:3d} LDS, "
              f"vmcnt waits {sorted(st['waits'])}{flag}")


if __name__ == "__main__":
    main(sys.argv[1:])
