#!/bin/sh
# oracle/build_ref.sh — the reference-side parity pin: builds oracle/ref_harness against the real crate (default
# /root/reference, override with WAA_REFERENCE_DIR) and renders the cases of tests/test_reference_dumps.py into
# oracle/_ref/dumps/ (git-ignored; travels to the GPU box with the snapshot).  Needs cargo + the crate's dependencies in
# the cargo registry.  Neither exists in the authoring container or on the GPU boxes (no Rust toolchain, no network,
# no Cargo.lock in the reference): the script then says so and exits 0 — building the checker is optional, and the
# tests that consume the dumps skip while they are absent.  On a machine with cargo:   sh oracle/build_ref.sh
# Stage 2 (also needs libwaa_hip.so and an MI355X): the Rust shim of shim/ built into a patched copy of the crate and the
# BASELINE graphs rendered through both the crate's CPU path and the shim, compared (shim/README.md).  UNBUILDABLE HERE.
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
REF=${WAA_REFERENCE_DIR:-/root/reference}
if ! command -v cargo >/dev/null 2>&1; then
  echo "build_ref.sh: cargo not found - the reference crate cannot be built here; parity stays pinned by the re-typed reference tests and golden vectors (tests/test_reference_kat.py), see DESIGN.md section 4"
  exit 0
fi
if [ ! -f "$REF/Cargo.toml" ]; then
  echo "build_ref.sh: no reference crate at $REF"
  exit 0
fi
mkdir -p "$HERE/_ref/inputs" "$HERE/_ref/dumps" "$HERE/_ref/target"
python3 "$HERE/../tools/ref_inputs.py" "$HERE/_ref/inputs"
# the harness is built from a copy of its manifest that points at $REF (sources stay where they lie)
WORK="$HERE/_ref/harness"
rm -rf "$WORK" && mkdir -p "$WORK/src"
sed "s#/root/reference#$REF#" "$HERE/ref_harness/Cargo.toml" > "$WORK/Cargo.toml"
cp "$HERE/ref_harness/src/main.rs" "$WORK/src/main.rs"
if ! CARGO_TARGET_DIR="$HERE/_ref/target" cargo build --release --manifest-path "$WORK/Cargo.toml"; then
  echo "build_ref.sh: cargo build failed (dependencies not in the registry / no network?) - no dumps written"
  exit 0
fi
"$HERE/_ref/target/release/waa-ref-harness" "$HERE/_ref/inputs" "$HERE/_ref/dumps" "$REF"
echo "build_ref.sh: dumps in $HERE/_ref/dumps (tests/test_reference_dumps.py picks them up)"

# ---- stage 2: the shim (shim/README.md).  A patched COPY of the crate (the reference tree itself is never written to),
# shim/harness built against it with the `hip` feature, every BASELINE graph rendered through both paths and compared.
LIBDIR=${WAA_HIP_LIB_DIR:-$HERE/../web-audio-api-rs_amd/csrc}
if [ ! -f "$LIBDIR/libwaa_hip.so" ]; then
  echo "build_ref.sh: no libwaa_hip.so in $LIBDIR - shim stage skipped (python __graft_entry__.py builds it)"
  exit 0
fi
CRATE="$HERE/_ref/crate"
rm -rf "$CRATE" && mkdir -p "$CRATE"
(cd "$REF" && tar cf - --exclude target --exclude .git .) | (cd "$CRATE" && tar xf -)
if ! (cd "$CRATE" && patch -p1 < "$HERE/../shim/reference.patch"); then
  echo "build_ref.sh: shim/reference.patch does not apply to $REF (another crate version?)"
  exit 1
fi
CHECK="$HERE/_ref/shim_check"
rm -rf "$CHECK" && mkdir -p "$CHECK/src"
sed "s#WAA_PATCHED_CRATE#$CRATE#" "$HERE/../shim/harness/Cargo.toml" > "$CHECK/Cargo.toml"
cp "$HERE/../shim/harness/src/main.rs" "$CHECK/src/main.rs"
if ! WAA_HIP_LIB_DIR="$LIBDIR" CARGO_TARGET_DIR="$HERE/_ref/target" cargo build --release --manifest-path "$CHECK/Cargo.toml"; then
  echo "build_ref.sh: the patched crate / shim did not build - see the compiler output above (the shim has never been compiled by its author: shim/README.md)"
  exit 1
fi
LD_LIBRARY_PATH="$LIBDIR:${LD_LIBRARY_PATH:-}" "$HERE/_ref/target/release/waa-shim-check" "$HERE/_ref/inputs" "$REF" "${WAA_DEVICE:-0}" | tee "$HERE/_ref/shim_check.txt"
