#!/bin/sh
# oracle/build_ref.sh — the reference-side parity pin: builds oracle/ref_harness against the real crate (default
# /root/reference, override with WAA_REFERENCE_DIR) and renders the cases of tests/test_reference_dumps.py into
# oracle/_ref/dumps/ (git-ignored; travels to the GPU box with the snapshot).  Needs cargo + the crate's dependencies in
# the cargo registry.  Neither exists in the authoring container or on the GPU boxes (no Rust toolchain, no network,
# no Cargo.lock in the reference): the script then says so and exits 0 — building the checker is optional, and the
# tests that consume the dumps skip while they are absent.  On a machine with cargo:   sh oracle/build_ref.sh
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
