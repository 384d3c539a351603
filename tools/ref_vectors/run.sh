#!/bin/bash
# run.sh <reference checkout> [output.json]: builds the dumper INSIDE the reference's workspace (so that every dependency resolves
# through the reference's own Cargo.toml / Cargo.lock pins) and writes the vectors. Needs the reference's toolchain
# (rust-toolchain: nightly-2025-05-22) and network access for its git dependencies -- neither exists in the builder's image.
set -euo pipefail
REF=${1:?usage: run.sh <reference checkout> [output.json]}
OUT=${2:-$(cd "$(dirname "$0")/../.." && pwd)/tests/golden/reference_vectors.json}
HERE=$(cd "$(dirname "$0")" && pwd)
rm -rf "$REF/ref_vectors" && mkdir -p "$REF/ref_vectors"
cp -r "$HERE/Cargo.toml" "$HERE/src" "$REF/ref_vectors/"
grep -q '"ref_vectors"' "$REF/Cargo.toml" || sed -i 's/^members = \[/members = [\n  "ref_vectors",/' "$REF/Cargo.toml"
( cd "$REF" && cargo run --release -p ref_vectors -- "$OUT" )
# the other configuration of the reference (its `original_poseidon` feature): only `default_hasher`, `identifier_block_column`,
# `polynomial_batch`, `challenger` and `proof` depend on it
( cd "$REF" && cargo run --release -p ref_vectors --features original_poseidon -- "${OUT%.json}_poseidon.json" )
echo "now: python -m pytest tests/test_reference_vectors.py -q   (and -m gpu on a GPU box)"
