#!/bin/bash
# Builds the REFERENCE's own test programs (tests/test_*.cu, unmodified, from
# where they lie under /root/reference) against THIS repo's headers and
# libyalla_hip.so, into oracle/_ref/ (git-ignored binaries that travel to the GPU
# box).  They are the strongest drop-in check there is: ya||a's tests, our
# engine.  No reference source is copied: a scratch tree of symlinks makes the
# tests' `#include "../include/x.cuh"` resolve to include/ of this repo.
#   test_dtypes test_solvers test_links test_inits test_vtk test_mesh   -> built
#   test_polarity  stale in the reference itself             -> skipped
# and the reference's model programs (examples/*.cu, unmodified too: the four BASELINE
# configurations' and the other 19) into oracle/_ref/examples/.
set -e
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(dirname "$HERE")
[ -d "$REF/tests" ] || { echo "no reference checkout at $REF: nothing to build"; exit 0; }
OUT=$HERE/_ref
TREE=$(mktemp -d)
trap 'rm -rf "$TREE"' EXIT
mkdir -p "$OUT" "$TREE/tests"
ln -s "$ROOT/include" "$TREE/include"
for f in "$REF"/tests/*; do ln -s "$f" "$TREE/tests/"; done
cd "$TREE/tests"
for t in test_dtypes test_solvers test_links test_inits test_vtk test_mesh; do
  /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -std=c++17 -O2 -ffp-contract=off -fno-slp-vectorize \
      -Wno-error=parentheses -w -include "$ROOT/include/compat/cuda_names.h" -I"$ROOT/include/compat" \
      $t.cu -L"$ROOT/yalla_amd" -lyalla_hip -Wl,-rpath,'$ORIGIN/../../yalla_amd' -o "$OUT/$t"
  echo "built oracle/_ref/$t"
done
mkdir -p "$OUT/examples" "$TREE/examples"
for f in "$REF"/examples/*; do ln -s "$f" "$TREE/examples/"; done
cd "$TREE/examples"
build_model() {
  /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -std=c++17 -O2 -ffp-contract=off -fno-slp-vectorize \
      -Wno-error=parentheses -w -include "$ROOT/include/compat/cuda_names.h" -I"$ROOT/include/compat" \
      $1.cu -L"$ROOT/yalla_amd" -lyalla_hip -lpthread -Wl,-rpath,'$ORIGIN/../../../yalla_amd' -o "$OUT/examples/$1" \
    && echo "built oracle/_ref/examples/$1"
}
export -f build_model
export ROOT OUT
# the four BASELINE configurations' programs, then the other 19 (polarization.cu calls
# bidirectional_polarization_force with a whole point for the partner, the spelling before `Polarity`,
# include/polarity.cuh keeps that overload)
printf "%s\n" springs sorting passive_growth branching apical_constriction bending epithelia_double_polarity \
    epithelium gradient growth_w_wall intercalation lineage_tracing migration random_walk sorting_prot turing \
    turing_w_noise wnt write_vtk_w_mask teapot intercalation_w_gradient model_features_sequential_addition polarization | xargs -P 6 -I{} bash -c 'build_model {}'
[ "$(ls "$OUT/examples" | wc -l)" -ge 23 ] || { echo "some example programs failed to build"; exit 1; }
# What must be present on the GPU box: the tests FAIL (not skip) there when a listed binary is missing.
( cd "$OUT" && { ls test_* | sed 's|^|oracle/_ref/|'; ls examples/* | sed 's|^|oracle/_ref/|'; } ) > "$HERE/ref_manifest.txt"
echo "wrote oracle/ref_manifest.txt ($(wc -l < "$HERE/ref_manifest.txt") binaries)"
