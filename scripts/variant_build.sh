#!/bin/bash
# usage: scripts/variant_build.sh NAME 'sed-expression' [extra hipcc flags]   -> scratch_so/libhf_NAME.so (never shipped)
# The variant's source is kept: its diff against the in-tree kernel file goes to profiles/variants/NAME.diff (tracked,
# so that a log of an A/B -- or of a fault -- can always be read next to the code that produced it).
set -euo pipefail
cd "$(dirname "$0")/../mitsuba3-differentiable-heightfield-rendering_amd"
name=$1; expr=$2; shift 2
mkdir -p ../scratch_so ../profiles/variants
sed "$expr" csrc/hf_kernels.hip > csrc/_variant_$name.hip
diff -u csrc/hf_kernels.hip csrc/_variant_$name.hip > ../profiles/variants/$name.diff || true
{ echo "# flags: $*"; echo "# base: $(git rev-parse --short HEAD)"; } >> ../profiles/variants/$name.diff
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math \
  -Wno-bitwise-instead-of-logical -Wno-unused-function -fno-slp-vectorize "$@" -I ../include csrc/_variant_$name.hip csrc/hf_capi.cpp -ldl \
  -o ../scratch_so/libhf_$name.so
rm -f csrc/_variant_$name.hip
