#!/bin/bash
# usage: scripts/vb.sh NAME [extra hipcc flags, e.g. -DHF_HOIST=1]   -> scratch_so/libhf_NAME.so (never shipped)
# builds the in-tree sources as they are with extra -D flags (variant_build.sh additionally applies a sed expression);
# the flags are recorded in profiles/variants/NAME.flags
set -euo pipefail
cd "$(dirname "$0")/../mitsuba3-differentiable-heightfield-rendering_amd"
name=$1; shift
mkdir -p ../scratch_so ../profiles/variants
echo "base $(git rev-parse --short HEAD)$(git diff --quiet -- csrc || echo '+dirty') flags: $*" > ../profiles/variants/$name.flags
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math \
  -Wno-bitwise-instead-of-logical -Wno-unused-function -fno-slp-vectorize "$@" -I ../include csrc/hf_kernels.hip csrc/hf_capi.cpp -ldl \
  -o ../scratch_so/libhf_$name.so
