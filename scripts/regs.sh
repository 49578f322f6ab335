#!/bin/bash
# usage: scripts/regs.sh lib.so [name filter]   VGPRs / spills / LDS of the kernels in a built library
# (compiles nothing: reads the code object's metadata notes)
set -e
so=$(readlink -f "$1"); pat=${2:-.}
tmp=$(mktemp -d); cd $tmp
/opt/rocm/bin/roc-obj-ls "$so" | awk '/gfx950/{print $NF}' | head -1 > uri
/opt/rocm/bin/roc-obj-extract -o co "$(cat uri)" >/dev/null 2>&1 || true
f=$(ls co* 2>/dev/null | head -1)
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$f" | awk '
  /\.group_segment_fixed_size:/ {lds=$2} /\.name:/ {name=$2} /\.sgpr_spill_count:/ {ss=$2} /\.vgpr_count:/ {v=$2}
  /\.vgpr_spill_count:/ {vs=$2; print name, "vgpr", v, "vgpr_spill", vs, "sgpr_spill", ss, "lds", lds}' | grep -E "$pat" | sort
rm -rf $tmp
