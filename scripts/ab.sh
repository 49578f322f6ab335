#!/bin/bash
# usage (GPU box): scripts/ab.sh OUTDIR "kinds" lib1 lib2 ...   ("main" = the in-tree libhf.so)
# A/B timing of kernel variants built into scratch_so/ on the bench workload.
OUT=$1; KINDS=$2; shift 2
mkdir -p gpurun_out/$OUT
for L in "$@"; do
  echo "== $L"
  if [ "$L" = main ]; then python scripts/prof_kernels.py --iters 10 $KINDS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/$OUT/$L.log
  else HF_LIB=$PWD/scratch_so/libhf_$L.so python scripts/prof_kernels.py --iters 10 $KINDS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/$OUT/$L.log; fi
done
