#!/bin/bash
# usage (GPU box): [KIND=sec_fwd_inc] scripts/pmc_sec.sh OUT variant...   SQ counters of the bounce-ray launch (prof_kernels.py $KIND, default sec_fwd) per variant
# (variant: main = the in-tree library, NAME = scratch_so/libhf_NAME.so); counters in their own passes, --kernel-trace only
set -euo pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; shift; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = main ]; then unset HF_LIB; else export HF_LIB=$R/scratch_so/libhf_$v.so; fi
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS -d $OUT/p1 -o run -- python $R/scripts/prof_kernels.py --iters 1 ${KIND:-sec_fwd} > $OUT/p1_$v.log 2>&1
  python $R/scripts/rocpd_summary.py pmc $OUT/p1/run_results.db hf_trace > $OUT/insts_$v.csv; rm -rf $OUT/p1
  rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM -d $OUT/p2 -o run -- python $R/scripts/prof_kernels.py --iters 1 ${KIND:-sec_fwd} > $OUT/p2_$v.log 2>&1
  python $R/scripts/rocpd_summary.py pmc $OUT/p2/run_results.db hf_trace > $OUT/sq_$v.csv; rm -rf $OUT/p2
  echo "== $v"; tail -n 14 $OUT/insts_$v.csv; tail -n 16 $OUT/sq_$v.csv
done
