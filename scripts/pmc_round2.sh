#!/bin/bash
# usage (on the GPU box, via gpurun): scripts/pmc_round2.sh TAG [kinds...]
# SQ issue/stall counters of the trace kernel on the bench wavefront, reduced to CSV.
set -euo pipefail
TAG=${1:-pmc}; shift || true
KINDS=${@:-fwd miss}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAVES" \
           "SQ_BUSY_CYCLES SQ_CYCLES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $SET -d $OUT/p$i -o run -- python $R/scripts/prof_kernels.py --iters 1 $KINDS > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; continue; }
  python $R/scripts/rocpd_summary.py pmc $OUT/p$i/run_results.db hf_trace > $OUT/pmc_sq_$i.csv
  rm -rf $OUT/p$i
done
ls -la $OUT
