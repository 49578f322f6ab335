#!/bin/bash
# usage (GPU box): scripts/prof_adj.sh TAG : SQ / TCP counters of hf_adjoint_kernel on the bench workload (separate --pmc passes)
set -euo pipefail
TAG=${1:-adj}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $OUT/p1 -o run -- python $R/scripts/prof_kernels.py --iters 1 adj > $OUT/p1.log 2>&1
python $R/scripts/rocpd_summary.py pmc $OUT/p1/run_results.db hf_adjoint > $OUT/adj_pmc_sq.csv
rm -rf $OUT/p1
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum  -d $OUT/p2 -o run -- python $R/scripts/prof_kernels.py --iters 1 adj > $OUT/p2.log 2>&1
python $R/scripts/rocpd_summary.py pmc $OUT/p2/run_results.db hf_adjoint > $OUT/adj_pmc_tcp.csv
rm -rf $OUT/p2
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum  -d $OUT/p3 -o run -- python $R/scripts/prof_kernels.py --iters 1 adj > $OUT/p3.log 2>&1
python $R/scripts/rocpd_summary.py pmc $OUT/p3/run_results.db hf_adjoint > $OUT/adj_pmc_tcc.csv
rm -rf $OUT/p3
cat $OUT/adj_pmc_sq.csv $OUT/adj_pmc_tcp.csv $OUT/adj_pmc_tcc.csv
