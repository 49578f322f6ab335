#!/bin/bash
# usage (GPU box): scripts/prof_kinds.sh TAG : rocprofv3 kernel stats of single launches of every hot-path entry on the bench workload
set -euo pipefail
TAG=${1:-kinds}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o run -- python $R/scripts/prof_kernels.py --iters 10 prelim test si fwd adj sec_fwd sec_test sec_fwd_inc > $OUT/kinds_under_rocprof.log 2>&1
python $R/scripts/rocpd_summary.py stats_ms $OUT/stats/run_results.db > $OUT/kinds_kernel_stats.csv
rm -rf $OUT/stats
grep "hf_" $OUT/kinds_kernel_stats.csv | cut -c1-140
