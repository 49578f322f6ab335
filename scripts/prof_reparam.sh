#!/bin/bash
# usage (GPU box): scripts/prof_reparam.sh TAG : rocprofv3 kernel stats of the reparameterisation backward (4 auxiliary rays, 67.1 M rays)
set -euo pipefail
TAG=${1:-reparam}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o run -- python $R/scripts/prof_kernels.py --iters 3 reparam > $OUT/reparam_under_rocprof.log 2>&1
python $R/scripts/rocpd_summary.py stats_ms $OUT/stats/run_results.db > $OUT/reparam_kernel_stats.csv
rm -rf $OUT/stats
cat $OUT/reparam_kernel_stats.csv
