#!/bin/bash
# usage (on the GPU box, via gpurun): scripts/profile_round.sh TAG [SOURCE]   (SOURCE: commit the profile is taken at)
# rocprofv3 kernel stats of the bench command + PMC traffic / instruction counters of the kernels, reduced to
# small CSV files under gpurun_out/TAG/ (the rocpd databases are deleted: gpurun copies back at most 64 MiB).
set -euo pipefail
TAG=${1:-prof}
SRC=${2:-unknown}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (the other-launch timings are switched off here so that every launch of a kernel in the table is the bench workload)
HF_BENCH_EXTRAS=0 rocprofv3 --kernel-trace --stats -d $OUT/stats -o run -- python $R/bench.py --steps 5 --warmup 2 > $OUT/bench_under_rocprof.log 2>&1
python $R/scripts/rocpd_summary.py stats_ms $OUT/stats/run_results.db > $OUT/kernel_stats.csv
grep '^{"metric"' $OUT/bench_under_rocprof.log > $OUT/bench.json
rm -rf $OUT/stats
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$C -o run -- python $R/scripts/prof_kernels.py --iters 1 fwd adj miss > $OUT/pmc_$C.log 2>&1
  python $R/scripts/rocpd_summary.py pmc $OUT/pmc_$C/run_results.db hf_ > $OUT/pmc_$C.csv
  rm -rf $OUT/pmc_$C
done
python $R/scripts/make_traffic_json.py $OUT "$SRC" > $OUT/traffic.json
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc_sq -o run -- python $R/scripts/prof_kernels.py --iters 1 fwd miss > $OUT/pmc_sq.log 2>&1
python $R/scripts/rocpd_summary.py pmc $OUT/pmc_sq/run_results.db hf_trace > $OUT/pmc_sq.csv
rm -rf $OUT/pmc_sq
# cache behaviour of the traversal gathers: L2 (TCC) hits / misses, L1 (TCP) accesses and its read requests to L2
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $OUT/pmc_tcc -o run -- python $R/scripts/prof_kernels.py --iters 1 fwd sec_fwd miss > $OUT/pmc_tcc.log 2>&1
python $R/scripts/rocpd_summary.py pmc $OUT/pmc_tcc/run_results.db hf_trace > $OUT/pmc_tcc.csv
rm -rf $OUT/pmc_tcc
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum -d $OUT/pmc_tcp -o run -- python $R/scripts/prof_kernels.py --iters 1 fwd sec_fwd miss > $OUT/pmc_tcp.log 2>&1
python $R/scripts/rocpd_summary.py pmc $OUT/pmc_tcp/run_results.db hf_trace > $OUT/pmc_tcp.csv
rm -rf $OUT/pmc_tcp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS -d $OUT/pmc_insts -o run -- python $R/scripts/prof_kernels.py --iters 1 fwd prelim miss adj > $OUT/pmc_insts.log 2>&1
python $R/scripts/rocpd_summary.py pmc $OUT/pmc_insts/run_results.db hf_ > $OUT/pmc_insts.csv
rm -rf $OUT/pmc_insts
cd $R && python bench.py --steps 20 --warmup 3 > $OUT/bench_20steps.json
ls -la $OUT
