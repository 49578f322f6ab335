set -euo pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp

rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS -d $OUT/p1 -o run -- python $R/scripts/prof_kernels.py --iters 1 fwd prelim miss > $OUT/p1.log 2>&1
python $R/scripts/rocpd_summary.py pmc $OUT/p1/run_results.db hf_trace > $OUT/pmc_insts.csv; rm -rf $OUT/p1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $OUT/p2 -o run -- python $R/scripts/prof_kernels.py --iters 1 fwd prelim miss > $OUT/p2.log 2>&1
python $R/scripts/rocpd_summary.py pmc $OUT/p2/run_results.db hf_trace > $OUT/pmc_sq.csv; rm -rf $OUT/p2
cat $OUT/pmc_insts.csv $OUT/pmc_sq.csv
