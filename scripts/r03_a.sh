#!/bin/bash
# round-3 batch A (GPU box): parity of the variants, A/B timing, counter list, L2 hit counters
set -e
OUT=gpurun_out/r03_a; mkdir -p $OUT
for L in hoist5 hoist6 hoist6s chunk16 one16; do
  echo "== parity $L"; HF_LIB=$PWD/scratch_so/libhf_$L.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_sheared_stress.py -x -q -m gpu 2>&1 | tail -2 | tee $OUT/parity_$L.log
done
scripts/ab.sh r03_a "fwd prelim sec_fwd" main hoist5 hoist6 hoist6s chunk8 chunk16 chunk32 chunk64 one1 one16
for L in main_ws hoist5_ws; do HF_LIB=$PWD/scratch_so/libhf_$L.so python scripts/wstats.py 4096 1024 64 | tee $OUT/wstats_$L.txt; done
rocprofv3 -L > $OUT/counters.txt 2>&1 || true
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $R/$OUT/pmc_tcc -o run -- python $R/scripts/prof_kernels.py --iters 1 fwd sec_fwd miss > $R/$OUT/pmc_tcc.log 2>&1
python $R/scripts/rocpd_summary.py pmc $R/$OUT/pmc_tcc/run_results.db hf_trace > $R/$OUT/pmc_tcc.csv
rm -rf $R/$OUT/pmc_tcc
cat $R/$OUT/pmc_tcc.csv
