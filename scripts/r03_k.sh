#!/bin/bash
# usage: r03_k.sh "kinds" lib...  : parity of each lib, interleaved A/B (3 reps) against v3, wstats where a _ws build exists
KINDS=$1; shift
OUT=gpurun_out/r03_k; mkdir -p $OUT; R=$GRAFT_REPO_ROOT
for L in "$@"; do
HF_LIB=$R/scratch_so/libhf_$L.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_sheared_stress.py tests/test_gpu_band.py -x -q -m gpu 2>&1 | tail -2 | tee $OUT/parity_$L.log
done
scripts/abr.sh r03_k 3 "$KINDS" base "$@"
for L in "$@"; do
if [ -f $R/scratch_so/libhf_${L}_ws.so ]; then HF_LIB=$R/scratch_so/libhf_${L}_ws.so python scripts/wstats.py 4096 1024 64 | tee $OUT/wstats_$L.txt; fi
done
