#!/bin/bash
# usage: r03_v.sh OUT "kinds" base lib...  : parity of each lib, interleaved A/B (3 reps) against `base`, wstats where _ws / _ws3 builds exist
OUT=$1; KINDS=$2; BASE=$3; shift 3
mkdir -p gpurun_out/$OUT; R=$GRAFT_REPO_ROOT
for L in "$@"; do
HF_LIB=$R/scratch_so/libhf_$L.so timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_sheared_stress.py tests/test_gpu_band.py -x -q -m gpu 2>&1 | tail -2 | tee gpurun_out/$OUT/parity_$L.log
done
timeout -k 10 500 scripts/abr.sh $OUT 3 "$KINDS" $BASE "$@"
for L in "$@"; do
if [ -f $R/scratch_so/libhf_${L}_ws.so ]; then HF_LIB=$R/scratch_so/libhf_${L}_ws.so timeout -k 10 120 python scripts/wstats.py 4096 1024 64 | tee gpurun_out/$OUT/wstats_$L.txt; fi
if [ -f $R/scratch_so/libhf_${L}_ws3.so ]; then HF_LIB=$R/scratch_so/libhf_${L}_ws3.so timeout -k 10 120 python scripts/wstats3.py 4096 1024 64 | tee gpurun_out/$OUT/wstats3_$L.txt; fi
done
