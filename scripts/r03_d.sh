#!/bin/bash
set -e
OUT=gpurun_out/r03_d; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
for L in "$@"; do
HF_LIB=$R/scratch_so/libhf_$L.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_sheared_stress.py tests/test_gpu_band.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/parity_$L.log
done
scripts/ab.sh r03_d "${KINDS:-fwd prelim test sec_fwd fwd_ymajor fwd_xmajor fwd_steep}" main "$@" main
for L in "$@"; do
if [ -f $R/scratch_so/libhf_${L}_ws.so ]; then HF_LIB=$R/scratch_so/libhf_${L}_ws.so python scripts/wstats.py 4096 1024 64 | tee $OUT/wstats_$L.txt; fi
done
