#!/bin/bash
set -e
OUT=gpurun_out/r03_c; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
HF_LIB=$R/scratch_so/libhf_adj1.so python -m pytest tests/test_gpu_parity.py tests/test_rectangle_ray_gradients.py tests/test_gpu_full_size.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/parity_adj1.log
scripts/ab.sh r03_c "adj fwd" main adj1
HF_LIB=$R/scratch_so/libhf_hoist6_ws.so python scripts/wstats.py 4096 1024 64 | tee $OUT/wstats_hoist6.txt
cd /tmp && export TMPDIR=/tmp
for L in main hoist5 hoist6; do
  LIB=$R/scratch_so/libhf_$L.so; [ $L = main ] && LIB=$R/mitsuba3-differentiable-heightfield-rendering_amd/libhf.so
  HF_LIB=$LIB rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $R/$OUT/p_$L -o run -- python $R/scripts/prof_kernels.py --iters 1 fwd prelim > $R/$OUT/p_$L.log 2>&1
  python $R/scripts/rocpd_summary.py pmc $R/$OUT/p_$L/run_results.db hf_trace > $R/$OUT/p_$L.csv
  rm -rf $R/$OUT/p_$L
  echo "== $L"; python3 - $R/$OUT/p_$L.csv <<'PY'
import sys,csv,collections
d=collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])): d[(r['kernel'][:40],r['dispatch'])][r['counter']]=float(r['value'])
for k,v in d.items(): print(k, ' '.join(f"{c[3:]}={x/1e6:.0f}M" for c,x in sorted(v.items())))
PY
done
