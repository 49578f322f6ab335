#!/bin/bash
# same-box comparison of bench.py (20 steps) between library builds: usage r03_bench_ab.sh lib... (main = in-tree)
OUT=gpurun_out/r03_bench_ab; mkdir -p $OUT
for rep in 1 2; do for L in "$@"; do
  LIB=$PWD/scratch_so/libhf_$L.so; [ $L = main ] && LIB=
  HF_BENCH_EXTRAS=0 HF_LIB=$LIB python bench.py --steps 20 --warmup 3 > $OUT/${L}_$rep.json
  python3 -c "
import json,sys; d=json.load(open('$OUT/${L}_$rep.json')); r=d['roofline']; print('$L', $rep, 'value', d['value'], 'ms/step', d['ms_per_step'], 'fwd', r['fwd_ms'], 'adj', r['adj_ms'], 'frac', r['fwd_frac'])"
done; done
