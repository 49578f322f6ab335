#!/bin/bash
# counters of the all-miss launch for library variants: usage r03_h.sh lib...
OUT=gpurun_out/r03_h; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  LIB=$R/scratch_so/libhf_$L.so; [ $L = main ] && LIB=$R/mitsuba3-differentiable-heightfield-rendering_amd/libhf.so
  HF_LIB=$LIB rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES -d $R/$OUT/p_$L -o run -- python $R/scripts/prof_kernels.py --iters 1 miss > $R/$OUT/p_$L.log 2>&1
  python $R/scripts/rocpd_summary.py pmc $R/$OUT/p_$L/run_results.db hf_trace > $R/$OUT/p_$L.csv
  rm -rf $R/$OUT/p_$L
  echo "== $L"; python3 - $R/$OUT/p_$L.csv <<'PY'
import sys,csv,collections
d=collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])): d[(r['kernel'][22:30],r['dispatch'])][r['counter']]=float(r['value'])
for k,v in list(d.items())[-2:]:
    print(k, ' '.join(f"{c[3:]}={x/1e6:.1f}M" for c,x in sorted(v.items())))
PY
done
