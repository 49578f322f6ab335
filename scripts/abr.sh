#!/bin/bash
# usage (GPU box): scripts/abr.sh OUTDIR REPS "kinds" lib1 lib2 ...   interleaved repetitions, median per (lib, kind)
OUT=$1; REPS=$2; KINDS=$3; shift 3
mkdir -p gpurun_out/$OUT
for rep in $(seq 1 $REPS); do for L in "$@"; do
  LIB=$PWD/scratch_so/libhf_$L.so; [ "$L" = main ] && LIB=
  HF_LIB=$LIB python scripts/prof_kernels.py --iters 10 $KINDS 2>&1 | grep -v amdgpu.ids | sed "s/^/$L /" >> gpurun_out/$OUT/all.log
done; done
python3 - gpurun_out/$OUT/all.log <<'PY'
import sys,collections,statistics
d=collections.defaultdict(list)
for l in open(sys.argv[1]):
    p=l.split()
    if len(p)>=4 and p[3]=='ms': d[(p[0],p[1])].append(float(p[2]))
libs=[]; kinds=[]
for (L,k) in d:
    if L not in libs: libs.append(L)
    if k not in kinds: kinds.append(k)
print('lib'.ljust(10)+''.join(k.rjust(22) for k in kinds))
for L in libs:
    print(L.ljust(10)+''.join((f"{statistics.median(d[(L,k)]):.3f} [{min(d[(L,k)]):.3f}-{max(d[(L,k)]):.3f}]").rjust(22) for k in kinds))
PY
