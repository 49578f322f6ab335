#!/bin/bash
# usage (GPU box): scripts/final_extras.sh TAG   -- the round's other records next to profile_round.sh's: rocprof kernel stats of
# every hot-path launch and of the reparameterisation backward, wave statistics and cycle stamps of the diagnostic builds
# (scratch_so/libhf_fin_ws.so, _ws4, _ts: scripts/vb.sh fin_ws -DHF_WSTATS=1 ...), per-kind HIP-event timings
set -euo pipefail
TAG=${1:-extras}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python scripts/prof_kernels.py --iters 10 fwd prelim test si adj miss mips sec_fwd sec_test fwd_ymajor fwd_xmajor fwd_steep sec_fwd_inc sec_test_inc 2>&1 | grep -v amdgpu.ids | tee $OUT/kinds.txt
python scripts/prof_kernels.py --iters 3 reparam 2>&1 | grep -v amdgpu.ids | tee -a $OUT/kinds.txt
python scripts/prof_kernels.py --iters 2 reparam16 2>&1 | grep -v amdgpu.ids | tee -a $OUT/kinds.txt
python scripts/prof_kernels.py --iters 10 --grid 2048 --film 512 --spp 16 fwd prelim 2>&1 | grep -v amdgpu.ids | sed 's/^/configs2-size /' | tee -a $OUT/kinds.txt
for m in 1 4; do
  suf=_ws; [ $m = 4 ] && suf=_ws4
  HF_LIB=$R/scratch_so/libhf_fin$suf.so python scripts/wstats.py --mode $m 4096 1024 64 2>&1 | grep -v amdgpu.ids | tee $OUT/wave_stats_mode$m.txt
done
HF_LIB=$R/scratch_so/libhf_fin_ts.so python scripts/tstats.py 4096 1024 64 2>&1 | grep -v amdgpu.ids | tee $OUT/tstats_prelim.txt
HF_LIB=$R/scratch_so/libhf_fin_ts.so python scripts/tstats.py 4096 1024 64 fused 2>&1 | grep -v amdgpu.ids | tee $OUT/tstats_fused.txt
bash scripts/prof_kinds.sh $TAG/kinds_rocprof
bash scripts/prof_reparam.sh $TAG/reparam_rocprof
