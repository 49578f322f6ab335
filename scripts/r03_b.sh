#!/bin/bash
# round-3 batch B (GPU box): new GPU tests + TA / TCP / TD counters of the forward launch
set -e
OUT=gpurun_out/r03_b; mkdir -p $OUT
python -m pytest tests/test_gpu_band.py tests/test_graph_capture.py tests/test_packet_and_params.py -x -q -m gpu 2>&1 | tail -5 | tee $OUT/tests.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
pass() { # name counters...
  local nm=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $R/$OUT/p_$nm -o run -- python $R/scripts/prof_kernels.py --iters 1 fwd sec_fwd miss > $R/$OUT/p_$nm.log 2>&1 || { echo "pass $nm failed"; tail -3 $R/$OUT/p_$nm.log; return 0; }
  python $R/scripts/rocpd_summary.py pmc $R/$OUT/p_$nm/run_results.db hf_trace > $R/$OUT/p_$nm.csv
  rm -rf $R/$OUT/p_$nm
}
pass busy GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_max TCP_GATE_EN1_sum TD_TD_BUSY_sum
pass stall TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
pass lat TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum
pass wf TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
pass sq SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass sq2 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SMEM
cat $R/$OUT/p_*.csv
