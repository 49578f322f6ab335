#!/bin/bash
# usage (GPU box): scripts/fuzz_round.sh OUT seed0 [first [count]]   ten fuzz_parity.py runs (about 5e8 rays: hierarchical and band
# checkers), one log each under gpurun_out/OUT; first / count: a part of the ten (a gpurun call is limited to 20 minutes)
set -uo pipefail
OUT=$1; S=$2; FIRST=${3:-0}; COUNT=${4:-10}; mkdir -p gpurun_out/$OUT
i=0
for cfg in "hier_si FUZZ_SI=1 3000 400 250000" "band FUZZ_BAND=1 3000 400 250000" "hier_small FUZZ_X=0 700 1500 60000" "band_small FUZZ_BAND=1 700 1500 60000" \
           "hier_tiny FUZZ_X=0 40 3000 30000" "band_tiny FUZZ_BAND=1 40 3000 30000" "hier_si2 FUZZ_SI=1 3000 400 250000" "band2 FUZZ_BAND=1 3000 400 250000" \
           "hier_mid FUZZ_X=0 1500 800 120000" "band_mid FUZZ_BAND=1 1500 800 120000"; do
  set -- $cfg; seed=$((S + i)); i=$((i + 1))
  if [ $((i - 1)) -lt $FIRST ] || [ $((i - 1)) -ge $((FIRST + COUNT)) ]; then continue; fi
  rc=0
  env $2 FUZZ_MAXDIM=$3 timeout -k 10 400 python tests/tools/fuzz_parity.py $4 $5 $seed > gpurun_out/$OUT/$1_$seed.log 2>&1 || rc=$?
  echo rc=$rc >> gpurun_out/$OUT/$1_$seed.log
  echo "$1 seed $seed: $(grep 'scenes x' gpurun_out/$OUT/$1_$seed.log | tail -1 || true) $(tail -1 gpurun_out/$OUT/$1_$seed.log)"
  # a run that timed out, aborted or faulted has told us something: no further GPU step in this call
  case $rc in 124|134|137|139) echo "stopping after rc=$rc"; exit $rc;; esac
done
