#!/bin/bash
# helper for gpurun: beam-5 decode loop alone (tools/decode_trace.py) under the experiment knobs of the f32 GEMM tile / the wide LSTM step's K split
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
for rep in 1 2; do
for cfg in ${CFGS:-"0 1" "0 3" "0 4" "0 6" "0 8"}; do
  set -- $cfg
  echo -n "f32 tile $1, LSTM-step K split $2, rep $rep: "
  SAT_GEMM_F32_TILE=$1 SAT_BEAM_KSPLIT=$2 timeout -k 10 120 python tools/decode_trace.py 5 40 2>/dev/null | tail -1
done
done
