#!/bin/bash
# helper for gpurun: beam-5 decode loop alone (tools/decode_trace.py) by the number of K slices of the wide LSTM step's gate GEMM (SAT_BEAM_KSPLIT)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
for rep in 1 2; do
for ks in ${KS:-1 2 3 4 6 8}; do
  echo -n "LSTM-step K slices $ks, rep $rep: "
  SAT_BEAM_KSPLIT=$ks timeout -k 10 120 python tools/decode_trace.py 5 40 2>/dev/null | tail -1
done
done
