#!/bin/bash
# helper for gpurun: layer-3 conv geometries with parts of the kernel switched off (SAT_CONV_DBG: 1 no loads, 2 no MFMA, 4 no epilogue)
for v in 16 3; do
  for d in 0 4 1 2 5 6 7; do
    echo "variant $v dbg $d"; SAT_VARIANT=$v SAT_CONV_DBG=$d python tools/microbench.py conv 2>&1 | grep "^conv"
  done
done
