#!/bin/bash
# store flavour of conv_xp_kernel: SAT_CONV_DBG 0 = write-through, 256 = write-back, 512 = non-temporal
for d in 0 256 512 0 256 512; do
  echo "== SAT_CONV_DBG=$d"
  SAT_CONV_DBG=$d timeout -k 10 200 python tools/microbench.py xp 2>&1 | grep -v amdgpu.ids | tail -6 || exit 1
done
