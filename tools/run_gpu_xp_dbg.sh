#!/bin/bash
# phase costs of conv_xp_kernel: SAT_CONV_DBG 16 = A panel from one line per row, 32 = no weight traffic, 64 = no output
for d in 0 16 32 64 48 112; do
  echo "== SAT_CONV_DBG=$d"
  SAT_CONV_DBG=$d timeout -k 10 200 python tools/microbench.py xp 2>&1 | grep -v amdgpu.ids | sed -n '1,7p' || exit 1
done
