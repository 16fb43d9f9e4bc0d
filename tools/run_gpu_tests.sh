#!/bin/bash
# helper for gpurun: kernel tests, then parity tests; stop if a step was killed by its timeout
mkdir -p gpurun_out
timeout -k 10 420 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 120 -p no:cacheprovider > gpurun_out/kernels.log 2>&1
rc=$?
tail -n 15 gpurun_out/kernels.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "kernel tests timed out"; exit $rc; fi
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dp.py -m gpu -q --timeout 600 --durations=8 -p no:cacheprovider > gpurun_out/parity.log 2>&1
rc2=$?
tail -n 30 gpurun_out/parity.log
if [ $rc2 -eq 124 ] || [ $rc2 -eq 137 ]; then echo "parity tests timed out"; exit $rc2; fi
timeout -k 10 1100 python -m pytest tests/test_gpu_parity_full.py -m gpu -q -s --timeout 1000 --durations=8 -p no:cacheprovider > gpurun_out/parity_full.log 2>&1
rc3=$?
grep -E "cfg1|cfg2|train pass|eval:|head output|passed|failed|Error" gpurun_out/parity_full.log | tail -n 40
exit $(( rc + rc2 + rc3 ))
