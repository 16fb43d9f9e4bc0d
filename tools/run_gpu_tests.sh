#!/bin/bash
# helper for gpurun: kernel tests, then parity tests; stop if a step was killed by its timeout
mkdir -p gpurun_out
timeout -k 10 420 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 120 -p no:cacheprovider > gpurun_out/kernels.log 2>&1
rc=$?
tail -n 40 gpurun_out/kernels.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "kernel tests timed out"; exit $rc; fi
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 --durations=12 -p no:cacheprovider > gpurun_out/parity.log 2>&1
rc2=$?
tail -n 60 gpurun_out/parity.log
exit $(( rc + rc2 ))
