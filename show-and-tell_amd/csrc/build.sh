#!/bin/bash
# Build libsat_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
OUT=../libsat_hip.so
FLAGS="${SAT_EXTRA_FLAGS} --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
mkdir -p build
pids=()
for f in sat_gemm sat_gemm_x3 sat_gemm_bf16 sat_conv_glds sat_lstm_persist sat_skinny sat_elementwise sat_attend sat_beam sat_gram sat_host; do
  stale=0
  [ -f build/$f.o ] || stale=1
  for d in $f.hip sat_common.h sat_internal.h ../../include/sat_hip.h sat_conv_*.inc; do [ $d -nt build/$f.o ] && stale=1; done
  if [ $stale = 1 ]; then
    hipcc $FLAGS -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
# the TEST build of the library: the same objects, sat_lstm_persist.hip once more with -DSAT_TESTHOOKS (fault injection for the
# status-word tests: SAT_LSTM_DEBUG_STALL).  The product library above does not contain the hook.
if [ ! -f build/sat_lstm_persist_testhooks.o ] || [ sat_lstm_persist.hip -nt build/sat_lstm_persist_testhooks.o ] || [ sat_common.h -nt build/sat_lstm_persist_testhooks.o ] || [ sat_internal.h -nt build/sat_lstm_persist_testhooks.o ] || [ ../../include/sat_hip.h -nt build/sat_lstm_persist_testhooks.o ]; then
  hipcc $FLAGS -DSAT_TESTHOOKS -c sat_lstm_persist.hip -o build/sat_lstm_persist_testhooks.o
fi
mkdir -p ../../tests/_build
hipcc --offload-arch=gfx950 -shared -fPIC build/sat_gemm.o build/sat_gemm_x3.o build/sat_gemm_bf16.o build/sat_conv_glds.o build/sat_lstm_persist_testhooks.o build/sat_skinny.o build/sat_elementwise.o build/sat_attend.o build/sat_beam.o build/sat_gram.o build/sat_host.o -o ../../tests/_build/libsat_hip_testhooks.so
hipcc --offload-arch=gfx950 -shared -fPIC build/sat_gemm.o build/sat_gemm_x3.o build/sat_gemm_bf16.o build/sat_conv_glds.o build/sat_lstm_persist.o build/sat_skinny.o build/sat_elementwise.o build/sat_attend.o build/sat_beam.o build/sat_gram.o build/sat_host.o -o $OUT
echo "built $OUT"
