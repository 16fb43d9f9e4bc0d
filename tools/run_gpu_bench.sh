#!/bin/bash
# helper for gpurun: bench line + rocprofv3 kernel trace of the same command
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
cd $R
# one variant table for the bench line and for both traces: the autotuner's timing runs are distorted under the profiler (it
# serialises the streams and adds per-dispatch overhead), so the profiled commands load the table the un-profiled run saved
export SAT_TUNE_FILE=/tmp/sat_tune_bench.json
rm -f $SAT_TUNE_FILE
timeout -k 10 600 python bench.py --steps ${STEPS:-20} --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err
rc=$?
cat gpurun_out/bench.json; tail -n 5 gpurun_out/bench.err
if [ $rc -ne 0 ]; then echo "bench failed rc=$rc"; exit $rc; fi
# ... and the strictly sequential schedule's own entries (there the ungrouped program leads the kernel choice)
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-mode --no-lookahead --repeats 1 > gpurun_out/bench_seq.json 2> gpurun_out/bench_seq.err || tail -3 gpurun_out/bench_seq.err
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-mode --repeats 1 > $R/gpurun_out/prof_run.log 2>&1
rc=$?
tail -n 3 $R/gpurun_out/prof_run.log
# the same command with strictly sequential steps (--no-lookahead): per-launch spans there are the kernels alone
rm -rf $R/gpurun_out/prof_seq
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_seq -o bench -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-mode --no-lookahead --repeats 1 > $R/gpurun_out/prof_seq_run.log 2>&1
tail -n 1 $R/gpurun_out/prof_seq_run.log | cut -c1-200
ft=$(find $R/gpurun_out/prof -name "*kernel_trace.csv" | head -1)
[ -n "$ft" ] && python3 $R/tools/overlap_summary.py "$ft" > $R/gpurun_out/overlap_summary.txt 2>&1; cat $R/gpurun_out/overlap_summary.txt
fs=$(find $R/gpurun_out/prof_seq -name "*kernel_trace.csv" | head -1)
[ -n "$fs" ] && python3 $R/tools/step_breakdown.py "$fs" 40 > $R/gpurun_out/one_step_breakdown.txt 2>&1; head -n 12 $R/gpurun_out/one_step_breakdown.txt
find $R/gpurun_out/prof_seq -name "*kernel_trace.csv" -size +20M -delete
find $R/gpurun_out/prof -name "*stats*" | head
f=$(find $R/gpurun_out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -n 12 "$f"
# keep the big trace out of the merge-back budget
find $R/gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
exit $rc
