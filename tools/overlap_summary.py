#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace (csv) of bench.py run WITH the encoder look-ahead: three streams are in flight (two conv
stacks + the decoder), so a launch's begin-end span shares the chip with others and no longer measures the kernel alone.
Prints, over the steady-state training steps (clamp+Adam to clamp+Adam): wall per step, summed kernel spans per step by class,
mean launches in flight, and the conv launches' mean span in the overlapped steps next to the mean of bench.py's in-sequence
roofline pass (the last conv-stack run of the trace, which runs alone).
    python tools/overlap_summary.py gpurun_out/prof/bench_kernel_trace.csv"""
import csv
import sys
from collections import defaultdict


CONV_KERNELS = ('conv_glds_kernel', 'conv_xp_kernel', 'conv_pr_kernel', 'conv_stem_kernel', 'conv_pw_kernel', 'conv_aw_kernel', 'conv_ap_kernel')


def cls(n):
    if any(k in n for k in CONV_KERNELS):
        return 'conv'
    if 'bn_act_kernel' in n or 'bn_relu' in n or 'bn_strided' in n:
        return 'bn apply'
    for k in ('maxpool', 'avgpool', 'image_prep', 'bn_finalize', 'bn_slab_to_acc', 'bn_running_apply'):
        if k in n:
            return 'encoder other'
    return 'head + decoder + optimizer'


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    for r in rows:
        r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    rows.sort(key=lambda r: r['s'])
    adam = [r for r in rows if 'clamp_adam' in r['Kernel_Name']]
    if len(adam) < 18:
        raise SystemExit("need a trace with at least 18 training steps")
    # steady state: skip the warm-up steps + the timed region's pipeline fill at the front and the drain at the back (the last
    # `lookahead_depth` = 6 steps start no new conv stack: their batches ran ahead already)
    i0, i1 = 6, len(adam) - 8
    a0, a1 = adam[i0], adam[i1]
    nsteps = i1 - i0
    t0, t1 = a0['e'], a1['e']
    win = [r for r in rows if r['s'] >= t0 and r['e'] <= t1]
    wall = (t1 - t0) / 1e6 / nsteps
    busy = defaultdict(float)
    cnt = defaultdict(int)
    for r in win:
        busy[cls(r['Kernel_Name'])] += (r['e'] - r['s']) / 1e6
        cnt[cls(r['Kernel_Name'])] += 1
    tot = sum(busy.values())
    print("steady state, %d training steps: wall %.3f ms/step; summed launch spans %.3f ms/step => %.2f launches in flight on average"
          % (nsteps, wall, tot / nsteps, tot / nsteps / wall))
    for k in sorted(busy, key=lambda k: -busy[k]):
        print("  %-28s %5d launches/step  %7.3f ms/step of spans" % (k, cnt[k] // nsteps, busy[k] / nsteps))
    is_conv = lambda r: any(k in r['Kernel_Name'] for k in CONV_KERNELS)
    conv_win = [r for r in win if is_conv(r)]
    # bench.py's roofline leg: the conv launches after the last optimizer step run in sequence, alone
    tail = [r for r in rows if r['s'] > adam[-1]['e'] and is_conv(r)]
    ip = [r for r in rows if r['s'] > adam[-1]['e'] and 'image_prep' in r['Kernel_Name']]
    if ip:
        tail = [r for r in tail if r['s'] > ip[-1]['s']]
    if conv_win:
        print("conv launches: mean span %.1f us in the training steps of the look-ahead run (the profiler serialises the streams)"
              % (sum(r['e'] - r['s'] for r in conv_win) / len(conv_win) / 1e3))
    # bench.py's roofline leg proper: the in-order passes of the GROUPED program (grid.y = batches per launch) behind the last step
    grouped = [r for r in rows if r['s'] > adam[-1]['e'] and is_conv(r) and int(r.get('Grid_Size_Y', 1) or 1) > 1]
    if grouped:
        per_pass = 155
        print("conv launches of the grouped program, in sequence (bench.py's `roofline`): %d launches (%d passes), mean %.2f us, %.3f ms per pass"
              % (len(grouped), len(grouped) // per_pass, sum(r['e'] - r['s'] for r in grouped) / len(grouped) / 1e3,
                 sum(r['e'] - r['s'] for r in grouped) / max(1, len(grouped) // per_pass) / 1e6))
    if tail:
        print("conv launches: %d in the last in-sequence pass, mean %.1f us, sum %.3f ms (bench.py's `roofline.one_batch_per_launch` pass: the ungrouped program)"
              % (len(tail), sum(r['e'] - r['s'] for r in tail) / len(tail) / 1e3, sum(r['e'] - r['s'] for r in tail) / 1e6))


if __name__ == "__main__":
    main()
