#!/usr/bin/env python3
"""Cut the last full training step out of a rocprofv3 kernel trace (csv) and print it grouped by (kernel, grid).
    python tools/step_breakdown.py gpurun_out/prof/bench_kernel_trace.csv [top_n]"""
import csv
import re
import sys
from collections import defaultdict


def short(n):
    m = re.search(r'conv_glds_kernel<([^>]*)>', n)
    if m:
        return 'conv<' + m.group(1).replace(' ', '') + '>'
    m = re.search(r'conv_xp_kernel<([^>]*)>', n)       # register-resident A panel (expansion 1x1 convs)
    if m:
        return 'conv<xp:' + m.group(1).replace(' ', '') + '>'
    m = re.search(r'conv_pr_kernel<([^>]*)>', n)       # LDS-resident input patch (3x3 convs)
    if m:
        return 'conv<pr:' + m.group(1).replace(' ', '') + '>'
    if 'conv_stem_kernel' in n:                         # persistent stem kernel (sat_conv_stem.inc)
        return 'conv<stem>'
    m = re.search(r'conv_pw_kernel<([^>]*)>', n)       # LDS-resident input patch, weights straight into registers (3x3 convs)
    if m:
        return 'conv<pw:' + m.group(1).replace(' ', '') + '>'
    m = re.search(r'conv_aw_kernel<([^>]*)>', n)       # 1x1 convs, weights straight into registers
    if m:
        return 'conv<aw:' + m.group(1).replace(' ', '') + '>'
    m = re.search(r'conv_ap_kernel<([^>]*)>', n)       # expansion 1x1 convs, weights resident in registers, persistent over row tiles
    if m:
        return 'conv<ap:' + m.group(1).replace(' ', '') + '>'
    for k in ('gram_kernel', 'gram_cov', 'bn_from_gram', 'gemm_bf16_nt'):      # bn3 from the Gram matrix of conv3's input (sat_gram.hip)
        if k in n:
            return k
    if 'bn_act_kernel' in n:
        return 'bn_add' if ('_Accum, bool' in n or 'Lb1' in n) else 'bn_relu'
    for k in ['bn_finalize', 'maxpool', 'avgpool', 'image_prep', 'lstm_persist', 'lstm_bwd_step', 'skinny', 'gemm_kernel', 'lstm_bwd_point', 'ce_rows',
              'validate_ids', 'bn_slab_to_acc',
              'clamp_adam', 'embed', 'colsum', 'bn1d', 'sum_slabs', 'pack_targets', 'sum_scale', 'beam']:
        if k in n:
            return k
    return n[:30]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    ip = [i for i, r in enumerate(rows) if 'image_prep' in r['Kernel_Name']]
    adam = [i for i, r in enumerate(rows) if 'clamp_adam' in r['Kernel_Name']]
    # the last FULL training step: the last clamp+Adam launch and the image prep that opened its step (bench.py's
    # roofline passes run the encoder again afterwards, without an optimizer step)
    last = adam[-1]
    s = [i for i in ip if i < last][-1]
    e = [last]
    step = rows[s:e[0] + 1]
    wall = (int(step[-1]['End_Timestamp']) - int(step[0]['Start_Timestamp'])) / 1e6
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step) / 1e6
    print("one training step: %d launches, wall %.3f ms, sum of kernel durations %.3f ms" % (len(step), wall, busy))
    agg = defaultdict(lambda: [0, 0.0])
    for r in step:
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        k = (short(r['Kernel_Name']), int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])))
        agg[k][0] += 1
        agg[k][1] += d
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print("%-28s grid %6d  n=%3d  avg %7.1f us  total %7.3f ms" % (k[0], k[1], v[0], v[1] / v[0], v[1] / 1e3))


if __name__ == "__main__":
    main()
