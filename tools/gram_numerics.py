#!/usr/bin/env python3
"""Numerics gate of the Gram-statistics route for bn3, on the CPU, before any kernel exists.

Train-mode bn3 needs the batch mean / variance of c3 = a2 W3^T (a2 = relu(bn2(c2)), the bf16 operand conv3 reads) BEFORE conv3's
epilogue can normalise.  They are a linear / quadratic form of conv3's INPUT:
    mean_c = w_c . mu,   var_c = w_c^T (G / M - mu mu^T) w_c,     mu = sum_m a2[m] / M,   G = a2^T a2
This script emulates what the kernels would do -- G and sum(a2) accumulated in f32 per row slab (the MFMA), slabs summed exactly
(fixed-point int64), covariance and quadratic form in f64 -- and compares with the f64 column statistics of the conv's own output,
next to an emulation of what the kernels do TODAY (f32 sum and sum of squares of the f32 accumulators per 128-row tile, tiles
summed exactly).  Geometries: conv3 of every ResNet-152 stage at batch 64, plus the cancellation case the quadratic form is
weakest on (every channel's variance ~1e-4 of its mean^2).      python tools/gram_numerics.py
"""
import sys

import torch

torch.manual_seed(0)
torch.set_num_threads(8)
EPS = 1e-5


def bf16(x):
    return x.to(torch.bfloat16).to(torch.float32)


def run(name, M, P, slab, scale=None, shift=None, wmode="he"):
    N = 4 * P
    g = torch.Generator().manual_seed(M + P)
    c2 = bf16(torch.randn(M, P, generator=g))                                   # raw conv2 output as stored
    sc = (0.5 + torch.rand(P, generator=g)) if scale is None else torch.full((P,), float(scale))
    sh = (0.4 * torch.randn(P, generator=g)) if shift is None else torch.full((P,), float(shift))
    a2 = bf16(torch.relu(torch.addcmul(sh, c2, sc)))                            # f32 fma -> bf16 (RNE) -> ReLU, the kernels' transform
    if wmode == "he":
        W = bf16(torch.randn(N, P, generator=g) * (2.0 / N) ** 0.5)
    else:                                                                       # every output channel ~ a positive mix: mean^2 >> var
        W = bf16((1.0 + 0.05 * torch.randn(N, P, generator=g)) / P)
    a64, W64 = a2.double(), W.double()
    c3 = a64 @ W64.t()                                                          # exact products, f64 sums
    mean_ref = c3.mean(0)
    var_ref = c3.var(0, unbiased=False)
    # ---- today: f32 accumulators (emulated: f32 matmul), per 128-row tile sum / sum of squares in f32, tiles summed exactly ----
    c3_32 = a2 @ W.t()
    s1 = torch.zeros(N, dtype=torch.float64)
    s2 = torch.zeros(N, dtype=torch.float64)
    for r in range(0, M, 128):
        t = c3_32[r:r + 128]
        s1 += t.sum(0).double()
        s2 += (t * t).sum(0).double()
    mean_now = s1 / M
    var_now = (s2 / M - mean_now ** 2).clamp_min(0)
    # ---- Gram route: per slab G and column sums in f32, slabs summed exactly, covariance + quadratic form in f64 ----
    G = torch.zeros(P, P, dtype=torch.float64)
    s = torch.zeros(P, dtype=torch.float64)
    for r in range(0, M, slab):
        t = a2[r:r + slab]
        G += (t.t() @ t).double()
        s += t.sum(0).double()
    mu = s / M
    cov = G / M - torch.outer(mu, mu)
    mean_g = W64 @ mu
    var_g = ((W64 @ cov) * W64).sum(1).clamp_min(0)
    # the same with the quadratic form on the f32 matrix pipe: T = cov32 W^T (f32), var = sum_i W[c,i] T[i,c] (f64 dot)
    T = (cov.float() @ W.t()).double()
    var_g32 = (W64 * T.t()).sum(1).clamp_min(0)

    def rel_var(v):
        return ((v - var_ref).abs() / (var_ref + EPS)).max().item()

    def rel_mean(m):
        return ((m - mean_ref).abs() / (var_ref + EPS).sqrt()).max().item()

    ratio = (var_ref / (mean_ref ** 2 + 1e-30)).median().item()
    print("%-34s M=%6d P=%3d slab=%4d  var/mean^2 (median) %.1e | today: dvar %.1e dmean/std %.1e | gram f64: dvar %.1e dmean/std %.1e | "
          "gram, f32 quadratic form: dvar %.1e" % (name, M, P, slab, ratio, rel_var(var_now), rel_mean(mean_now), rel_var(var_g),
                                                   rel_mean(mean_g), rel_var(var_g32)), flush=True)
    return rel_var(var_g), rel_mean(mean_g), rel_var(var_now)


rows = []
for name, hw, P in (("layer1 conv3 56x56 64->256", 56, 64), ("layer2 conv3 28x28 128->512", 28, 128),
                    ("layer3 conv3 14x14 256->1024", 14, 256), ("layer4 conv3 7x7 512->2048", 7, 512)):
    M = 64 * hw * hw
    rows.append(run(name, M, P, 392))
rows.append(run("layer3, slab 98", 64 * 196, 256, 98))
rows.append(run("layer3, slab 1568", 64 * 196, 256, 1568))
# the cancellation case: a2 = relu(0.16 c2 + 1) has a coefficient of variation of 0.16 and every output channel is a positive
# mix of all of them: var_c / mean_c^2 ~ 0.16^2 / 256 = 1e-4
rows.append(run("layer3, var ~ 1e-4 mean^2", 64 * 196, 256, 392, scale=0.16, shift=1.0, wmode="positive"))
rows.append(run("layer3, var ~ 1e-6 mean^2", 64 * 196, 256, 392, scale=0.016, shift=1.0, wmode="positive"))
ok = all(r[0] < 5e-3 for r in rows[:6])
print("gate (He-init geometries: relative variance error < 5e-3, i.e. below one bf16 ulp of the normalised output): %s" % ("PASS" if ok else "FAIL"))
sys.exit(0 if ok else 1)
