"""GPU: train-mode bn3 from the Gram matrix of conv3's input (csrc/sat_gram.hip; `self.resnet(images)`, models.py:27 with BatchNorm2d in
train mode) -- the four ops SAT_OP_GRAM / GRAM_COV / GEMM_BF16_NT / BN_FROM_GRAM against the f64 column statistics of the conv's OWN
output, on the conv3 geometries of ResNet-152 at batch 64 (layers 2-4: the ones the fused form runs), on ragged sizes, grouped, and on
the cancellation case the quadratic form is weakest on (every output channel's variance ~1e-4 of its mean^2).  Then the whole stack:
the fused program against the three-launch form and against the CPU oracle.  Tolerances are written next to each check."""
import importlib
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

sat = importlib.import_module("show-and-tell_amd")
L = sat._lib
EPS, MOM = 1e-5, 0.1


def _ops(*ops):
    return (L.SatOp * len(ops))(*ops), len(ops)


def _gram_table(lib, c2, sc, sh, W, gamma, beta, groups=1, acc=None, bn2=None, running=None):
    """c2 bf16 [G][M][P] on the device; operand BatchNorm as a (scale, shift) table (ungrouped) or as integer sums `acc`
    [G][2][2][P] + (gamma2, beta2, count); W bf16 [N][P].  Returns (table [G][2][N], a2 bf16 [G][M][P] as the normalise+ReLU kernel
    forms it, slabs, mu)."""
    G, M, P = c2.shape
    N = W.shape[0]
    dev = c2.device
    slabs = torch.full((G * lib.sat_gram_slab_floats(M, P),), float("nan"), device=dev)
    cov3 = torch.empty(G * 3 * P * P, dtype=torch.bfloat16, device=dev)
    mu = torch.empty(G * P, dtype=torch.float64, device=dev)
    T = torch.empty(G * 3 * P * N, device=dev)
    table = torch.full((G, 2, N), float("nan"), device=dev)
    gr = L.SatOp()
    gr.kind, gr.dtype, gr.groups = L.OP_GRAM, L.SAT_BF16, G
    gr.in0, gr.out = c2.data_ptr(), slabs.data_ptr()
    gr.N, gr.Hout, gr.Wout, gr.Cout = M, 1, 1, P
    if acc is None:
        gr.scale0, gr.shift0 = sc.data_ptr(), sh.data_ptr()
    else:
        gr.stat_acc1, gr.gamma1, gr.beta1 = acc.data_ptr(), bn2[0].data_ptr(), bn2[1].data_ptr()
        gr.count, gr.eps = bn2[2], EPS
    co = L.SatOp()
    co.kind, co.dtype, co.groups = L.OP_GRAM_COV, L.SAT_BF16, G
    co.in0, co.out, co.scale_out = slabs.data_ptr(), cov3.data_ptr(), mu.data_ptr()
    co.N, co.Hout, co.Wout, co.Cout = M, 1, 1, P
    gm = L.SatOp()
    gm.kind, gm.dtype = L.OP_GEMM_BF16_NT, L.SAT_BF16
    gm.in0, gm.w, gm.out = cov3.data_ptr(), W.data_ptr(), T.data_ptr()
    gm.N, gm.Hout, gm.Wout, gm.Cin, gm.Cout = G * 3 * P, 1, 1, P, N
    fb = L.SatOp()
    fb.kind, fb.dtype, fb.groups = L.OP_BN_FROM_GRAM, L.SAT_BF16, G
    fb.in0, fb.in1, fb.w, fb.scale_out = T.data_ptr(), mu.data_ptr(), W.data_ptr(), table.data_ptr()
    fb.gamma, fb.beta = gamma.data_ptr(), beta.data_ptr()
    if running is not None:
        fb.running_mean, fb.running_var = running[0].data_ptr(), running[1].data_ptr()
    fb.Cin, fb.Cout, fb.count, fb.momentum, fb.eps = P, N, M, MOM, EPS
    ops, n = _ops(gr, co, gm, fb)
    L.check(lib.sat_run_ops_parity(ops, n, 0, L.stream()), "gram chain")
    torch.cuda.synchronize()
    return table, slabs, mu


def _a2_by_the_normalise_kernel(lib, c2, sc, sh):
    """relu(bn(c2)) as the stand-alone normalise+ReLU launch forms it (the conv kernels' operand transform is bit for bit this)"""
    G, M, P = c2.shape
    out = torch.empty_like(c2)
    for g in range(G):
        a = L.SatOp()
        a.kind, a.dtype = L.OP_BN_RELU, L.SAT_BF16
        a.in0, a.out, a.scale0, a.shift0 = c2[g].data_ptr(), out[g].data_ptr(), sc[g].data_ptr(), sh[g].data_ptr()
        a.N, a.Hout, a.Wout, a.Cout = M, 1, 1, P
        ops, n = _ops(a)
        L.check(lib.sat_run_ops(ops, n, L.stream()), "bn_relu")
    torch.cuda.synchronize()
    return out


def _reference(a2, W, gamma, beta):
    """f64 column statistics of c3 = a2 W^T (exact products of bf16 values, f64 sums) -> mean, biased var, (scale, shift)"""
    c3 = a2.double().cpu() @ W.double().cpu().t()
    mean, var = c3.mean(0), c3.var(0, unbiased=False)
    scale = gamma.double().cpu() / torch.sqrt(var + EPS)
    return mean, var, scale, beta.double().cpu() - mean * scale


@pytest.mark.parametrize("M,P", [(64 * 28 * 28, 128), (64 * 14 * 14, 256), (64 * 7 * 7, 512), (1000, 128), (777, 256), (130, 384)])
def test_bn3_from_the_gram_matrix_equals_the_f64_statistics_of_the_conv_output(M, P):
    lib = L.load()
    N = 4 * P
    g = torch.Generator().manual_seed(M + P)
    c2 = torch.randn(1, M, P, generator=g).to(torch.bfloat16).cuda()
    sc = (0.5 + torch.rand(1, P, generator=g)).cuda()
    sh = (0.4 * torch.randn(1, P, generator=g)).cuda()
    W = (torch.randn(N, P, generator=g) * (2.0 / N) ** 0.5).to(torch.bfloat16).cuda()
    gamma, beta = (0.5 + torch.rand(N, generator=g)).cuda(), torch.randn(N, generator=g).cuda()
    rm, rv = torch.zeros(N, device="cuda"), torch.ones(N, device="cuda")
    table, slabs, mu = _gram_table(lib, c2, sc[0], sh[0], W, gamma, beta, running=(rm, rv))
    a2 = _a2_by_the_normalise_kernel(lib, c2, sc, sh)[0]
    mean, var, scale, shift = _reference(a2, W, gamma, beta)
    assert torch.isfinite(slabs).all() and torch.isfinite(table).all()
    # the column sums of a2 are sums of bf16 values in f32 per slab, exact across slabs: mu to 1e-6 of the largest entry
    mu_ref = a2.double().mean(0).cpu()
    assert (mu.cpu() - mu_ref).abs().max().item() < 1e-6 * mu_ref.abs().max().item() + 1e-9
    got_scale, got_shift = table[0, 0].double().cpu(), table[0, 1].double().cpu()
    # (scale, shift) are f32 values of gamma / sqrt(var + eps): relative 2e-6 = a few f32 ulps + the ~3e-7 of the variance itself
    assert ((got_scale - scale).abs() / scale.abs()).max().item() < 2e-6
    assert ((got_shift - shift).abs() / (shift.abs() + scale.abs() * var.sqrt())).max().item() < 2e-6
    # running statistics: momentum update with the batch mean / UNBIASED variance
    unb = var * M / (M - 1)
    assert (rm.double().cpu() - MOM * mean).abs().max().item() < 1e-6 * (mean.abs().max().item() + 1e-3)
    assert ((rv.double().cpu() - (1 - MOM) - MOM * unb).abs() / (unb + 1e-3)).max().item() < 1e-5


def test_gram_route_on_the_cancellation_case():
    """every input channel ~ relu(0.16 x + 1) (coefficient of variation 0.16) and every output channel a positive mix of all of them:
    var_c ~ 1e-4 mean_c^2.  Sums of squares of the OUTPUTS lose 4 digits there; the Gram route centres the covariance of the INPUTS in
    f64 first: relative variance error < 1e-5 (tools/gram_numerics.py: 5e-7 on the CPU emulation, 3e-4 for the sum-of-squares route)"""
    lib = L.load()
    M, P = 64 * 14 * 14, 256
    N = 4 * P
    g = torch.Generator().manual_seed(99)
    c2 = torch.randn(1, M, P, generator=g).to(torch.bfloat16).cuda()
    sc, sh = torch.full((1, P), 0.16).cuda(), torch.full((1, P), 1.0).cuda()
    W = ((1.0 + 0.05 * torch.randn(N, P, generator=g)) / P).to(torch.bfloat16).cuda()
    gamma, beta = torch.ones(N).cuda(), torch.zeros(N).cuda()
    table, _, _ = _gram_table(lib, c2, sc[0], sh[0], W, gamma, beta)
    a2 = _a2_by_the_normalise_kernel(lib, c2, sc, sh)[0]
    mean, var, scale, shift = _reference(a2, W, gamma, beta)
    assert 3e-5 < (var / mean ** 2).median().item() < 3e-4            # the case is what it claims to be
    got_var = 1.0 / table[0, 0].double().cpu() ** 2 - EPS               # gamma = 1: scale = 1 / sqrt(var + eps)
    assert ((got_var - var).abs() / (var + EPS)).max().item() < 1e-5


def test_gram_chain_grouped_and_from_integer_sums_is_bitwise_the_ungrouped_one():
    """two batches in one launch (sat_op.groups = 2), the operand's BatchNorm derived from conv2's integer sums as in the program:
    each group's table is bit for bit the table of its own ungrouped run (the slabs depend on (M, P) only)"""
    lib = L.load()
    G, M, P = 2, 1500, 256
    N = 4 * P
    g = torch.Generator().manual_seed(5)
    c2 = torch.randn(G, M, P, generator=g).to(torch.bfloat16).cuda()
    W = (torch.randn(N, P, generator=g) * (2.0 / N) ** 0.5).to(torch.bfloat16).cuda()
    gamma, beta = (0.5 + torch.rand(N, generator=g)).cuda(), torch.randn(N, generator=g).cuda()
    gamma2, beta2 = (0.5 + torch.rand(P, generator=g)).cuda(), (0.3 * torch.randn(P, generator=g)).cuda()
    # integer sums [G][2 parities][2][P] of c2 itself (what conv2's epilogue leaves), parity 0
    acc = torch.zeros(G, 2, 2, P, dtype=torch.int64, device="cuda")
    c2d = c2.double()
    acc[:, 0, 0] = (c2d.sum(1) * 4194304.0).round().long()
    acc[:, 0, 1] = ((c2d * c2d).sum(1) * 4194304.0).round().long()
    both, _, _ = _gram_table(lib, c2, None, None, W, gamma, beta, groups=G, acc=acc, bn2=(gamma2, beta2, M))
    for k in range(G):
        one, _, _ = _gram_table(lib, c2[k:k + 1].contiguous(), None, None, W, gamma, beta, acc=acc[k:k + 1].contiguous(),
                                bn2=(gamma2, beta2, M))
        assert torch.equal(one[0], both[k]), k
    # ... and they are the statistics of relu(bn2(c2)) W^T
    mean2, var2 = c2d.mean(1), c2d.var(1, unbiased=False)
    sc = (gamma2.double() / torch.sqrt(var2 + EPS)).float()
    sh = (beta2.double() - mean2 * sc.double()).float()
    a2 = _a2_by_the_normalise_kernel(lib, c2, sc, sh)
    for k in range(G):
        _, _, scale, shift = _reference(a2[k], W, gamma, beta)
        # the device derives (scale2, shift2) in its own f64 / f32 arithmetic: a2 may differ from this host table's in single bf16
        # roundings, so the bound here is loose; the tight bound is the table test above
        assert ((both[k, 0].double().cpu() - scale).abs() / scale.abs()).max().item() < 1e-3


ARCH = dict(layers=(1, 2, 1, 1), width=128)       # layer 2's second block: planes 256, no projection -> the fused form


def _stack(dtype="bf16"):
    from oracle import encoder as OE
    gen = torch.Generator().manual_seed(21)
    ep, eb = OE.init_encoder_params(32, ARCH, generator=gen, randomize_bn=True, conditioning="trained_like")
    enc = sat.EncoderCNN(32, arch=ARCH, compute_dtype=dtype)
    sd = dict(ep)
    sd.update(eb)
    enc.load_state_dict(sd)
    return enc.cuda().train(), ep, eb, torch.randn(8, 3, 64, 64, generator=gen)


def test_fused_bottleneck_program_against_the_three_launch_form_and_the_oracle(monkeypatch):
    """a stack with a bottleneck the fused form runs (planes 256, no projection): pooled features of the fused program against the
    three-launch program (conv3 -> statistics of its output -> normalise + add + ReLU) and against the f32 CPU oracle; running
    statistics of bn3 from the Gram route against the oracle's"""
    from oracle import encoder as OE
    monkeypatch.setenv("SAT_GRAM_MAX_PLANES", "512")          # (the default fuses planes <= 128 only: where it was measured to pay)
    enc, ep, eb, images = _stack()
    x = images.cuda()
    with torch.no_grad():
        prog = enc._program(x)
        assert prog.gram_blocks == 1
        fused = enc.pooled_features(x).clone()
        rm_f = enc.resnet.layer2[1].bn3.running_mean.clone()
        rv_f = enc.resnet.layer2[1].bn3.running_var.clone()
    monkeypatch.setenv("SAT_GRAM_BN3", "0")
    enc2, _, _, _ = _stack()
    with torch.no_grad():
        assert enc2._program(x).gram_blocks == 0
        plain = enc2.pooled_features(x).clone()
        rm_p = enc2.resnet.layer2[1].bn3.running_mean.clone()
        rv_p = enc2.resnet.layer2[1].bn3.running_var.clone()
    bufs = {k: v.clone() for k, v in eb.items()}
    ref = OE.resnet_forward(ep, bufs, images, ARCH, training=True)[0]
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    # two bf16 programs that differ in where bn3's rounding happens: within bf16 noise of each other and of the f32 oracle
    assert rel(fused.cpu(), ref) < 2e-2 and rel(plain.cpu(), ref) < 2e-2, (rel(fused.cpu(), ref), rel(plain.cpu(), ref))
    assert rel(fused, plain) < 2e-2
    # bn3's batch statistics: the Gram route sees the f32 accumulators' exact statistics, the other route sums them: equal to bf16-input noise
    key = "resnet.layer2.1.bn3."
    assert rel(rm_f.cpu(), bufs[key + "running_mean"]) < 2e-2 and rel(rv_f.cpu(), bufs[key + "running_var"]) < 2e-2
    assert rel(rm_f, rm_p) < 1e-2 and rel(rv_f, rv_p) < 1e-2


def test_fused_bottleneck_grouped_lookahead_is_bitwise_the_sequential_run(monkeypatch):
    """two batches through ONE grouped program run (every launch covers both, the Gram chain included) against each batch's own
    ungrouped run: pooled features and bn3's running statistics bit for bit"""
    monkeypatch.setenv("SAT_GRAM_MAX_PLANES", "512")
    enc, _, _, images = _stack()
    g = torch.Generator().manual_seed(3)
    a, b = images.cuda(), torch.randn(8, 3, 64, 64, generator=g).cuda()
    with torch.no_grad():
        enc.prefetch_many([a, b])
        pa, pb = enc.pooled_features(a).clone(), enc.pooled_features(b).clone()
    rm = enc.resnet.layer2[1].bn3.running_mean.clone()
    enc2, _, _, _ = _stack()
    enc2.lookahead_depth = 0
    with torch.no_grad():
        assert enc2._program(a).gram_blocks == 1
        qa, qb = enc2.pooled_features(a).clone(), enc2.pooled_features(b).clone()
    assert torch.equal(pa, qa) and torch.equal(pb, qb)
    assert torch.equal(rm, enc2.resnet.layer2[1].bn3.running_mean)
