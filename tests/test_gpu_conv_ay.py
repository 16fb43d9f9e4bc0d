"""GPU: conv_ay_kernel (variants 36 / 37, csrc/sat_conv_ay.inc, SAT_CONV_IN_RESIDUAL) -- conv1 of a bottleneck that also finishes the
bottleneck in front of it (`self.resnet(images)`, models.py:27, train-mode BatchNorm): its operand y = relu(bn3(c3) + residual) is built
on the way to LDS and written out as it goes.  Against the two launches it replaces -- SAT_OP_BN_ADD_RELU, then the plain conv_aw_kernel
on its output -- every result must be BITWISE equal: y, the conv output, the conv's integer column sums, bn3's running statistics and
the cleared accumulators.  Geometries: the conv1s of ResNet-152's layers 2-4 at batch 64 (rows scaled down), ragged last tiles, both
forms of the operand BatchNorm, in place, grouped.  Then the whole stack: the deferred program against the three-launch program, bit
for bit, sequential and as a grouped look-ahead."""
import ctypes as C
import importlib

import pytest
import torch

from test_gpu_kernels import _pack_weights, cu, st, sync

pytestmark = pytest.mark.gpu
sat = importlib.import_module("show-and-tell_amd")
L = sat._lib
AW4, AY4, AY8 = 33, 36, 37


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "needs the MI355X"
    return L.load()


def _case(lib, G, N, H, W, Cin, Cout, seed):
    g = torch.Generator().manual_seed(seed)
    M = N * H * W
    c3 = (torch.randn(G, M, Cin, generator=g) * 1.3 + 0.1).bfloat16()
    res = torch.clamp(torch.randn(G, M, Cin, generator=g) + 0.2, min=0).bfloat16()
    w = (torch.randn(Cout, Cin, generator=g) / Cin ** 0.5).bfloat16()
    gamma, beta = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.2
    acc = torch.zeros(G, 2, 2, Cin, dtype=torch.int64)
    for q in range(G):
        xf = c3[q].double()
        acc[q, 0, 0] = torch.round(xf.sum(0) * 4194304.0).long()
        acc[q, 0, 1] = torch.round((xf ** 2).sum(0) * 4194304.0).long()
        acc[q, 1] = 4242
    wd = cu(w.contiguous())
    return dict(G=G, N=N, H=H, W=W, M=M, Cin=Cin, Cout=Cout, c3=c3, res=res, wd=wd, wp=_pack_weights(lib, wd, Cout, Cin, 1),
                gamma=cu(gamma), beta=cu(beta), acc=acc)


def _conv(k, x, out, acc_out, variant):
    o = L.SatOp()
    o.kind, o.dtype, o.groups = L.OP_CONV, L.SAT_BF16, k["G"]
    o.in0, o.w, o.w_packed, o.out = x.data_ptr(), k["wd"].data_ptr(), k["wp"].data_ptr(), out.data_ptr()
    o.N, o.Hin, o.Win, o.Cin, o.Hout, o.Wout, o.Cout = k["N"], k["H"], k["W"], k["Cin"], k["H"], k["W"], k["Cout"]
    o.KH, o.KW, o.stride, o.pad = 1, 1, 1, 0
    o.sN, o.sH, o.sW = k["H"] * k["W"] * k["Cin"], k["W"] * k["Cin"], k["Cin"]
    o.stat_acc = acc_out.data_ptr()
    o.variant = variant
    return o


def _bn_source(o, k, st_, first, table=None):
    sfx = "" if first else "1"
    if table is not None:
        o.scale0, o.shift0 = table[0].data_ptr(), table[1].data_ptr()
        return
    setattr(o, "stat_acc" + sfx, st_["acc"].data_ptr())
    setattr(o, "gamma" + sfx, k["gamma"].data_ptr())
    setattr(o, "beta" + sfx, k["beta"].data_ptr())
    setattr(o, "running_mean" + sfx, st_["run"][0, 0].data_ptr())
    setattr(o, "running_var" + sfx, st_["run"][0, 1].data_ptr())
    o.count, o.momentum, o.eps = k["M"], 0.1, 1e-5


def _state(k):
    G, Cin, Cout = k["G"], k["Cin"], k["Cout"]
    run = torch.zeros(G, 2, Cin, device="cuda")
    run[:, 1] = 1.0
    return dict(acc=cu(k["acc"].clone()), run=run, c3=cu(k["c3"].clone()), res=cu(k["res"].clone()),
                y=torch.full((G, k["M"], Cin), float("nan"), device="cuda", dtype=torch.bfloat16),
                c1=torch.full((G, k["M"], Cout), float("nan"), device="cuda", dtype=torch.bfloat16),
                oacc=torch.zeros(G, 2, 2, Cout, dtype=torch.int64, device="cuda"))


def _two_launches(lib, k, table=None):
    s = _state(k)
    act = L.SatOp()
    act.kind, act.dtype, act.groups = L.OP_BN_ADD_RELU, L.SAT_BF16, k["G"]
    act.in0, act.in1, act.out = s["c3"].data_ptr(), s["res"].data_ptr(), s["y"].data_ptr()
    _bn_source(act, k, s, True, table)
    act.N, act.Hout, act.Wout, act.Cout = k["N"], k["H"], k["W"], k["Cin"]
    ops = (L.SatOp * 2)(act, _conv(k, s["y"], s["c1"], s["oacc"], AW4))
    L.check(lib.sat_run_ops_parity(ops, 2, 0, st()), "bn_add + conv_aw")
    sync()
    return s


def _one_launch(lib, k, variant, table=None, inplace=False):
    s = _state(k)
    o = _conv(k, s["c3"], s["c1"], s["oacc"], variant)
    _bn_source(o, k, s, False, table)
    o.in1, o.out1, o.flags = s["res"].data_ptr(), (s["c3"] if inplace else s["y"]).data_ptr(), L.CONV_IN_RESIDUAL
    L.check(lib.sat_run_ops_parity(C.pointer(o), 1, 0, st()), "conv_ay")
    sync()
    if inplace:
        s["y"] = s["c3"]
    return s


def _same(a, b, derived=True):
    assert torch.isfinite(a["y"].float()).all() and torch.isfinite(a["c1"].float()).all()
    assert torch.equal(a["y"], b["y"])
    assert torch.equal(a["c1"], b["c1"])
    assert torch.equal(a["oacc"], b["oacc"]) and int(a["oacc"][:, 0].abs().sum()) > 0
    if derived:
        assert torch.equal(a["run"], b["run"]) and float((a["run"][:, 0]).abs().sum()) > 0
        assert torch.equal(a["acc"], b["acc"]) and int(a["acc"][:, 1].abs().sum()) == 0


@pytest.mark.parametrize("variant", [AY4, AY8])
@pytest.mark.parametrize("N,H,W,Cin,Cout", [(8, 14, 14, 1024, 256), (3, 28, 28, 512, 128), (5, 7, 7, 2048, 512), (1, 9, 13, 512, 256),
                                            (2, 10, 13, 1024, 768)])
def test_conv_ay_is_bitwise_the_normalise_add_launch_then_conv_aw(lib, N, H, W, Cin, Cout, variant):
    if variant == AY8 and Cout % 256:
        pytest.skip("the eight-wave form has 256-column tiles")
    k = _case(lib, 1, N, H, W, Cin, Cout, N * 31 + W + Cout)
    want = _two_launches(lib, k)
    _same(_one_launch(lib, k, variant), want)
    # against the definition, in f64 on the kernel's bf16 operands (the tolerance is bf16 output rounding at |y|, |c1| < ~8)
    xf = k["c3"][0].double()
    mean, var = xf.mean(0), xf.var(0, unbiased=False)
    sc = k["gamma"].cpu().double() / torch.sqrt(var + 1e-5)
    y = (xf * sc + (k["beta"].cpu().double() - mean * sc) + k["res"][0].double()).clamp(min=0)
    assert (want["y"][0].float().cpu().double() - y).abs().max().item() < 6e-2
    c1 = want["y"][0].float().cpu().double() @ k["wd"].float().cpu().double().t()
    assert (want["c1"][0].float().cpu().double() - c1).abs().max().item() < 3e-2 + 4e-3 * c1.abs().max().item()


def test_conv_ay_with_a_precomputed_table_in_place_and_grouped(lib):
    k = _case(lib, 1, 4, 14, 14, 1024, 256, 5)
    g = torch.Generator().manual_seed(6)
    table = (cu(torch.rand(1024, generator=g) + 0.3), cu(torch.randn(1024, generator=g) * 0.3))
    want = _two_launches(lib, k, table)
    _same(_one_launch(lib, k, AY4, table), want, derived=False)
    _same(_one_launch(lib, k, AY8, table), want, derived=False)
    # in place: y overwrites the raw tensor -- only where one column tile covers Cout (the eight-wave form here; the library refuses the other)
    want = _two_launches(lib, k)
    _same(_one_launch(lib, k, AY8, inplace=True), want)
    s = _state(k)
    o = _conv(k, s["c3"], s["c1"], s["oacc"], AY4)
    _bn_source(o, k, s, False)
    o.in1, o.out1, o.flags = s["res"].data_ptr(), s["c3"].data_ptr(), L.CONV_IN_RESIDUAL
    ops = (L.SatOp * 1)(o)
    assert lib.sat_run_ops_parity(ops, 1, 0, st()) == 0 and ops[0].variant == AY4      # (runs: the launch falls back on a variant that may alias)
    sync()
    assert torch.equal(s["c3"], want["y"]) and torch.equal(s["c1"], want["c1"])
    k128 = _case(lib, 1, 3, 28, 28, 512, 128, 8)
    _same(_one_launch(lib, k128, AY4, inplace=True), _two_launches(lib, k128))
    # grouped: every group is the ungrouped launch on its own batch, with its own statistics and running-statistics log
    k2 = _case(lib, 2, 3, 14, 14, 1024, 256, 11)
    want2 = _two_launches(lib, k2)
    for v in (AY4, AY8):
        got2 = _one_launch(lib, k2, v)
        _same(got2, want2)
        for q in range(2):
            kq = dict(k2, G=1, c3=k2["c3"][q:q + 1], res=k2["res"][q:q + 1], acc=k2["acc"][q:q + 1])
            one = _one_launch(lib, kq, v)
            assert torch.equal(one["y"][0], got2["y"][q]) and torch.equal(one["c1"][0], got2["c1"][q])
            assert torch.equal(one["oacc"][0], got2["oacc"][q]) and torch.equal(one["run"][0], got2["run"][q])


def test_conv_ay_refuses_what_it_cannot_run(lib):
    k = _case(lib, 1, 1, 8, 8, 512, 128, 3)
    s = _state(k)
    o = _conv(k, s["c3"], s["c1"], s["oacc"], 0)
    _bn_source(o, k, s, False)
    o.in1, o.out1, o.flags = s["res"].data_ptr(), s["y"].data_ptr(), L.CONV_IN_RESIDUAL
    assert lib.sat_run_ops_parity(C.pointer(o), 1, 0, st()) == 0            # variant 0: the library's own choice runs it
    o.out1 = None
    assert lib.sat_run_ops_parity(C.pointer(o), 1, 0, st()) == 1001
    o.out1 = s["res"].data_ptr()                                             # y over the residual: other workgroups still read it
    assert lib.sat_run_ops_parity(C.pointer(o), 1, 0, st()) == 1003
    o.out1, o.w_packed = s["y"].data_ptr(), None
    assert lib.sat_run_ops_parity(C.pointer(o), 1, 0, st()) == 1003
    sync()


ARCH = {"name": "tiny-bottleneck-defer", "layers": (2, 3, 4, 3), "width": 64}


def _stack(dtype="bf16"):
    from oracle import encoder as OE
    gen = torch.Generator().manual_seed(11)
    ep, eb = OE.init_encoder_params(32, ARCH, generator=gen, randomize_bn=True, conditioning="trained_like")
    enc = sat.EncoderCNN(32, arch=ARCH, compute_dtype=dtype)
    sd = dict(ep)
    sd.update(eb)
    enc.load_state_dict(sd)
    return enc.cuda().train(), torch.randn(8, 3, 64, 64, generator=gen)


def _force_aw(prog):
    """every conv1 of a non-first bottleneck on conv_aw_kernel: the kernel whose column sums conv_ay_kernel reproduces (statistics
    signature 5000) -- in a real model the leader program's signatures pin that; here neither program has a table entry"""
    for i in range(prog.n_ops):
        o = prog.ops[i]
        if (o.kind == L.OP_CONV and o.KH == 1 and o.stride == 1 and o.Cin == 4 * o.Cout and o.Cout % 128 == 0 and
                not (o.flags & L.CONV_IN_RESIDUAL)):
            o.variant = AW4


def _bn_state(enc):
    return {n: b.clone() for n, b in enc.resnet.named_buffers() if "running" in n}


@pytest.mark.parametrize("inplace", ["0", "1"])
def test_deferred_program_is_bitwise_the_three_launch_program(monkeypatch, inplace):
    """a stack whose layers 2 and 3 have bottlenecks the deferred form runs: pooled features and every running statistic of the
    program with bn3 + add + ReLU inside the next conv1 against the program with the separate launches, two steps (both parities)"""
    monkeypatch.setenv("SAT_GRAM_BN3", "0")
    monkeypatch.setenv("SAT_DEFER_BN3", "1")
    monkeypatch.setenv("SAT_DEFER_INPLACE", inplace)
    enc, images = _stack()
    x = images.cuda()
    with torch.no_grad():
        prog = enc._program(x)
        assert prog.deferred_blocks == 4                                  # layer 2: 1 of 3 bottlenecks, layer 3: 2 of 4, layer 4: 1 of 3
        _force_aw(prog)
        got = [enc.pooled_features(x).clone() for _ in range(2)]
    st_got = _bn_state(enc)
    monkeypatch.setenv("SAT_DEFER_BN3", "0")
    enc2, _ = _stack()
    with torch.no_grad():
        assert enc2._program(x).deferred_blocks == 0
        _force_aw(enc2._program(x))
        want = [enc2.pooled_features(x).clone() for _ in range(2)]
    st_want = _bn_state(enc2)
    assert torch.isfinite(got[0]).all()
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    for n in st_want:
        assert torch.equal(st_got[n], st_want[n]), n


def test_deferred_program_grouped_lookahead_is_bitwise_the_sequential_run(monkeypatch):
    """two batches through ONE grouped run of the deferred program (every launch covers both; conv_ay_kernel with two groups)
    against each batch's own ungrouped run of the deferred program: pooled features and running statistics bit for bit"""
    monkeypatch.setenv("SAT_GRAM_BN3", "0")
    monkeypatch.setenv("SAT_DEFER_BN3", "1")
    enc, images = _stack()
    g = torch.Generator().manual_seed(3)
    a, b = images.cuda(), torch.randn(8, 3, 64, 64, generator=g).cuda()
    with torch.no_grad():
        enc.prefetch_many([a, b])
        pa, pb = enc.pooled_features(a).clone(), enc.pooled_features(b).clone()
    st_g = _bn_state(enc)
    enc2, _ = _stack()
    enc2.lookahead_depth = 0
    with torch.no_grad():
        assert enc2._program(a).deferred_blocks == 4
        qa, qb = enc2.pooled_features(a).clone(), enc2.pooled_features(b).clone()
    assert torch.equal(pa, qa) and torch.equal(pb, qb)
    st_s = _bn_state(enc2)
    for n in st_s:
        assert torch.equal(st_g[n], st_s[n]), n
