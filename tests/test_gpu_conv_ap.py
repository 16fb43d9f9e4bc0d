"""GPU: conv_ap_kernel (variant 35, csrc/sat_conv_ap.inc) -- the expansion 1x1 conv of a bottleneck (K = 256 -> N = 4 K, conv3 of
`self.resnet(images)`, models.py:27) with the weights resident in registers and the workgroup persistent over row tiles -- against the
ring kernel (variant 1: same MFMA, same K order, so the output tensor is BITWISE equal; the column sums to rounding) and the f64
definition: the layer-3 geometry at batch 64, ragged last tiles, fewer row tiles than workers, many tiles per worker, statistics as
slabs / integer atomics / none, the operand's BatchNorm + ReLU precomputed or derived from the producer's integer sums, grouped."""
import ctypes as C
import importlib

import pytest
import torch
import torch.nn.functional as F

from test_gpu_kernels import _conv_op, _pack_weights, cu, st, sync

pytestmark = pytest.mark.gpu
sat = importlib.import_module("show-and-tell_amd")
L = sat._lib
AP = 35


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "needs the MI355X"
    return L.load()


@pytest.mark.parametrize("mode", ["slab", "atomic", "none"])
@pytest.mark.parametrize("affine", ["no", "table", "derive"])
@pytest.mark.parametrize("N,H,W,Cout", [(64, 14, 14, 1024), (3, 9, 13, 512), (1, 10, 13, 128), (200, 14, 14, 256), (17, 7, 7, 1024)])
def test_conv_ap_is_bit_identical_to_the_ring_kernel(lib, N, H, W, Cout, affine, mode):
    Cin = 256
    g = torch.Generator().manual_seed(N * 13 + W + Cout)
    x = (torch.randn(N, Cin, H, W, generator=g) * 1.5 + 0.2).bfloat16()
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5).bfloat16().float()
    gamma, beta = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.2 + 0.3
    xf = x.float().permute(0, 2, 3, 1).reshape(-1, Cin).double()
    M = xf.shape[0]
    mean, var = xf.mean(0), xf.var(0, unbiased=False)
    scale = (gamma.double() / torch.sqrt(var + 1e-5)).float()
    shift = (beta.double() - mean * scale.double()).float()
    a = x.float() if affine == "no" else torch.clamp(x.float() * scale[None, :, None, None] + shift[None, :, None, None], min=0).bfloat16().float()
    ref = F.conv2d(a.double(), w.double()).permute(0, 2, 3, 1).reshape(-1, Cout)

    def run(v):
        o, keep, _ = _conv_op(L.SAT_BF16, x.float().permute(0, 2, 3, 1), w.permute(0, 2, 3, 1), 1, 0, stats=(mode == "slab"))
        o.variant = v
        extra = {"wp": _pack_weights(lib, keep[1], Cout, Cin, 1)}
        o.w_packed = extra["wp"].data_ptr()
        if mode == "atomic":
            extra["acc"] = torch.zeros(2, 2, Cout, dtype=torch.int64, device="cuda")
            o.stat_acc = extra["acc"].data_ptr()
        if affine == "derive":
            iacc = torch.zeros(2, 2, Cin, dtype=torch.int64, device="cuda")
            iacc[0, 0] = torch.round(xf.sum(0) * 4194304.0).long().cuda()
            iacc[0, 1] = torch.round((xf ** 2).sum(0) * 4194304.0).long().cuda()
            iacc[1] = 777
            gd, bd, rm, rv = cu(gamma), cu(beta), cu(torch.zeros(Cin)), cu(torch.ones(Cin))
            o.stat_acc1, o.gamma1, o.beta1 = iacc.data_ptr(), gd.data_ptr(), bd.data_ptr()
            o.running_mean1, o.running_var1 = rm.data_ptr(), rv.data_ptr()
            o.count, o.momentum, o.eps = M, 0.1, 1e-5
            extra.update(iacc=iacc, gd=gd, bd=bd, rm=rm, rv=rv)
        elif affine == "table":
            sd, td = cu(scale), cu(shift)
            o.scale0, o.shift0 = sd.data_ptr(), td.data_ptr()
            extra.update(sd=sd, td=td)
        L.check(lib.sat_run_ops_parity(C.pointer(o), 1, 0, st()))
        sync()
        return keep, extra

    want, wx = run(1)
    got, gx = run(AP)
    assert torch.isfinite(got[2].float()).all()
    assert torch.equal(got[2], want[2])
    assert (got[2].float().cpu().double() - ref).abs().max().item() < 3e-2 + 4e-3 * ref.abs().max().item()
    if mode == "slab":
        torch.testing.assert_close(got[3], want[3], rtol=1e-4, atol=2e-3)
    elif mode == "atomic":
        torch.testing.assert_close(gx["acc"].double() / 2 ** 22, wx["acc"].double() / 2 ** 22, rtol=1e-4, atol=5e-3)
        assert int(gx["acc"][1].abs().sum()) == 0
        # the sums are those of the f32 accumulators (f32 partial sums per lane and worker): against the f64 column sums of the exact
        # products to 1e-3 relative / 1e-5 per row absolute
        torch.testing.assert_close(gx["acc"][0, 0].double().cpu() / 2 ** 22, ref.sum(0), rtol=1e-3, atol=1e-5 * M + 2e-2)
    if affine == "derive":
        assert int(gx["iacc"][1].abs().sum()) == 0
        assert torch.equal(gx["rm"], wx["rm"]) and torch.equal(gx["rv"], wx["rv"])


def test_conv_ap_grouped_launch_equals_one_launch_per_batch(lib):
    """sat_op.groups = 2: every group is bit for bit the ungrouped launch on its batch -- output AND the integer column sums (the
    worker split depends on (M, N) only)"""
    N, H, W, Cin, Cout, G = 9, 14, 14, 256, 512, 2
    g = torch.Generator().manual_seed(4)
    xs = [(torch.randn(N, H, W, Cin, generator=g)).bfloat16() for _ in range(G)]
    w = (torch.randn(Cout, 1, 1, Cin, generator=g) / 16).bfloat16()
    wd = cu(w.reshape(Cout, Cin).contiguous())
    wp = _pack_weights(lib, wd, Cout, Cin, 1)
    M = N * H * W

    def op(x, out, acc, groups):
        o = L.SatOp()
        o.kind, o.dtype, o.groups, o.variant = L.OP_CONV, L.SAT_BF16, groups, AP
        o.in0, o.w, o.out, o.w_packed, o.stat_acc = x.data_ptr(), wd.data_ptr(), out.data_ptr(), wp.data_ptr(), acc.data_ptr()
        o.N, o.Hin, o.Win, o.Cin, o.Hout, o.Wout, o.Cout = N, H, W, Cin, H, W, Cout
        o.KH, o.KW, o.stride, o.pad = 1, 1, 1, 0
        o.sN, o.sH, o.sW = H * W * Cin, W * Cin, Cin
        return o

    xg = cu(torch.stack(xs).contiguous())
    outg = torch.full((G, M, Cout), float("nan"), device="cuda", dtype=torch.bfloat16)
    accg = torch.zeros(G, 2, 2, Cout, dtype=torch.int64, device="cuda")
    L.check(lib.sat_run_ops_parity(C.pointer(op(xg, outg, accg, G)), 1, 0, st()))
    sync()
    for k in range(G):
        out1 = torch.full((M, Cout), float("nan"), device="cuda", dtype=torch.bfloat16)
        acc1 = torch.zeros(2, 2, Cout, dtype=torch.int64, device="cuda")
        x1 = xg[k].contiguous()
        L.check(lib.sat_run_ops_parity(C.pointer(op(x1, out1, acc1, 1)), 1, 0, st()))
        sync()
        assert torch.equal(out1, outg[k]) and torch.equal(acc1, accg[k]), k
