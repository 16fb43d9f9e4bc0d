"""GPU: conv_rs_kernel (variant 38, csrc/sat_conv_rs.inc) -- the 3 x 3 / stride 1 convs over 32 channels of the Inception-v3 stem
(Conv2d_2a_3x3: 32 -> 32 without padding, Conv2d_2b_3x3: 32 -> 64 with padding 1; BASELINE configs[3]) with the weights in registers and
whole input rows in LDS -- against the ring kernel (variant 1: same MFMA, same K order, so the output tensor is BITWISE equal; the column
sums to rounding) and the f64 definition: the stem's own map sizes at a small batch, maps whose rows are barely 128 pixels wide, tiles that
straddle two rows and two images, a ragged last tile, statistics as slabs / integer atomics / none, grouped."""
import ctypes as C
import importlib

import pytest
import torch
import torch.nn.functional as F

from test_gpu_kernels import _conv_op, _stem_op, cu, st, sync

pytestmark = pytest.mark.gpu
sat = importlib.import_module("show-and-tell_amd")
L = sat._lib
RS = 38
RS64 = 39
RS8 = 40
RS_STEM = 41


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "needs the MI355X"
    return L.load()


@pytest.mark.parametrize("mode", ["slab", "atomic", "none"])
@pytest.mark.parametrize("N,H,W,Cout,pad", [(2, 149, 149, 32, 0), (2, 147, 147, 64, 1), (1, 5, 131, 32, 0), (3, 4, 128, 64, 1), (1, 3, 300, 64, 0)])
def test_conv_rs_is_bit_identical_to_the_ring_kernel(lib, N, H, W, Cout, pad, mode):
    Cin = 32
    g = torch.Generator().manual_seed(N * 13 + W + Cout)
    x = (torch.randn(N, Cin, H, W, generator=g) * 1.5 + 0.2).bfloat16().float()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5).bfloat16().float()
    ref = F.conv2d(x.double(), w.double(), None, 1, pad).permute(0, 2, 3, 1).reshape(-1, Cout)

    def run(v):
        o, keep, _ = _conv_op(L.SAT_BF16, x.permute(0, 2, 3, 1), w.permute(0, 2, 3, 1), 1, pad, stats=(mode == "slab"))
        o.variant = v
        extra = {}
        if mode == "atomic":
            extra["acc"] = torch.zeros(2, 2, Cout, dtype=torch.int64, device="cuda")
            o.stat_acc = extra["acc"].data_ptr()
        ops = (L.SatOp * 1)(o)
        L.check(lib.sat_run_ops_parity(ops, 1, 0, st()))
        sync()
        return keep, extra

    want, wx = run(1)
    got, gx = run(RS)
    assert torch.isfinite(got[2].float()).all()
    assert torch.equal(got[2], want[2])
    assert (got[2].float().cpu().double() - ref).abs().max().item() < 3e-2 + 4e-3 * ref.abs().max().item()
    if mode == "slab":
        torch.testing.assert_close(got[3].sum(0), want[3].sum(0), rtol=1e-4, atol=2e-2)
        acc_f = got[2].float().cpu().double()
        torch.testing.assert_close(got[3].sum(0)[0].cpu().double(), ref.sum(0), rtol=1e-3, atol=1e-5 * ref.shape[0] + 2e-2)
        del acc_f
    elif mode == "atomic":
        torch.testing.assert_close(gx["acc"].double() / 2 ** 22, wx["acc"].double() / 2 ** 22, rtol=1e-4, atol=2e-2)
        assert int(gx["acc"][1].abs().sum()) == 0


def test_conv_rs_is_the_default_for_its_geometry_and_runs_grouped(lib):
    """no table entry: sat_conv_default_variant names it; two groups in one launch = each group's own launch, bit for bit"""
    N, H, W, Cin, Cout = 2, 6, 140, 32, 64
    g = torch.Generator().manual_seed(5)
    xs = [(torch.randn(N, H, W, Cin, generator=g) + 0.1).bfloat16() for _ in range(2)]
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) / 17.0).bfloat16()
    o, keep, _ = _conv_op(L.SAT_BF16, xs[0].float(), w.float(), 1, 1, stats=False)
    assert lib.sat_conv_default_variant(C.byref(o), -1) == RS
    assert lib.sat_conv_variant_signature(RS) == 7000
    outs = []
    for x in xs:
        o1, k1, _ = _conv_op(L.SAT_BF16, x.float(), w.float(), 1, 1, stats=False)
        acc = torch.zeros(2, 2, Cout, dtype=torch.int64, device="cuda")
        o1.stat_acc, o1.variant = acc.data_ptr(), RS
        L.check(lib.sat_run_ops_parity(C.pointer(o1), 1, 0, st()))
        sync()
        outs.append((k1[2].clone(), acc.clone()))
    xg = cu(torch.stack(xs))
    og, kg, _ = _conv_op(L.SAT_BF16, xs[0].float(), w.float(), 1, 1, stats=False)
    outg = torch.full((2, N * H * W, Cout), float("nan"), device="cuda", dtype=torch.bfloat16)
    accg = torch.zeros(2, 2, 2, Cout, dtype=torch.int64, device="cuda")
    og.in0, og.out, og.stat_acc, og.groups, og.variant = xg.data_ptr(), outg.data_ptr(), accg.data_ptr(), 2, RS
    L.check(lib.sat_run_ops_parity(C.pointer(og), 1, 0, st()))
    sync()
    for q in range(2):
        assert torch.equal(outg[q], outs[q][0]) and torch.equal(accg[q], outs[q][1])


def test_conv_rs_refuses_other_geometries(lib):
    g = torch.Generator().manual_seed(6)
    for (H, W, Cin, Cout, stride, pad) in [(8, 100, 32, 32, 1, 0), (8, 140, 64, 32, 1, 0), (8, 140, 32, 128, 1, 1), (9, 141, 32, 32, 2, 0)]:
        x = torch.randn(1, H, W, Cin, generator=g)
        w = torch.randn(Cout, 3, 3, Cin, generator=g)
        o, keep, _ = _conv_op(L.SAT_BF16, x, w, stride, pad, stats=False)
        assert lib.sat_conv_default_variant(C.byref(o), -1) != RS
        o.variant = RS                                    # asked for anyway: the launch falls back on a variant that runs it
        L.check(lib.sat_run_ops_parity(C.pointer(o), 1, 0, st()))
        sync()
        ref = F.conv2d(x.bfloat16().double().permute(0, 3, 1, 2), w.bfloat16().double().permute(0, 3, 1, 2), None, stride, pad)
        assert (keep[2].float().cpu().double() - ref.permute(0, 2, 3, 1).reshape(-1, Cout)).abs().max().item() < 0.5


@pytest.mark.parametrize("mode", ["slab", "atomic", "none"])
@pytest.mark.parametrize("N,H,W", [(4, 56, 56), (3, 8, 50), (1, 2, 64), (9, 6, 49)])
def test_conv_rs64_is_bit_identical_to_the_ring_kernel(lib, N, H, W, mode):
    """conv_rs64_kernel (variant 39): conv2 of ResNet's layer 1 -- 3 x 3 / padding 1, 64 -> 64 channels, two output rows per step, the
    column sums per workgroup"""
    Cin = Cout = 64
    g = torch.Generator().manual_seed(N * 7 + W)
    x = (torch.randn(N, Cin, H, W, generator=g) * 1.5 + 0.2).bfloat16().float()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / 24.0).bfloat16().float()
    ref = F.conv2d(x.double(), w.double(), None, 1, 1).permute(0, 2, 3, 1).reshape(-1, Cout)

    def run(v):
        o, keep, _ = _conv_op(L.SAT_BF16, x.permute(0, 2, 3, 1), w.permute(0, 2, 3, 1), 1, 1, stats=(mode == "slab"))
        o.variant = v
        extra = {}
        if mode == "atomic":
            extra["acc"] = torch.zeros(2, 2, Cout, dtype=torch.int64, device="cuda")
            o.stat_acc = extra["acc"].data_ptr()
        ops = (L.SatOp * 1)(o)
        L.check(lib.sat_run_ops_parity(ops, 1, 0, st()))
        sync()
        return keep, extra

    want, wx = run(1)
    got, gx = run(RS64)
    slabs_ok = mode != "slab" or lib.sat_conv_tiles_m(N * H * W) >= min(N * H // 2, 512)
    if not slabs_ok:
        pytest.skip("fewer statistics slabs than workgroups: the library falls back on another variant")
    assert torch.isfinite(got[2].float()).all()
    assert torch.equal(got[2], want[2])
    assert (got[2].float().cpu().double() - ref).abs().max().item() < 3e-2 + 4e-3 * ref.abs().max().item()
    if mode == "slab":
        assert torch.isfinite(got[3]).all()
        torch.testing.assert_close(got[3].sum(0), want[3].sum(0), rtol=1e-4, atol=3e-2)
    elif mode == "atomic":
        torch.testing.assert_close(gx["acc"].double() / 2 ** 22, wx["acc"].double() / 2 ** 22, rtol=1e-4, atol=3e-2)


def test_conv_rs64_default_signature_and_groups(lib):
    N, H, W, C_ = 3, 4, 56, 64
    g = torch.Generator().manual_seed(15)
    xs = [(torch.randn(N, H, W, C_, generator=g) + 0.1).bfloat16() for _ in range(2)]
    w = (torch.randn(C_, 3, 3, C_, generator=g) / 24.0).bfloat16()
    o, keep, _ = _conv_op(L.SAT_BF16, xs[0].float(), w.float(), 1, 1, stats=False)
    assert lib.sat_conv_default_variant(C.byref(o), -1) == RS64 and lib.sat_conv_variant_signature(RS64) == 7001
    outs = []
    for x in xs:
        o1, k1, _ = _conv_op(L.SAT_BF16, x.float(), w.float(), 1, 1, stats=False)
        acc = torch.zeros(2, 2, C_, dtype=torch.int64, device="cuda")
        o1.stat_acc, o1.variant = acc.data_ptr(), RS64
        L.check(lib.sat_run_ops_parity(C.pointer(o1), 1, 0, st()))
        sync()
        outs.append((k1[2].clone(), acc.clone()))
    xg = cu(torch.stack(xs))
    og, kg, _ = _conv_op(L.SAT_BF16, xs[0].float(), w.float(), 1, 1, stats=False)
    outg = torch.full((2, N * H * W, C_), float("nan"), device="cuda", dtype=torch.bfloat16)
    accg = torch.zeros(2, 2, 2, C_, dtype=torch.int64, device="cuda")
    og.in0, og.out, og.stat_acc, og.groups, og.variant = xg.data_ptr(), outg.data_ptr(), accg.data_ptr(), 2, RS64
    L.check(lib.sat_run_ops_parity(C.pointer(og), 1, 0, st()))
    sync()
    for q in range(2):
        assert torch.equal(outg[q], outs[q][0]) and torch.equal(accg[q], outs[q][1])


@pytest.mark.parametrize("mode", ["slab", "atomic", "none"])
@pytest.mark.parametrize("N,H,W", [(2, 299, 299), (3, 9, 261), (1, 3, 321), (5, 21, 259)])
def test_conv_rs8_is_bit_identical_to_the_ring_kernel(lib, N, H, W, mode):
    """conv_rs8_kernel (variant 40): the first conv of the Inception stem -- 3 x 3 / stride 2 over the image's 3 channels padded to 8 -> 32"""
    Cin, Cout = 8, 32
    g = torch.Generator().manual_seed(N * 5 + W)
    x = (torch.randn(N, Cin, H, W, generator=g) * 1.5 + 0.2).bfloat16().float()
    x[:, 3:] = 0.0
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / 5.0).bfloat16().float()
    ref = F.conv2d(x.double(), w.double(), None, 2, 0).permute(0, 2, 3, 1).reshape(-1, Cout)

    def run(v):
        o, keep, _ = _conv_op(L.SAT_BF16, x.permute(0, 2, 3, 1), w.permute(0, 2, 3, 1), 2, 0, stats=(mode == "slab"))
        o.variant = v
        extra = {}
        if mode == "atomic":
            extra["acc"] = torch.zeros(2, 2, Cout, dtype=torch.int64, device="cuda")
            o.stat_acc = extra["acc"].data_ptr()
        ops = (L.SatOp * 1)(o)
        if v == RS8:
            assert lib.sat_conv_default_variant(C.byref(o), -1) == RS8
        L.check(lib.sat_run_ops_parity(ops, 1, 0, st()))
        sync()
        return keep, extra

    want, wx = run(1)
    got, gx = run(RS8)
    assert torch.isfinite(got[2].float()).all()
    assert torch.equal(got[2], want[2])
    assert (got[2].float().cpu().double() - ref).abs().max().item() < 3e-2 + 4e-3 * ref.abs().max().item()
    if mode == "slab":
        assert torch.isfinite(got[3]).all()
        torch.testing.assert_close(got[3].sum(0), want[3].sum(0), rtol=1e-4, atol=3e-2)
    elif mode == "atomic":
        torch.testing.assert_close(gx["acc"].double() / 2 ** 22, wx["acc"].double() / 2 ** 22, rtol=1e-4, atol=3e-2)


@pytest.mark.parametrize("N,H,W,groups", [(3, 224, 224, 1), (2, 160, 192, 1), (1, 130, 250, 1), (2, 224, 224, 2)])
def test_conv_rs_stem_is_bit_identical_to_the_stem_kernel(lib, N, H, W, groups):
    """conv_rs_stem_kernel (variant 41): ResNet's 7 x 7 / stride 2 stem in the op program's layout as a rolling window of seven input
    rows, against conv_stem_kernel (variant 31): output bitwise equal, the column sums (per workgroup here, per tile there) equal in total"""
    g = torch.Generator().manual_seed(N * 7 + W)
    Hp, Wp = H + 6, (W + 8 + 1) // 2 * 2
    Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    x = torch.zeros(groups * N, Hp, Wp, 4)
    x[:, 3:3 + H, 3:3 + W, :3] = torch.randn(groups * N, H, W, 3, generator=g)
    w = torch.zeros(64, 7, 8, 4)
    w[:, :, :7, :3] = torch.randn(64, 7, 7, 3, generator=g) / 12.0
    xd, wd = x.bfloat16().cuda(), w.reshape(64, 224).bfloat16().cuda()
    res = {}
    for v in (31, RS_STEM):
        o, out, part = _stem_op(xd, wd, Ho, Wo, groups)
        o.variant = v
        L.check(lib.sat_run_ops(C.pointer(o), 1, st()))
        sync()
        res[v] = (out.clone(), part.clone())
    assert lib.sat_conv_variant_signature(RS_STEM) == 7002
    assert torch.isfinite(res[RS_STEM][0].float()).all() and torch.isfinite(res[RS_STEM][1]).all()
    assert torch.equal(res[RS_STEM][0], res[31][0])
    torch.testing.assert_close(res[RS_STEM][1].sum(1), res[31][1].sum(1), rtol=1e-4, atol=3e-2)
    if groups == 2:                                      # each group = its own launch
        for q in range(2):
            o, out, part = _stem_op(xd[q * N:(q + 1) * N].contiguous(), wd, Ho, Wo, 1)
            o.variant = RS_STEM
            L.check(lib.sat_run_ops(C.pointer(o), 1, st()))
            sync()
            assert torch.equal(out, res[RS_STEM][0][q * N * Ho * Wo:(q + 1) * N * Ho * Wo])
            assert torch.equal(part[0], res[RS_STEM][1][q])


@pytest.mark.parametrize("kind", ["rs32", "rs64", "rs8", "stem"])
def test_rolling_window_kernels_run_the_inference_epilogue(lib, kind):
    """eval mode: BatchNorm as a fixed affine of the output column + ReLU in the conv's epilogue (scale1 / shift1 / flags bit 0) --
    bitwise the ring kernel's"""
    g = torch.Generator().manual_seed(len(kind) * 3)
    if kind == "stem":
        N, H, W = 2, 224, 224
        Hp, Wp, Ho, Wo = H + 6, 232, 112, 112
        x = torch.zeros(N, Hp, Wp, 4)
        x[:, 3:3 + H, 3:3 + W, :3] = torch.randn(N, H, W, 3, generator=g)
        w = torch.zeros(64, 7, 8, 4)
        w[:, :, :7, :3] = torch.randn(64, 7, 7, 3, generator=g) / 12.0
        xd, wd = x.bfloat16().cuda(), w.reshape(64, 224).bfloat16().cuda()
        Cout, variant, base = 64, RS_STEM, 10

        def make():
            o, out, _ = _stem_op(xd, wd, Ho, Wo)
            o.stat_partial, o.tiles_m = None, 0
            return o, out
    else:
        N, H, W, Cin, Cout, stride, pad, variant = {"rs32": (2, 20, 149, 32, 32, 1, 0, RS), "rs64": (3, 56, 56, 64, 64, 1, 1, RS64),
                                                    "rs8": (2, 41, 299, 8, 32, 2, 0, RS8)}[kind]
        base = 1
        x = (torch.randn(N, H, W, Cin, generator=g) + 0.1).bfloat16().float()
        w = (torch.randn(Cout, 3, 3, Cin, generator=g) / (3.0 * Cin ** 0.5)).bfloat16().float()

        def make():
            o, keep, _ = _conv_op(L.SAT_BF16, x, w, stride, pad, stats=False)
            make.keep = keep
            return o, keep[2]
    sc, sh = cu(torch.rand(Cout, generator=g) + 0.5), cu(torch.randn(Cout, generator=g) * 0.3)
    outs = {}
    for v in (base, variant):
        o, out = make()
        o.scale1, o.shift1, o.flags, o.variant = sc.data_ptr(), sh.data_ptr(), 1, v
        if v == variant:
            assert lib.sat_conv_default_variant(C.byref(o), -1) == variant          # (no table entry: the geometry's default)
        L.check(lib.sat_run_ops(C.pointer(o), 1, st()))
        sync()
        outs[v] = out.clone()
    assert torch.isfinite(outs[variant].float()).all() and float(outs[variant].float().min()) >= 0.0
    assert torch.equal(outs[variant], outs[base])
