"""GPU (MI355X) kernel-level parity: every libsat_hip.so entry point called through the C ABI (ctypes) and
compared with CPU fp32/fp64 arithmetic on the same seeded inputs.  Tolerances are written next to each check."""
import ctypes as C
import importlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

sat = importlib.import_module("show-and-tell_amd")
L = sat._lib


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "needs the MI355X"
    return L.load()


def cu(t):
    return t.cuda()


def st():
    return L.stream()


def sync():
    torch.cuda.synchronize()


# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("amode,bmode", [(0, 0), (0, 1), (2, 1), (2, 0)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (200, 72, 100), (64, 2048, 256), (1216, 512, 64), (33, 20, 8),
                                   (300, 1000, 516)])
def test_gemm_f32(lib, amode, bmode, M, N, K):
    if amode == 2 and M % 4:
        M = (M + 3) // 4 * 4
    if bmode == 1 and N % 4:
        N = (N + 3) // 4 * 4
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(N, K, generator=g)
    bias = torch.randn(N, generator=g)
    bias2 = torch.randn(N, generator=g)
    ref = (A.double() @ B.double().t() + bias.double() + bias2.double())
    Ad = cu(A if amode == 0 else A.t().contiguous())
    Bd = cu(B if bmode == 0 else B.t().contiguous())
    lda = K if amode == 0 else M
    ldb = K if bmode == 0 else N
    Cd = torch.full((M, N), float("nan"), device="cuda")
    b1d, b2d = cu(bias), cu(bias2)          # keep the device buffers alive across the launch
    L.check(lib.sat_gemm_f32(amode, bmode, L.ptr(Ad), lda, L.ptr(Bd), ldb, L.ptr(Cd), N, L.ptr(b1d), L.ptr(b2d),
                             M, N, K, st()))
    sync()
    out = Cd.cpu().double()
    # exact-f32 MFMA: error ~ 1e-7 * sum|a*b|
    tol = 3e-6 * (A.abs().double() @ B.abs().double().t()).max().item()
    assert torch.isfinite(out).all()
    assert (out - ref).abs().max().item() < tol


def _conv_op(dtype, x_nhwc, w_ohwi, stride, pad, stats=True):
    N, H, W, Cin = x_nhwc.shape
    Cout, KH, KW, _ = w_ohwi.shape
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    td = torch.bfloat16 if dtype == L.SAT_BF16 else torch.float32
    xd = cu(x_nhwc.to(td).contiguous())
    wd = cu(w_ohwi.to(td).reshape(Cout, -1).contiguous())
    out = torch.full((N * Ho * Wo, Cout), float("nan"), device="cuda", dtype=td)
    tiles = L.load().sat_conv_tiles_m(N * Ho * Wo)
    part = torch.full((tiles, 2, Cout), float("nan"), device="cuda") if stats else None
    o = L.SatOp()
    o.kind, o.dtype = L.OP_CONV, dtype
    o.in0, o.w, o.out = xd.data_ptr(), wd.data_ptr(), out.data_ptr()
    o.N, o.Hin, o.Win, o.Cin, o.Hout, o.Wout, o.Cout = N, H, W, Cin, Ho, Wo, Cout
    o.KH, o.KW, o.stride, o.pad = KH, KW, stride, pad
    o.sN, o.sH, o.sW = H * W * Cin, W * Cin, Cin
    if stats:
        o.stat_partial, o.tiles_m = part.data_ptr(), tiles
    return o, (xd, wd, out, part), (N, Ho, Wo, Cout)


@pytest.mark.parametrize("dtype", [L.SAT_F32, L.SAT_BF16])
@pytest.mark.parametrize("N,H,W,Cin,Cout,k,stride,pad", [
    (2, 8, 8, 64, 64, 1, 1, 0), (2, 9, 7, 64, 128, 3, 1, 1), (3, 10, 10, 128, 64, 3, 2, 1), (2, 8, 8, 256, 72, 1, 2, 0),
    (2, 12, 12, 8, 16, 3, 1, 1), (1, 6, 6, 16, 8, 1, 1, 0), (2, 7, 7, 24, 40, 3, 2, 1), (4, 14, 14, 256, 256, 3, 1, 1)])
def test_conv_fwd_and_stats(lib, dtype, N, H, W, Cin, Cout, k, stride, pad):
    g = torch.Generator().manual_seed(N * H + Cin + Cout + k)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    if dtype == L.SAT_BF16:          # compare on bf16-rounded operands: what the kernel sees
        x, w = x.bfloat16().float(), w.bfloat16().float()
    ref = F.conv2d(x.double(), w.double(), None, stride, pad).permute(0, 2, 3, 1).reshape(-1, Cout)
    o, keep, (n, ho, wo, co) = _conv_op(dtype, x.permute(0, 2, 3, 1), w.permute(0, 2, 3, 1), stride, pad)
    L.check(lib.sat_run_ops(C.pointer(o), 1, st()))
    sync()
    out = keep[2].float().cpu().double()
    assert torch.isfinite(out).all()
    tol = 2e-5 if dtype == L.SAT_F32 else 2e-2       # bf16 output rounding: 2^-8 relative on |y| <~ 4
    assert (out - ref).abs().max().item() < tol
    part = keep[3].cpu().double()
    np.testing.assert_allclose(part[:, 0].sum(0).numpy(), ref.sum(0).numpy(), rtol=0, atol=1e-3 * ref.shape[0] ** 0.5 + 1e-4)
    np.testing.assert_allclose(part[:, 1].sum(0).numpy(), (ref ** 2).sum(0).numpy(), rtol=2e-4, atol=1e-3)


NUM_CONV_VARIANTS = 41
PLAIN_CONV_VARIANTS = list(range(1, 27))      # ring kernel variants (27..29: conv_xp_kernel, 30: conv_pr_kernel -- their own tests)


@pytest.mark.parametrize("variant", PLAIN_CONV_VARIANTS)
@pytest.mark.parametrize("N,H,W,Cin,Cout,k,stride,pad", [(3, 15, 13, 64, 192, 3, 1, 1), (2, 9, 9, 24, 72, 3, 2, 1),
                                                         (5, 8, 8, 320, 64, 1, 1, 0)])
def test_conv_bf16_every_kernel_variant(lib, variant, N, H, W, Cin, Cout, k, stride, pad):
    """every (tile, ring depth, waves) variant the autotuner may pick computes the same convolution + statistics"""
    g = torch.Generator().manual_seed(variant * 31 + Cin)
    x = (torch.randn(N, Cin, H, W, generator=g)).bfloat16().float()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).bfloat16().float()
    ref = F.conv2d(x.double(), w.double(), None, stride, pad).permute(0, 2, 3, 1).reshape(-1, Cout)
    o, keep, _ = _conv_op(L.SAT_BF16, x.permute(0, 2, 3, 1), w.permute(0, 2, 3, 1), stride, pad)
    o.variant = variant
    L.check(lib.sat_run_ops(C.pointer(o), 1, st()))
    sync()
    out = keep[2].float().cpu().double()
    assert torch.isfinite(out).all()
    assert (out - ref).abs().max().item() < 2e-2
    part = keep[3].cpu().double()
    np.testing.assert_allclose(part[:, 0].sum(0).numpy(), ref.sum(0).numpy(), rtol=0, atol=1e-3 * ref.shape[0] ** 0.5 + 1e-4)
    np.testing.assert_allclose(part[:, 1].sum(0).numpy(), (ref ** 2).sum(0).numpy(), rtol=2e-4, atol=1e-3)


@pytest.mark.parametrize("stat_mode", ["slab", "atomic"])
@pytest.mark.parametrize("affine", [False, True])
@pytest.mark.parametrize("variant,N,H,W,Cin,Cout", [
    (27, 5, 12, 12, 256, 1024), (28, 5, 12, 12, 256, 1024), (29, 3, 14, 14, 256, 1024), (27, 2, 16, 16, 128, 512),
    (28, 7, 9, 9, 128, 512), (29, 2, 28, 28, 128, 512), (27, 3, 20, 20, 64, 256), (28, 1, 30, 30, 64, 256), (28, 64, 14, 14, 256, 1024)])
def test_conv_expansion_1x1_register_resident_panel_is_bit_identical(lib, variant, N, H, W, Cin, Cout, affine, stat_mode):
    """conv_xp_kernel (variants 27-29: 1 / 2 / 4 column tiles per workgroup, the A panel of an expansion 1x1 conv held in MFMA
    fragment registers, the operand's BatchNorm + ReLU applied there once) against the 128x128-tile ring kernel (variant 3) on
    the same op: output tensor BITWISE equal, per-tile statistics slabs and integer-atomic sums equal to rounding; ragged M (rows past the last
    full 128-row tile), the derive-from-sums table with its running-statistics update; models.py:27."""
    g = torch.Generator().manual_seed(variant * 7 + Cin + N)
    x = (torch.randn(N, H, W, Cin, generator=g) * 1.5 + 0.2).bfloat16()
    w = (torch.randn(Cout, Cin, generator=g) / Cin ** 0.5).bfloat16()
    gamma, beta = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.2
    xf = x.float().reshape(-1, Cin).double()
    M = xf.shape[0]

    def run(v):
        o, keep, _ = _conv_op(L.SAT_BF16, x.float(), w.float().reshape(Cout, 1, 1, Cin), 1, 0, stats=(stat_mode == "slab"))
        o.variant = v
        extra = {}
        if stat_mode == "atomic":
            acc = torch.zeros(2, 2, Cout, dtype=torch.int64, device="cuda")
            o.stat_acc = acc.data_ptr()
            extra["acc"] = acc
        if affine:
            iacc = torch.zeros(2, 2, Cin, dtype=torch.int64, device="cuda")
            iacc[0, 0] = torch.round(xf.sum(0) * 4194304.0).long().cuda()
            iacc[0, 1] = torch.round((xf ** 2).sum(0) * 4194304.0).long().cuda()
            iacc[1] = 777
            gd, bd, rm, rv = cu(gamma), cu(beta), cu(torch.zeros(Cin)), cu(torch.ones(Cin))
            o.stat_acc1, o.gamma1, o.beta1 = iacc.data_ptr(), gd.data_ptr(), bd.data_ptr()
            o.running_mean1, o.running_var1 = rm.data_ptr(), rv.data_ptr()
            o.count, o.momentum, o.eps = M, 0.1, 1e-5
            extra.update(iacc=iacc, gd=gd, bd=bd, rm=rm, rv=rv)
        L.check(lib.sat_run_ops_parity(C.pointer(o), 1, 0, st()))
        sync()
        return keep, extra

    want, wx = run(3)
    got, gx = run(variant)
    assert torch.isfinite(got[2].float()).all()
    assert torch.equal(got[2], want[2])
    # the column sums pair even / odd rows (two rows per packed instruction): the same numbers in another fixed order
    if stat_mode == "slab":
        torch.testing.assert_close(got[3], want[3], rtol=2e-5, atol=2e-4)
    else:
        torch.testing.assert_close(gx["acc"].double() / 2 ** 22, wx["acc"].double() / 2 ** 22, rtol=2e-5, atol=2e-3)
        assert int(gx["acc"][0].abs().sum()) > 0
    if affine:
        assert int(gx["iacc"][1].abs().sum()) == 0
        assert torch.equal(gx["rm"], wx["rm"]) and torch.equal(gx["rv"], wx["rv"])
    # and against the f64 definition
    a = x.float().reshape(-1, Cin)
    if affine:
        mean, var = xf.mean(0), xf.var(0, unbiased=False)
        scale = (gamma.double() / torch.sqrt(var + 1e-5)).float()
        shift = (beta.double() - mean * scale.double()).float()
        a = torch.clamp(a * scale + shift, min=0).bfloat16().float()
    ref = a.double() @ w.float().double().t()
    assert (got[2].float().cpu().double() - ref).abs().max().item() < 3e-2 + 4e-3 * ref.abs().max().item()


@pytest.mark.parametrize("stat_mode", ["slab", "atomic"])
@pytest.mark.parametrize("N,H,W,Cin,Cout", [(3, 15, 13, 64, 192), (64, 14, 14, 256, 256), (5, 28, 28, 128, 128), (7, 7, 7, 512, 136),
                                            (2, 5, 31, 64, 128), (1, 3, 3, 128, 256)])
def test_conv3x3_lds_resident_patch_matches_the_ring_kernel(lib, N, H, W, Cin, Cout, stat_mode):
    """conv_pr_kernel (variant 30: 3x3 / stride 1 / pad 1 with the input patch of a 128-row tile resident in LDS, only the weights
    streaming) against the f64 definition and against the ring kernel (variant 1): the K axis is walked channel-block major,
    so outputs agree to f32 summation order (then one bf16 rounding), not bitwise; image borders, rows of two images in one tile,
    ragged M and N, the widest supported image (W = 31); models.py:27."""
    g = torch.Generator().manual_seed(N * 17 + W + Cin)
    x = torch.randn(N, Cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).bfloat16().float()
    ref = F.conv2d(x.double(), w.double(), None, 1, 1).permute(0, 2, 3, 1).reshape(-1, Cout)

    def run(v):
        o, keep, _ = _conv_op(L.SAT_BF16, x.permute(0, 2, 3, 1), w.permute(0, 2, 3, 1), 1, 1, stats=(stat_mode == "slab"))
        o.variant = v
        acc = None
        if stat_mode == "atomic":
            acc = torch.zeros(2, 2, Cout, dtype=torch.int64, device="cuda")
            o.stat_acc = acc.data_ptr()
        L.check(lib.sat_run_ops_parity(C.pointer(o), 1, 0, st()))
        sync()
        return keep, acc

    want, wacc = run(1)
    got, gacc = run(30)
    out = got[2].float().cpu().double()
    assert torch.isfinite(out).all()
    assert (out - ref).abs().max().item() < 2e-2
    # one bf16 ulp at most between the two kernels, and only on a small fraction of the elements
    d = (got[2].float() - want[2].float()).abs()
    assert d.max().item() <= 2.0 ** -6 * max(1.0, want[2].float().abs().max().item())
    assert (d > 0).float().mean().item() < 0.05
    if stat_mode == "slab":
        torch.testing.assert_close(got[3], want[3], rtol=1e-4, atol=2e-3)
    else:
        torch.testing.assert_close(gacc.double() / 2 ** 22, wacc.double() / 2 ** 22, rtol=1e-4, atol=5e-3)


@pytest.mark.parametrize("derive", [False, True])
@pytest.mark.parametrize("N,H,W,Cin,Cout", [(64, 14, 14, 256, 256), (5, 28, 28, 128, 128), (3, 9, 13, 64, 192), (2, 7, 7, 512, 136)])
def test_conv3x3_lds_resident_patch_with_fused_input_bn_relu(lib, N, H, W, Cin, Cout, derive):
    """conv_pr_kernel with the operand's BatchNorm + ReLU (bn1 of a bottleneck) applied to each 64-channel slice of the LDS-resident
    patch by the loader waves: against relu(bn(x)) -> bf16 -> conv in f64 (outputs and the per-tile statistics slabs);
    zero padding stays zero (shift != 0), table precomputed or derived from the producer's integer sums (running statistics
    updated once, the other parity cleared); models.py:27."""
    g = torch.Generator().manual_seed(N * 13 + W + Cin)
    x = (torch.randn(N, Cin, H, W, generator=g) * 1.5 + 0.2).bfloat16()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).bfloat16().float()
    gamma, beta = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.2 + 0.3
    xf = x.float().permute(0, 2, 3, 1).reshape(-1, Cin).double()
    M = xf.shape[0]
    mean, var = xf.mean(0), xf.var(0, unbiased=False)
    scale = (gamma.double() / torch.sqrt(var + 1e-5)).float()
    shift = (beta.double() - mean * scale.double()).float()
    a = torch.clamp(x.float() * scale[None, :, None, None] + shift[None, :, None, None], min=0).bfloat16().float()
    ref = F.conv2d(a.double(), w.double(), None, 1, 1).permute(0, 2, 3, 1).reshape(-1, Cout)

    def run(v):
        o, keep, _ = _conv_op(L.SAT_BF16, x.float().permute(0, 2, 3, 1), w.permute(0, 2, 3, 1), 1, 1)
        o.variant = v
        extra = {}
        if derive:
            iacc = torch.zeros(2, 2, Cin, dtype=torch.int64, device="cuda")
            iacc[0, 0] = torch.round(xf.sum(0) * 4194304.0).long().cuda()
            iacc[0, 1] = torch.round((xf ** 2).sum(0) * 4194304.0).long().cuda()
            iacc[1] = 777
            gd, bd, rm, rv = cu(gamma), cu(beta), cu(torch.zeros(Cin)), cu(torch.ones(Cin))
            o.stat_acc1, o.gamma1, o.beta1 = iacc.data_ptr(), gd.data_ptr(), bd.data_ptr()
            o.running_mean1, o.running_var1 = rm.data_ptr(), rv.data_ptr()
            o.count, o.momentum, o.eps = M, 0.1, 1e-5
            extra.update(iacc=iacc, gd=gd, bd=bd, rm=rm, rv=rv)
        else:
            sd, td = cu(scale), cu(shift)
            o.scale0, o.shift0 = sd.data_ptr(), td.data_ptr()
            extra.update(sd=sd, td=td)
        L.check(lib.sat_run_ops_parity(C.pointer(o), 1, 0, st()))
        sync()
        return keep, extra

    got, gx = run(30)
    again, ax = run(0)                                 # the heuristic picks the same kernel for a 3x3 conv with a fused input BatchNorm
    out = got[2].float().cpu().double()
    assert torch.isfinite(out).all()
    assert (out - ref).abs().max().item() < 3e-2 + 4e-3 * ref.abs().max().item()
    assert torch.equal(got[2], again[2]) and torch.equal(got[3], again[3])
    part = got[3].cpu().double()
    np.testing.assert_allclose(part[:, 0].sum(0).numpy(), ref.sum(0).numpy(), rtol=0, atol=2e-3 * ref.shape[0] ** 0.5 + 1e-3)
    np.testing.assert_allclose(part[:, 1].sum(0).numpy(), (ref ** 2).sum(0).numpy(), rtol=5e-4, atol=5e-3)
    if derive:
        assert int(gx["iacc"][1].abs().sum()) == 0
        np.testing.assert_allclose(gx["rm"].cpu().numpy(), (0.1 * mean).numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(gx["rv"].cpu().numpy(), (0.9 + 0.1 * var * M / (M - 1)).numpy(), rtol=1e-4)



def _pack_weights(lib, wd, Cout, Cin, taps):
    """the fragment-ordered copy of bf16 weights [Cout][taps*Cin] that conv_pw_kernel streams into registers"""
    packed = torch.empty_like(wd)
    L.check(lib.sat_conv_pack_weights(wd.data_ptr(), packed.data_ptr(), Cout, Cin, taps, st()))
    return packed


def test_pack_weights_is_the_mfma_fragment_order(lib):
    """sat_conv_pack_weights: packed[nb][cb][tap][ks][lane = h*32 + r][e] = w[32 nb + r][tap][64 cb + 16 ks + 8 h + e]"""
    Cout, Cin, taps = 64, 128, 9
    w = torch.arange(Cout * taps * Cin, dtype=torch.float32).reshape(Cout, taps, Cin) % 251
    wd = cu(w.bfloat16().reshape(Cout, -1).contiguous())
    got = _pack_weights(lib, wd, Cout, Cin, taps).float().cpu().reshape(Cout // 32, Cin // 64, taps, 4, 2, 32, 8)
    sync()
    src = w.bfloat16().float().reshape(Cout // 32, 32, taps, Cin // 64, 4, 2, 8)          # nb r tap cb ks h e
    assert torch.equal(got, src.permute(0, 3, 2, 4, 5, 1, 6).contiguous())
    assert lib.sat_conv_pack_weights(wd.data_ptr(), wd.data_ptr(), 48, Cin, taps, st()) != 0      # Cout % 32
    assert lib.sat_conv_pack_weights(wd.data_ptr(), wd.data_ptr(), Cout, 96, taps, st()) != 0      # Cin % 64


@pytest.mark.parametrize("stat_mode", ["slab", "atomic", "eval"])
@pytest.mark.parametrize("N,H,W,Cin,Cout", [(64, 14, 14, 256, 256), (5, 28, 28, 128, 128), (7, 7, 7, 512, 128), (2, 5, 31, 64, 128),
                                            (1, 3, 3, 128, 256), (3, 15, 13, 64, 384)])
def test_conv3x3_weights_in_registers_is_bit_identical_to_the_lds_patch_kernel(lib, N, H, W, Cin, Cout, stat_mode):
    """conv_pw_kernel (variant 32: the input patch in LDS, the weights streamed straight into registers from the fragment-ordered
    copy, two workgroups per CU) against conv_pr_kernel (variant 30: same K order, so the output is BITWISE equal; the column sums
    agree to rounding) and the f64 definition; `eval`: the fixed BatchNorm affine + ReLU in the epilogue (against the ring kernel's);
    image borders, rows of several images in one tile, a ragged last tile, one to eight channel blocks; models.py:27."""
    g = torch.Generator().manual_seed(N * 19 + W + Cin)
    x = torch.randn(N, Cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).bfloat16().float()
    ref = F.conv2d(x.double(), w.double(), None, 1, 1).permute(0, 2, 3, 1).reshape(-1, Cout)
    osc, osh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.3

    def run(v):
        o, keep, _ = _conv_op(L.SAT_BF16, x.permute(0, 2, 3, 1), w.permute(0, 2, 3, 1), 1, 1, stats=(stat_mode == "slab"))
        o.variant = v
        extra = [_pack_weights(lib, keep[1], Cout, Cin, 9)]
        o.w_packed = extra[0].data_ptr()
        acc = None
        if stat_mode == "atomic":
            acc = torch.zeros(2, 2, Cout, dtype=torch.int64, device="cuda")
            o.stat_acc = acc.data_ptr()
        if stat_mode == "eval":
            extra += [cu(osc), cu(osh)]
            o.scale1, o.shift1, o.flags = extra[1].data_ptr(), extra[2].data_ptr(), 1
        L.check(lib.sat_run_ops_parity(C.pointer(o), 1, 0, st()))
        sync()
        return keep, acc

    want, wacc = run(1 if stat_mode == "eval" else 30)
    got, gacc = run(32)
    out = got[2].float().cpu().double()
    assert torch.isfinite(out).all()
    if stat_mode == "eval":
        ref = torch.clamp(ref * osc.double() + osh.double(), min=0)
        d = (got[2].float() - want[2].float()).abs()            # the ring kernel walks K tap major: equal to f32 summation order
        assert d.max().item() <= 2.0 ** -6 * max(1.0, want[2].float().abs().max().item())
        assert (d > 0).float().mean().item() < 0.05
    else:
        assert torch.equal(got[2], want[2])
    assert (out - ref).abs().max().item() < 3e-2
    if stat_mode == "slab":
        torch.testing.assert_close(got[3], want[3], rtol=1e-4, atol=2e-3)
    elif stat_mode == "atomic":
        torch.testing.assert_close(gacc.double() / 2 ** 22, wacc.double() / 2 ** 22, rtol=1e-4, atol=5e-3)
        assert int(gacc[1].abs().sum()) == 0


@pytest.mark.parametrize("derive", [False, True])
@pytest.mark.parametrize("N,H,W,Cin,Cout", [(64, 14, 14, 256, 256), (5, 28, 28, 128, 128), (3, 9, 13, 64, 256), (2, 7, 7, 512, 128)])
def test_conv3x3_weights_in_registers_with_fused_input_bn_relu(lib, N, H, W, Cin, Cout, derive):
    """conv_pw_kernel with the operand's BatchNorm + ReLU (bn1 of a bottleneck) applied to each 64-channel slice of the patch in LDS:
    output and statistics slabs BITWISE / to rounding those of conv_pr_kernel (same transform, same K order), running statistics
    updated once and the other parity cleared when the table is derived from the producer's integer sums; models.py:27."""
    g = torch.Generator().manual_seed(N * 13 + W + Cin)
    x = (torch.randn(N, Cin, H, W, generator=g) * 1.5 + 0.2).bfloat16()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).bfloat16().float()
    gamma, beta = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.2 + 0.3
    xf = x.float().permute(0, 2, 3, 1).reshape(-1, Cin).double()
    M = xf.shape[0]
    mean, var = xf.mean(0), xf.var(0, unbiased=False)
    scale = (gamma.double() / torch.sqrt(var + 1e-5)).float()
    shift = (beta.double() - mean * scale.double()).float()

    def run(v):
        o, keep, _ = _conv_op(L.SAT_BF16, x.float().permute(0, 2, 3, 1), w.permute(0, 2, 3, 1), 1, 1)
        o.variant = v
        extra = {"wp": _pack_weights(lib, keep[1], Cout, Cin, 9)}
        o.w_packed = extra["wp"].data_ptr()
        if derive:
            iacc = torch.zeros(2, 2, Cin, dtype=torch.int64, device="cuda")
            iacc[0, 0] = torch.round(xf.sum(0) * 4194304.0).long().cuda()
            iacc[0, 1] = torch.round((xf ** 2).sum(0) * 4194304.0).long().cuda()
            iacc[1] = 777
            gd, bd, rm, rv = cu(gamma), cu(beta), cu(torch.zeros(Cin)), cu(torch.ones(Cin))
            o.stat_acc1, o.gamma1, o.beta1 = iacc.data_ptr(), gd.data_ptr(), bd.data_ptr()
            o.running_mean1, o.running_var1 = rm.data_ptr(), rv.data_ptr()
            o.count, o.momentum, o.eps = M, 0.1, 1e-5
            extra.update(iacc=iacc, gd=gd, bd=bd, rm=rm, rv=rv)
        else:
            sd, td = cu(scale), cu(shift)
            o.scale0, o.shift0 = sd.data_ptr(), td.data_ptr()
            extra.update(sd=sd, td=td)
        L.check(lib.sat_run_ops_parity(C.pointer(o), 1, 0, st()))
        sync()
        return keep, extra

    want, wx = run(30)
    got, gx = run(32)
    assert torch.isfinite(got[2].float()).all()
    assert torch.equal(got[2], want[2])
    torch.testing.assert_close(got[3], want[3], rtol=1e-4, atol=2e-3)
    if derive:
        assert int(gx["iacc"][1].abs().sum()) == 0
        assert torch.equal(gx["rm"], wx["rm"]) and torch.equal(gx["rv"], wx["rv"])



@pytest.mark.parametrize("variant", [33, 34])
@pytest.mark.parametrize("mode", ["slab", "atomic", "eval", "eval_residual"])
@pytest.mark.parametrize("N,H,W,Cin,Cout,stride", [(64, 14, 14, 1024, 256, 1), (5, 28, 28, 128, 512, 1), (7, 7, 7, 512, 2048, 1), (3, 9, 13, 64, 128, 1),
                                                   (4, 14, 14, 256, 512, 2), (2, 15, 13, 128, 128, 2), (1, 3, 3, 320, 256, 1)])
def test_conv1x1_weights_in_registers_is_bit_identical_to_the_ring_kernel(lib, N, H, W, Cin, Cout, stride, mode, variant):
    """conv_aw_kernel (variant 33: 1x1 convs with the activations staged global -> registers -> LDS and the weights streamed straight
    into registers from the fragment-ordered copy, two workgroups per CU; variant 34: eight waves, 256-column tiles, the activations
    staged once per row tile) against the ring kernel (variant 1: same K order, so the
    output is BITWISE equal; column sums to rounding) and the f64 definition; stride 2 (the projection shortcuts), ragged last tiles,
    1 to 16 K-steps (the tail of the three-step pipeline), the inference epilogue with and without the residual; models.py:27."""
    if variant == 34 and Cout % 256:
        pytest.skip("the eight-wave form needs Cout % 256 == 0")
    g = torch.Generator().manual_seed(N * 23 + W + Cin + stride)
    x = torch.randn(N, Cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5).bfloat16().float()
    ref = F.conv2d(x.double(), w.double(), None, stride, 0).permute(0, 2, 3, 1).reshape(-1, Cout)
    osc, osh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.3
    res = torch.randn(ref.shape[0], Cout, generator=g).bfloat16()

    def run(v):
        o, keep, _ = _conv_op(L.SAT_BF16, x.permute(0, 2, 3, 1), w.permute(0, 2, 3, 1), stride, 0, stats=(mode == "slab"))
        o.variant = v
        extra = [_pack_weights(lib, keep[1], Cout, Cin, 1)]
        o.w_packed = extra[0].data_ptr()
        acc = None
        if mode == "atomic":
            acc = torch.zeros(2, 2, Cout, dtype=torch.int64, device="cuda")
            o.stat_acc = acc.data_ptr()
        if mode.startswith("eval"):
            extra += [cu(osc), cu(osh)]
            o.scale1, o.shift1, o.flags = extra[1].data_ptr(), extra[2].data_ptr(), 1
            if mode == "eval_residual":
                extra.append(cu(res))
                o.in1 = extra[3].data_ptr()
        L.check(lib.sat_run_ops_parity(C.pointer(o), 1, 0, st()))
        sync()
        return keep, acc

    want, wacc = run(1)
    got, gacc = run(variant)
    assert torch.isfinite(got[2].float()).all()
    assert torch.equal(got[2], want[2])
    want_ref = ref
    if mode.startswith("eval"):
        want_ref = ref * osc.double() + osh.double()
        if mode == "eval_residual":
            want_ref = want_ref.bfloat16().double() + res.double()
        want_ref = torch.clamp(want_ref, min=0)
    assert (got[2].float().cpu().double() - want_ref).abs().max().item() < 4e-2
    if mode == "slab":
        torch.testing.assert_close(got[3], want[3], rtol=1e-4, atol=2e-3)
    elif mode == "atomic":
        torch.testing.assert_close(gacc.double() / 2 ** 22, wacc.double() / 2 ** 22, rtol=1e-4, atol=5e-3)
        assert int(gacc[1].abs().sum()) == 0


@pytest.mark.parametrize("variant", [33, 34])
@pytest.mark.parametrize("derive", [False, True])
@pytest.mark.parametrize("N,H,W,Cin,Cout", [(64, 14, 14, 256, 1024), (5, 28, 28, 128, 512), (16, 7, 7, 512, 2048), (3, 9, 13, 64, 256)])
def test_conv1x1_weights_in_registers_with_fused_input_bn_relu(lib, N, H, W, Cin, Cout, derive, variant):
    """conv_aw_kernel with the operand's BatchNorm + ReLU (bn2 in front of conv3) applied to the activation registers on their way
    to LDS: output BITWISE that of the ring kernel's in-LDS transform (variant 1; same scalar form, same K order), statistics slabs
    to rounding; rows past M (ragged last tile: 16 x 49 rows) stay zero in the column sums; table precomputed or derived from the
    producer's integer sums (running statistics updated once, the other parity cleared); models.py:27."""
    g = torch.Generator().manual_seed(N * 13 + W + Cin)
    x = (torch.randn(N, Cin, H, W, generator=g) * 1.5 + 0.2).bfloat16()
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5).bfloat16().float()
    gamma, beta = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.2 + 0.3
    xf = x.float().permute(0, 2, 3, 1).reshape(-1, Cin).double()
    M = xf.shape[0]
    mean, var = xf.mean(0), xf.var(0, unbiased=False)
    scale = (gamma.double() / torch.sqrt(var + 1e-5)).float()
    shift = (beta.double() - mean * scale.double()).float()
    a = torch.clamp(x.float() * scale[None, :, None, None] + shift[None, :, None, None], min=0).bfloat16().float()
    ref = F.conv2d(a.double(), w.double()).permute(0, 2, 3, 1).reshape(-1, Cout)

    def run(v):
        o, keep, _ = _conv_op(L.SAT_BF16, x.float().permute(0, 2, 3, 1), w.permute(0, 2, 3, 1), 1, 0)
        o.variant = v
        extra = {"wp": _pack_weights(lib, keep[1], Cout, Cin, 1)}
        o.w_packed = extra["wp"].data_ptr()
        if derive:
            iacc = torch.zeros(2, 2, Cin, dtype=torch.int64, device="cuda")
            iacc[0, 0] = torch.round(xf.sum(0) * 4194304.0).long().cuda()
            iacc[0, 1] = torch.round((xf ** 2).sum(0) * 4194304.0).long().cuda()
            iacc[1] = 777
            gd, bd, rm, rv = cu(gamma), cu(beta), cu(torch.zeros(Cin)), cu(torch.ones(Cin))
            o.stat_acc1, o.gamma1, o.beta1 = iacc.data_ptr(), gd.data_ptr(), bd.data_ptr()
            o.running_mean1, o.running_var1 = rm.data_ptr(), rv.data_ptr()
            o.count, o.momentum, o.eps = M, 0.1, 1e-5
            extra.update(iacc=iacc, gd=gd, bd=bd, rm=rm, rv=rv)
        else:
            sd, td = cu(scale), cu(shift)
            o.scale0, o.shift0 = sd.data_ptr(), td.data_ptr()
            extra.update(sd=sd, td=td)
        L.check(lib.sat_run_ops_parity(C.pointer(o), 1, 0, st()))
        sync()
        return keep, extra

    want, wx = run(1)
    got, gx = run(variant)
    assert torch.isfinite(got[2].float()).all()
    assert torch.equal(got[2], want[2])
    assert (got[2].float().cpu().double() - ref).abs().max().item() < 3e-2 + 4e-3 * ref.abs().max().item()
    torch.testing.assert_close(got[3], want[3], rtol=1e-4, atol=2e-3)
    if derive:
        assert int(gx["iacc"][1].abs().sum()) == 0
        assert torch.equal(gx["rm"], wx["rm"]) and torch.equal(gx["rv"], wx["rv"])


def _stem_op(x_pad, w, Ho, Wo, groups=1):
    """the op program's stem conv (resnet.ConvStackProgram): a 7 x 1 kernel over rows of 8 pixels x 4 channels of a zero-bordered
    NHWC4 image, stride 2; x_pad bf16 [G*N][Hp][Wp][4] on the device, w bf16 [64][224]"""
    GN, Hp, Wp, _ = x_pad.shape
    N = GN // groups
    out = torch.full((GN * Ho * Wo, 64), float("nan"), device="cuda", dtype=torch.bfloat16)
    tiles = L.load().sat_conv_tiles_m(N * Ho * Wo)
    part = torch.full((groups, tiles, 2, 64), float("nan"), device="cuda")
    o = L.SatOp()
    o.kind, o.dtype, o.groups = L.OP_CONV, L.SAT_BF16, groups
    o.in0, o.w, o.out = x_pad.data_ptr(), w.data_ptr(), out.data_ptr()
    o.N, o.Hin, o.Win, o.Cin, o.Hout, o.Wout, o.Cout = N, Hp, Wp, 32, Ho, Wo, 64
    o.KH, o.KW, o.stride, o.pad = 7, 1, 2, 0
    o.sN, o.sH, o.sW = Hp * Wp * 4, Wp * 4, 4
    o.stat_partial, o.tiles_m = part.data_ptr(), tiles
    return o, out, part


@pytest.mark.parametrize("N,H,W", [(3, 224, 224), (2, 160, 192), (1, 130, 128), (5, 224, 224)])
def test_conv_stem_kernel_is_bit_identical_to_the_ring_kernel(lib, N, H, W):
    """conv_stem_kernel (variant 31: persistent workgroups, weights in registers, input row segments in LDS) on the op program's
    stem layout against the ring kernel (variant 10) and the f64 definition: output BITWISE equal (same MFMA, same K order),
    per-tile statistics slabs equal to rounding; tiles that span two and three (image, output row) pairs, a ragged last tile"""
    g = torch.Generator().manual_seed(N * 7 + W)
    Hp, Wp = H + 6, (W + 8 + 1) // 2 * 2
    Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    x = torch.zeros(N, Hp, Wp, 4)
    x[:, 3:3 + H, 3:3 + W, :3] = torch.randn(N, H, W, 3, generator=g)
    w = torch.zeros(64, 7, 8, 4)
    w[:, :, :7, :3] = torch.randn(64, 7, 7, 3, generator=g) / 12.0
    xd, wd = x.bfloat16().cuda(), w.reshape(64, 224).bfloat16().cuda()
    res = {}
    for v in (10, 31, 0):                               # 0: the heuristic must pick the stem kernel for this layout
        o, out, part = _stem_op(xd, wd, Ho, Wo)
        o.variant = v
        L.check(lib.sat_run_ops(C.pointer(o), 1, st()))
        sync()
        res[v] = (out.clone(), part.clone())
    assert torch.isfinite(res[31][0].float()).all()
    assert torch.equal(res[31][0], res[10][0]) and torch.equal(res[0][0], res[31][0])
    torch.testing.assert_close(res[31][1], res[10][1], rtol=2e-5, atol=2e-4)
    assert torch.equal(res[0][1], res[31][1])
    # f64 definition on the same bf16 operands: windows of 8 pixels x 4 channels per kernel row
    xf, wf = xd.float().cpu().double(), wd.float().cpu().double().reshape(64, 7, 32)
    rows = torch.stack([xf[:, kh:kh + 2 * Ho:2].reshape(N, Ho, Wp * 4) for kh in range(7)], 2)      # [N, Ho, 7, Wp*4]
    win = torch.stack([rows[..., 8 * wo:8 * wo + 32] for wo in range(Wo)], 2)                        # [N, Ho, Wo, 7, 32]
    ref = torch.einsum("nhwkc,okc->nhwo", win, wf).reshape(-1, 64)
    assert (res[31][0].float().cpu().double() - ref).abs().max().item() < 2e-2
    np.testing.assert_allclose(res[31][1][0, :, 0].cpu().double().sum(0).numpy(), ref.sum(0).numpy(), rtol=0, atol=2e-3 * ref.shape[0] ** 0.5 + 1e-3)


@pytest.mark.parametrize("kind", ["ring", "ring_wide", "xp", "pr", "pw", "aw", "aw8", "stem"])
def test_grouped_conv_launch_equals_one_launch_per_batch(lib, kind):
    """sat_op.groups = 3: three batches in ONE launch (grid.y = group; activations, statistics slabs / integer accumulators and the
    operand-BatchNorm accumulators + running-statistics log moved by their group strides, weights shared) against three launches,
    one per batch, of the same variant: outputs, slabs, integer sums and the running-statistics log BITWISE equal per batch --
    every kernel family of the conv stack (models.py:27 under the look-ahead of DESIGN 3.1d)"""
    G = 3
    g = torch.Generator().manual_seed(len(kind) * 13)
    if kind == "stem":
        N, H, W = 2, 224, 224
        Hp, Wp, Ho, Wo = H + 6, 232, 112, 112
        x = torch.zeros(G * N, Hp, Wp, 4)
        x[:, 3:3 + H, 3:3 + W, :3] = torch.randn(G * N, H, W, 3, generator=g)
        w = torch.zeros(64, 7, 8, 4)
        w[:, :, :7, :3] = torch.randn(64, 7, 7, 3, generator=g) / 12.0
        xd, wd = x.bfloat16().cuda(), w.reshape(64, 224).bfloat16().cuda()
        og, outg, partg = _stem_op(xd, wd, Ho, Wo, groups=G)
        og.variant = 31
        L.check(lib.sat_run_ops(C.pointer(og), 1, st()))
        sync()
        M = N * Ho * Wo
        for k in range(G):
            o1, out1, part1 = _stem_op(xd[k * N:(k + 1) * N].contiguous(), wd, Ho, Wo)
            o1.variant = 31
            L.check(lib.sat_run_ops(C.pointer(o1), 1, st()))
            sync()
            assert torch.equal(outg[k * M:(k + 1) * M], out1) and torch.equal(partg[k], part1[0])
        return
    geo = {"ring": (4, 13, 11, 128, 192, 1, 2), "ring_wide": (4, 14, 14, 256, 512, 1, 22), "xp": (5, 12, 12, 256, 1024, 1, 28),
           "pr": (5, 14, 14, 128, 256, 3, 30), "pw": (5, 14, 14, 128, 256, 3, 32), "aw": (5, 12, 12, 256, 1024, 1, 33), "aw8": (5, 12, 12, 256, 1024, 1, 34)}[kind]
    N, H, W, Cin, Cout, k, variant = geo
    pad = 1 if k == 3 else 0
    x = (torch.randn(G * N, H, W, Cin, generator=g) * 1.5 + 0.2).bfloat16()
    w = (torch.randn(Cout, k, k, Cin, generator=g) / (Cin * k * k) ** 0.5).bfloat16()
    gamma, beta = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.2
    M = N * H * W
    with_in_bn = kind in ("xp", "pr", "pw", "aw", "aw8")

    def run(xs, groups):
        o, keep, _ = _conv_op(L.SAT_BF16, xs.float(), w.float(), 1, pad, stats=False)
        o.N, o.groups, o.variant = N, groups, variant
        acc = torch.zeros(groups, 2, 2, Cout, dtype=torch.int64, device="cuda")
        o.stat_acc = acc.data_ptr()
        extra = {"acc": acc}
        if kind in ("pw", "aw", "aw8"):
            extra["wp"] = _pack_weights(lib, keep[1], Cout, Cin, k * k)
            o.w_packed = extra["wp"].data_ptr()
        if with_in_bn:
            iacc = torch.zeros(groups, 2, 2, Cin, dtype=torch.int64, device="cuda")
            for q in range(groups):
                xf = xs[q * N:(q + 1) * N].float().reshape(-1, Cin).double()
                iacc[q, 0, 0] = torch.round(xf.sum(0) * 4194304.0).long().cuda()
                iacc[q, 0, 1] = torch.round((xf ** 2).sum(0) * 4194304.0).long().cuda()
            iacc[:, 1] = 777
            gd, bd, log = cu(gamma), cu(beta), torch.zeros(groups, 2, Cin, device="cuda")
            o.stat_acc1, o.gamma1, o.beta1 = iacc.data_ptr(), gd.data_ptr(), bd.data_ptr()
            o.running_mean1, o.running_var1 = log[0, 0].data_ptr(), log[0, 1].data_ptr()
            o.count, o.momentum, o.eps = M, 1.0, 1e-5
            extra.update(iacc=iacc, gd=gd, bd=bd, log=log)
        L.check(lib.sat_run_ops_parity(C.pointer(o), 1, 0, st()))
        sync()
        return keep[2], extra

    outg, xg = run(x, G)
    assert torch.isfinite(outg.float()).all()
    for q in range(G):
        out1, x1 = run(x[q * N:(q + 1) * N].contiguous(), 1)
        assert torch.equal(outg[q * M:(q + 1) * M], out1[:M]), q
        assert torch.equal(xg["acc"][q], x1["acc"][0]), q
        if with_in_bn:
            assert torch.equal(xg["log"][q], x1["log"][0]) and int(xg["iacc"][q, 1].abs().sum()) == 0
    if with_in_bn:                                      # the batches differ, so do their operand statistics
        assert not torch.equal(xg["log"][0], xg["log"][1])


def test_conv_autotune_sets_a_variant_and_keeps_results(lib):
    g = torch.Generator().manual_seed(77)
    x = torch.randn(4, 128, 14, 14, generator=g).bfloat16().float()
    w = (torch.randn(256, 128, 3, 3, generator=g) / 34.0).bfloat16().float()
    ref = F.conv2d(x.double(), w.double(), None, 1, 1).permute(0, 2, 3, 1).reshape(-1, 256)
    o, keep, _ = _conv_op(L.SAT_BF16, x.permute(0, 2, 3, 1), w.permute(0, 2, 3, 1), 1, 1)
    ops = (L.SatOp * 1)(o)
    scratch = torch.empty(4096, device="cuda")          # the tuner works in caller-owned memory
    assert lib.sat_conv_autotune(ops, 1, 2, None, 0, st()) == 1002
    L.check(lib.sat_conv_autotune(ops, 1, 2, scratch.data_ptr(), scratch.numel() * 4, st()))
    assert 1 <= ops[0].variant <= NUM_CONV_VARIANTS
    L.check(lib.sat_run_ops(ops, 1, st()))
    sync()
    assert (keep[2].float().cpu().double() - ref).abs().max().item() < 2e-2


@pytest.mark.parametrize("dtype", [L.SAT_F32, L.SAT_BF16])
def test_stem_conv_via_padded_nhwc4(lib, dtype):
    """image prep + 7x7/2 conv on the padded NHWC4 image == conv2d(images, w, stride 2, pad 3)"""
    N, H, W, width = 2, 32, 40, 16
    g = torch.Generator().manual_seed(5)
    img = torch.randn(N, 3, H, W, generator=g)
    w = torch.randn(width, 3, 7, 7, generator=g) / 12.0
    if dtype == L.SAT_BF16:
        img_r, w = img.bfloat16().float(), w.bfloat16().float()
    else:
        img_r = img
    ref = F.conv2d(img_r.double(), w.double(), None, 2, 3).permute(0, 2, 3, 1).reshape(-1, width)
    td = torch.bfloat16 if dtype == L.SAT_BF16 else torch.float32
    Hp, Wp = H + 6, (W + 8 + 1) // 2 * 2
    Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    pad_img = torch.zeros(N, Hp, Wp, 4, device="cuda", dtype=td)
    imgd = cu(img)
    p = L.SatOp()
    p.kind, p.dtype = L.OP_IMAGE_PREP, dtype
    p.in0, p.out = imgd.data_ptr(), pad_img.data_ptr()
    p.N, p.Hin, p.Win, p.Hout, p.Wout, p.pad = N, H, W, Hp, Wp, 3
    wst = torch.zeros(width, 7, 8, 4)
    wst[:, :, :7, :3] = w.permute(0, 2, 3, 1)
    wd = cu(wst.reshape(width, 224).to(td))
    out = torch.full((N * Ho * Wo, width), float("nan"), device="cuda", dtype=td)
    c = L.SatOp()
    c.kind, c.dtype = L.OP_CONV, dtype
    c.in0, c.w, c.out = pad_img.data_ptr(), wd.data_ptr(), out.data_ptr()
    c.N, c.Hin, c.Win, c.Cin, c.Hout, c.Wout, c.Cout = N, Hp, Wp, 32, Ho, Wo, width
    c.KH, c.KW, c.stride, c.pad = 7, 1, 2, 0
    c.sN, c.sH, c.sW = Hp * Wp * 4, Wp * 4, 4
    ops = (L.SatOp * 2)(p, c)
    L.check(lib.sat_run_ops(ops, 2, st()))
    sync()
    got = out.float().cpu().double()
    assert (got - ref).abs().max().item() < (2e-5 if dtype == L.SAT_F32 else 3e-2)


@pytest.mark.parametrize("dtype", [L.SAT_F32, L.SAT_BF16])
def test_bn_finalize_apply_pool(lib, dtype):
    td = torch.bfloat16 if dtype == L.SAT_BF16 else torch.float32
    N, H, W, Cc = 3, 10, 10, 32
    g = torch.Generator().manual_seed(9)
    x = (torch.randn(N, H, W, Cc, generator=g) * 2 + 0.5).to(td)
    idt = torch.randn(N, H, W, Cc, generator=g).to(td)
    gamma, beta = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.1
    xf = x.float().reshape(-1, Cc)
    M = xf.shape[0]
    tiles = 3
    part = torch.zeros(tiles, 2, Cc)
    for t in range(tiles):
        rows = xf[t::tiles]
        part[t, 0], part[t, 1] = rows.sum(0), (rows ** 2).sum(0)
    rm, rv = cu(torch.zeros(Cc)), cu(torch.ones(Cc))
    sc, sh = torch.empty(Cc, device="cuda"), torch.empty(Cc, device="cuda")
    pd, gd, bd = cu(part), cu(gamma), cu(beta)
    f = L.SatOp()
    f.kind, f.dtype = L.OP_BN_FINALIZE, dtype
    f.stat_partial, f.gamma, f.beta = pd.data_ptr(), gd.data_ptr(), bd.data_ptr()
    f.running_mean, f.running_var, f.scale_out, f.shift_out = rm.data_ptr(), rv.data_ptr(), sc.data_ptr(), sh.data_ptr()
    f.Cout, f.count, f.tiles_m, f.training, f.momentum, f.eps = Cc, M, tiles, 1, 0.1, 1e-5
    xd, idd = cu(x), cu(idt)
    y1 = torch.empty_like(xd)
    a = L.SatOp()
    a.kind, a.dtype = L.OP_BN_RELU, dtype
    a.in0, a.out, a.scale0, a.shift0 = xd.data_ptr(), y1.data_ptr(), sc.data_ptr(), sh.data_ptr()
    a.N, a.Hout, a.Wout, a.Cout = N, H, W, Cc
    y2 = torch.empty_like(xd)
    b = L.SatOp()
    b.kind, b.dtype = L.OP_BN_ADD_RELU, dtype
    b.in0, b.in1, b.out, b.scale0, b.shift0 = xd.data_ptr(), idd.data_ptr(), y2.data_ptr(), sc.data_ptr(), sh.data_ptr()
    b.N, b.Hout, b.Wout, b.Cout = N, H, W, Cc
    y3 = torch.empty_like(xd)
    b2 = L.SatOp()
    b2.kind, b2.dtype = L.OP_BN_ADD_RELU, dtype
    b2.in0, b2.in1, b2.out = xd.data_ptr(), idd.data_ptr(), y3.data_ptr()
    b2.scale0, b2.shift0, b2.scale1, b2.shift1 = sc.data_ptr(), sh.data_ptr(), sc.data_ptr(), sh.data_ptr()
    b2.N, b2.Hout, b2.Wout, b2.Cout = N, H, W, Cc
    Hq, Wq = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y4 = torch.empty(N, Hq, Wq, Cc, device="cuda", dtype=td)
    m = L.SatOp()
    m.kind, m.dtype = L.OP_BN_RELU_MAXPOOL, dtype
    m.in0, m.out, m.scale0, m.shift0 = xd.data_ptr(), y4.data_ptr(), sc.data_ptr(), sh.data_ptr()
    m.N, m.Hin, m.Win, m.Cout, m.Hout, m.Wout = N, H, W, Cc, Hq, Wq
    y5 = torch.empty(N, Cc, device="cuda")
    p = L.SatOp()
    p.kind, p.dtype = L.OP_AVGPOOL, dtype
    p.in0, p.out = xd.data_ptr(), y5.data_ptr()
    p.N, p.Hin, p.Win, p.Cout = N, H, W, Cc
    ops = (L.SatOp * 6)(f, a, b, b2, m, p)
    L.check(lib.sat_run_ops(ops, 6, st()))
    sync()
    mean, var = xf.double().mean(0), xf.double().var(0, unbiased=False)
    scale = gamma.double() / torch.sqrt(var + 1e-5)
    shift = beta.double() - mean * scale
    np.testing.assert_allclose(sc.cpu().numpy(), scale.numpy(), rtol=2e-5)
    np.testing.assert_allclose(sh.cpu().numpy(), shift.numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(rm.cpu().numpy(), (0.1 * mean).numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rv.cpu().numpy(), (0.9 + 0.1 * var * M / (M - 1)).numpy(), rtol=1e-5)
    tol = 1e-5 if dtype == L.SAT_F32 else 4e-2
    xb = x.double() * scale + shift
    ib = idt.double()
    assert (y1.float().cpu().double() - xb.clamp(min=0)).abs().max() < tol
    assert (y2.float().cpu().double() - (xb + ib).clamp(min=0)).abs().max() < tol
    assert (y3.float().cpu().double() - (xb + ib * scale + shift).clamp(min=0)).abs().max() < 2 * tol
    mp = F.max_pool2d(xb.clamp(min=0).permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1)
    assert (y4.float().cpu().double() - mp).abs().max() < tol
    assert (y5.cpu().double() - x.double().mean((1, 2))).abs().max() < 1e-5


# ------------------------------------------------------------------------------------------------------
def test_clamp_adam_matches_torch(lib):
    g = torch.Generator().manual_seed(3)
    n = 10007
    p0 = torch.randn(n, generator=g)
    ref_p = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref_p], lr=1e-3)
    npad = (n + 3) // 4 * 4
    p, m, v = cu(torch.cat([p0, torch.zeros(npad - n)])), torch.zeros(npad, device="cuda"), torch.zeros(npad, device="cuda")
    for step in range(1, 4):
        gr = torch.randn(n, generator=g) * 0.3
        ref_p.grad = gr.clone().clamp_(-0.1, 0.1)
        opt.step()
        gd = cu(torch.cat([gr, torch.zeros(npad - n)]))
        L.check(lib.sat_clamp_adam_step(L.ptr(p), L.ptr(gd), L.ptr(m), L.ptr(v), n, 1e-3, 0.9, 0.999, 1e-8, 0.1, step, st()))
        sync()
        np.testing.assert_allclose(gd.cpu()[:n].numpy(), ref_p.grad.numpy(), rtol=0, atol=0)     # clamp is exact
        np.testing.assert_allclose(p.cpu()[:n].numpy(), ref_p.detach().numpy(), rtol=0, atol=2e-7)


def test_ce_rows_and_colsum(lib):
    g = torch.Generator().manual_seed(11)
    N, V = 37, 1003
    logits = torch.randn(N, V, generator=g) * 3
    logits[5, 17] = 40.0          # force a dominant column (large max, exp underflow elsewhere)
    tgt = torch.randint(0, V, (N,), generator=g)
    ld = cu(logits.clone())
    rl, lo = torch.empty(N, device="cuda"), torch.empty(1, device="cuda")
    inv = 1.0 / 50.0
    tgd = cu(tgt)
    L.check(lib.sat_ce_rows(L.ptr(ld), V, L.ptr(tgd), N, V, inv, 1, L.ptr(rl), L.ptr(lo), st()))
    sync()
    lg = logits.double().requires_grad_(True)
    ref = F.cross_entropy(lg, tgt, reduction="sum") * inv
    ref.backward()
    assert abs(lo.item() - ref.item()) < 1e-5 * abs(ref.item())
    np.testing.assert_allclose(ld.cpu().numpy(), lg.grad.numpy(), rtol=0, atol=2e-8 + 1e-6 * inv)
    cs = torch.empty(V, device="cuda")
    L.check(lib.sat_colsum_f32(L.ptr(ld), V, N, V, L.ptr(cs), st()))
    sync()
    np.testing.assert_allclose(cs.cpu().numpy(), lg.grad.sum(0).numpy(), rtol=0, atol=1e-7)


def test_vocab_argmax_first_max(lib):
    g = torch.Generator().manual_seed(13)
    B, H, V = 5, 64, 1003
    h, w, b = torch.randn(B, H, generator=g), torch.randn(V, H, generator=g) * 0.1, torch.randn(V, generator=g) * 0.01
    w[900] = w[30]
    b[900] = b[30]               # exact tie between columns 30 and 900 -> first index must win
    h[2] = w[30] * 50
    ref = (h @ w.t() + b).max(1)[1]
    ids = torch.full((B, 20), -1, dtype=torch.int64, device="cuda")
    wsb = lib.sat_vocab_argmax_ws_bytes(B, V)
    ws = torch.empty(wsb // 4, device="cuda")
    col = ids[:, 3]
    hd, wd, bd = cu(h), cu(w), cu(b)
    L.check(lib.sat_vocab_argmax(L.ptr(hd), L.ptr(wd), L.ptr(bd), B, H, V, col.data_ptr(), 20, L.ptr(ws), wsb, st()))
    sync()
    assert ids[2, 3].item() == 30
    assert torch.equal(ids[:, 3].cpu(), ref)
    assert (ids[:, 0] == -1).all()


def test_fc_bn1d_fwd_bwd(lib):
    from oracle import encoder as OE
    g = torch.Generator().manual_seed(17)
    B, Fd, E = 12, 256, 32
    params, buffers = OE.init_encoder_params(E, dict(layers=(1, 1, 1, 1), width=8), generator=g, randomize_bn=True)
    pooled = torch.rand(B, Fd, generator=g)
    dy = torch.randn(B, E, generator=g)
    bufs = {k: v.clone() for k, v in buffers.items()}
    y, tape = OE.head_forward(params, bufs, pooled)
    gr = OE.head_backward(params, tape, dy)
    rm, rv = cu(buffers["bn.running_mean"]), cu(buffers["bn.running_var"])
    feats, xhat, rstd = torch.empty(B, E, device="cuda"), torch.empty(B, E, device="cuda"), torch.empty(E, device="cuda")
    wsb = lib.sat_fc_bn1d_ws_bytes(B, Fd, E)
    ws = torch.empty(max(wsb // 4, B * E), device="cuda")
    pd, wd, bd = cu(pooled), cu(params["resnet.fc.weight"]), cu(params["resnet.fc.bias"])
    gd, bed = cu(params["bn.weight"]), cu(params["bn.bias"])
    L.check(lib.sat_fc_bn1d_fwd(L.ptr(pd), L.ptr(wd), L.ptr(bd), L.ptr(gd), L.ptr(bed), L.ptr(rm), L.ptr(rv), 0.01, 1e-5, 1,
                                B, Fd, E, L.ptr(feats), L.ptr(xhat), L.ptr(rstd), L.ptr(ws), ws.numel() * 4, st()))
    dw, db, dg, dbe = torch.empty(E, Fd, device="cuda"), torch.empty(E, device="cuda"), torch.empty(E, device="cuda"), torch.empty(E, device="cuda")
    dyd = cu(dy)
    L.check(lib.sat_fc_bn1d_bwd(L.ptr(dyd), L.ptr(pd), L.ptr(xhat), L.ptr(rstd), L.ptr(gd), B, Fd, E, L.ptr(dw), L.ptr(db),
                                L.ptr(dg), L.ptr(dbe), L.ptr(ws), ws.numel() * 4, st()))
    sync()
    np.testing.assert_allclose(feats.cpu().numpy(), y.numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(rm.cpu().numpy(), bufs["bn.running_mean"].numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(rv.cpu().numpy(), bufs["bn.running_var"].numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(dg.cpu().numpy(), gr["bn.weight"].numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(dbe.cpu().numpy(), gr["bn.bias"].numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(dw.cpu().numpy(), gr["resnet.fc.weight"].numpy(), rtol=1e-3, atol=2e-5)
    # d(fc.bias) = sum_b dz is mathematically ZERO under train-mode BatchNorm: both sides hold rounding noise of size
    # ~1e-7 * sum_b |dz| (|dz| is O(10) here), so only its smallness can be compared
    assert db.cpu().abs().max().item() < 5e-4 and gr["resnet.fc.bias"].abs().max().item() < 5e-4


def test_conv_atomic_stats_then_consumer_derives_affine(lib):
    """bf16 conv adds fixed-point column sums with integer atomics; BN_RELU / BN_ADD_RELU derive scale/shift from them, update
    the running statistics once, clear the other parity's accumulators; results are bitwise reproducible"""
    N, H, W, Cin, Cout = 6, 20, 20, 64, 192
    g = torch.Generator().manual_seed(93)
    x = (torch.randn(N, Cin, H, W, generator=g) + 0.3).bfloat16().float()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / 24.0).bfloat16().float()
    idt = torch.randn(N, H, W, Cout, generator=g).bfloat16()
    gamma, beta = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    ref = F.conv2d(x.double(), w.double(), None, 1, 1).permute(0, 2, 3, 1).reshape(-1, Cout)
    M = ref.shape[0]
    conv, keep, _ = _conv_op(L.SAT_BF16, x.permute(0, 2, 3, 1), w.permute(0, 2, 3, 1), 1, 1, stats=False)
    acc = torch.zeros(2, 2, Cout, dtype=torch.int64, device="cuda")
    acc[1] = 12345                                     # stale other-parity half: must be cleared by the consumer
    conv.stat_acc = acc.data_ptr()
    gd, bd = cu(gamma), cu(beta)
    rm, rv = cu(torch.zeros(Cout)), cu(torch.ones(Cout))
    idd = cu(idt)
    y1 = torch.empty(M, Cout, device="cuda", dtype=torch.bfloat16)
    act = L.SatOp()
    act.kind, act.dtype = L.OP_BN_ADD_RELU, L.SAT_BF16
    act.in0, act.in1, act.out = keep[2].data_ptr(), idd.data_ptr(), y1.data_ptr()
    act.stat_acc, act.gamma, act.beta = acc.data_ptr(), gd.data_ptr(), bd.data_ptr()
    act.running_mean, act.running_var = rm.data_ptr(), rv.data_ptr()
    act.count, act.momentum, act.eps = M, 0.1, 1e-5
    act.N, act.Hout, act.Wout, act.Cout = N, H, W, Cout
    ops = (L.SatOp * 2)(conv, act)
    L.check(lib.sat_run_ops_parity(ops, 2, 0, st()))
    sync()
    c_bf = keep[2].float().cpu().double()
    mean, var = ref.mean(0), ref.var(0, unbiased=False)
    scale = gamma.double() / torch.sqrt(var + 1e-5)
    want = (c_bf * scale + (beta.double() - mean * scale) + idt.double().reshape(M, Cout)).clamp(min=0)
    assert (y1.float().cpu().double() - want).abs().max().item() < 5e-2
    np.testing.assert_allclose(rm.cpu().numpy(), (0.1 * mean).numpy(), rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(rv.cpu().numpy(), (0.9 + 0.1 * var * M / (M - 1)).numpy(), rtol=1e-3)
    assert int(acc[1].abs().sum()) == 0 and int(acc[0].abs().sum()) > 0
    np.testing.assert_allclose((acc[0, 0].cpu().double() / 2 ** 22).numpy(), ref.sum(0).numpy(), rtol=0, atol=2e-2 * M ** 0.5)
    # next step uses parity 1 and clears parity 0; same data => bit-identical output
    y_first = y1.clone()
    L.check(lib.sat_run_ops_parity(ops, 2, 1, st()))
    sync()
    assert torch.equal(y1, y_first)
    assert int(acc[0].abs().sum()) == 0 and int(acc[1].abs().sum()) > 0


@pytest.mark.parametrize("derive", [False, True])
@pytest.mark.parametrize("variant", [0, 1, 3, 6, 9])
def test_conv1x1_with_input_bn_relu_fused(lib, derive, variant):
    """1x1 bf16 conv whose A operand is relu(bn(x)) applied to the landed LDS stage: table precomputed or derived from
    the producer's integer sums (which workgroup 0 then clears / folds into the running statistics)"""
    N, H, W, Cin, Cout = 5, 12, 12, 128, 192
    g = torch.Generator().manual_seed(97 + variant)
    x = (torch.randn(N, H, W, Cin, generator=g) * 1.5 + 0.2).bfloat16()
    w = (torch.randn(Cout, Cin, generator=g) / Cin ** 0.5).bfloat16()
    gamma, beta = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.2
    xf = x.float().reshape(-1, Cin).double()
    M = xf.shape[0]
    mean, var = xf.mean(0), xf.var(0, unbiased=False)
    scale = (gamma.double() / torch.sqrt(var + 1e-5)).float()
    shift = (beta.double() - mean * scale.double()).float()
    a = torch.clamp(x.float().reshape(-1, Cin) * scale + shift, min=0).bfloat16().float().double()    # what the LDS stage holds
    ref = a @ w.float().double().t()
    o, keep, _ = _conv_op(L.SAT_BF16, x.float(), w.float().reshape(Cout, 1, 1, Cin), 1, 0)
    o.variant = variant
    extra = []
    if derive:
        acc = torch.zeros(2, 2, Cin, dtype=torch.int64, device="cuda")
        acc[0, 0] = torch.round(xf.sum(0) * 4194304.0).long().cuda()
        acc[0, 1] = torch.round((xf ** 2).sum(0) * 4194304.0).long().cuda()
        acc[1] = 777
        gd, bd, rm, rv = cu(gamma), cu(beta), cu(torch.zeros(Cin)), cu(torch.ones(Cin))
        o.stat_acc1, o.gamma1, o.beta1 = acc.data_ptr(), gd.data_ptr(), bd.data_ptr()
        o.running_mean1, o.running_var1 = rm.data_ptr(), rv.data_ptr()
        o.count, o.momentum, o.eps = M, 0.1, 1e-5
        extra = [acc, gd, bd, rm, rv]
    else:
        sd, td = cu(scale), cu(shift)
        o.scale0, o.shift0 = sd.data_ptr(), td.data_ptr()
        extra = [sd, td]
    L.check(lib.sat_run_ops_parity(C.pointer(o), 1, 0, st()))
    sync()
    out = keep[2].float().cpu().double()
    assert (out - ref).abs().max().item() < 3e-2 + 4e-3 * ref.abs().max().item()
    part = keep[3].cpu().double()
    np.testing.assert_allclose(part[:, 0].sum(0).numpy(), ref.sum(0).numpy(), rtol=2e-3, atol=0.3)
    if derive:
        acc, _, _, rm, rv = extra
        assert int(acc[1].abs().sum()) == 0
        np.testing.assert_allclose(rm.cpu().numpy(), (0.1 * mean).numpy(), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("variant", [0, 2, 5, 8, 12, 16, 19])
@pytest.mark.parametrize("k,stride,pad,relu,resid", [(1, 1, 0, True, True), (3, 1, 1, True, False), (1, 2, 0, False, False)])
def test_conv_inference_epilogue_affine_residual_relu(lib, variant, k, stride, pad, relu, resid):
    """inference fusion: out = [relu](conv(x)*scale + shift [+ residual]) in the conv epilogue (eval-mode BatchNorm +
    add + ReLU of a bottleneck, models.py:27 under model.eval()) vs the same chain in torch"""
    g = torch.Generator().manual_seed(300 + variant + k)
    N, H, W, Cin, Cout = 3, 14, 14, 64, 136
    x = torch.randn(N, Cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).bfloat16().float()
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    conv = F.conv2d(x.double(), w.double(), None, stride, pad).permute(0, 2, 3, 1).reshape(-1, Cout)
    z = torch.randn(conv.shape, generator=g).bfloat16()
    ref = conv * sc.double() + sh.double()
    if resid:
        ref = ref + z.double()
    if relu:
        ref = ref.clamp_min(0)
    o, keep, _ = _conv_op(L.SAT_BF16, x.permute(0, 2, 3, 1), w.permute(0, 2, 3, 1), stride, pad, stats=False)
    scd, shd, zd = cu(sc), cu(sh), z.cuda()
    o.scale1, o.shift1, o.flags, o.variant = scd.data_ptr(), shd.data_ptr(), 1 if relu else 0, variant
    if resid:
        o.in1 = zd.data_ptr()
    L.check(lib.sat_run_ops(C.pointer(o), 1, st()))
    sync()
    out = keep[2].float().cpu().double()
    assert (out - ref).abs().max().item() < 3e-2 + 8e-3 * ref.abs().max().item()
    if relu:
        assert float(out.min()) >= 0.0
    # statistics and a fixed output affine exclude each other; the f32 path has no fused epilogue
    o2, keep2, _ = _conv_op(L.SAT_BF16, x.permute(0, 2, 3, 1), w.permute(0, 2, 3, 1), stride, pad, stats=True)
    o2.scale1, o2.shift1 = scd.data_ptr(), shd.data_ptr()
    assert lib.sat_run_ops(C.pointer(o2), 1, st()) == 1001


# ------------------------------------------------------------------------------------------------------
# entry points that had no test in round 1 + the round-2 additions
@pytest.mark.parametrize("dtype", [L.SAT_F32, L.SAT_BF16])
def test_conv_bn_relu_fwd_one_call(lib, dtype):
    """sat_conv_bn_relu_fwd: conv (statistics in the epilogue) -> finalize -> normalise + ReLU, vs fp64 arithmetic"""
    g = torch.Generator().manual_seed(81)
    N, H, W, Cin, Cout = 3, 9, 9, 64, 72
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / 24.0
    if dtype == L.SAT_BF16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    gamma, beta = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    o, keep, (n, ho, wo, co) = _conv_op(dtype, x.permute(0, 2, 3, 1), w.permute(0, 2, 3, 1), 1, 1)
    rm, rv = cu(torch.zeros(Cout)), cu(torch.ones(Cout))
    sc, sh, gd, bd = torch.empty(Cout, device="cuda"), torch.empty(Cout, device="cuda"), cu(gamma), cu(beta)
    f = L.SatOp()
    f.kind, f.dtype = L.OP_BN_FINALIZE, dtype
    f.stat_partial, f.gamma, f.beta = keep[3].data_ptr(), gd.data_ptr(), bd.data_ptr()
    f.running_mean, f.running_var, f.scale_out, f.shift_out = rm.data_ptr(), rv.data_ptr(), sc.data_ptr(), sh.data_ptr()
    f.Cout, f.count, f.tiles_m, f.training, f.momentum, f.eps = Cout, n * ho * wo, o.tiles_m, 1, 0.1, 1e-5
    y = torch.empty_like(keep[2])
    a = L.SatOp()
    a.kind, a.dtype = L.OP_BN_RELU, dtype
    a.in0, a.out, a.scale0, a.shift0 = keep[2].data_ptr(), y.data_ptr(), sc.data_ptr(), sh.data_ptr()
    a.N, a.Hout, a.Wout, a.Cout = n, ho, wo, co
    L.check(lib.sat_conv_bn_relu_fwd(C.pointer(o), C.pointer(f), C.pointer(a), st()))
    sync()
    c = F.conv2d(x.double(), w.double(), None, 1, 1).permute(0, 2, 3, 1).reshape(-1, Cout)
    mean, var = c.mean(0), c.var(0, unbiased=False)
    ref = ((c - mean) / torch.sqrt(var + 1e-5) * gamma.double() + beta.double()).clamp(min=0)
    tol = 2e-4 if dtype == L.SAT_F32 else 6e-2
    assert (y.float().cpu().double() - ref).abs().max().item() < tol
    bad = L.SatOp()
    bad.kind = L.OP_BN_RELU
    assert lib.sat_conv_bn_relu_fwd(C.pointer(o), C.pointer(f), C.pointer(bad), st()) == 1001   # bnrelu.in0 != conv.out


def test_casts_round_to_nearest_even_and_back(lib):
    g = torch.Generator().manual_seed(82)
    x = torch.randn(100003, generator=g) * 3
    x[:4] = torch.tensor([0.0, -0.0, 1.00390625, 65504.0])       # 1 + 2^-8: a tie, rounds to even (1.0)
    xd = cu(x)
    b = torch.empty(x.numel(), dtype=torch.bfloat16, device="cuda")
    L.check(lib.sat_cast_f32_bf16(xd.data_ptr(), b.data_ptr(), x.numel(), st()))
    back = torch.empty_like(xd)
    L.check(lib.sat_cast_bf16_f32(b.data_ptr(), back.data_ptr(), x.numel(), st()))
    sync()
    assert torch.equal(b.cpu(), x.bfloat16())
    assert torch.equal(back.cpu(), x.bfloat16().float())


def test_validate_ids_flags_only_out_of_range(lib):
    ids = torch.randint(0, 50, (7, 12))
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    d = cu(ids)
    L.check(lib.sat_validate_ids(d.data_ptr(), 12, 7, 12, 0, 50, status.data_ptr(), st()))
    assert int(status.item()) == 0
    ids2 = ids.clone()
    ids2[6, 11] = 50
    d2 = cu(ids2)
    L.check(lib.sat_validate_ids(d2.data_ptr(), 12, 7, 11, 0, 50, status.data_ptr(), st()))     # bad id outside the checked columns
    assert int(status.item()) == 0
    L.check(lib.sat_validate_ids(d2.data_ptr(), 12, 7, 12, 0, 50, status.data_ptr(), st()))
    assert int(status.item()) == 1
    status.zero_()
    ids3 = ids.clone()
    ids3[0, 0] = -3
    d3 = cu(ids3)
    L.check(lib.sat_validate_ids(d3.data_ptr(), 12, 7, 12, 0, 50, status.data_ptr(), st()))
    assert int(status.item()) == 1
    assert lib.sat_validate_ids(d3.data_ptr(), 4, 7, 12, 0, 50, status.data_ptr(), st()) == 1001    # stride < cols


def test_run_ops_timed_reports_conv_durations_and_computes_the_same(lib):
    g = torch.Generator().manual_seed(83)
    x = torch.randn(8, 128, 14, 14, generator=g).bfloat16().float()
    w = (torch.randn(256, 128, 3, 3, generator=g) / 34.0).bfloat16().float()
    o, keep, _ = _conv_op(L.SAT_BF16, x.permute(0, 2, 3, 1), w.permute(0, 2, 3, 1), 1, 1)
    ops = (L.SatOp * 1)(o)
    L.check(lib.sat_run_ops(ops, 1, st()))
    sync()
    ref = keep[2].clone()
    keep[2].fill_(float("nan"))
    us = (C.c_float * 1)()
    L.check(lib.sat_run_ops_timed(ops, 1, 0, st(), us))
    assert torch.equal(keep[2], ref)
    assert 1.0 < us[0] < 500.0, us[0]             # a 0.9 GFLOP conv: a few microseconds, not 0 and not a millisecond


def test_embed_concat_bwd_beyond_the_default_lds_limit(lib):
    """N = 20000 packed rows: the token list needs 80 KB of dynamic LDS (opt-in above 48 KB); first-occurrence scatter
    equals index_add in fp64"""
    g = torch.Generator().manual_seed(84)
    B, T, E, V = 1000, 21, 16, 300
    lengths = [T - 1] * B                                      # packed steps T-1 = 20 -> N = 20000
    caps = torch.randint(0, V, (B, T), generator=g)
    pi = sat.PackInfo.get(lengths, "cuda")
    N = pi.N
    dX = torch.randn(N, E, generator=g)
    dXd, cd = cu(dX), cu(caps)
    d_embed = torch.full((V, E), float("nan"), device="cuda")
    d_feat = torch.full((B, E), float("nan"), device="cuda")
    L.check(lib.sat_embed_concat_bwd(dXd.data_ptr(), cd.data_ptr(), cd.stride(0), pi.prefix_dev.data_ptr(), pi.T, N, B, E, V,
                                     d_embed.data_ptr(), d_feat.data_ptr(), st()))
    sync()
    ref = torch.zeros(V, E, dtype=torch.float64)
    for t in range(1, pi.T):
        rows = dX[pi.prefix[t]:pi.prefix[t + 1]].double()
        ref.index_add_(0, caps[:pi.batch_sizes[t], t - 1], rows)
    np.testing.assert_allclose(d_embed.cpu().double().numpy(), ref.numpy(), rtol=0, atol=5e-5)
    assert torch.equal(d_feat.cpu(), dX[:B])


@pytest.mark.parametrize("B,T,In,H,ragged", [(64, 19, 256, 512, False), (64, 19, 64, 512, True), (13, 7, 32, 64, True),
                                             (8, 5, 32, 32, False), (40, 12, 96, 256, True), (3, 9, 32, 128, True),
                                             (64, 19, 512, 1024, False), (64, 12, 1024, 1024, True), (21, 6, 64, 1024, True)])
def test_lstm_fwd_persistent_equals_per_step_launches(lib, B, T, In, H, ragged, monkeypatch):
    """sat_lstm_fwd with the exchange workspace (ONE persistent launch: W_hh in registers, granule hand-off per group)
    against the same entry point without it (one launch per step): every tape bit-identical (same MFMA, same K order per
    output element is NOT guaranteed -- the per-step kernel splits K over 8 waves -- so: 1e-6), and against fp64.
    H = 1024 (the reference's default hidden size, config.py:28; BASELINE configs[3]): 16-row groups, 4 groups x 64 members at
    batch 64, the W_hh slice across VGPRs + AGPRs (round 5)."""
    g = torch.Generator().manual_seed(B * 7 + T + H)
    lengths = sorted([int(x) for x in torch.randint(1, T + 1, (B,), generator=g)], reverse=True) if ragged else [T] * B
    lengths[0] = T
    pi = sat.PackInfo.get(lengths, "cuda")
    N = pi.N
    k = 1.0 / H ** 0.5
    X = torch.randn(N, In, generator=g)
    w_ih = torch.empty(4 * H, In).uniform_(-k, k, generator=g)
    w_hh = torch.empty(4 * H, H).uniform_(-k, k, generator=g)
    b_ih = torch.empty(4 * H).uniform_(-k, k, generator=g)
    b_hh = torch.empty(4 * H).uniform_(-k, k, generator=g)
    d = [cu(t) for t in (X, w_ih, w_hh, b_ih, b_hh)]

    def run(with_ws):
        GA, CS = torch.full((N, 4 * H), float("nan"), device="cuda"), torch.full((N, H), float("nan"), device="cuda")
        HS, HP = torch.full((N, H), float("nan"), device="cuda"), torch.full((N, H), float("nan"), device="cuda")
        cst = torch.empty(B, H, device="cuda")
        wsb = lib.sat_lstm_fwd_ws_bytes(B, H) if with_ws else 0
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device="cuda")
        L.check(lib.sat_lstm_fwd(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(), pi.bs_c, pi.T, In, H,
                                 GA.data_ptr(), CS.data_ptr(), HS.data_ptr(), HP.data_ptr(), cst.data_ptr(),
                                 ws.data_ptr() if with_ws else None, wsb, st()))
        sync()
        err = int(ws[wsb - 64:wsb - 60].view(torch.int32).item()) if with_ws and wsb else 0
        return GA.cpu(), CS.cpu(), HS.cpu(), HP.cpu(), err

    assert lib.sat_lstm_fwd_ws_bytes(B, H) > 0
    pers = run(True)
    step = run(False)
    assert pers[4] == 0, "persistent recurrence timed out"
    for a, b, name in zip(pers[:4], step[:4], ("GA", "CS", "HS", "HP")):
        assert torch.isfinite(a).all(), name
        np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=0, atol=2e-6, err_msg=name)
    # fp64 reference of the recurrence
    Xg = X.double() @ w_ih.double().t() + b_ih.double() + b_hh.double()
    h, c = torch.zeros(B, H, dtype=torch.float64), torch.zeros(B, H, dtype=torch.float64)
    for t, n in enumerate(pi.batch_sizes):
        gts = Xg[pi.prefix[t]:pi.prefix[t] + n] + h[:n] @ w_hh.double().t()
        i, f, gg, o = gts[:, :H].sigmoid(), gts[:, H:2 * H].sigmoid(), gts[:, 2 * H:3 * H].tanh(), gts[:, 3 * H:].sigmoid()
        c = torch.cat([f * c[:n] + i * gg, c[n:]])
        h = torch.cat([o * c[:n].tanh(), h[n:]])
        np.testing.assert_allclose(pers[2][pi.prefix[t]:pi.prefix[t] + n].double().numpy(), h[:n].numpy(), rtol=0, atol=5e-6)


@pytest.mark.parametrize("N,H,V", [(1216, 512, 10000), (76, 64, 500), (33, 32, 1003), (200, 128, 130)])
def test_vocab_ce_fwd_and_bwd_vs_fp64(lib, N, H, V):
    """sat_vocab_ce_fwd (projection + row-wise CE, d(loss)/d(logits) written over the logits) and sat_vocab_ce_bwd against
    fp64 softmax cross entropy and its gradients; V not a multiple of 4 / of the tile, N below one tile (models.py:53,
    train.py:53,143-144)"""
    g = torch.Generator().manual_seed(N + V)
    Hs = torch.randn(N, H, generator=g) * 0.5
    W = torch.empty(V, H).uniform_(-0.1, 0.1, generator=g)
    b = torch.randn(V, generator=g) * 0.1
    tg = torch.randint(0, V, (N,), generator=g)
    inv = 1.0 / N
    Hd, Wd, bd, tgd = cu(Hs), cu(W), cu(b), cu(tg)
    ldl = (V + 3) // 4 * 4
    logits = torch.zeros(N, ldl, device="cuda")
    rl, lo = torch.empty(N, device="cuda"), torch.empty(1, device="cuda")
    assert lib.sat_vocab_ce_fwd(L.ptr(Hd), L.ptr(Wd), L.ptr(bd), None, N, H, V, inv, L.ptr(logits), ldl, L.ptr(rl), L.ptr(lo), st()) == 1001
    L.check(lib.sat_vocab_ce_fwd(L.ptr(Hd), L.ptr(Wd), L.ptr(bd), L.ptr(tgd), N, H, V, inv, L.ptr(logits), ldl, L.ptr(rl), L.ptr(lo), st()))
    sync()
    Hq, Wq = Hs.double().requires_grad_(True), W.double().requires_grad_(True)
    bq = b.double().requires_grad_(True)
    ref_logits = (Hq @ Wq.t() + bq)
    ref_logits.retain_grad()
    ref_rows = torch.logsumexp(ref_logits, 1) - ref_logits[torch.arange(N), tg]
    loss = ref_rows.sum() * inv
    loss.backward()
    np.testing.assert_allclose(rl.cpu().numpy(), ref_rows.detach().numpy(), rtol=0, atol=1e-5)
    assert abs(lo.item() - loss.item()) < 2e-6
    np.testing.assert_allclose(logits[:, :V].cpu().numpy(), ref_logits.grad.numpy(), rtol=1e-4, atol=2e-9)      # the gradient, in place
    if ldl > V:
        assert float(logits[:, V:].abs().sum()) == 0.0
    dw, db, dH = torch.full((V, H), float("nan"), device="cuda"), torch.full((V,), float("nan"), device="cuda"), torch.full((N, H), float("nan"), device="cuda")
    bwsb = lib.sat_vocab_ce_bwd_ws_bytes(N, H, V)
    bws = torch.empty(max(bwsb // 4, 4), device="cuda")
    L.check(lib.sat_vocab_ce_bwd(L.ptr(logits), ldl, L.ptr(Hd), L.ptr(Wd), N, H, V, L.ptr(dw), L.ptr(db), L.ptr(dH), L.ptr(bws), bwsb, st()))
    sync()
    np.testing.assert_allclose(dw.cpu().numpy(), Wq.grad.numpy(), rtol=1e-4, atol=2e-8)
    np.testing.assert_allclose(db.cpu().numpy(), bq.grad.numpy(), rtol=1e-4, atol=2e-8)
    np.testing.assert_allclose(dH.cpu().numpy(), Hq.grad.numpy(), rtol=1e-4, atol=2e-8)


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,wkm,bias", [(64, 512, 1024, 0, True), (64, 1024, 4096, 1, False), (37, 1536, 4096, 1, False),
                                            (5, 1024, 512, 1, True), (128, 48, 64, 0, False), (1, 10000, 512, 0, True)])
def test_skinny_gemm_f32(M, N, K, wkm, bias):
    """sat_skinny_gemm_f32: the split-K few-rows GEMM of a decode / BPTT step (model2.py:54-62) against float64."""
    lib = L.load()
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(M, K, generator=g)
    W = torch.randn((K, N) if wkm else (N, K), generator=g) * 0.05
    b = torch.randn(N, generator=g) if bias else None
    want = A.double() @ (W.double() if wkm else W.double().t())
    if bias:
        want = want + b.double()
    Ad, Wd = A.cuda(), W.cuda()
    bd = b.cuda() if bias else None
    out = torch.full((M, N), float("nan"), device="cuda")
    need = lib.sat_skinny_gemm_ws_bytes(M, N, K)
    ws = torch.empty(max(need // 4, 1), device="cuda")
    L.check(lib.sat_skinny_gemm_f32(Ad.data_ptr(), K, Wd.data_ptr(), N if wkm else K, wkm, M, N, K, L.ptr(bd), out.data_ptr(), N,
                                    ws.data_ptr(), need, L.stream()))
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), want.numpy(), rtol=2e-5, atol=2e-5 * float(K) ** 0.5 * 0.05 * 4)
    if need:
        assert lib.sat_skinny_gemm_f32(Ad.data_ptr(), K, Wd.data_ptr(), N if wkm else K, wkm, M, N, K, None, out.data_ptr(), N,
                                       None, 0, L.stream()) == 1002


@pytest.mark.gpu
def test_skinny_gemm2_sum_of_two_products():
    """sat_skinny_gemm2_f32: out = A W + A2 W2 with K-major weights (dh_{t-1} through the LSTMCell and the attention projection)"""
    lib = L.load()
    g = torch.Generator().manual_seed(4)
    M, N, K, K2 = 37, 1024, 4096, 512
    A, W = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g) * 0.03
    A2, W2 = torch.randn(M, K2, generator=g), torch.randn(K2, N, generator=g) * 0.03
    want = A.double() @ W.double() + A2.double() @ W2.double()
    d = [t.cuda() for t in (A, W, A2, W2)]
    out = torch.empty(M, N, device="cuda")
    need = lib.sat_skinny_gemm_ws_bytes(M, N, K)
    ws = torch.empty(max(need // 4, 1), device="cuda")
    L.check(lib.sat_skinny_gemm2_f32(d[0].data_ptr(), K, d[1].data_ptr(), N, K, d[2].data_ptr(), K2, d[3].data_ptr(), N, K2, 1, M, N,
                                     None, out.data_ptr(), N, ws.data_ptr(), need, L.stream()))
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), want.numpy(), rtol=2e-5, atol=5e-5)


@pytest.mark.gpu
def test_fixed_point_statistics_hold_for_small_magnitude_channels(lib):
    """The conv's BatchNorm sums are 2^-22 fixed point per tile column (2.4e-7 absolute per tile sum).  Channels whose
    activations are 1e-1 ... 1e-4 of the others: the normalised output and the running statistics still match float64 --
    a channel small enough for the quantisation to show in its variance (var << eps = 1e-5) is dominated by eps in
    rsqrt(var + eps), one above that has thousands of quanta per tile."""
    N, H, W, Cin, Cout = 8, 28, 28, 64, 128
    g = torch.Generator().manual_seed(17)
    x = (torch.randn(N, Cin, H, W, generator=g) + 0.2).bfloat16().float()
    mag = torch.tensor([10.0 ** -(c % 5) for c in range(Cout)])
    w = ((torch.randn(Cout, Cin, 1, 1, generator=g) / 8.0) * mag.view(-1, 1, 1, 1)).bfloat16().float()
    gamma, beta = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    ref = F.conv2d(x.double(), w.double()).permute(0, 2, 3, 1).reshape(-1, Cout)
    M = ref.shape[0]
    conv, keep, _ = _conv_op(L.SAT_BF16, x.permute(0, 2, 3, 1), w.permute(0, 2, 3, 1), 1, 0, stats=False)
    acc = torch.zeros(2, 2, Cout, dtype=torch.int64, device="cuda")
    conv.stat_acc = acc.data_ptr()
    gd, bd = cu(gamma), cu(beta)
    rm, rv = cu(torch.zeros(Cout)), cu(torch.zeros(Cout))
    y1 = torch.empty(M, Cout, device="cuda", dtype=torch.bfloat16)
    act = L.SatOp()
    act.kind, act.dtype = L.OP_BN_RELU, L.SAT_BF16
    act.in0, act.out = keep[2].data_ptr(), y1.data_ptr()
    act.stat_acc, act.gamma, act.beta = acc.data_ptr(), gd.data_ptr(), bd.data_ptr()
    act.running_mean, act.running_var = rm.data_ptr(), rv.data_ptr()
    act.count, act.momentum, act.eps = M, 1.0, 1e-5          # momentum 1: the running buffers receive the batch statistics
    act.N, act.Hout, act.Wout, act.Cout = N, H, W, Cout
    ops = (L.SatOp * 2)(conv, act)
    L.check(lib.sat_run_ops_parity(ops, 2, 0, st()))
    sync()
    c_bf = keep[2].float().cpu().double()                    # what the kernel normalises: its own bf16 conv output
    mean, var = ref.mean(0), ref.var(0, unbiased=False)
    scale = gamma.double() / torch.sqrt(var + 1e-5)
    want = (c_bf * scale + (beta.double() - mean * scale)).clamp(min=0)
    err = (y1.float().cpu().double() - want).abs().max(0).values
    assert err.max().item() < 3e-2, err                      # bf16 output rounding only, in every magnitude class
    unb = var * M / (M - 1)
    got_m, got_v = rm.cpu().double(), rv.cpu().double()
    # mean: absolute error <= 1.2e-7 per tile sum / 128 rows; variance: relative to (var + eps), which is what the normalisation sees
    assert (got_m - mean).abs().max().item() < 2e-6, (got_m - mean).abs().max()
    assert (((got_v - unb).abs()) / (unb + 1e-5)).max().item() < 2e-3


def test_lstm_fwd_16_row_groups_at_hidden_512_in_a_fresh_process():
    """SAT_LSTM_ROWS=16 (read once per process): the cfg-2 layer on 4 groups x 32 members = 128 workgroups -- full 16-row MFMA tiles,
    half the CUs -- must equal the per-step launches like the default 8-row form does"""
    import os
    import subprocess
    import sys
    env = dict(os.environ, SAT_LSTM_ROWS="16")
    here = os.path.abspath(__file__)
    r = subprocess.run([sys.executable, "-m", "pytest", here, "-q", "-x", "-m", "gpu", "-k",
                        "test_lstm_fwd_persistent_equals_per_step_launches and (512 or 256)"], env=env, capture_output=True, text=True,
                       timeout=600, cwd=os.path.dirname(os.path.dirname(here)))
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_persistent_lstm_timeout_is_reported_not_swallowed(hooks_lib, monkeypatch):
    """ADVICE r2 / VERDICT r2 (robustness 10): a spin timeout inside the persistent recurrence sets the workspace's status word;
    `watch.ResidencyWatch` reads it back behind the call and raises, and the process falls back to one launch per step.
    SAT_LSTM_DEBUG_STALL=1 makes workgroup 0 withhold its hidden state (TEST build of the library only: `hooks_lib`),
    SAT_LSTM_SPIN_LIMIT shortens the wait (models.py:52)."""
    import importlib
    lib = hooks_lib
    sat = importlib.import_module("show-and-tell_amd")
    M = importlib.import_module("show-and-tell_amd.models")
    torch.manual_seed(11)
    dec = sat.DecoderRNN(32, 64, 100, 1).cuda()
    feats = torch.randn(16, 32, device="cuda")
    caps = torch.randint(4, 100, (16, 9), device="cuda")
    lengths = [10] * 16
    watch = sat.watch.ResidencyWatch.get(feats.device)
    with torch.no_grad():
        want = dec(feats, caps, lengths).clone()
        watch.poll(block=True)                                   # clean run: nothing raised
        try:
            monkeypatch.setenv("SAT_LSTM_DEBUG_STALL", "1")
            monkeypatch.setenv("SAT_LSTM_SPIN_LIMIT", "128")
            dec(feats, caps, lengths)
            with pytest.raises(RuntimeError, match="persistent LSTM"):
                watch.poll(block=True)
            monkeypatch.delenv("SAT_LSTM_DEBUG_STALL")
            monkeypatch.delenv("SAT_LSTM_SPIN_LIMIT")
            assert lib.sat_lstm_persist_enable(0) == 0            # the watch switched the process to per-step launches ...
            got = dec(feats, caps, lengths).clone()               # ... which give the same logits
            watch.poll(block=True)
            assert torch.allclose(got, want, rtol=0, atol=2e-6)   # another summation order inside the recurrent GEMM
        finally:
            lib.sat_lstm_persist_enable(1)
        again = dec(feats, caps, lengths).clone()                 # persistent again, clean
        watch.poll(block=True)
        assert torch.equal(again, want)


@pytest.mark.parametrize("stall", ["1", "2"])
def test_a_stalled_train_step_never_reaches_the_parameters(hooks_lib, monkeypatch, stall):
    """a persistent LSTM launch that gives up (forward: SAT_LSTM_DEBUG_STALL=1, backward: =2) leaves garbage gradients;
    `TrainStep` folds the status words into the step's fault flag ON THE DEVICE (sat_step_fault_flag) and clamp + Adam read it
    (sat_clamp_adam_step_guarded): parameters, Adam moments and the step count must be bit for bit those before the faulted step
    -- also for a step submitted after it and before the host has seen the flag -- the RuntimeError must say so, and training
    continues with per-step launches.  train.py:144-146 / models.py:52"""
    import importlib
    lib = hooks_lib
    sat = importlib.import_module("show-and-tell_amd")
    torch.manual_seed(3)
    model = sat.ShowAndTell(32, 64, 150, 1, arch=dict(layers=(1, 1, 1, 1), width=8), compute_dtype="f32").cuda().train()
    ts = sat.TrainStep(model, lr=1e-3, grad_clip=0.1)
    g = torch.Generator().manual_seed(5)
    B = 16
    images = torch.randn(B, 3, 64, 64, generator=g).cuda()
    caps = torch.randint(4, 150, (B, 10), generator=g).cuda()
    caps[:, 0], caps[:, -1] = 1, 2
    lengths = [10] * B
    try:
        ts.step(images, caps, lengths)
        ts.step(images, caps, lengths)
        ts.check_ids()                                           # two clean steps
        before = [t.clone() for t in (ts.flat.params, ts.flat.m, ts.flat.v)]
        count = ts.step_count
        monkeypatch.setenv("SAT_LSTM_DEBUG_STALL", stall)
        monkeypatch.setenv("SAT_LSTM_SPIN_LIMIT", "128")
        fwd = ts.forward_backward((images, caps, lengths), 1.0 / (B * 9))        # the faulted step ...
        ts.optimizer_step()
        monkeypatch.delenv("SAT_LSTM_DEBUG_STALL")
        monkeypatch.delenv("SAT_LSTM_SPIN_LIMIT")
        del fwd
        # ... and a clean one behind it, submitted without looking (forward_backward / optimizer_step do not block): sticky flag
        ts.forward_backward((images, caps, lengths), 1.0 / (B * 9))
        with pytest.raises(RuntimeError, match="SKIPPED on the device"):
            ts.optimizer_step()
            ts.check_ids()
        torch.cuda.synchronize()
        for a, b in zip(before, (ts.flat.params, ts.flat.m, ts.flat.v)):
            assert torch.equal(a, b)                              # nothing of the faulted step (or the one behind it) got through
        assert ts.step_count == count
        assert lib.sat_lstm_persist_enable(0) == 0                # the process now runs one launch per step
        ts.step(images, caps, lengths)                            # training goes on, updates flow again
        ts.check_ids()
        assert ts.step_count == count + 1 and not torch.equal(before[0], ts.flat.params)
        # ... and equals the step a never-faulted engine takes from the same state (per-step launches on both sides)
        torch.manual_seed(3)
        model2 = sat.ShowAndTell(32, 64, 150, 1, arch=dict(layers=(1, 1, 1, 1), width=8), compute_dtype="f32").cuda().train()
        ts2 = sat.TrainStep(model2, lr=1e-3, grad_clip=0.1)
        for _ in range(3):
            ts2.step(images, caps, lengths)
        ts2.check_ids()
        assert (ts2.flat.params - ts.flat.params).abs().max().item() < 5e-4      # persistent vs per-step recurrence: another summation order (Adam amplifies a near-zero gradient element)
    finally:
        lib.sat_lstm_persist_enable(1)


@pytest.mark.parametrize("variant", [22, 23, 24])
def test_wide_tile_output_stores_are_stable_over_many_launches(lib, variant):
    """Regression pin for the `store16_wt` hazard (robustness 13): the write-through output stores are inline asm,
    and a VMEM store of more than 64 bits needs wait states before its data VGPRs are overwritten -- the compiler's hazard
    recognizer does not look inside asm.  Without the `s_nop 1` in `store16_wt` (sat_common.h) the 256-wide tiles -- whose
    store loop is the longest and reuses its registers at once -- produced occasional wrong 16-byte chunks.  The geometry
    below (several 256-wide tiles, M and N tails) repeated 40 times must give the reference every time, bit for bit equal
    between launches.  The static half of the pin is tests/test_cabi_and_host.py::test_write_through_store_keeps_its_wait_states."""
    g = torch.Generator().manual_seed(variant)
    N, H, W, Cin, Cout, k = 3, 15, 13, 64, 320, 3
    x = torch.randn(N, Cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).bfloat16().float()
    ref = F.conv2d(x.double(), w.double(), None, 1, 1).permute(0, 2, 3, 1).reshape(-1, Cout)
    o, keep, _ = _conv_op(L.SAT_BF16, x.permute(0, 2, 3, 1), w.permute(0, 2, 3, 1), 1, 1)
    o.variant = variant
    first = None
    for rep in range(40):
        keep[2].fill_(float("nan"))
        L.check(lib.sat_run_ops(C.pointer(o), 1, st()))
        out = keep[2].clone()
        if first is None:
            first = out
            assert (out.float().cpu().double() - ref).abs().max().item() < 2e-2
        else:
            assert torch.equal(out, first), rep


@pytest.mark.parametrize("kind", ["pw", "aw", "aw8"])
def test_weights_in_registers_kernels_repeat_bit_for_bit_under_full_occupancy(lib, kind):
    """conv_pw_kernel / conv_aw_kernel hand LDS buffers over with raw barriers and counted waits (triple-buffered patch slices with the
    fetch issued from inline asm; activations staged through registers into three LDS buffers): a race there would show as launches
    that differ.  The real layer-3 geometry with the fused operand BatchNorm at the row count of a two-batch launch (392 / 1568
    workgroups: CUs with two resident workgroups next to CUs with one), 30 launches: every launch bit-identical to the first,
    statistics included."""
    g = torch.Generator().manual_seed(len(kind))
    k, Cin, Cout, variant = (3, 256, 256, 32) if kind == "pw" else (1, 256, 1024, 33 if kind == "aw" else 34)
    N, H, W, G = 128, 14, 14, 1
    x = (torch.randn(G * N, H, W, Cin, generator=g) * 1.5 + 0.2).bfloat16()
    w = (torch.randn(Cout, k, k, Cin, generator=g) / (Cin * k * k) ** 0.5).bfloat16()
    o, keep, _ = _conv_op(L.SAT_BF16, x.float(), w.float(), 1, k // 2, stats=False)
    wp = _pack_weights(lib, keep[1], Cout, Cin, k * k)
    acc = torch.zeros(G, 2, 2, Cout, dtype=torch.int64, device="cuda")
    sc, sh = cu(torch.rand(Cin, generator=g) + 0.5), cu(torch.randn(Cin, generator=g) * 0.2)
    o.N, o.groups, o.variant, o.w_packed, o.stat_acc = N, G, variant, wp.data_ptr(), acc.data_ptr()
    o.scale0, o.shift0 = sc.data_ptr(), sh.data_ptr()
    first = None
    for rep in range(30):
        keep[2].fill_(float("nan"))
        acc.zero_()
        L.check(lib.sat_run_ops_parity(C.pointer(o), 1, 0, st()))
        out, a = keep[2].clone(), acc.clone()
        if first is None:
            first = (out, a)
            assert torch.isfinite(out.float()).all() and int(a[:, 0].abs().sum()) > 0
        else:
            assert torch.equal(out, first[0]) and torch.equal(a, first[1]), rep


@pytest.mark.parametrize("M,N,K,ks,bias", [(128, 128, 64, 1, False), (200, 260, 192, 1, True), (1216, 1000, 512, 1, True),
                                           (300, 512, 1280, 3, False), (77, 64, 4096, 16, False)])
def test_gemm_bf16_nt_vs_f64_of_the_same_bf16_operands(lib, M, N, K, ks, bias):
    """`sat_gemm_bf16_nt` (the bf16-mode vocab GEMMs, models.py:53): C = A B^T (+ bias), f32 accumulate of exact bf16 products --
    tile tails in M and N, short and long K, split-K slabs"""
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).bfloat16()
    B = torch.randn(N, K, generator=g).bfloat16()
    b = torch.randn(N, generator=g) if bias else None
    ref = A.double() @ B.double().t() + (b.double() if bias else 0.0)
    Ad, Bd = cu(A), cu(B)
    nsl = -(-(K // 64) // (-(-(K // 64) // ks)))
    out = torch.full((nsl, M, N), float("nan"), device="cuda")
    bd = cu(b) if bias else None
    L.check(lib.sat_gemm_bf16_nt(Ad.data_ptr(), K, Bd.data_ptr(), K, out.data_ptr(), N, L.ptr(bd), M, N, K, ks, M * N, st()))
    sync()
    got = out.double().sum(0).cpu() if ks > 1 else out[0].double().cpu()
    assert torch.isfinite(got).all()
    tol = 2e-6 * (A.abs().double() @ B.abs().double().t()).max().item() * max(1.0, (K / 512) ** 0.5) + 1e-5
    assert (got - ref).abs().max().item() < tol, ((got - ref).abs().max().item(), tol)


@pytest.mark.parametrize("N,H,V", [(200, 128, 1000), (76, 64, 500), (1216, 512, 10000)])
def test_vocab_ce_bf16_path_vs_exact_f32_path(lib, N, H, V):
    """bf16 throughput mode of the vocab projection + CE + its backward (`sat_vocab_ce_fwd_bf16` / `_bwd_bf16`) against the
    exact-f32 kernels of the parity mode on the same inputs: logits to bf16 operand rounding, mean CE 1e-3, gradients 1 %
    relative L2 (train.py:143-144)"""
    g = torch.Generator().manual_seed(N + V)
    Hs = cu(torch.tanh(torch.randn(N, H, generator=g)))
    W = cu(torch.empty(V, H).uniform_(-0.1, 0.1, generator=g))
    b = cu(torch.randn(V, generator=g) * 0.01)
    tg = cu(torch.randint(0, V, (N,), generator=g))
    inv = 1.0 / N
    ldl = V
    # exact-f32 reference path
    lg32 = torch.empty(N, ldl, device="cuda")
    L.check(lib.sat_vocab_logits_fwd(Hs.data_ptr(), W.data_ptr(), b.data_ptr(), N, H, V, lg32.data_ptr(), ldl, st()))
    rl32, loss32 = torch.empty(N, device="cuda"), torch.zeros(1, device="cuda")
    logits32 = lg32.clone()
    L.check(lib.sat_ce_rows(lg32.data_ptr(), ldl, tg.data_ptr(), N, V, inv, 1, rl32.data_ptr(), loss32.data_ptr(), st()))
    wsb = lib.sat_vocab_ce_bwd_ws_bytes(N, H, V)
    ws = torch.empty(max(wsb // 4, 4), device="cuda")
    dW32, db32, dH32 = torch.empty(V, H, device="cuda"), torch.empty(V, device="cuda"), torch.empty(N, H, device="cuda")
    L.check(lib.sat_vocab_ce_bwd(lg32.data_ptr(), ldl, Hs.data_ptr(), W.data_ptr(), N, H, V, dW32.data_ptr(), db32.data_ptr(), dH32.data_ptr(),
                                 ws.data_ptr(), wsb, st()))
    # bf16 path
    wb = lib.sat_vocab_bf16_ws_bytes(N, H, V)
    assert wb > 0
    wsb16 = torch.empty(wb, dtype=torch.uint8, device="cuda")
    lg16 = torch.full((N, ldl), float("nan"), device="cuda")
    rl16, loss16 = torch.empty(N, device="cuda"), torch.zeros(1, device="cuda")
    L.check(lib.sat_vocab_ce_fwd_bf16(Hs.data_ptr(), W.data_ptr(), b.data_ptr(), tg.data_ptr(), N, H, V, inv, lg16.data_ptr(), ldl,
                                      rl16.data_ptr(), loss16.data_ptr(), wsb16.data_ptr(), wb, st()))
    dW16, db16, dH16 = (torch.full((V, H), float("nan"), device="cuda"), torch.full((V,), float("nan"), device="cuda"),
                        torch.full((N, H), float("nan"), device="cuda"))
    L.check(lib.sat_vocab_ce_bwd_bf16(N, H, V, dW16.data_ptr(), db16.data_ptr(), dH16.data_ptr(), wsb16.data_ptr(), wb, st()))
    sync()
    for t in (lg16, dW16, db16, dH16, loss16):
        assert torch.isfinite(t).all()
    # logits: bf16-rounded operands, f32 accumulate == f64 product of the rounded operands
    ref = (Hs.bfloat16().double() @ W.bfloat16().double().t() + b.double()).cpu()
    assert (lg16.double().cpu() - ref).abs().max().item() < 1e-5
    assert (lg16 - logits32).abs().max().item() < 2e-2
    assert abs(loss16.item() - loss32.item()) < 1e-3

    def rel(a, c):
        return ((a - c).norm() / c.norm()).item()
    print("bf16 vocab path N=%d H=%d V=%d: |dCE| %.2e, rel-L2 dW %.2e db %.2e dHs %.2e"
          % (N, H, V, abs(loss16.item() - loss32.item()), rel(dW16, dW32), rel(db16, db32), rel(dH16, dH32)))
    assert rel(dW16, dW32) < 1e-2 and rel(db16, db32) < 1e-2 and rel(dH16, dH32) < 1e-2
    assert lib.sat_vocab_bf16_ws_bytes(N, 100, V) == 0 and lib.sat_vocab_bf16_ws_bytes(N, H, 20000) == 0      # unsupported shapes say so


@pytest.mark.parametrize("B,T,In,H", [(8, 6, 32, 64), (64, 19, 256, 512), (5, 4, 36, 48)])
def test_lstm_layer_bf16_pipe_gemms_vs_exact_f32(lib, B, T, In, H):
    """`sat_lstm_fwd_bf16` / `sat_lstm_bwd_bf16` (the layer's batched GEMMs -- x-gates, dW_ih, dW_hh, dX -- on the bf16 matrix pipe
    from bf16 copies of the f32 operands; recurrence and gate arithmetic f32) against `sat_lstm_fwd` / `sat_lstm_bwd`: tapes and
    gradients within bf16 operand rounding (1 % relative L2); ragged batch, K tails (models.py:52, train.py:144)."""
    g = torch.Generator().manual_seed(B + T + H)
    lengths = sorted([int(x) for x in torch.randint(1, T + 1, (B,), generator=g)], reverse=True)
    lengths[0] = T
    pi = sat.PackInfo.get(lengths, "cuda")
    N = pi.N
    k = 1.0 / H ** 0.5
    X = torch.randn(N, In, generator=g)
    ws_ = [torch.empty(4 * H, In).uniform_(-k, k, generator=g), torch.empty(4 * H, H).uniform_(-k, k, generator=g),
           torch.empty(4 * H).uniform_(-k, k, generator=g), torch.empty(4 * H).uniform_(-k, k, generator=g)]
    dH = torch.randn(N, H, generator=g) * 0.1
    d = [cu(t) for t in [X] + ws_ + [dH]]
    mb = lib.sat_lstm_mixed_ws_bytes(N, In, H)
    assert mb > 0
    mixed = torch.empty(mb, dtype=torch.uint8, device="cuda")

    def run(bf):
        GA, CS = torch.full((N, 4 * H), float("nan"), device="cuda"), torch.full((N, H), float("nan"), device="cuda")
        HS, HP = torch.full((N, H), float("nan"), device="cuda"), torch.full((N, H), float("nan"), device="cuda")
        cst = torch.empty(B, H, device="cuda")
        wsb = lib.sat_lstm_fwd_ws_bytes(B, H)
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device="cuda")
        args = (d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(), pi.bs_c, pi.T, In, H,
                GA.data_ptr(), CS.data_ptr(), HS.data_ptr(), HP.data_ptr(), cst.data_ptr(), ws.data_ptr(), wsb)
        if bf:
            L.check(lib.sat_lstm_fwd_bf16(*args, mixed.data_ptr(), mb, st()))
        else:
            L.check(lib.sat_lstm_fwd(*args, st()))
        DG = torch.full((N, 4 * H), float("nan"), device="cuda")
        dwi, dwh = torch.full((4 * H, In), float("nan"), device="cuda"), torch.full((4 * H, H), float("nan"), device="cuda")
        dbi, dbh, dX = torch.empty(4 * H, device="cuda"), torch.empty(4 * H, device="cuda"), torch.full((N, In), float("nan"), device="cuda")
        bwsb = lib.sat_lstm_bwd_ws_bytes_full(N, B, In, H)
        bws = torch.empty(bwsb // 4, device="cuda")
        bargs = (d[5].data_ptr(), d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), GA.data_ptr(), CS.data_ptr(), HP.data_ptr(), pi.bs_c, pi.T,
                 In, H, DG.data_ptr(), dwi.data_ptr(), dwh.data_ptr(), dbi.data_ptr(), dbh.data_ptr(), dX.data_ptr(), bws.data_ptr(), bwsb)
        if bf:
            L.check(lib.sat_lstm_bwd_bf16(*bargs, mixed.data_ptr(), mb, st()))
        else:
            L.check(lib.sat_lstm_bwd(*bargs, st()))
        sync()
        return dict(HS=HS.cpu(), CS=CS.cpu(), dwi=dwi.cpu(), dwh=dwh.cpu(), dbi=dbi.cpu(), dX=dX.cpu())
    ref, got = run(False), run(True)
    for key in ref:
        assert torch.isfinite(got[key]).all(), key
        rel = ((got[key] - ref[key]).norm() / (ref[key].norm() + 1e-20)).item()
        assert rel < 1e-2, (key, rel)
    assert lib.sat_lstm_fwd_bf16(*([None] * 5), pi.bs_c, pi.T, In, H, *([None] * 6), 0, None, 0, st()) == 1001


@pytest.mark.parametrize("B,T,In,H,ragged", [(64, 19, 256, 512, False), (64, 19, 256, 512, True), (16, 7, 32, 64, True), (5, 4, 36, 48, True),
                                             (24, 12, 64, 128, True)])
def test_lstm_bwd_persistent_recurrence_equals_per_step_launches(lib, B, T, In, H, ragged, monkeypatch, request):
    """sat_lstm_bwd with the FULL workspace (the backward recurrence as ONE persistent launch: W_hh quarter per wave in registers,
    per-group exchange of the d(pre-activation) rows, dc in a register) against the same entry point with the minimal workspace
    (one fused launch per step): DG and every gradient to 2e-6 relative of the largest entry (another K summation order); status
    word clean; and a withheld hand-off (SAT_LSTM_DEBUG_STALL=2) is REPORTED through the status word.  train.py:144 / models.py:52"""
    g = torch.Generator().manual_seed(B * 5 + T + H)
    lengths = sorted([int(x) for x in torch.randint(1, T + 1, (B,), generator=g)], reverse=True) if ragged else [T] * B
    lengths[0] = T
    pi = sat.PackInfo.get(lengths, "cuda")
    N = pi.N
    k = 1.0 / H ** 0.5
    X = torch.randn(N, In, generator=g)
    ws_ = [torch.empty(4 * H, In).uniform_(-k, k, generator=g), torch.empty(4 * H, H).uniform_(-k, k, generator=g),
           torch.empty(4 * H).uniform_(-k, k, generator=g), torch.empty(4 * H).uniform_(-k, k, generator=g)]
    dH = torch.randn(N, H, generator=g) * 0.1
    d = [cu(t) for t in [X] + ws_ + [dH]]
    GA, CS = torch.empty(N, 4 * H, device="cuda"), torch.empty(N, H, device="cuda")
    HS, HP = torch.empty(N, H, device="cuda"), torch.empty(N, H, device="cuda")
    cst = torch.empty(B, H, device="cuda")
    L.check(lib.sat_lstm_fwd(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(), pi.bs_c, pi.T, In, H,
                             GA.data_ptr(), CS.data_ptr(), HS.data_ptr(), HP.data_ptr(), cst.data_ptr(), None, 0, st()))
    full = lib.sat_lstm_bwd_ws_bytes_full(N, B, In, H)
    soff = lib.sat_lstm_bwd_status_offset(N, B, In, H)
    # the exchange region and the status word come first, at offsets that depend on (B, H) only (one buffer serves every N)
    assert 0 <= soff < full and soff == lib.sat_lstm_bwd_status_offset(N + 7, B, In + 4, H)
    assert lib.sat_lstm_bwd_ws_bytes_max(N, B, In, H) >= full

    def run(nbytes, lib=lib):
        DG = torch.full((N, 4 * H), float("nan"), device="cuda")
        dwi, dwh = torch.full((4 * H, In), float("nan"), device="cuda"), torch.full((4 * H, H), float("nan"), device="cuda")
        dbi, dbh, dX = torch.empty(4 * H, device="cuda"), torch.empty(4 * H, device="cuda"), torch.full((N, In), float("nan"), device="cuda")
        bws = torch.full((nbytes // 4,), float("nan"), device="cuda")
        L.check(lib.sat_lstm_bwd(d[5].data_ptr(), d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), GA.data_ptr(), CS.data_ptr(), HP.data_ptr(),
                                 pi.bs_c, pi.T, In, H, DG.data_ptr(), dwi.data_ptr(), dwh.data_ptr(), dbi.data_ptr(), dbh.data_ptr(),
                                 dX.data_ptr(), bws.data_ptr(), nbytes, st()))
        sync()
        status = int(bws.view(torch.int32)[soff // 4].item()) if nbytes >= full else 0
        return dict(DG=DG.cpu(), dwi=dwi.cpu(), dwh=dwh.cpu(), dbi=dbi.cpu(), dX=dX.cpu()), status
    step, _ = run(lib.sat_lstm_bwd_ws_bytes(B, H))
    pers, status = run(full)
    assert status == 0, "persistent backward recurrence timed out"
    for key in step:
        assert torch.isfinite(pers[key]).all(), key
        scale = step[key].abs().max().item() + 1e-30
        assert (pers[key] - step[key]).abs().max().item() < 2e-6 * scale + 1e-9, (key, (pers[key] - step[key]).abs().max().item(), scale)
    if lib.sat_lstm_fwd_ws_bytes(B, H) > 0 and B >= 16:
        monkeypatch.setenv("SAT_LSTM_DEBUG_STALL", "2")
        _, status = run(full)
        assert status == 0                       # the PRODUCT library has no fault-injection switch: the variable does nothing
        monkeypatch.setenv("SAT_LSTM_SPIN_LIMIT", "128")
        _, status = run(full, request.getfixturevalue("hooks_lib"))      # the test build (-DSAT_TESTHOOKS) withholds a hand-off
        assert status != 0                       # reported, not swallowed (watch.ResidencyWatch raises on it)
