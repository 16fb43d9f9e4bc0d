"""GPU: sat_gemm_f32x3 (csrc/sat_gemm_x3.hip) -- an f32 GEMM on the bf16 matrix pipe by a three-way bf16 split of both operands (the vocab
projection of the beam decode loop, `self.linear(hiddens)`, models.py:53 / :63).  Against the f64 product of the SAME f32 operands, with the
error measured the way f32 GEMMs are: relative to sum_k |a||w| per output.  Bound written here: 2e-6 for K <= 1024 (f32 accumulation of K
exact products: ~sqrt(K) 2^-24), AND no worse than the exact-f32 MFMA GEMM of the library run beside it on the same inputs (measured at
320 x 10000 x 512: 6e-7 against 2e-6 -- the split sums 16 products per MFMA before it rounds, the f32 pipe 2).  Shapes: the decode
projection (320 x 10000 x 512), ragged rows / columns, K from 256 to 1024, wide dynamic range, zeros; then what the entry points refuse."""
import importlib

import pytest
import torch

from test_gpu_kernels import cu, st, sync

pytestmark = pytest.mark.gpu
sat = importlib.import_module("show-and-tell_amd")
L = sat._lib


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "needs the MI355X"
    return L.load()


def _run(lib, a, w, bias, ldc=None):
    M, K = a.shape
    N = w.shape[0]
    ldc = ldc or N
    ad, wd, bd = cu(a), cu(w), (cu(bias) if bias is not None else None)
    packed = torch.empty(lib.sat_gemm_f32x3_packed_bytes(N, K), dtype=torch.uint8, device="cuda")
    out = torch.full((M, ldc), float("nan"), device="cuda")
    L.check(lib.sat_gemm_f32x3_pack(wd.data_ptr(), N, K, packed.data_ptr(), st()), "pack")
    L.check(lib.sat_gemm_f32x3(ad.data_ptr(), K, packed.data_ptr(), bd.data_ptr() if bd is not None else None, out.data_ptr(), ldc,
                               M, N, K, st()), "gemm x3")
    ref32 = torch.full((M, N), float("nan"), device="cuda")
    L.check(lib.sat_gemm_f32(0, 0, ad.data_ptr(), K, wd.data_ptr(), K, ref32.data_ptr(), N, bd.data_ptr() if bd is not None else None,
                             None, M, N, K, st()), "gemm f32")
    sync()
    return out, ref32


@pytest.mark.parametrize("M,N,K,scale", [(320, 10000, 512, 1.0), (130, 1000, 256, 1.0), (64, 128, 1024, 1.0), (257, 636, 384, 1.0),
                                         (5, 12, 256, 1.0), (200, 516, 512, 1e12), (200, 516, 512, 1e-12),
                                         (256, 32768, 256, 1.0)])       # (the last one: >= 512 tiles of 128 rows -> the one-workgroup-per-CU form)
def test_gemm_f32x3_has_the_accuracy_of_an_f32_gemm(lib, M, N, K, scale):
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, K, generator=g) * 2.0) * scale      # wide dynamic range per element
    w = torch.randn(N, K, generator=g) * 0.1
    a[::7, ::5] = 0.0
    w[::3, ::11] = 0.0
    bias = torch.randn(N, generator=g)
    ldc = N + 4 if N % 8 else N
    out, ref32 = _run(lib, a, w, bias, ldc)
    ref = a.double() @ w.double().t() + bias.double()
    mag = a.double().abs() @ w.double().abs().t() + bias.double().abs()
    got = out[:, :N].cpu().double()
    assert torch.isfinite(got).all()
    err = ((got - ref).abs() / mag).max().item()
    err32 = ((ref32.cpu().double() - ref).abs() / mag).max().item()
    print("M=%d N=%d K=%d: x3 %.2e, exact-f32 pipe %.2e (relative to sum |a||w|)" % (M, N, K, err, err32))
    assert err < 2e-6 and err <= 1.25 * err32 + 1e-7, (err, err32)
    if ldc > N:
        assert torch.isnan(out[:, N:]).all()          # the pad columns of a wider row are not written


def test_gemm_f32x3_without_bias_and_exact_cases(lib):
    # operands that ARE bf16 numbers: one product term, exact sums of few terms -> the result is exact
    g = torch.Generator().manual_seed(4)
    a = torch.randint(-8, 9, (128, 256), generator=g).float()
    w = torch.randint(-4, 5, (256, 256), generator=g).float()
    out, ref32 = _run(lib, a, w, None)
    assert torch.equal(out.cpu(), (a.double() @ w.double().t()).float())
    assert torch.equal(ref32.cpu(), out.cpu())


def test_gemm_f32x3_refuses_what_it_cannot_run(lib):
    assert lib.sat_gemm_f32x3_packed_bytes(1000, 192) == 0 and lib.sat_gemm_f32x3_packed_bytes(1000, 128) == 0
    assert lib.sat_gemm_f32x3_packed_bytes(1000, 256) == 1024 * 256 * 6
    a, w = torch.zeros(8, 256, device="cuda"), torch.zeros(16, 256, device="cuda")
    p = torch.empty(lib.sat_gemm_f32x3_packed_bytes(16, 256), dtype=torch.uint8, device="cuda")
    c = torch.empty(8, 16, device="cuda")
    assert lib.sat_gemm_f32x3_pack(w.data_ptr(), 16, 192, p.data_ptr(), st()) == 1001
    assert lib.sat_gemm_f32x3(a.data_ptr(), 256, p.data_ptr(), None, c.data_ptr(), 16, 8, 14, 256, st()) == 1003      # N % 4
    assert lib.sat_gemm_f32x3(a.data_ptr(), 256, p.data_ptr(), None, c.data_ptr(), 8, 8, 16, 256, st()) == 1003       # ldc < N
    assert lib.sat_gemm_f32x3(None, 256, p.data_ptr(), None, c.data_ptr(), 16, 8, 16, 256, st()) == 1001
    assert lib.sat_gemm_f32x3(a.data_ptr(), 256, p.data_ptr(), None, c.data_ptr(), 16, 8, 16, 256, st()) == 0
    sync()
