"""GPU (MI355X): the Inception-v3 encoder of BASELINE.json configs[3] (299x299, with the 2-layer hidden-1024 decoder) on the HIP
path against the CPU oracle (`oracle/inception.py`; "parity unpinned": the reference has no Inception)."""
import ctypes as C
import importlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

sat = importlib.import_module("show-and-tell_amd")
L = sat._lib
from oracle import decoder as OD  # noqa: E402
from oracle import encoder as OE  # noqa: E402
from oracle import inception as OI  # noqa: E402
from oracle import train_step as OT  # noqa: E402


def st():
    return L.stream()


@pytest.mark.parametrize("dtype", [L.SAT_F32, L.SAT_BF16])
@pytest.mark.parametrize("kh,kw,ph,pw,stride", [(1, 7, 0, 3, 1), (7, 1, 3, 0, 1), (1, 3, 0, 1, 1), (3, 1, 1, 0, 1), (5, 5, 2, 2, 1), (3, 3, 0, 0, 2)])
def test_rectangular_conv_per_axis_padding_into_a_channel_slice(dtype, kh, kw, ph, pw, stride):
    """SAT_CONV_PADW + sat_op.ldc: 1x7 / 7x1 / 1x3 / 3x1 kernels with per-axis padding, output written into a channel slice of a
    wider NHWC tensor (the Inception concatenation), statistics slabs as usual"""
    lib = L.load()
    td = torch.bfloat16 if dtype == L.SAT_BF16 else torch.float32
    g = torch.Generator().manual_seed(kh * 10 + kw)
    N, H, W, Cin, Cout, Ctot, off = 3, 17, 17, 160, 192, 448, 128
    x = torch.randn(N, H, W, Cin, generator=g).to(td)
    w = (torch.randn(Cout, kh, kw, Cin, generator=g) / (Cin * kh * kw) ** 0.5).to(td)
    Ho, Wo = (H + 2 * ph - kh) // stride + 1, (W + 2 * pw - kw) // stride + 1
    xd, wd = x.cuda(), w.reshape(Cout, -1).contiguous().cuda()
    out = torch.full((N, Ho, Wo, Ctot), 7.0, device="cuda", dtype=td)
    tiles = lib.sat_conv_tiles_m(N * Ho * Wo)
    part = torch.zeros(tiles, 2, Cout, device="cuda")
    o = L.SatOp()
    o.kind, o.dtype = L.OP_CONV, dtype
    o.in0, o.w, o.out = xd.data_ptr(), wd.data_ptr(), out.data_ptr() + off * x.element_size()
    o.N, o.Hin, o.Win, o.Cin, o.Hout, o.Wout, o.Cout = N, H, W, Cin, Ho, Wo, Cout
    o.KH, o.KW, o.stride, o.pad, o.pad_w, o.flags, o.ldc = kh, kw, stride, ph, pw, L.CONV_PADW, Ctot
    o.sN, o.sH, o.sW = H * W * Cin, W * Cin, Cin
    o.stat_partial, o.tiles_m = part.data_ptr(), tiles
    L.check(lib.sat_run_ops(C.pointer(o), 1, st()))
    torch.cuda.synchronize()
    ref = F.conv2d(x.float().permute(0, 3, 1, 2).double(), w.float().permute(0, 3, 1, 2).double(), None, stride, (ph, pw)).permute(0, 2, 3, 1)
    got = out.float().cpu().double()
    tol = 2e-5 if dtype == L.SAT_F32 else 2e-2
    assert (got[..., off:off + Cout] - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item())
    assert float((got[..., :off] - 7.0).abs().max()) == 0 and float((got[..., off + Cout:] - 7.0).abs().max()) == 0     # neighbours untouched
    np.testing.assert_allclose(part[:, 0].sum(0).cpu().double().numpy(), ref.reshape(-1, Cout).sum(0).numpy(), rtol=0, atol=3e-2 + 1e-3 * (N * Ho * Wo) ** 0.5)


@pytest.mark.parametrize("dtype", [L.SAT_F32, L.SAT_BF16])
def test_pool3_ops_and_strided_bn_relu(dtype):
    lib = L.load()
    td = torch.bfloat16 if dtype == L.SAT_BF16 else torch.float32
    g = torch.Generator().manual_seed(5)
    N, H, W, Cc, Ctot, off = 2, 17, 17, 64, 160, 96
    x = torch.randn(N, H, W, Cc, generator=g).to(td)
    xd = x.cuda()
    Ho = (H - 3) // 2 + 1
    mo = torch.full((N, Ho, Ho, Ctot), 3.0, device="cuda", dtype=td)
    m = L.SatOp()
    m.kind, m.dtype = L.OP_MAXPOOL3S2, dtype
    m.in0, m.out, m.ldc = xd.data_ptr(), mo.data_ptr() + off * x.element_size(), Ctot
    m.N, m.Hin, m.Win, m.Cout = N, H, W, Cc
    ao = torch.empty(N, H, W, Cc, device="cuda", dtype=td)
    a = L.SatOp()
    a.kind, a.dtype = L.OP_AVGPOOL3, dtype
    a.in0, a.out = xd.data_ptr(), ao.data_ptr()
    a.N, a.Hin, a.Win, a.Cout = N, H, W, Cc
    sc, sh = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g)
    scd, shd = sc.cuda(), sh.cuda()
    bo = torch.full((N, H, W, Ctot), 3.0, device="cuda", dtype=td)
    b = L.SatOp()
    b.kind, b.dtype = L.OP_BN_RELU, dtype
    b.in0, b.out, b.ldc = xd.data_ptr(), bo.data_ptr() + off * x.element_size(), Ctot
    b.scale0, b.shift0 = scd.data_ptr(), shd.data_ptr()
    b.N, b.Hout, b.Wout, b.Cout = N, H, W, Cc
    ops = (L.SatOp * 3)(m, a, b)
    L.check(lib.sat_run_ops(ops, 3, st()))
    torch.cuda.synchronize()
    xf = x.float().permute(0, 3, 1, 2)
    assert torch.equal(mo[..., off:off + Cc].float().cpu(), F.max_pool2d(xf, 3, 2).permute(0, 2, 3, 1))
    assert float((mo[..., :off].float() - 3).abs().max()) == 0
    tol = 1e-6 if dtype == L.SAT_F32 else 2e-2
    assert (ao.float().cpu() - F.avg_pool2d(xf, 3, 1, 1).permute(0, 2, 3, 1)).abs().max().item() < tol
    ref = torch.clamp(x.float() * sc + sh, min=0)
    assert (bo[..., off:off + Cc].float().cpu() - ref).abs().max().item() < (1e-5 if dtype == L.SAT_F32 else 4e-2)
    assert float((bo[..., :off].float() - 3).abs().max()) == 0


def _encoder(dtype, seed=3, E=64):
    g = torch.Generator().manual_seed(seed)
    params, buffers = OI.init_inception_params(E, generator=g, randomize_bn=True)
    enc = sat.EncoderCNN(E, arch="inception_v3", compute_dtype=dtype)
    sd = dict(params)
    sd.update(buffers)
    enc.load_state_dict(sd)
    return enc.cuda(), params, buffers


@pytest.mark.timeout(600)
def test_inception_f32_matches_oracle_train_and_eval():
    """94 BasicConv2d, 11 Inception blocks, 299x299 (batch 4): f32 MFMA stack vs the CPU oracle, batch statistics then
    running statistics"""
    enc, params, buffers = _encoder("f32")
    x = torch.randn(4, 3, 299, 299, generator=torch.Generator().manual_seed(4))
    bufs = {k: v.clone() for k, v in buffers.items()}
    ref = OI.inception_forward(params, bufs, x, training=True)
    got = enc.train().pooled_features(x.cuda())
    err = (got.cpu() - ref).abs().max().item()
    assert got.shape == (4, 2048) and err < 2e-3 * max(1.0, ref.abs().max().item()), err
    sd = enc.state_dict()
    for k in ("resnet.Conv2d_1a_3x3.bn.running_mean", "resnet.Mixed_6c.branch7x7dbl_3.bn.running_var", "resnet.Mixed_7c.branch_pool.bn.running_mean"):
        np.testing.assert_allclose(sd[k].cpu().numpy(), bufs[k].numpy(), rtol=2e-3, atol=2e-5, err_msg=k)
    for _ in range(30):                       # let the running statistics converge, then compare eval passes
        OI.inception_forward(params, bufs, x, training=True)
        enc.pooled_features(x.cuda())
    ref_e = OI.inception_forward(params, bufs, x, training=False)
    got_e = enc.eval().pooled_features(x.cuda())
    assert (got_e.cpu() - ref_e).abs().max().item() < 5e-3 * max(1.0, ref_e.abs().max().item())


@pytest.mark.timeout(600)
def test_inception_bf16_vs_oracle_within_the_bf16_noise_floor():
    enc, params, buffers = _encoder("bf16", seed=6)
    x = torch.randn(8, 3, 299, 299, generator=torch.Generator().manual_seed(7))
    bufs = {k: v.clone() for k, v in buffers.items()}
    ref = OI.inception_forward(params, bufs, x, training=True)
    ref_bf = OI.inception_forward(params, bufs, x, training=True, bf16_storage=True)
    got = [enc.train().pooled_features(x.cuda()).cpu() for _ in range(3)]          # eager, eager (other parity), graph replay
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()   # noqa: E731
    floor = rel(ref_bf, ref)
    print("inception bf16: vs bf16-storage oracle %.4f, vs f32 oracle %.4f, oracle floor %.4f" % (rel(got[0], ref_bf), rel(got[0], ref), floor))
    assert torch.isfinite(got[0]).all()
    assert rel(got[0], ref_bf) < 1.25 * floor + 0.02 and rel(got[0], ref) < 1.5 * floor + 0.02
    for g in got[1:]:
        assert torch.equal(g, got[0])


def test_building_an_inception_program_leaves_the_model_untouched():
    """the shared tuner times whole-program passes on random images before the first real forward (ConvStackProgram._pick_in_program):
    running statistics, counters and parameters of the stack stay as they were, the statistics accumulators are handed over zeroed"""
    enc, _, _ = _encoder("bf16", seed=8)
    enc.train()
    before = {k: v.clone() for k, v in enc.state_dict().items()}
    x = torch.randn(4, 3, 299, 299, generator=torch.Generator().manual_seed(9)).cuda()
    prog = enc._program(x)
    torch.cuda.synchronize()
    after = enc.state_dict()
    for k, v in before.items():
        assert torch.equal(v, after[k]), k
    assert prog._parity == 0 and prog._runs == [0, 0] and prog.stat_accs
    for acc in prog.stat_accs:
        assert int(acc.abs().sum()) == 0
    assert all(int(prog.ops[i].variant) > 0 for i in range(prog.n_ops) if prog.ops[i].kind == sat._lib.OP_CONV)


@pytest.mark.timeout(900)
def test_cfg4_train_step_inception_encoder_two_layer_lstm_vs_oracle():
    """BASELINE configs[3]: Inception-v3 299x299 + embed 512 / hidden 1024 / 2 LSTM layers, one whole train.py:126-146 iteration
    (f32 mode, batch 4) against the CPU oracle; then the bf16 mode runs the same step (loss near ln V, finite)"""
    E, H, V, Lh, B, T = 512, 1024, 10000, 2, 4, 20
    g = torch.Generator().manual_seed(9)
    ep, eb = OI.init_inception_params(E, generator=g)
    dp = OD.init_decoder_params(E, H, V, Lh, generator=g)
    images = torch.randn(B, 3, 299, 299, generator=g)
    caps = torch.randint(4, V, (B, T), generator=g)
    caps[:, 0], caps[:, -1] = 1, 2
    lengths = [T] * B
    bufs = {k: v.clone() for k, v in eb.items()}
    pooled = OI.inception_forward(ep, bufs, images, training=True)
    feats, tape = OE.head_forward(ep, bufs, pooled, training=True)
    ref_loss, ref_grads, d_feat, _ = OT.decoder_loss_and_grads(dp, feats, caps, lengths, Lh)
    hg = OE.head_backward(ep, tape, d_feat)
    losses = {}
    for dtype in ("f32", "bf16"):
        model = sat.ShowAndTell(E, H, V, Lh, arch="inception_v3", compute_dtype=dtype)
        sd = dict(ep)
        sd.update(eb)
        model.encoder.load_state_dict(sd)
        model.decoder.load_state_dict(dp)
        model.cuda().train()
        ts = sat.TrainStep(model)
        loss = ts.forward_backward((images.cuda(), caps.cuda(), lengths), 1.0 / (B * (T - 1))).clone()
        losses[dtype] = loss.item()
        if dtype == "f32":
            assert abs(loss.item() - ref_loss.item()) < 1e-4
            np.testing.assert_allclose(ts.flat.grad("decoder.lstm.weight_hh_l1").cpu().numpy(), ref_grads["lstm.weight_hh_l1"].numpy(), rtol=5e-3, atol=2e-7)
            np.testing.assert_allclose(ts.flat.grad("encoder.bn.weight").cpu().numpy(), hg["bn.weight"].numpy(), rtol=5e-2, atol=1e-6)
        ts.optimizer_step()
        ts.check_ids()
    print("cfg4 loss: oracle %.5f f32 %.5f bf16 %.5f" % (ref_loss.item(), losses["f32"], losses["bf16"]))
    assert abs(losses["bf16"] - ref_loss.item()) < 5e-3


@pytest.mark.timeout(600)
def test_inception_grouped_lookahead_is_bitwise_the_sequential_run():
    """BASELINE configs[3] under the look-ahead: three batches (the Inception default since round 5) through ONE grouped Inception
    program run (sat_op.groups = 3: every conv, statistics reducer and normalise+ReLU launch -- the channel-slice ones included -- covers
    all of them, the pools and the image prep see 3 x N images) against each batch's own ungrouped run: pooled features and every
    BatchNorm's running statistics bit for bit (models.py:27 with BatchNorm2d in train mode; per-batch statistics).  Then the eval-mode
    form (the batches concatenate)."""
    enc, _, _ = _encoder("bf16", seed=12)
    enc.train()
    g = torch.Generator().manual_seed(13)
    a, b, c = (torch.randn(4, 3, 299, 299, generator=g).cuda() for _ in range(3))
    assert enc.lookahead_groups == 3 and enc.lookahead_depth == 6
    with torch.no_grad():
        assert enc.prefetch_many([a, b, c]) == 3
        assert enc._inflight[0]["prog"].groups == 3
        pa, pb, pc = (enc.pooled_features(x).clone() for x in (a, b, c))
    sd = {k: v.clone() for k, v in enc.state_dict().items()}
    enc2, _, _ = _encoder("bf16", seed=12)
    enc2.train()
    with torch.no_grad():
        qa, qb, qc = (enc2.pooled_features(x).clone() for x in (a, b, c))
    assert torch.isfinite(pa).all() and torch.equal(pa, qa) and torch.equal(pb, qb) and torch.equal(pc, qc)
    for k, v in enc2.state_dict().items():
        assert torch.equal(v, sd[k]), k
    enc.eval(), enc2.eval()
    with torch.no_grad():
        assert enc.prefetch_many([a, b, c]) == 3
        ea, eb_, ec = (enc.pooled_features(x).clone() for x in (a, b, c))
        fa, fb, fc = (enc2.pooled_features(x).clone() for x in (a, b, c))
    assert torch.equal(ec, fc)
    assert torch.equal(ea, fa) and torch.equal(eb_, fb)
