"""GPU (MI355X) parity AT THE BENCHMARKED CONFIGURATION: the real ResNet-152 [3,8,36,3] stack at 224x224 in the bf16
mode `bench.py` measures -- against the CPU oracle with bf16 storage emulated (`oracle.encoder.
resnet_forward_bf16_storage`), against the f32 oracle, and against this library's own f32 parity mode -- and whole
training steps at BASELINE configs[1] (batch 64) in both modes against `oracle.train_step.full_step`.

Tolerances (measured values are printed; DESIGN.md section 4 quotes them):
  * f32 mode: mean CE within 1e-4 of the oracle on every step (north_star's bar); measured 2e-6.
  * bf16 mode: the conv stack stores bf16 (2^-8 relative rounding per stored tensor), 155 convs deep with a
    BatchNorm renormalising after each.  With random weights that map is CHAOTIC at bf16 resolution: two CPU
    implementations of the very same bf16-storage arithmetic that differ only in accumulation precision (f32 vs f64
    accumulate inside each conv -- a legitimate change of summation order) end 0.16 apart in relative L2 at the pooled
    features (cfg1), the same distance either has from the f32 result (0.18).  No implementation can be pinned tighter
    than that floor at full depth, so the full-depth test measures the floor itself (two oracle runs) and requires the
    HIP path to sit within it; the tight bound (<1 %) lives where it can hold: at 6 bottlenecks
    (`test_gpu_parity.py::test_encoder_bf16_close_to_oracle`) and per conv geometry at batch 64
    (`test_resnet152_conv_geometries_batch64_autotuned_vs_cpu` below: each real layer shape, autotuned variant,
    integer-atomic statistics, against fp32 CPU convolution of the same bf16 operands).  The CE, which at these weights
    sees the features only through the BatchNorm1d-normalised head, moves by 8.5e-4 at cfg2 (bound 2e-3).
Encoder parity is build-internal either way ("parity unpinned": torchvision is absent)."""
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu

sat = importlib.import_module("show-and-tell_amd")
L = sat._lib
from oracle import decoder as OD  # noqa: E402
from oracle import encoder as OE  # noqa: E402
from oracle import train_step as OT  # noqa: E402


def _rel(a, b):
    return ((a - b).norm() / b.norm()).item()


def _cos(a, b):
    return torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0).item()


def _encoder(arch, E, params, buffers, dtype):
    enc = sat.EncoderCNN(E, arch=arch, compute_dtype=dtype)
    sd = dict(params)
    sd.update(buffers)
    enc.load_state_dict(sd)
    return enc.cuda()


@pytest.mark.timeout(900)
def test_resnet152_bf16_train_mode_cfg1_vs_bf16_storage_oracle():
    """BASELINE cfg-1 shape (batch 4, 224x224), all 152 layers, bf16 TRAIN mode (batch statistics): models.py:25-29"""
    arch, E, B = OE.RESNET152, 256, 4
    gen = torch.Generator().manual_seed(41)
    params, buffers = OE.init_encoder_params(E, arch, generator=gen, randomize_bn=True)
    x = torch.randn(B, 3, 224, 224, generator=torch.Generator().manual_seed(42))
    enc = _encoder(arch, E, params, buffers, "bf16").train()
    got = [enc.pooled_features(x.cuda()).cpu() for _ in range(3)]      # eager, eager (other parity), graph replay
    ref_bf = OE.resnet_forward_bf16_storage(params, x, arch)
    bufs = {k: v.clone() for k, v in buffers.items()}
    ref_f32, _ = OE.resnet_forward(params, bufs, x, arch, training=True)
    # the noise floor of the bf16-storage arithmetic itself at this depth: the same emulation with every conv
    # accumulated in f64 instead of f32 (summation order / last-bit differences before each bf16 rounding)
    import torch.nn.functional as F
    conv2d = F.conv2d
    try:
        F.conv2d = lambda a, w, b=None, stride=1, padding=0: conv2d(a.double(), w.double(), None, stride, padding).float()
        ref_bf64 = OE.resnet_forward_bf16_storage(params, x, arch)
    finally:
        F.conv2d = conv2d
    floor = _rel(ref_bf64, ref_bf)
    r_bf, r_f32, r_or = _rel(got[0], ref_bf), _rel(got[0], ref_f32), _rel(ref_bf, ref_f32)
    print("cfg1 bf16 pooled: rel-L2 vs bf16-storage oracle %.4f, vs f32 oracle %.4f | oracle noise floor (f32- vs f64-accumulate "
          "bf16-storage oracle) %.4f, bf16-storage oracle vs f32 oracle %.4f | cos %.5f"
          % (r_bf, r_f32, floor, r_or, _cos(got[0], ref_bf)))
    assert torch.isfinite(got[0]).all()
    # within the floor: no further from the emulation than the emulation is from its own re-ordered self (+25 %)
    assert r_bf < 1.25 * floor + 0.02, (r_bf, floor)
    assert r_f32 < 1.25 * r_or + 0.02, (r_f32, r_or)
    assert _cos(got[0], ref_bf) > 0.97
    for g in got[1:]:                                      # same input, same statistics: every pass reproduces the first
        assert torch.equal(g, got[0])


def _cfg2_models(dtype, seed=123, B=64, conditioning=None):
    gen = torch.Generator().manual_seed(seed)
    ep, eb = OE.init_encoder_params(256, OE.RESNET152, generator=gen, conditioning=conditioning)
    dp = OD.init_decoder_params(256, 512, 10000, 1, generator=gen)
    images = torch.randn(B, 3, 224, 224, generator=gen)
    caps = torch.randint(4, 10000, (B, 20), generator=gen)
    caps[:, 0], caps[:, 19] = 1, 2
    model = sat.ShowAndTell(256, 512, 10000, 1, compute_dtype=dtype)
    sd = dict(ep)
    sd.update(eb)
    model.encoder.load_state_dict(sd)
    model.decoder.load_state_dict(dp)
    return model.cuda().train(), ep, eb, dp, images, caps, [20] * B


@pytest.mark.timeout(1500)
def test_cfg2_whole_train_steps_f32_and_bf16_vs_oracle():
    """BASELINE configs[1]: batch 64, 224x224, E=256 H=512 V=10000 -- three whole train.py:126-146 iterations (both step
    parities and the captured hipGraph are exercised) in the f32 parity mode AND the benchmarked bf16 mode against
    `oracle.train_step.full_step` on the same weights and batch."""
    model32, ep, eb, dp, images, caps, lengths = _cfg2_models("f32")
    model16, _, _, _, _, _, _ = _cfg2_models("bf16")
    ts32, ts16 = sat.TrainStep(model32), sat.TrainStep(model16)
    di, dc = images.cuda(), caps.cuda()
    ebo = {k: v.clone() for k, v in eb.items()}
    epo = {k: v.clone() for k, v in ep.items()}
    dpo = {k: v.clone() for k, v in dp.items()}
    state = {}
    ref, l32, l16 = [], [], []
    for step in range(3):
        rl, _ = OT.full_step(epo, ebo, dpo, images, caps, lengths, state)
        ref.append(rl.item())
        l32.append(ts32.step(di, dc, lengths).item())
        l16.append(ts16.step(di, dc, lengths).item())
    print("cfg2 CE per step: oracle %s | f32 %s | bf16 %s" % (ref, l32, l16))
    print("cfg2 |dCE| f32 %.2e, bf16 %.2e" % (max(abs(a - b) for a, b in zip(l32, ref)), max(abs(a - b) for a, b in zip(l16, ref))))
    for a, b in zip(l32, ref):
        assert abs(a - b) < 1e-4, (l32, ref)                 # north_star: CE within 1e-4 fp32
    for a, b in zip(l16, ref):
        assert abs(a - b) < 2e-3, (l16, ref)                 # bf16 conv stack: stated tolerance (module docstring)
    # parameters after three clamp+Adam steps (f32 mode; resnet.fc.bias has a mathematically zero gradient under
    # train-mode BatchNorm1d: Adam turns its rounding noise into +-lr steps in torch as here -- excluded)
    # Adam's first steps move every element by ~lr * sign(g): where a gradient element is at rounding-noise level its
    # sign -- hence a +-lr step -- is arbitrary in torch as here, so a small fraction of elements may sit 1..3 lr apart
    got = {k: v.detach().cpu() for k, v in model32.decoder.state_dict().items()}
    got["resnet.fc.weight"] = model32.encoder.resnet.fc.weight.detach().cpu()
    want = dict(dpo)
    want["resnet.fc.weight"] = epo["resnet.fc.weight"]
    for k in want:
        d = (got[k] - want[k]).abs()
        assert d.max().item() <= 3 * 2e-3 + 1e-6, (k, d.max().item())          # never more than 3 steps x 2 lr
        frac = (d > 1e-5).float().mean().item()
        print("cfg2 params after 3 steps %-22s max|d| %.2e, fraction beyond 1e-5: %.2e" % (k, d.max().item(), frac))
        # the encoder output (2e-3 conv-stack difference, amplified by the batch-of-64 BatchNorm1d) is the LSTM's step-0
        # input, so the recurrent chain's small gradient elements change sign in both W_ih and W_hh (~10 %); the vocabulary
        # side (embed, linear) sees it only through h: a fraction of a percent.
        # STRESS CHECK, NOT THE PARITY CLAIM: on this He-init stack the bound below cannot catch a wiring error on
        # the encoder side -- the parameter parity claim (<= 1e-3 of the elements beyond 1e-5, LSTM and fc included) is
        # test_cfg2_trained_like_whole_train_steps... below, on the well-conditioned stack; here only "nothing exploded"
        assert frac < (0.2 if ("lstm" in k or k == "resnet.fc.weight") else 5e-3), (k, frac)
    # head output of the two HIP modes on the SAME (now trained-for-3-steps-apart) weights is not comparable; compare
    # encoder features on the f32 model's weights instead
    sd = model32.encoder.state_dict()
    model16.encoder.load_state_dict(sd)
    p32 = model32.encoder.pooled_features(di)
    p16 = model16.encoder.pooled_features(di)
    print("cfg2 pooled bf16 vs f32 HIP: rel-L2 %.4f cos %.5f" % (_rel(p16.cpu(), p32.cpu()), _cos(p16, p32)))
    # (stress check on the chaotic He-init stack: garbage gives cos ~ 0; the tight bf16 bounds live in the trained-like tests)
    assert _rel(p16.cpu(), p32.cpu()) < 0.30 and _cos(p16, p32) > 0.95     # the chaos floor at full depth (module docstring)


@pytest.mark.timeout(900)
def test_cfg2_bf16_vs_f32_hip_train_two_steps_and_eval():
    """STRESS CASE (He-init weights: the 152-layer map is chaotic at bf16 resolution, so these bounds only say "finite, and
    pointing the same way" -- the parity bounds are the trained-like tests at the end of this file).  Full size, autotuned
    variants, atomic statistics, slab-to-acc, hipGraph replay: bf16 pooled features and head output against this library's f32
    mode (itself oracle-checked at cfg1/cfg2 above) on the same weights -- train mode (3 passes: both parities + the captured
    graph) and eval mode (fused epilogues)."""
    model32, ep, eb, dp, images, caps, lengths = _cfg2_models("f32", seed=7)
    model16, _, _, _, _, _, _ = _cfg2_models("bf16", seed=7)
    di = images.cuda()
    e32, e16 = model32.encoder.train(), model16.encoder.train()
    with torch.no_grad():
        for i in range(3):
            p32, p16 = e32.pooled_features(di), e16.pooled_features(di)
            r, c = _rel(p16.cpu(), p32.cpu()), _cos(p16, p32)
            print("train pass %d: pooled rel-L2 %.4f cos %.5f" % (i, r, c))
            assert r < 0.30 and c > 0.95, (i, r, c)           # the chaos floor at full depth (module docstring); garbage gives cos ~ 0
        f32o, f16o = e32(di), e16(di)
        r = _rel(f16o.cpu(), f32o.cpu())
        # NOT asserted: with random weights the 152-layer stack maps every image to nearly the same pooled vector (the
        # per-image variation is a few percent of it), BatchNorm1d then subtracts that common part and rescales what is
        # left -- which is of the size of the bf16 noise itself.  The head output of a random-weight model is therefore not
        # comparable across precisions (measured 1.3 = two uncorrelated vectors); the pooled features above are.
        print("train head output rel-L2 %.4f (informational)" % r)
        assert torch.isfinite(f16o).all()
        # eval: let the running statistics converge on the f32 model (momentum 0.1), copy them, compare eval passes
        for _ in range(40):
            e32.pooled_features(di)
        e16.load_state_dict(e32.state_dict())
        e32.eval()
        e16.eval()
        q32, q16 = e32.pooled_features(di), e16.pooled_features(di)
        assert torch.isfinite(q32).all() and torch.isfinite(q16).all()
        r, c = _rel(q16.cpu(), q32.cpu()), _cos(q16, q32)
        print("eval: pooled rel-L2 %.4f cos %.5f" % (r, c))
        assert r < 0.30 and c > 0.95, (r, c)


def test_encoder_sees_weights_loaded_into_the_stack_after_a_forward():
    """`encoder.resnet.load_state_dict(...)` (how the pretrained ResNet-152 of models.py:13 would be loaded) after a
    forward must rebuild the op program's kernel-layout weight copies: output == a fresh model with those weights"""
    arch, E, B = dict(layers=(1, 1, 1, 1), width=8), 32, 4
    gen = torch.Generator().manual_seed(5)
    p1, b1 = OE.init_encoder_params(E, arch, generator=gen, randomize_bn=True)
    p2, b2 = OE.init_encoder_params(E, arch, generator=gen, randomize_bn=True)
    x = torch.randn(B, 3, 64, 64, generator=gen).cuda()
    for dtype in ("f32", "bf16"):
        enc = _encoder(arch, E, p1, b1, dtype).eval()
        y1 = enc.pooled_features(x)
        stack_sd = {k[len("resnet."):]: v for k, v in {**p2, **b2}.items() if k.startswith("resnet.")}
        enc.resnet.load_state_dict(stack_sd)                   # the PARENT's hook does not fire here
        y2 = enc.pooled_features(x)
        fresh = _encoder(arch, E, p2, b2, dtype).eval()
        assert torch.equal(y2, fresh.pooled_features(x)) and not torch.equal(y1, y2)
        # in-place write through the Parameter (version counter) is seen too
        with torch.no_grad():
            enc.resnet.conv1.weight.mul_(0.5)
            fresh.resnet.conv1.weight.mul_(0.5)
        fresh.refresh_weights()
        assert torch.equal(enc.pooled_features(x), fresh.pooled_features(x))


def test_two_forwards_before_one_backward_keep_their_own_activations():
    """grad accumulation on the drop-in path: model(a); model(b); backward of both -- the head's saved `pooled` must not
    alias the program's output buffer"""
    arch, E = dict(layers=(1, 1, 1, 1), width=8), 32
    gen = torch.Generator().manual_seed(9)
    p, b = OE.init_encoder_params(E, arch, generator=gen, randomize_bn=True)
    xa, xb = torch.randn(4, 3, 64, 64, generator=gen).cuda(), torch.randn(4, 3, 64, 64, generator=gen).cuda()
    enc = _encoder(arch, E, p, b, "f32").train()
    ya = enc(xa)
    ya.sum().backward()
    ref = enc.resnet.fc.weight.grad.clone()
    enc2 = _encoder(arch, E, p, b, "f32").train()
    ya2 = enc2(xa)
    enc2(xb)                                  # second forward of the same shape before the backward
    ya2.sum().backward()
    assert torch.allclose(enc2.resnet.fc.weight.grad, ref, rtol=1e-5, atol=1e-7)


def test_out_of_range_caption_id_raises_instead_of_training_silently():
    model = sat.ShowAndTell(32, 64, 100, 1, arch=dict(layers=(1, 1, 1, 1), width=8), compute_dtype="f32").cuda().train()
    ts = sat.TrainStep(model)
    images = torch.randn(4, 3, 64, 64, device="cuda")
    caps = torch.randint(4, 100, (4, 8), device="cuda")
    ts.step(images, caps, [8] * 4)
    ts.check_ids()                            # fine
    bad = caps.clone()
    bad[2, 3] = 100                           # == V
    ts.step(images, bad, [8] * 4)
    with pytest.raises(IndexError):
        ts.check_ids()
    ts.step(images, caps, [8] * 4)            # the guard is usable again
    ts.check_ids()
    neg = caps.clone()
    neg[0, 1] = -1
    dec = model.decoder
    feats = torch.randn(4, 32, device="cuda")
    dec(feats, neg[:, :-1], [8] * 4)
    with pytest.raises(IndexError):
        dec.id_guard().poll(block=True)
    with pytest.raises(ValueError):
        ts.step(images, caps, [9] * 4)        # a length beyond the caption matrix: sat_pack_targets would read past the row
    with pytest.raises(ValueError):
        ts.step(images, caps, [8, 8, 8, 1])   # no target token


RESNET152_GEOMS_B64 = [   # (H, W, Cin, Cout, k, stride, pad): every distinct conv geometry of the [3,8,36,3] stack at 224x224
    (56, 56, 64, 64, 1, 1, 0), (56, 56, 64, 64, 3, 1, 1), (56, 56, 64, 256, 1, 1, 0), (56, 56, 256, 64, 1, 1, 0),
    (56, 56, 256, 128, 1, 1, 0), (56, 56, 128, 128, 3, 2, 1), (28, 28, 128, 512, 1, 1, 0), (56, 56, 256, 512, 1, 2, 0),
    (28, 28, 512, 128, 1, 1, 0), (28, 28, 128, 128, 3, 1, 1), (28, 28, 512, 256, 1, 1, 0), (28, 28, 256, 256, 3, 2, 1),
    (14, 14, 256, 1024, 1, 1, 0), (28, 28, 512, 1024, 1, 2, 0), (14, 14, 1024, 256, 1, 1, 0), (14, 14, 256, 256, 3, 1, 1),
    (14, 14, 1024, 512, 1, 1, 0), (14, 14, 512, 512, 3, 2, 1), (7, 7, 512, 2048, 1, 1, 0), (14, 14, 1024, 2048, 1, 2, 0),
    (7, 7, 2048, 512, 1, 1, 0), (7, 7, 512, 512, 3, 1, 1)]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("geom", RESNET152_GEOMS_B64, ids=lambda g: "x".join(str(v) for v in g))
def test_resnet152_conv_geometries_batch64_autotuned_vs_cpu(geom):
    """Each real layer shape at batch 64 with the variant the autotuner picks and the integer-atomic BatchNorm
    statistics of the training program, against fp32 CPU convolution of the same bf16 operands: output to bf16
    rounding, column sums / sums of squares to 1e-3.  (This is where a wrong variant or statistics path at the real
    geometries shows, free of the full-depth chaos.)"""
    import ctypes as C
    import torch.nn.functional as F
    L = sat._lib
    lib = L.load()
    H, W, Cin, Cout, k, stride, pad = geom
    N = 64
    g = torch.Generator().manual_seed(H * 1000 + Cin + Cout + k)
    x = torch.randn(N, H, W, Cin, generator=g).bfloat16()
    w = (torch.randn(Cout, k, k, Cin, generator=g) / (Cin * k * k) ** 0.5).bfloat16()
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    M = N * Ho * Wo
    xd, wd = x.cuda(), w.reshape(Cout, -1).contiguous().cuda()
    out = torch.full((M, Cout), float("nan"), device="cuda", dtype=torch.bfloat16)
    tiles = lib.sat_conv_tiles_m(M)
    acc = torch.zeros(2, 2, Cout, dtype=torch.int64, device="cuda")
    part = torch.zeros(tiles, 2, Cout, device="cuda")
    o = L.SatOp()
    o.kind, o.dtype = L.OP_CONV, L.SAT_BF16
    o.in0, o.w, o.out = xd.data_ptr(), wd.data_ptr(), out.data_ptr()
    o.N, o.Hin, o.Win, o.Cin, o.Hout, o.Wout, o.Cout = N, H, W, Cin, Ho, Wo, Cout
    o.KH, o.KW, o.stride, o.pad = k, k, stride, pad
    o.sN, o.sH, o.sW = H * W * Cin, W * Cin, Cin
    atomic = tiles <= 128
    if atomic:
        o.stat_acc = acc.data_ptr()
    else:
        o.stat_partial, o.tiles_m = part.data_ptr(), tiles
    ops = (L.SatOp * 1)(o)
    scratch = torch.empty(4096, device="cuda")
    L.check(lib.sat_conv_autotune(ops, 1, 3, scratch.data_ptr(), 16384, L.stream()))
    assert ops[0].variant >= 1
    L.check(lib.sat_run_ops_parity(ops, 1, 0, L.stream()))
    torch.cuda.synchronize()
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), None, stride, pad).permute(0, 2, 3, 1).reshape(M, Cout)
    got = out.float().cpu()
    assert torch.isfinite(got).all()
    err = (got - ref).abs().max().item()
    assert err < 2 ** -7 * max(1.0, ref.abs().max().item()), (err, ops[0].variant)      # one bf16 rounding of the output
    if atomic:
        s1 = acc[0, 0].cpu().double() / 2 ** 22
        s2 = acc[0, 1].cpu().double() / 2 ** 22
    else:
        s1, s2 = part[:, 0].cpu().double().sum(0), part[:, 1].cpu().double().sum(0)
    rs, rq = ref.double().sum(0), (ref.double() ** 2).sum(0)
    assert (s1 - rs).abs().max().item() < 1e-3 * M ** 0.5 + 2e-2, (s1 - rs).abs().max().item()
    assert ((s2 - rq).abs() / rq).max().item() < 1e-3


def test_default_program_fuses_bn1_where_the_patch_kernel_runs():
    """The default bf16 training program of ResNet-152: conv2 of every stride-1 bottleneck with 128..512 planes and
    rows of <= 31 pixels carries bn1 + ReLU itself (conv_pr_kernel's LDS-resident patch) -- 44 of the 50 normalise+ReLU launches are
    gone (7 + 35 + 2 in layers 2, 3, 4), six remain (layer 1: 64 planes; the stride-2 first blocks of layers 2-4)."""
    torch.manual_seed(3)
    enc = sat.EncoderCNN(64, compute_dtype="bf16").cuda().train()
    with torch.no_grad():
        enc.pooled_features(torch.randn(4, 3, 224, 224, device="cuda"))
    prog = next(v for k, v in enc._programs.items() if k[7] is None)
    ops = [prog.ops[i] for i in range(prog.n_ops)]
    fused3x3 = [o for o in ops if o.kind == L.OP_CONV and o.KH == 3 and o.stat_acc1]
    assert len(fused3x3) == 44
    assert all(o.stride == 1 and o.Win <= 31 and 128 <= o.Cout <= 512 for o in fused3x3)
    assert sum(1 for o in ops if o.kind == L.OP_BN_RELU) == 6


def test_lookahead_is_bitwise_identical_at_the_benchmarked_configuration():
    """BASELINE configs[1] itself (batch 64, 224x224, E=256 / H=512 / V=10000, bf16 stack, autotuned variants, hipGraph replay): seven
    steps with the next six batches' conv stacks running ahead on side streams as GROUPED programs (two batches per launch) give
    the same losses, parameters and BatchNorm running statistics, bit for bit, as seven strictly sequential steps."""
    def run(lookahead):
        torch.manual_seed(123)
        model = sat.ShowAndTell(256, 512, 10000, 1, compute_dtype="bf16").cuda().train()
        ts = sat.TrainStep(model, lr=1e-3, grad_clip=0.1)
        g = torch.Generator().manual_seed(77)
        nb = model.encoder.lookahead_depth + 1
        batches = [torch.randn(64, 3, 224, 224, generator=g).cuda() for _ in range(nb)]
        caps = torch.randint(4, 10000, (64, 20), generator=g)
        caps[:, 0], caps[:, -1] = 1, 2
        caps = caps.cuda()
        lengths = [20] * 64
        losses = []
        n = 7
        for i in range(n):
            nxt = [batches[j % nb] for j in range(i + 1, i + 1 + model.encoder.lookahead_depth) if j < n] if lookahead else None
            losses.append(ts.step(batches[i % nb], caps, lengths, next_images=nxt or None))
        torch.cuda.synchronize()
        rs = torch.cat([torch.cat([bn.running_mean, bn.running_var]) for bn in model.encoder.resnet.bns()])
        return torch.cat(losses).cpu(), ts.flat.params.clone().cpu(), rs.cpu()
    a, b = run(False), run(True)
    assert torch.isfinite(a[0]).all() and abs(float(a[0][0]) - 9.21) < 0.3          # ln(10000) = 9.21 at initialisation
    for x, y in zip(a, b):
        assert torch.equal(x, y)


# ------------------------------------------------------------------------------------------------------------------------
# Full depth, WELL-CONDITIONED stack (`conditioning="trained_like"`, oracle/encoder.py): the He-init stack above is
# chaotic at bf16 resolution and maps every image to nearly the same pooled vector, so the tests above cannot see the
# encoder through the head or the loss.  With the last BatchNorm gamma of every bottleneck small (as in trained nets) the
# 152-layer map is stable (bf16-storage floor 0.7 % instead of 16 %) and the pooled vectors differ per image by 15-18 %
# of their norm, so the head output, the CE and the post-Adam parameters all depend measurably on the conv stack: a
# wrong residual wiring, statistics path or layer order now fails these tests.
def _floor_f64(params, x, arch):
    """noise floor of the bf16-storage arithmetic: the same emulation with every conv accumulated in f64"""
    import torch.nn.functional as F
    conv2d = F.conv2d
    try:
        F.conv2d = lambda a, w, b=None, stride=1, padding=0: conv2d(a.double(), w.double(), None, stride, padding).float()
        return OE.resnet_forward_bf16_storage(params, x, arch)
    finally:
        F.conv2d = conv2d


@pytest.mark.timeout(900)
def test_cfg1_trained_like_stack_pooled_and_head_f32_tight_bf16_within_measured_floor():
    """BASELINE cfg-1 shape (batch 4, 224x224, all 152 layers, train-mode statistics), models.py:25-29: f32 mode against the
    f32 oracle (pooled 1e-4, head output 1e-3 relative L2), bf16 mode against the bf16-storage oracle with bounds taken
    from the floor measured in the same test (two summation orders of the oracle itself)."""
    arch, E, B = OE.RESNET152, 256, 4
    gen = torch.Generator().manual_seed(41)
    params, buffers = OE.init_encoder_params(E, arch, generator=gen, conditioning="trained_like")
    x = torch.randn(B, 3, 224, 224, generator=torch.Generator().manual_seed(42))
    bufs = {k: v.clone() for k, v in buffers.items()}
    ref32, _ = OE.resnet_forward(params, bufs, x, arch, training=True)
    head32, _ = OE.head_forward(params, {k: v.clone() for k, v in buffers.items()}, ref32, True)
    ref_bf = OE.resnet_forward_bf16_storage(params, x, arch)
    head_bf, _ = OE.head_forward(params, {k: v.clone() for k, v in buffers.items()}, ref_bf, True)
    ref_bf64 = _floor_f64(params, x, arch)
    head_bf64, _ = OE.head_forward(params, {k: v.clone() for k, v in buffers.items()}, ref_bf64, True)
    floor_p, floor_h = _rel(ref_bf64, ref_bf), _rel(head_bf64, head_bf)
    var = ((ref32 - ref32.mean(0, keepdim=True)).norm() / ref32.norm()).item()
    assert var > 0.10, var                                   # the images are told apart at the pooled features
    e32 = _encoder(arch, E, params, buffers, "f32").train()
    e16 = _encoder(arch, E, params, buffers, "bf16").train()
    with torch.no_grad():
        p32 = e32.pooled_features(x.cuda()).cpu()
        rv32 = e32.state_dict()["resnet.layer4.2.bn3.running_var"].cpu().clone()       # after exactly one training pass
        h32 = e32(x.cuda()).cpu()
        p16 = e16.pooled_features(x.cuda()).cpu()
        h16 = e16(x.cuda()).cpu()
    print("cfg1 trained-like: per-image variation %.3f | f32 pooled %.2e head %.2e | bf16 pooled %.4f (floor %.4f) head %.4f "
          "(floor %.4f) | bf16 vs f32 oracle pooled %.4f head %.4f"
          % (var, _rel(p32, ref32), _rel(h32, head32), _rel(p16, ref_bf), floor_p, _rel(h16, head_bf), floor_h,
             _rel(p16, ref32), _rel(h16, head32)))
    assert _rel(p32, ref32) < 1e-5 and _rel(h32, head32) < 1e-4          # measured 7e-7 and 6e-6
    assert _rel(p16, ref_bf) < 1.5 * floor_p + 0.003, (_rel(p16, ref_bf), floor_p)
    assert _rel(h16, head_bf) < 1.5 * floor_h + 0.02, (_rel(h16, head_bf), floor_h)
    assert _rel(p16, ref_bf) < 0.03 and _rel(h16, head_bf) < 0.25         # absolute caps (He init: 0.16 and 1.4)
    # the running statistics of the deepest BatchNorm after the f32 pass (every layer's batch statistics fed it)
    assert torch.allclose(rv32, bufs["resnet.layer4.2.bn3.running_var"], rtol=1e-4, atol=1e-6)


@pytest.mark.timeout(1800)
def test_cfg2_trained_like_whole_train_steps_are_sensitive_to_the_encoder():
    """BASELINE configs[1] (batch 64, 224x224, E=256 H=512 V=10000), three train.py:126-146 iterations on the
    well-conditioned stack.  f32 mode vs `oracle.train_step.full_step`: pooled, head output, CE (1e-4, north_star) and the
    parameters after three clamp+Adam steps (<= 5e-3 of the elements of ANY tensor beyond 1e-5, the LSTM and fc included).
    bf16 mode vs the same oracle with bf16 storage emulated in the conv stack: head output and CE."""
    model32, ep, eb, dp, images, caps, lengths = _cfg2_models("f32", conditioning="trained_like")
    model16, _, _, _, _, _, _ = _cfg2_models("bf16", conditioning="trained_like")
    ts32, ts16 = sat.TrainStep(model32), sat.TrainStep(model16)
    di, dc = images.cuda(), caps.cuda()
    o32 = dict(ep={k: v.clone() for k, v in ep.items()}, eb={k: v.clone() for k, v in eb.items()},
               dp={k: v.clone() for k, v in dp.items()}, st={})
    o16 = dict(ep={k: v.clone() for k, v in ep.items()}, eb={k: v.clone() for k, v in eb.items()},
               dp={k: v.clone() for k, v in dp.items()}, st={})
    r32, r16, l32, l16 = [], [], [], []
    for step in range(3):
        out32, out16 = {}, {}
        r32.append(OT.full_step(o32["ep"], o32["eb"], o32["dp"], images, caps, lengths, o32["st"], out=out32)[0].item())
        r16.append(OT.full_step(o16["ep"], o16["eb"], o16["dp"], images, caps, lengths, o16["st"], encoder_storage="bf16",
                                out=out16)[0].item())
        l32.append(ts32.step(di, dc, lengths).item())
        p32, f32 = ts32.last_pooled.cpu(), ts32.last_features.cpu()
        l16.append(ts16.step(di, dc, lengths).item())
        p16, f16 = ts16.last_pooled.cpu(), ts16.last_features.cpu()
        print("cfg2 trained-like step %d: f32 pooled %.2e head %.2e dCE %.2e | bf16 vs bf16-storage oracle pooled %.4f head %.4f "
              "dCE %.2e | bf16 vs f32 oracle head %.4f dCE %.2e"
              % (step, _rel(p32, out32["pooled"]), _rel(f32, out32["features"]), abs(l32[-1] - r32[-1]),
                 _rel(p16, out16["pooled"]), _rel(f16, out16["features"]), abs(l16[-1] - r16[-1]),
                 _rel(f16, out32["features"]), abs(l16[-1] - r32[-1])))
        assert _rel(p32, out32["pooled"]) < 1e-5 and _rel(f32, out32["features"]) < 2e-4
        assert abs(l32[-1] - r32[-1]) < 1e-4                                # north_star: CE within 1e-4 fp32
        # bf16: measured floor of the emulation itself 0.7 % (pooled) / 4-7 % (head, batch 4-16); 16 % / 140 % with He init.
        # Measured here: pooled 0.73 %, head 4.1 %, |dCE| 4e-6 vs the bf16-storage oracle and 6e-5 vs the f32 oracle
        assert _rel(p16, out16["pooled"]) < 0.015 and _rel(f16, out16["features"]) < 0.08
        assert abs(l16[-1] - r16[-1]) < 1e-4 and abs(l16[-1] - r32[-1]) < 5e-4
    var = ((out32["pooled"] - out32["pooled"].mean(0, keepdim=True)).norm() / out32["pooled"].norm()).item()
    assert var > 0.10, var
    got = {k: v.detach().cpu() for k, v in model32.decoder.state_dict().items()}
    got["resnet.fc.weight"] = model32.encoder.resnet.fc.weight.detach().cpu()
    got["bn.weight"], got["bn.bias"] = model32.encoder.bn.weight.detach().cpu(), model32.encoder.bn.bias.detach().cpu()
    want = dict(o32["dp"])
    for k in ("resnet.fc.weight", "bn.weight", "bn.bias"):
        want[k] = o32["ep"][k]
    for k in want:
        d = (got[k] - want[k]).abs()
        frac = (d > 1e-5).float().mean().item()
        print("cfg2 trained-like params after 3 steps %-22s max|d| %.2e, fraction beyond 1e-5: %.2e" % (k, d.max().item(), frac))
        assert d.max().item() <= 3 * 2e-3 + 1e-6, (k, d.max().item())
        assert frac < 1e-3, (k, frac)                        # measured <= 9.2e-5 (resnet.fc.weight); VERDICT r2 asked <= 5e-3
