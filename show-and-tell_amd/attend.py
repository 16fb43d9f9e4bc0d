"""Show-Attend-Tell behind the same boundary: `ShowAttendTellModel` with the reference's constructor,
`forward(images, captions, lengths)` and `sample(images, states)` (`/root/reference/model2.py:9-111`, the model
`train.py:37` constructs), every tensor op a libsat_hip.so kernel.

    encoder  : VGG16 `features[:-3]` (model2.py:15-16), frozen (model2.py:17, 87-89) -- the conv op program of the ResNet
               path (implicit-GEMM conv with the bias + ReLU riding in the bf16 conv epilogue, SAT_OP_MAXPOOL2)
    decoder  : context_encode = features @ image_att_w (sat_gemm_f32); init_lstm; per packed step: weight_hh projection,
               sat_attention_fwd (tanh / softmax / weighted mean, model2.py:73-78), sat_lstmcell_fwd (model2.py:58);
               output_layer batched over all packed rows after the loop (model2.py:80-85)
    backward : hand-written (sat_attention_bwd, LSTMCell BPTT, batched weight-gradient GEMMs) behind torch.autograd, so
               `loss.backward()` (train.py:144) works unchanged; `finetune(allow=True)` (model2.py:87-89) adds the conv-stack
               backward (f32 mode): dgrad = the forward conv kernel on flipped weights, wgrad = split-K GEMMs over the flat
               zero-bordered pixel index, ReLU mask / max-pool routing kernels.

state_dict keys equal the reference's: `encoder.{0,2,5,...}.weight/bias`, `image_att_w`, `init_hidden.*`, `init_memory.*`,
`weight_hh.*`, `weight_att`, `embedding.weight`, `lstmcell.{weight_ih,weight_hh,bias_ih,bias_hh}`, `context2out.*`,
`hidden2tout.*`, `classifier.*`.  GPU only: there is no CPU fallback.
"""
import math
import os

import torch
import torch.nn as nn

from . import _lib as L
from . import tune as T
from .models import IdGuard
from .pack import PackInfo

VGG16_FEATURES = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512]    # vgg16.features[:-3]


class _ConvB(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.cin, self.cout = cin, cout
        w = torch.empty(cout, cin, 3, 3)
        nn.init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu")          # torchvision vgg init
        self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(torch.zeros(cout))


class VggFeatures(nn.Module):
    """`nn.Sequential(*list(vgg16.features)[:-3])` as a parameter tree with the same child names ("0", "2", "5", ...)."""

    def __init__(self, cfg=VGG16_FEATURES):
        super().__init__()
        self.cfg = list(cfg)
        i, c = 0, 3
        self.conv_names = []
        for v in self.cfg:
            if v == "M":
                i += 1
            else:
                self.add_module(str(i), _ConvB(c, v))
                self.conv_names.append(str(i))
                c, i = v, i + 2
        self.out_channels = c

    def convs(self):
        return [getattr(self, n) for n in self.conv_names]


class VggProgram:
    """Device buffers + sat_op array of the frozen VGG stack for one (batch, H, W, dtype): images f32 NCHW ->
    features f32 [N, P, C] (model2.py:44-45's view + transpose is the NHWC flattening) and their mean over P."""

    def __init__(self, stack, N, H, W, dtype, device):
        self.N, self.H, self.W, self.dtype, self.stack = N, H, W, dtype, stack
        td = torch.bfloat16 if dtype == L.SAT_BF16 else torch.float32
        ch = 8 if dtype == L.SAT_BF16 else 4
        self.keep, ops = [], []

        def alloc(shape, dt=td, zero=False):
            t = (torch.zeros if zero else torch.empty)(shape, dtype=dt, device=device)
            self.keep.append(t)
            return t

        # 3-channel input: zero-bordered NHWC image with the channels padded to one 16-byte chunk per pixel
        cpad = ch
        self.img_pad = alloc((N, H + 2, W + 2, cpad), zero=True)
        o = L.SatOp()
        o.kind, o.dtype = L.OP_IMAGE_PREP, dtype
        o.out = self.img_pad.data_ptr()
        o.N, o.Hin, o.Win, o.Hout, o.Wout, o.pad, o.Cout = N, H, W, H + 2, W + 2, 1, cpad
        ops.append(o)
        x, h, w, c = self.img_pad, H, W, cpad
        first = True
        self.layers, self.run_id = [], 0            # (kind, ...) in forward order: the tapes of the backward
        self.wcopies = []                           # (conv, kernel-layout weight copy, bias copy, Cin): refresh_weights()
        ones = {}
        convs = iter(stack.convs())
        for v in stack.cfg:
            if v == "M":
                out = alloc((N, h // 2, w // 2, c))
                o = L.SatOp()
                o.kind, o.dtype = L.OP_MAXPOOL2, dtype
                o.in0, o.out = x.data_ptr(), out.data_ptr()
                o.N, o.Hin, o.Win, o.Cout = N, h, w, c
                ops.append(o)
                self.layers.append(("pool", x, out, h, w, c))
                x, h, w = out, h // 2, w // 2
                continue
            conv = next(convs)
            wt = conv.weight.detach().to(device=device, dtype=torch.float32)
            if first:                                    # pad Cin 3 -> chunk width with zero weights
                wp = torch.zeros(v, cpad, 3, 3, device=device)
                wp[:, :3] = wt
                wt = wp
            wk = wt.permute(0, 2, 3, 1).contiguous().to(td).reshape(v, -1)
            bias = conv.bias.detach().to(device=device, dtype=torch.float32).clone()     # a copy: never an alias of the live parameter
            self.keep += [wk, bias]
            self.wcopies.append((conv, wk, bias, 3 if first else c))
            out = alloc((N, h, w, v))
            o = L.SatOp()
            o.kind, o.dtype = L.OP_CONV, dtype
            o.in0, o.w, o.out = x.data_ptr(), wk.data_ptr(), out.data_ptr()
            cin = c
            o.N, o.Cin, o.Hout, o.Wout, o.Cout, o.KH, o.KW, o.stride = N, cin, h, w, v, 3, 3, 1
            if first:                                    # the border is in the image: no padding arithmetic in the kernel
                o.Hin, o.Win, o.pad = h + 2, w + 2, 0
                o.sN, o.sH, o.sW = (h + 2) * (w + 2) * cin, (w + 2) * cin, cin
            else:
                o.Hin, o.Win, o.pad = h, w, 1
                o.sN, o.sH, o.sW = h * w * cin, w * cin, cin
            if v not in ones:
                ones[v] = alloc((v,), torch.float32)
                ones[v].fill_(1.0)
            if dtype == L.SAT_BF16:                      # bias + ReLU ride in the conv epilogue (out = relu(acc*1 + bias))
                o.scale1, o.shift1, o.flags = ones[v].data_ptr(), bias.data_ptr(), 1
                ops.append(o)
            else:                                        # f32 parity mode: conv, then the elementwise affine + ReLU kernel
                raw = alloc((N, h, w, v))
                o.out = raw.data_ptr()
                ops.append(o)
                a = L.SatOp()
                a.kind, a.dtype = L.OP_BN_RELU, dtype
                a.in0, a.out, a.scale0, a.shift0 = raw.data_ptr(), out.data_ptr(), ones[v].data_ptr(), bias.data_ptr()
                a.N, a.Hout, a.Wout, a.Cout = N, h, w, v
                ops.append(a)
            self.layers.append(("conv", conv, x, out, h, w, cin, v, first))
            x, c, first = out, v, False
        self.P, self.C = h * w, c
        self.fmap = x
        self.fmean = alloc((N, c), torch.float32)
        ap = L.SatOp()
        ap.kind, ap.dtype = L.OP_AVGPOOL, dtype
        ap.in0, ap.out = x.data_ptr(), self.fmean.data_ptr()
        ap.N, ap.Hin, ap.Win, ap.Cout = N, h, w, c
        ops.append(ap)
        self.features = self.fmap.view(N, self.P, c) if dtype == L.SAT_F32 else alloc((N, self.P, c), torch.float32)
        self.ops = (L.SatOp * len(ops))(*ops)
        self.n_ops = len(ops)
        if dtype == L.SAT_BF16:
            # kernel variant per conv geometry: the committed table, the geometry-only default for anything it does not name;
            # timing only on request (tune.py: SAT_AUTOTUNE=1 / force)
            missing = T.assign(self.ops, self.n_ops)
            if missing and T.mode() in ("time", "force"):
                scratch = alloc((4096,), torch.float32)
                L.check(L.load().sat_conv_autotune(self.ops, self.n_ops, 3, scratch.data_ptr(), scratch.numel() * 4, L.stream()),
                        "sat_conv_autotune")
                torch.cuda.synchronize()
                T.save(self.ops, self.n_ops)
            elif missing:
                T.defaults(self.ops, missing)

    @torch.no_grad()
    def refresh_weights(self):
        """Re-derive the kernel-layout copies ([Cout][KH][KW][Cin], the stack's dtype) and the bias copies from the live
        parameters IN PLACE: one strided cast-copy per conv, no rebuild, no re-tune.  Fine-tuning (model2.py:87-89) calls this
        before every forward: an optimizer that updates the parameters through raw pointers (`FusedClampAdam`) or in place
        (`torch.optim.Adam`) is then always seen, and forward and backward use the same weights."""
        for conv, wk, bias, cin in self.wcopies:
            v = wk.shape[0]
            wk.view(v, 3, 3, -1)[..., :cin].copy_(conv.weight.detach().permute(0, 2, 3, 1))
            bias.copy_(conv.bias.detach())

    def run(self, images):
        L.require_gpu(images, "images")
        if images.dtype != torch.float32 or tuple(images.shape) != (self.N, 3, self.H, self.W):
            raise ValueError("images must be float32 [%d,3,%d,%d]" % (self.N, self.H, self.W))
        images = images.contiguous()
        lib = L.load()
        self.ops[0].in0 = images.data_ptr()
        L.check(lib.sat_run_ops(self.ops, self.n_ops, L.stream()), "sat_run_ops")
        self.run_id += 1
        if self.dtype == L.SAT_BF16:
            L.check(lib.sat_cast_bf16_f32(self.fmap.data_ptr(), self.features.data_ptr(), self.features.numel(), L.stream()),
                    "sat_cast_bf16_f32")
        return self.features, self.fmean


def _vgg_backward(self, d_feats, d_fmean):
    """Gradient of the conv stack (f32 NHWC): per 3x3 conv layer the zero-bordered d(pre-activation) (`sat_pad_nhwc_f32` with the
    ReLU mask), the bias gradient (`sat_colsum_f32`), nine split-K GEMMs over the flat padded pixel index for the weight
    gradient, and the forward conv kernel on flipped weights for the input gradient; `sat_maxpool2_bwd_f32` for the pools.
    Returns [dW, db] per conv in forward order (parameter layout)."""
    lib, st = L.load(), L.stream()
    N = self.N
    dev = d_feats.device
    bf = self.dtype == L.SAT_BF16
    # bf16 stack (mixed precision, f32 master weights -- the parameters themselves): the forward ran on bf16 copies of the weights
    # and stored bf16 activations.  Backward: gradients travel between layers in f32; the input gradient runs on the bf16 matrix
    # pipe (the forward conv kernel on flipped bf16 weights over the bf16-rounded zero-bordered d(pre-activation)); the weight
    # gradient stays an exact-f32 split-K GEMM over f32 casts of the stored activations, so dW is accumulated in f32 from bf16
    # activations and f32 gradients, and the optimizer updates f32 masters.

    def f32_of(t):
        if not bf:
            return t
        o = torch.empty(t.shape, dtype=torch.float32, device=dev)
        L.check(lib.sat_cast_bf16_f32(t.data_ptr(), o.data_ptr(), t.numel(), st), "sat_cast_bf16_f32")
        return o
    dY = d_feats.contiguous().clone()                     # [N, P, C] == NHWC of the last map
    if d_fmean is not None:                               # fmean = mean over positions (model2.py:68)
        L.check(lib.sat_bcast_add_f32(d_fmean.contiguous().data_ptr(), N, self.P, self.C, 1.0 / self.P, dY.data_ptr(), st), "sat_bcast_add_f32")
    grads = {}
    for layer in reversed(self.layers):
        if layer[0] == "pool":
            _, x, out, h, w, c = layer
            x = f32_of(x)
            dX = torch.empty_like(x)
            L.check(lib.sat_maxpool2_bwd_f32(x.data_ptr(), dY.data_ptr(), N, h, w, c, dX.data_ptr(), st), "sat_maxpool2_bwd_f32")
            dY = dX
            continue
        _, conv, x, out, h, w, cin, cout, first = layer
        x, out = f32_of(x), f32_of(out)
        hp, wp = h + 2, w + 2
        npix = N * hp * wp
        dZp = torch.empty(npix, cout, device=dev)
        L.check(lib.sat_pad_nhwc_f32(dY.data_ptr(), out.data_ptr(), N, h, w, cout, 1, dZp.data_ptr(), st), "sat_pad_nhwc_f32")
        db = torch.empty(cout, device=dev)
        L.check(lib.sat_colsum_f32(dZp.data_ptr(), cout, npix, cout, db.data_ptr(), st), "sat_colsum_f32")
        # zero-bordered input with a margin of one padded row (+1 pixel) at both ends: a tap is a constant flat offset
        margin = wp + 1
        Xp = torch.zeros(npix + 2 * margin, cin, device=dev)
        inner = Xp.data_ptr() + margin * cin * 4
        if first:                                         # the stem's input is the already padded image
            L.check(lib.sat_rows_copy(x.data_ptr(), cin, None, 0, npix, npix, cin, inner, cin, st), "sat_rows_copy")
        else:
            L.check(lib.sat_pad_nhwc_f32(x.data_ptr(), None, N, h, w, cin, 1, inner, st), "sat_pad_nhwc_f32")
        tiles = ((cout + 63) // 64) * ((cin + 63) // 64)
        ks = max(1, min(64, 512 // tiles, npix // 256))
        slab = cout * cin
        wsl = torch.empty(ks * slab, device=dev)
        tap_out = torch.empty(cout, cin, device=dev)
        dWk = torch.empty(cout, 9 * cin, device=dev)
        for kh in range(3):
            for kw in range(3):
                shift = (kh - 1) * wp + (kw - 1)
                L.check(lib.sat_gemm_f32_splitk(2, 1, dZp.data_ptr(), cout, inner + shift * cin * 4, cin, wsl.data_ptr(), cin, None, None,
                                                cout, cin, npix, ks, slab, st), "sat_gemm_f32_splitk")
                L.check(lib.sat_sum_slabs_f32(wsl.data_ptr(), ks, slab, slab, tap_out.data_ptr(), st), "sat_sum_slabs_f32")
                L.check(lib.sat_rows_copy(tap_out.data_ptr(), cin, None, 0, cout, cout, cin, dWk.data_ptr() + (kh * 3 + kw) * cin * 4,
                                          9 * cin, st), "sat_rows_copy")
        dW = dWk.view(cout, 3, 3, cin)[..., :conv.cin].permute(0, 3, 1, 2).contiguous()       # kernel layout -> [Cout, Cin, 3, 3]
        grads[conv] = (dW, db)
        if first:
            break
        # input gradient = conv of the zero-bordered d(pre-activation) with the flipped, transposed weights (forward kernel)
        wflip = conv.weight.detach().flip(2, 3).permute(1, 2, 3, 0).contiguous().view(cin, 9 * cout)
        dX = torch.empty(N, h, w, cin, device=dev)
        if bf and cin % 8 == 0 and cout % 8 == 0:
            # input gradient on the bf16 matrix pipe: bf16 copies of dZp and of the flipped weights, bf16 result cast back to f32
            dZb = torch.empty(npix, cout, dtype=torch.bfloat16, device=dev)
            L.check(lib.sat_cast_f32_bf16(dZp.data_ptr(), dZb.data_ptr(), dZp.numel(), st), "sat_cast_f32_bf16")
            wfb = wflip.to(torch.bfloat16)
            dXb = torch.empty(N, h, w, cin, dtype=torch.bfloat16, device=dev)
            o = L.SatOp()
            o.kind, o.dtype = L.OP_CONV, L.SAT_BF16
            o.in0, o.w, o.out = dZb.data_ptr(), wfb.data_ptr(), dXb.data_ptr()
            o.N, o.Hin, o.Win, o.Cin, o.Hout, o.Wout, o.Cout = N, hp, wp, cout, h, w, cin
            o.KH, o.KW, o.stride, o.pad = 3, 3, 1, 0
            o.sN, o.sH, o.sW = hp * wp * cout, wp * cout, cout
            ops = (L.SatOp * 1)(o)
            L.check(lib.sat_run_ops(ops, 1, st), "sat_run_ops")
            L.check(lib.sat_cast_bf16_f32(dXb.data_ptr(), dX.data_ptr(), dX.numel(), st), "sat_cast_bf16_f32")
            dY = dX
            continue
        o = L.SatOp()
        o.kind, o.dtype = L.OP_CONV, L.SAT_F32
        o.in0, o.w, o.out = dZp.data_ptr(), wflip.data_ptr(), dX.data_ptr()
        o.N, o.Hin, o.Win, o.Cin, o.Hout, o.Wout, o.Cout = N, hp, wp, cout, h, w, cin
        o.KH, o.KW, o.stride, o.pad = 3, 3, 1, 0
        o.sN, o.sH, o.sW = hp * wp * cout, wp * cout, cout
        ops = (L.SatOp * 1)(o)
        L.check(lib.sat_run_ops(ops, 1, st), "sat_run_ops")
        dY = dX
    out = []
    for conv in self.stack.convs():
        out += list(grads[conv])
    return out


VggProgram.backward = _vgg_backward


class _Lin(nn.Module):
    def __init__(self, fin, fout):
        super().__init__()
        k = 1.0 / math.sqrt(fin)
        self.weight = nn.Parameter(torch.empty(fout, fin).uniform_(-k, k))         # nn.Linear default init
        self.bias = nn.Parameter(torch.empty(fout).uniform_(-k, k))


class _Emb(nn.Module):
    def __init__(self, v, e):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(v, e).normal_(0, 1))                # nn.Embedding default init


class _Cell(nn.Module):
    def __init__(self, fin, h):
        super().__init__()
        k = 1.0 / math.sqrt(h)
        self.weight_ih = nn.Parameter(torch.empty(4 * h, fin).uniform_(-k, k))     # nn.LSTMCell default init
        self.weight_hh = nn.Parameter(torch.empty(4 * h, h).uniform_(-k, k))
        self.bias_ih = nn.Parameter(torch.empty(4 * h).uniform_(-k, k))
        self.bias_hh = nn.Parameter(torch.empty(4 * h).uniform_(-k, k))


_SKINNY_WS = {}
_SKINNY_ROWS = int(os.environ.get("SAT_SKINNY_ROWS", "128"))   # per-step GEMMs with at most this many rows take the split-K kernel


def _ptr(x):
    return x if isinstance(x, int) or x is None else x.data_ptr()


def _gemm(lib, amode, bmode, A, lda, B, ldb, Cout, ldc, M, N, K, bias=None, bias2=None):
    """C[M,N] = op(A) op(B) + bias (+ bias2).  A 64 x 64-tiled GEMM of a 64-row decode step runs on N/64 workgroups; those go to
    sat_skinny_gemm_f32, which splits K over waves and grid slices instead."""
    if amode == 0 and M <= _SKINNY_ROWS and bias2 is None and ldc == N and K % 4 == 0 and lda % 4 == 0 and (bmode == 1 or ldb % 4 == 0):
        need = lib.sat_skinny_gemm_ws_bytes(M, N, K)
        dev = torch.cuda.current_device()
        ws = _SKINNY_WS.get(dev)
        if ws is None or ws.numel() * 4 < need:
            ws = _SKINNY_WS[dev] = torch.empty(max(need // 4, 1 << 20), dtype=torch.float32, device="cuda")
        L.check(lib.sat_skinny_gemm_f32(_ptr(A), lda, _ptr(B), ldb, bmode, M, N, K, L.ptr(bias), _ptr(Cout), ldc,
                                        ws.data_ptr(), ws.numel() * 4, L.stream()), "sat_skinny_gemm_f32")
        return
    L.check(lib.sat_gemm_f32(amode, bmode, _ptr(A), lda, _ptr(B), ldb, _ptr(Cout), ldc, L.ptr(bias), L.ptr(bias2), M, N, K,
                             L.stream()), "sat_gemm_f32")


def _p(t, row0=0):
    """device pointer of row `row0` of a 2-D contiguous f32 tensor"""
    return t.data_ptr() + row0 * t.shape[1] * 4


class _AttendFn(torch.autograd.Function):
    """decoder half of model2.py:38-65 given the encoder features (the conv stack is frozen, model2.py:17)."""

    @staticmethod
    def forward(ctx, model, features, fmean, captions, pi, *params):
        lib = L.load()
        m = model
        dev = features.device
        st = L.stream()
        B, P, C = features.shape
        E, H, V, Hin = m.embed_size, m.hidden_size, m.vocab_size, m.hidden_size
        N, T = pi.N, pi.T
        f2 = features.view(B * P, C)
        ctx_enc = torch.empty(B * P, C, device=dev)
        _gemm(lib, 0, 1, f2, C, m.image_att_w, C, ctx_enc, C, B * P, C, C)                      # model2.py:46
        HSX = torch.empty(B + N, H, device=dev)            # [h_0 ; h of every packed row]: h_{t-1} of any row is one gather away
        h0, c0 = HSX[:B], torch.empty(B, H, device=dev)
        _gemm(lib, 0, 0, fmean, C, m.init_hidden.weight, C, h0, H, B, H, C, m.init_hidden.bias)  # model2.py:67-71
        _gemm(lib, 0, 0, fmean, C, m.init_memory.weight, C, c0, H, B, H, C, m.init_memory.bias)
        c = c0.clone()
        HS, PROJ = HSX[B:], torch.empty(N, C, device=dev)
        X, GATES = torch.empty(N, Hin, device=dev), torch.empty(N, 4 * H, device=dev)
        CS, ALPHA = torch.empty(N, H, device=dev), torch.empty(N, P, device=dev)
        watt = m.weight_att.view(-1)
        att_ws = torch.empty(B * P, device=dev)
        # the embedding half of every step's LSTMCell input [emb | ctx] (model2.py:55-57) in one gather
        toks = torch.empty(N, dtype=torch.int64, device=dev)
        L.check(lib.sat_pack_tokens(captions.data_ptr(), captions.stride(0), pi.prefix_dev.data_ptr(), T, N, 0, toks.data_ptr(), st),
                "sat_pack_tokens")
        L.check(lib.sat_rows_copy(L.ptr(m.embedding.weight), E, toks.data_ptr(), 1, V, N, E, L.ptr(X), Hin, st), "sat_rows_copy")
        for t, bs in enumerate(pi.batch_sizes):                                                  # model2.py:54-62
            r0 = pi.prefix[t]
            hprev = h0.data_ptr() if t == 0 else _p(HS, pi.prefix[t - 1])
            _gemm(lib, 0, 0, hprev, H, m.weight_hh.weight, H, _p(PROJ, r0), C, bs, C, H, m.weight_hh.bias)
            # the context lands in its half of the LSTMCell input row (ld = Hin)
            L.check(lib.sat_attention_fwd(L.ptr(ctx_enc), L.ptr(f2), _p(PROJ, r0), C, L.ptr(watt), bs, P, C, _p(ALPHA, r0),
                                          _p(X, r0) + E * 4, Hin, att_ws.data_ptr(), att_ws.numel() * 4, st), "sat_attention_fwd")
            L.check(lib.sat_lstmcell_fwd(_p(X, r0), hprev, L.ptr(c), L.ptr(m.lstmcell.weight_ih), L.ptr(m.lstmcell.weight_hh),
                                         L.ptr(m.lstmcell.bias_ih), L.ptr(m.lstmcell.bias_hh), bs, Hin, H, _p(HS, r0),
                                         _p(GATES, r0), _p(CS, r0), st), "sat_lstmcell_fwd")
        # output_layer over all packed rows at once (model2.py:80-85): z = [ctx | h] [W_c2o | W_h2o]^T + b1 + b2
        Zin, Wz = torch.empty(N, C + H, device=dev), torch.empty(E, C + H, device=dev)
        L.check(lib.sat_rows_copy(X.data_ptr() + E * 4, Hin, None, 0, N, N, C, L.ptr(Zin), C + H, st), "sat_rows_copy")
        L.check(lib.sat_rows_copy(L.ptr(HS), H, None, 0, N, N, H, Zin.data_ptr() + C * 4, C + H, st), "sat_rows_copy")
        L.check(lib.sat_rows_copy(L.ptr(m.context2out.weight), C, None, 0, E, E, C, L.ptr(Wz), C + H, st), "sat_rows_copy")
        L.check(lib.sat_rows_copy(L.ptr(m.hidden2tout.weight), H, None, 0, E, E, H, Wz.data_ptr() + C * 4, C + H, st), "sat_rows_copy")
        Z = torch.empty(N, E, device=dev)
        _gemm(lib, 0, 0, Zin, C + H, Wz, C + H, Z, E, N, E, C + H, m.context2out.bias, m.hidden2tout.bias)
        ldl = (V + 3) // 4 * 4
        logits = torch.zeros(N, ldl, device=dev) if V % 4 else torch.empty(N, V, device=dev)
        _gemm(lib, 0, 0, Z, E, m.classifier.weight, E, logits, ldl, N, V, E, m.classifier.bias)
        ctx.m, ctx.pi, ctx.captions = m, pi, captions
        ctx.tapes = dict(f2=f2, fmean=fmean, ctx_enc=ctx_enc, h0=h0, c0=c0, PROJ=PROJ, HS=HS, X=X, GATES=GATES, CS=CS, ALPHA=ALPHA,
                         Zin=Zin, Wz=Wz, Z=Z, toks=toks, HSX=HSX)
        return logits if ldl == V else logits[:, :V]

    @staticmethod
    def backward(ctx, dlogits):
        from .attend_bwd import attend_backward
        want = ctx.needs_input_grad[1]                     # features carry a graph only when the conv stack is fine-tuned
        grads, d_feats, d_fmean = attend_backward(ctx.m, ctx.pi, ctx.captions, ctx.tapes, dlogits, want_dfeat=want)
        return (None, d_feats, d_fmean if ctx.needs_input_grad[2] else None, None, None) + tuple(grads)


class _VggFn(torch.autograd.Function):
    """the conv stack WITH a backward (fine-tuning, model2.py:87-89 `finetune(allow=True)`): f32 parity mode, or bf16 forward /
    bf16 input-gradient convs with f32 master weights and f32 weight gradients (compute_dtype='bf16')"""

    @staticmethod
    def forward(ctx, prog, images, *params):
        feats, fmean = prog.run(images)
        ctx.prog, ctx.run_id = prog, prog.run_id
        return feats.clone(), fmean.clone()

    @staticmethod
    def backward(ctx, d_feats, d_fmean):
        prog = ctx.prog
        if prog.run_id != ctx.run_id:
            raise RuntimeError("the conv stack ran again before this backward: its activation tapes were overwritten "
                               "(fine-tuning keeps one forward per backward)")
        return (None, None) + tuple(prog.backward(d_feats, d_fmean))


PARAM_ORDER = ("image_att_w", "init_hidden.weight", "init_hidden.bias", "init_memory.weight", "init_memory.bias",
               "weight_hh.weight", "weight_hh.bias", "weight_att", "embedding.weight", "lstmcell.weight_ih", "lstmcell.weight_hh",
               "lstmcell.bias_ih", "lstmcell.bias_hh", "context2out.weight", "context2out.bias", "hidden2tout.weight",
               "hidden2tout.bias", "classifier.weight", "classifier.bias")


class ShowAttendTellModel(nn.Module):
    """model2.py:9-111.  `compute_dtype` ('bf16' conv stack / 'f32' parity) and `vgg_cfg` are build extensions."""

    def __init__(self, hidden_size, context_size, vocab_size, embed_size, opt=None, feature_size=(196, 512),
                 compute_dtype="bf16", vgg_cfg=VGG16_FEATURES):
        super().__init__()
        feat = int(feature_size[1])
        if embed_size + feat != hidden_size:
            raise ValueError("the LSTMCell input is cat[embedding, context] (model2.py:57-58): hidden_size must equal "
                             "embed_size + %d" % feat)
        if context_size != feat:
            raise ValueError("weight_hh(hidden) is added to context_encode (model2.py:74): context_size must equal %d" % feat)
        if feat % 4 or hidden_size % 4 or embed_size % 4:
            raise ValueError("feature, hidden and embed sizes must be multiples of 4")
        self.opt = opt
        self.encoder = VggFeatures(vgg_cfg)                                             # model2.py:15-16
        if self.encoder.out_channels != feat:
            raise ValueError("the conv stack ends in %d channels, feature_size says %d" % (self.encoder.out_channels, feat))
        self.compute_dtype = compute_dtype
        self.finetune(allow=False)                                                      # model2.py:17
        self.image_att_w = nn.Parameter(torch.empty(feat, feat).normal_(0, 0.05))       # model2.py:20 (uninitialised there)
        self.init_hidden, self.init_memory = _Lin(feat, hidden_size), _Lin(feat, hidden_size)
        self.weight_hh = _Lin(hidden_size, context_size)
        self.weight_att = nn.Parameter(torch.empty(feat, 1).normal_(0, 0.05))           # model2.py:25 (uninitialised there)
        self.embedding = _Emb(vocab_size, embed_size)
        self.lstmcell = _Cell(hidden_size, hidden_size)
        self.context2out, self.hidden2tout = _Lin(context_size, embed_size), _Lin(hidden_size, embed_size)
        self.classifier = _Lin(embed_size, vocab_size)
        self.hidden_size, self.embed_size, self.vocab_size, self.feat = hidden_size, embed_size, vocab_size, feat
        self.compute_dtype = compute_dtype
        self._programs, self._guard = {}, None
        self._pf_list = []          # features in flight: [(images, feats, fmean, event, weights signature, instance)]
        self.register_load_state_dict_post_hook(lambda mod, k: mod._programs.clear())
        self.encoder.register_load_state_dict_post_hook(lambda mod, k: self._programs.clear())

    def finetune(self, allow=False):
        """model2.py:87-89: (un)freeze the conv stack.  Fine-tuning runs the stack with a hand-written backward (dgrad through
        the forward conv kernel on flipped weights, wgrad as split-K GEMMs, ReLU / max-pool routing) in the f32 parity mode."""
        # compute_dtype='bf16': mixed precision -- the parameters ARE the f32 master weights; the stack runs on bf16 copies
        # refreshed from them before every forward (VggProgram.refresh_weights), gradients come back in f32 (_vgg_backward)
        for p in self.encoder.parameters():
            p.requires_grad = True if allow else False

    def _apply(self, fn, *a, **k):
        self._programs.clear()
        return super()._apply(fn, *a, **k)

    PF_DEPTH = 2      # batches whose features may be in flight (own program instance and side stream each)

    def prefetch_features(self, images):
        """Start the FROZEN conv stack (model2.py:17 `finetune(allow=False)`) of a LATER batch on a side stream, under the current
        batch's decoder forward / backward / optimizer (hundreds of small launches that leave most of the chip idle).  The
        features depend on the images and the frozen weights only, so this changes the schedule, not a bit of the result;
        `forward(images, ...)` / `sample(images)` of the SAME tensor object picks them up; other tensors are computed as usual
        and leave the batches in flight alone.  No-op while fine-tuning."""
        if images is None or any(p.requires_grad for p in self.encoder.parameters()):
            return False
        if any(e[0] is images for e in self._pf_list) or len(self._pf_list) >= self.PF_DEPTH:
            return False
        from .models import lookahead_stream
        busy = {e[5] for e in self._pf_list}
        inst = next(i for i in range(self.PF_DEPTH) if i not in busy)
        stream = lookahead_stream(images.device, inst)
        main = torch.cuda.current_stream(images.device)
        stream.wait_stream(main)
        ready = getattr(images, "_sat_ready_event", None)      # a DevicePrefetcher copy still in flight on its own stream
        if ready is not None:
            stream.wait_event(ready)
        with torch.cuda.stream(stream), torch.no_grad():
            feats, fmean = self._program_for(images, instance=inst).run(images)
            feats, fmean = feats.clone(), fmean.clone()
            ev = torch.cuda.Event()
            ev.record(stream)
        feats.record_stream(main)
        fmean.record_stream(main)
        self._pf_list.append((images, feats, fmean, ev, self._encoder_sig(), inst))
        return True

    def _encoder_sig(self):
        """Frozen stack: changes when a weight is replaced or written in place (the program's kernel-layout copies are then
        rebuilt).  While fine-tuning the weights change EVERY step, so the program is keyed on the storage only and its copies
        are refreshed in place before each forward (`VggProgram.refresh_weights`)."""
        if any(p.requires_grad for p in self.encoder.parameters()):
            return -1 - (sum(p.data_ptr() & 0xffffffff for p in self.encoder.parameters()) & ((1 << 40) - 1))
        return sum(p._version * 7 + (p.data_ptr() & 0xffffffff) for p in self.encoder.parameters())

    def _program_for(self, images, instance=None):
        """instance None: the program `forward` runs; 0, 1: independent copies (own buffers) for batches in flight"""
        N, _, H, W = images.shape
        dt = L.SAT_BF16 if self.compute_dtype == "bf16" else L.SAT_F32
        sig = self._encoder_sig()
        key = (N, H, W, dt, str(images.device), sig, instance)
        prog = self._programs.get(key)
        if prog is None:
            for k in [k for k in self._programs if k[5] != sig or len(self._programs) >= 6]:
                del self._programs[k]
            prog = self._programs[key] = VggProgram(self.encoder, N, H, W, dt, images.device)
        return prog

    def _encode(self, images):
        L.require_gpu(images, "images")
        frozen = not any(p.requires_grad for p in self.encoder.parameters())
        for k, e in enumerate(self._pf_list):
            if e[0] is images:
                del self._pf_list[k]
                torch.cuda.current_stream(images.device).wait_event(e[3])
                if frozen and e[4] == self._encoder_sig():
                    return e[1], e[2]
                break                                  # weights changed meanwhile: compute again
        prog = self._program_for(images)
        tuned = [p for p in self.encoder.parameters() if p.requires_grad]
        if tuned:
            prog.refresh_weights()
        if tuned and torch.is_grad_enabled():
            if len(tuned) != 2 * len(self.encoder.conv_names):
                raise NotImplementedError("fine-tune all of the conv stack or none of it (model2.py:87-89)")
            params = []
            for cv in self.encoder.convs():
                params += [cv.weight, cv.bias]
            feats, fmean = _VggFn.apply(prog, images, *params)
        else:
            with torch.no_grad():
                feats, fmean = prog.run(images)
            feats, fmean = feats.clone(), fmean.clone()   # the program's buffers are overwritten by the next forward
        if feats.shape[2] != self.feat:
            raise ValueError("encoder features have %d channels, expected %d" % (feats.shape[2], self.feat))
        return feats, fmean

    def _params(self):
        d = dict(self.named_parameters())
        return [d[k] for k in PARAM_ORDER]

    def forward(self, images, captions, lengths):
        """model2.py:38-65: logits f32 [sum(lengths), V], rows in time-major packed order."""
        feats, fmean = self._encode(images)
        return self.decode(feats, fmean, captions, lengths)

    def decode(self, features, fmean, captions, lengths):
        L.require_gpu(captions, "captions")
        if len(lengths) != features.shape[0]:
            raise ValueError("len(lengths) != batch size")
        pi = PackInfo.get(lengths, features.device)
        if pi.T > captions.shape[1]:
            raise ValueError("a length exceeds captions.shape[1]")
        if captions.dtype != torch.int64 or captions.stride(1) != 1:
            captions = captions.long().contiguous()
        if self._guard is None or self._guard.status.device != features.device:
            self._guard = IdGuard(features.device)
        self._guard.submit(captions, pi.T, self.vocab_size, "captions")
        return _AttendFn.apply(self, features, fmean, captions, pi, *self._params())

    @torch.no_grad()
    def sample(self, images, states=None):
        """Greedy search, 20 steps (model2.py:91-111; torch-0.1 keepdim semantics): i64 [B,20].  `states`: None = zeros (what
        eval.py:82-83 passes), a (h, c) pair of [B,H] tensors, or eval.py:89's stacked [2,B,H] tensor."""
        feats, _ = self._encode(images)
        return self.sample_features(feats, states).squeeze()      # model2.py:111: [20] at batch 1

    @torch.no_grad()
    def sample_features(self, features, states=None, steps=20, start_id=1):
        lib = L.load()
        m, dev, st = self, features.device, L.stream()
        B, P, C = features.shape
        E, H, V, Hin = m.embed_size, m.hidden_size, m.vocab_size, m.hidden_size
        f2 = features.contiguous().view(B * P, C)
        ctx_enc = torch.empty(B * P, C, device=dev)
        _gemm(lib, 0, 1, f2, C, m.image_att_w, C, ctx_enc, C, B * P, C, C)
        if states is None:
            h, c = torch.zeros(B, H, device=dev), torch.zeros(B, H, device=dev)
        else:
            if len(states) != 2 or any(tuple(s.shape) != (B, H) for s in states):
                raise ValueError("states must be (h, c), each [B=%d, H=%d] (model2.py:99), got %s"
                                 % (B, H, [tuple(s.shape) for s in states]))
            h, c = states[0].to(dev).float().contiguous().clone(), states[1].to(dev).float().contiguous().clone()
        h2 = torch.empty(B, H, device=dev)
        proj, X = torch.empty(B, C, device=dev), torch.empty(B, Hin, device=dev)
        ctxb, Zin, Z = torch.empty(B, C, device=dev), torch.empty(B, C + H, device=dev), torch.empty(B, E, device=dev)
        Wz = torch.empty(E, C + H, device=dev)
        L.check(lib.sat_rows_copy(L.ptr(m.context2out.weight), C, None, 0, E, E, C, L.ptr(Wz), C + H, st), "sat_rows_copy")
        L.check(lib.sat_rows_copy(L.ptr(m.hidden2tout.weight), H, None, 0, E, E, H, Wz.data_ptr() + C * 4, C + H, st), "sat_rows_copy")
        ids = torch.full((B, steps), int(start_id), dtype=torch.int64, device=dev)
        start = torch.full((B,), int(start_id), dtype=torch.int64, device=dev)
        wsb = lib.sat_vocab_argmax_ws_bytes(B, V)
        ws = torch.empty(wsb // 4, device=dev)
        watt = m.weight_att.view(-1)
        att_ws = torch.empty(B * P, device=dev)
        for i in range(steps):
            _gemm(lib, 0, 0, h, H, m.weight_hh.weight, H, proj, C, B, C, H, m.weight_hh.bias)
            L.check(lib.sat_attention_fwd(L.ptr(ctx_enc), L.ptr(f2), L.ptr(proj), C, L.ptr(watt), B, P, C, None, L.ptr(ctxb), C,
                                          att_ws.data_ptr(), att_ws.numel() * 4, st), "sat_attention_fwd")
            if i == 0:                                                               # model2.py:101-102
                L.check(lib.sat_rows_copy(L.ptr(m.embedding.weight), E, start.data_ptr(), 1, V, B, E, L.ptr(X), Hin, st), "sat_rows_copy")
                L.check(lib.sat_rows_copy(L.ptr(ctxb), C, None, 0, B, B, C, X.data_ptr() + E * 4, Hin, st), "sat_rows_copy")
            L.check(lib.sat_lstmcell_fwd(L.ptr(X), L.ptr(h), L.ptr(c), L.ptr(m.lstmcell.weight_ih), L.ptr(m.lstmcell.weight_hh),
                                         L.ptr(m.lstmcell.bias_ih), L.ptr(m.lstmcell.bias_hh), B, Hin, H, L.ptr(h2), None, None, st),
                    "sat_lstmcell_fwd")
            h, h2 = h2, h
            L.check(lib.sat_rows_copy(L.ptr(ctxb), C, None, 0, B, B, C, L.ptr(Zin), C + H, st), "sat_rows_copy")
            L.check(lib.sat_rows_copy(L.ptr(h), H, None, 0, B, B, H, Zin.data_ptr() + C * 4, C + H, st), "sat_rows_copy")
            _gemm(lib, 0, 0, Zin, C + H, Wz, C + H, Z, E, B, E, C + H, m.context2out.bias, m.hidden2tout.bias)
            col = ids[:, i]
            L.check(lib.sat_vocab_argmax(L.ptr(Z), L.ptr(m.classifier.weight), L.ptr(m.classifier.bias), B, E, V, col.data_ptr(),
                                         ids.stride(0), L.ptr(ws), wsb, st), "sat_vocab_argmax")
            # model2.py:107-108: the NEXT LSTM input = [embedding(predicted), THIS step's context]
            L.check(lib.sat_rows_copy(L.ptr(m.embedding.weight), E, col.data_ptr(), ids.stride(0), V, B, E, L.ptr(X), Hin, st), "sat_rows_copy")
            L.check(lib.sat_rows_copy(L.ptr(ctxb), C, None, 0, B, B, C, X.data_ptr() + E * 4, Hin, st), "sat_rows_copy")
        return ids


@torch.no_grad()
def _sample_beam_features(self, features, beam_size=5, states=None, end_id=None, steps=20, start_id=1, return_all=False):
    """Beam search over `sample`'s loop (model2.py:91-111; the reference's `sample_beam` is a stub, model2.py:113-114, so parity
    is pinned only at beam_size=1 == the greedy goldens).  Rows are (image b, hypothesis k) = b*K + k: the features and their
    attention encoding are replicated per hypothesis once (data movement), every step runs the greedy step's kernels on B*K rows,
    `sat_beam_step` keeps the best K of the K*V candidates per image, and h, c and the carried context follow their parent
    (`sat_beam_gather_rows`).  Returns ids i64 [B,steps] of the best hypothesis (return_all: ids [B,K,steps] best-first, scores)."""
    lib = L.load()
    m, dev, st = self, features.device, L.stream()
    B, P, C = features.shape
    K = int(beam_size)
    if K < 1 or K > 8:
        raise ValueError("beam_size must be in 1..8")
    E, H, V, Hin = m.embed_size, m.hidden_size, m.vocab_size, m.hidden_size
    R = B * K
    feats = features.contiguous().repeat_interleave(K, 0).contiguous()           # [R, P, C]
    f2 = feats.view(R * P, C)
    ctx_enc = torch.empty(R * P, C, device=dev)
    _gemm(lib, 0, 1, f2, C, m.image_att_w, C, ctx_enc, C, R * P, C, C)
    if states is None:
        h, c = torch.zeros(R, H, device=dev), torch.zeros(R, H, device=dev)
    else:
        h = states[0].to(dev).float().repeat_interleave(K, 0).contiguous()
        c = states[1].to(dev).float().repeat_interleave(K, 0).contiguous()
    h2, c2 = torch.empty(R, H, device=dev), torch.empty(R, H, device=dev)
    proj, X = torch.empty(R, C, device=dev), torch.empty(R, Hin, device=dev)
    ctxb, ctx2 = torch.empty(R, C, device=dev), torch.empty(R, C, device=dev)
    Zin, Z = torch.empty(R, C + H, device=dev), torch.empty(R, E, device=dev)
    Wz = torch.empty(E, C + H, device=dev)
    L.check(lib.sat_rows_copy(L.ptr(m.context2out.weight), C, None, 0, E, E, C, L.ptr(Wz), C + H, st), "sat_rows_copy")
    L.check(lib.sat_rows_copy(L.ptr(m.hidden2tout.weight), H, None, 0, E, E, H, Wz.data_ptr() + C * 4, C + H, st), "sat_rows_copy")
    ldl = (V + 3) // 4 * 4
    logits = torch.zeros(R, ldl, device=dev)
    scores = torch.full((B, K), float("-inf"), device=dev)
    scores[:, 0] = 0.0
    scores2 = torch.empty(B, K, device=dev)
    bws = torch.empty(lib.sat_beam_step_ws_bytes(B, K), dtype=torch.uint8, device=dev)
    parents = torch.empty(steps, R, dtype=torch.int32, device=dev)
    tokens = torch.empty(steps, R, dtype=torch.int64, device=dev)
    start = torch.full((R,), int(start_id), dtype=torch.int64, device=dev)
    watt = m.weight_att.view(-1)
    att_ws = torch.empty(R * P, device=dev)
    eid = -1 if end_id is None else int(end_id)
    for i in range(steps):
        _gemm(lib, 0, 0, h, H, m.weight_hh.weight, H, proj, C, R, C, H, m.weight_hh.bias)
        L.check(lib.sat_attention_fwd(L.ptr(ctx_enc), L.ptr(f2), L.ptr(proj), C, L.ptr(watt), R, P, C, None, L.ptr(ctxb), C,
                                      att_ws.data_ptr(), att_ws.numel() * 4, st), "sat_attention_fwd")
        if i == 0:                                                                   # model2.py:101-102
            L.check(lib.sat_rows_copy(L.ptr(m.embedding.weight), E, start.data_ptr(), 1, V, R, E, L.ptr(X), Hin, st), "sat_rows_copy")
            L.check(lib.sat_rows_copy(L.ptr(ctxb), C, None, 0, R, R, C, X.data_ptr() + E * 4, Hin, st), "sat_rows_copy")
        L.check(lib.sat_lstmcell_fwd(L.ptr(X), L.ptr(h), L.ptr(c), L.ptr(m.lstmcell.weight_ih), L.ptr(m.lstmcell.weight_hh),
                                     L.ptr(m.lstmcell.bias_ih), L.ptr(m.lstmcell.bias_hh), R, Hin, H, L.ptr(h2), None, None, st),
                "sat_lstmcell_fwd")
        h, h2 = h2, h
        L.check(lib.sat_rows_copy(L.ptr(ctxb), C, None, 0, R, R, C, L.ptr(Zin), C + H, st), "sat_rows_copy")
        L.check(lib.sat_rows_copy(L.ptr(h), H, None, 0, R, R, H, Zin.data_ptr() + C * 4, C + H, st), "sat_rows_copy")
        _gemm(lib, 0, 0, Zin, C + H, Wz, C + H, Z, E, R, E, C + H, m.context2out.bias, m.hidden2tout.bias)
        L.check(lib.sat_vocab_logits_fwd(L.ptr(Z), L.ptr(m.classifier.weight), L.ptr(m.classifier.bias), R, E, V, L.ptr(logits), ldl, st),
                "sat_vocab_logits_fwd")
        last = tokens[i - 1].data_ptr() if (i > 0 and eid >= 0) else None
        L.check(lib.sat_beam_step(L.ptr(logits), ldl, L.ptr(scores), last, eid, B, K, V, parents[i].data_ptr(), tokens[i].data_ptr(),
                                  L.ptr(scores2), L.ptr(bws), bws.numel(), st), "sat_beam_step")
        scores, scores2 = scores2, scores
        src_ctx = ctxb
        if K > 1:                                    # the survivors' state: h, c and THIS step's context follow their parent
            for (a, b_) in ((h, h2), (c, c2)):
                L.check(lib.sat_beam_gather_rows(L.ptr(a), parents[i].data_ptr(), B, K, H, L.ptr(b_), st), "sat_beam_gather_rows")
            h, h2, c, c2 = h2, h, c2, c
            L.check(lib.sat_beam_gather_rows(L.ptr(ctxb), parents[i].data_ptr(), B, K, C, L.ptr(ctx2), st), "sat_beam_gather_rows")
            src_ctx = ctx2
        # model2.py:107-108: the NEXT LSTM input = [embedding(token), THIS step's context]
        L.check(lib.sat_rows_copy(L.ptr(m.embedding.weight), E, tokens[i].data_ptr(), 1, V, R, E, L.ptr(X), Hin, st), "sat_rows_copy")
        L.check(lib.sat_rows_copy(L.ptr(src_ctx), C, None, 0, R, R, C, X.data_ptr() + E * 4, Hin, st), "sat_rows_copy")
    ids = torch.empty(B, K, steps, dtype=torch.int64, device=dev)
    L.check(lib.sat_beam_backtrack(L.ptr(parents), L.ptr(tokens), steps, B, K, L.ptr(ids), st), "sat_beam_backtrack")
    if return_all:
        return ids, scores
    return ids[:, 0].contiguous()


@torch.no_grad()
def _sample_beam(self, images, beam_size=5, states=None, end_id=None, return_all=False):
    """`sample_beam(images, ...)`: the method the reference leaves as a stub (model2.py:113-114); BASELINE configs[4] asks beam 5."""
    feats, _ = self._encode(images)
    return self.sample_beam_features(feats, beam_size, states, end_id, return_all=return_all)


ShowAttendTellModel.sample_beam_features = _sample_beam_features
ShowAttendTellModel.sample_beam = _sample_beam
