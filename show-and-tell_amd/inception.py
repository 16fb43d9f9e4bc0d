"""Inception-v3 conv stack (BASELINE.json configs[3]: "Inception-v3 encoder (299x299)") as an op program for libsat_hip.so,
pluggable into `EncoderCNN`'s slot (`/root/reference/models.py:9-29`; the reference itself only has ResNet-152 / VGG16, so
this is a build extension: SURVEY 8f.3).  Parameter names are torchvision's (`Conv2d_1a_3x3.conv.weight`, `Mixed_5b.branch1x1.bn.*`,
..., `fc.*`), so an `inception_v3` state_dict loads as is (its AuxLogits head is not part of the feature path).

Every BasicConv2d = implicit-GEMM conv (asymmetric 1x7 / 7x1 / 1x3 / 3x1 kernels via per-axis padding) with the BatchNorm
statistics in its epilogue + one normalise/ReLU launch that writes straight into the channel slice of the block's
concatenated output (sat_op.ldc); 3x3 max / average pools are their own small kernels.  Train-mode batch statistics
(nothing calls .eval() in train.py), integer-atomic statistics in bf16 as on the ResNet path; eval uses the running ones."""
import os

import torch
import torch.nn as nn

from . import _lib as L
from .resnet import _BN, _Conv, ConvStackProgram

BN_EPS = 1e-3
BN_MOMENTUM = 0.1

BLOCKS = [("Mixed_5b", "A", 192, 32, 256), ("Mixed_5c", "A", 256, 64, 288), ("Mixed_5d", "A", 288, 64, 288),
          ("Mixed_6a", "B", 288, None, 768), ("Mixed_6b", "C", 768, 128, 768), ("Mixed_6c", "C", 768, 160, 768),
          ("Mixed_6d", "C", 768, 160, 768), ("Mixed_6e", "C", 768, 192, 768), ("Mixed_7a", "D", 768, None, 1280),
          ("Mixed_7b", "E", 1280, None, 2048), ("Mixed_7c", "E", 2048, None, 2048)]


class _ConvI(_Conv):
    """conv weight holder with a rectangular kernel / per-axis padding (frozen, like the ResNet stack: models.py:14-15)"""

    def __init__(self, cin, cout, k, stride=1, pad=0):
        nn.Module.__init__(self)
        self.kh, self.kw = (k, k) if isinstance(k, int) else k
        self.ph, self.pw = (pad, pad) if isinstance(pad, int) else pad
        self.cin, self.cout, self.stride = cin, cout, stride
        w = torch.empty(cout, cin, self.kh, self.kw).normal_(0, (2.0 / (cin * self.kh * self.kw)) ** 0.5)
        self.weight = nn.Parameter(w, requires_grad=False)


class BasicConv2d(nn.Module):
    def __init__(self, cin, cout, k, stride=1, pad=0):
        super().__init__()
        self.conv, self.bn = _ConvI(cin, cout, k, stride, pad), _BN(cout)


def _block(kind, cin, arg):
    m = nn.Module()
    B = BasicConv2d
    if kind == "A":
        m.branch1x1, m.branch5x5_1, m.branch5x5_2 = B(cin, 64, 1), B(cin, 48, 1), B(48, 64, 5, 1, 2)
        m.branch3x3dbl_1, m.branch3x3dbl_2, m.branch3x3dbl_3 = B(cin, 64, 1), B(64, 96, 3, 1, 1), B(96, 96, 3, 1, 1)
        m.branch_pool = B(cin, arg, 1)
    elif kind == "B":
        m.branch3x3 = B(cin, 384, 3, 2)
        m.branch3x3dbl_1, m.branch3x3dbl_2, m.branch3x3dbl_3 = B(cin, 64, 1), B(64, 96, 3, 1, 1), B(96, 96, 3, 2)
    elif kind == "C":
        c7 = arg
        m.branch1x1 = B(cin, 192, 1)
        m.branch7x7_1, m.branch7x7_2, m.branch7x7_3 = B(cin, c7, 1), B(c7, c7, (1, 7), 1, (0, 3)), B(c7, 192, (7, 1), 1, (3, 0))
        m.branch7x7dbl_1, m.branch7x7dbl_2 = B(cin, c7, 1), B(c7, c7, (7, 1), 1, (3, 0))
        m.branch7x7dbl_3, m.branch7x7dbl_4 = B(c7, c7, (1, 7), 1, (0, 3)), B(c7, c7, (7, 1), 1, (3, 0))
        m.branch7x7dbl_5 = B(c7, 192, (1, 7), 1, (0, 3))
        m.branch_pool = B(cin, 192, 1)
    elif kind == "D":
        m.branch3x3_1, m.branch3x3_2 = B(cin, 192, 1), B(192, 320, 3, 2)
        m.branch7x7x3_1, m.branch7x7x3_2 = B(cin, 192, 1), B(192, 192, (1, 7), 1, (0, 3))
        m.branch7x7x3_3, m.branch7x7x3_4 = B(192, 192, (7, 1), 1, (3, 0)), B(192, 192, 3, 2)
    else:
        m.branch1x1 = B(cin, 320, 1)
        m.branch3x3_1, m.branch3x3_2a, m.branch3x3_2b = B(cin, 384, 1), B(384, 384, (1, 3), 1, (0, 1)), B(384, 384, (3, 1), 1, (1, 0))
        m.branch3x3dbl_1, m.branch3x3dbl_2 = B(cin, 448, 1), B(448, 384, 3, 1, 1)
        m.branch3x3dbl_3a, m.branch3x3dbl_3b = B(384, 384, (1, 3), 1, (0, 1)), B(384, 384, (3, 1), 1, (1, 0))
        m.branch_pool = B(cin, 192, 1)
    return m


class _FC(nn.Module):
    def __init__(self, fin, fout):
        super().__init__()
        self.in_features, self.out_features = fin, fout
        self.weight = nn.Parameter(torch.empty(fout, fin).normal_(0.0, 0.02))   # models.py:22
        self.bias = nn.Parameter(torch.zeros(fout))                             # models.py:23


class InceptionStack(nn.Module):
    """torchvision `inception_v3` parameter tree (feature path) + the trainable `fc` the reference puts on its encoder."""
    feature_dim = 2048

    def __init__(self, embed_size):
        super().__init__()
        B = BasicConv2d
        self.Conv2d_1a_3x3, self.Conv2d_2a_3x3, self.Conv2d_2b_3x3 = B(3, 32, 3, 2), B(32, 32, 3), B(32, 64, 3, 1, 1)
        self.Conv2d_3b_1x1, self.Conv2d_4a_3x3 = B(64, 80, 1), B(80, 192, 3)
        for name, kind, cin, arg, _ in BLOCKS:
            setattr(self, name, _block(kind, cin, arg))
        self.fc = _FC(self.feature_dim, embed_size)
        self.arch = "inception_v3"

    @property
    def conv1(self):           # EncoderCNN keys its program cache on the first conv
        return self.Conv2d_1a_3x3.conv

    def bns(self):
        return [m.bn for m in self.modules() if isinstance(m, BasicConv2d)]

    def program(self, N, H, W, dtype, training, device, groups=1, signatures=None):
        return InceptionProgram(self, N, H, W, dtype, training, device, groups=groups, signatures=signatures)


class InceptionProgram(ConvStackProgram):
    """op program of the Inception-v3 stack for one (batch, H, W, dtype, training); `run` / `run_timed` / hipGraph replay are
    the ResNet program's."""

    def __init__(self, stack, N, H, W, dtype, training, device, groups=1, signatures=None):
        """groups = G > 1 (bf16): G look-ahead batches per launch, as in the ResNet program -- train mode: `sat_op.groups` = G, every
        per-batch buffer G consecutive copies, per-group BatchNorm statistics, deferred running statistics; eval mode: the batches
        concatenate into one program over G * N images."""
        self.N, self.H, self.W, self.dtype, self.training, self.stack = N, H, W, dtype, training, stack
        self.groups = int(groups)
        self._n_prep = self.groups                         # ops[0 .. groups) are the image preps (ConvStackProgram.run)
        if self.groups > 1 and dtype != L.SAT_BF16:
            raise ValueError("grouped programs are for the bf16 stack")
        Nb = N                                             # images per batch
        if self.groups > 1 and not training:
            N, G = self.groups * N, 1
        else:
            G = self.groups
        GN = G * N                                         # images every per-image op (pools, image prep, the final pool) sees
        self.keep, self.bn_list, self.stat_accs, self._want_sigs = [], [], [], dict(signatures or {})
        td = torch.bfloat16 if dtype == L.SAT_BF16 else torch.float32
        esz = 2 if dtype == L.SAT_BF16 else 4
        ch = 16 // esz
        lib = L.load()
        ops = []

        def alloc(shape, dt=td, zero=False):
            t = (torch.zeros if zero else torch.empty)(shape, dtype=dt, device=device)
            self.keep.append(t)
            return t

        bns = stack.bns()
        flat = torch.stack([bn.num_batches_tracked.detach().to(device) for bn in bns])
        for i, bn in enumerate(bns):
            bn.num_batches_tracked = flat[i]
        object.__setattr__(stack, "_nbt_flat", flat)
        atomic = training and dtype == L.SAT_BF16
        max_part = [0]
        self.partial = None
        part_users = []

        def basic(m, x, h, w, out=None, out_off=0, out_ld=0):
            """BasicConv2d: conv (+ statistics) -> normalise + ReLU into `out` (a dense tensor or a channel slice).
            x: dense NHWC tensor [N,h,w,cin].  Returns (activation tensor or None when sliced, ho, wo)."""
            cv, bn = m.conv, m.bn
            cin = x.shape[3]
            ho, wo = (h + 2 * cv.ph - cv.kh) // cv.stride + 1, (w + 2 * cv.pw - cv.kw) // cv.stride + 1
            wt = cv.weight.detach().to(device=device, dtype=torch.float32)
            if wt.shape[1] != cin:                       # the stem's 3 input channels padded to one 16-byte chunk
                wp = torch.zeros(cv.cout, cin, cv.kh, cv.kw, device=device)
                wp[:, :wt.shape[1]] = wt
                wt = wp
            wk = wt.permute(0, 2, 3, 1).contiguous().to(td).reshape(cv.cout, -1)
            raw = alloc((GN, ho, wo, cv.cout))
            self.keep.append(wk)
            o = L.SatOp()
            o.kind, o.dtype, o.groups = L.OP_CONV, dtype, G
            o.in0, o.w, o.out = x.data_ptr(), wk.data_ptr(), raw.data_ptr()
            o.N, o.Hin, o.Win, o.Cin, o.Hout, o.Wout, o.Cout = N, h, w, cin, ho, wo, cv.cout
            o.KH, o.KW, o.stride, o.pad = cv.kh, cv.kw, cv.stride, cv.ph
            if cv.ph != cv.pw:
                o.flags, o.pad_w = L.CONV_PADW, cv.pw
            o.sN, o.sH, o.sW = h * w * cin, w * cin, cin
            M = N * ho * wo
            tiles = lib.sat_conv_tiles_m(M)
            a = L.SatOp()
            a.kind, a.dtype, a.groups = L.OP_BN_RELU, dtype, G
            a.in0 = raw.data_ptr()
            a.N, a.Hout, a.Wout, a.Cout = N, ho, wo, cv.cout
            if out is None:
                act = alloc((GN, ho, wo, cv.cout))
                a.out = act.data_ptr()
            else:
                act = None
                a.out, a.ldc = out.data_ptr() + out_off * esz, out_ld
            self.bn_list.append(bn)
            if not training:                             # eval: running statistics -> (scale, shift) table
                s, t = alloc((cv.cout,), torch.float32), alloc((cv.cout,), torch.float32)
                f = L.SatOp()
                f.kind, f.dtype = L.OP_BN_FINALIZE, dtype
                f.gamma, f.beta = bn.weight.data_ptr(), bn.bias.data_ptr()
                f.running_mean, f.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
                f.scale_out, f.shift_out = s.data_ptr(), t.data_ptr()
                f.Cout, f.count, f.tiles_m, f.training, f.momentum, f.eps = cv.cout, M, 0, 0, BN_MOMENTUM, BN_EPS
                a.scale0, a.shift0 = s.data_ptr(), t.data_ptr()
                ops.extend([o, f, a])
            elif atomic:
                acc = alloc((G, 2, 2, cv.cout), torch.int64, zero=True)      # [G][2 parities][2][C]
                self.stat_accs.append(acc)
                if tiles <= 128:
                    o.stat_acc = acc.data_ptr()
                    ops.append(o)
                else:                                    # many row tiles: per-tile slabs + the wide reducer into the same sums
                    max_part[0] = max(max_part[0], G * tiles * 2 * cv.cout)
                    o.tiles_m = tiles
                    f = L.SatOp()
                    f.kind, f.dtype, f.groups = L.OP_BN_FINALIZE, dtype, G
                    f.stat_acc = acc.data_ptr()
                    f.Cout, f.tiles_m, f.training = cv.cout, tiles, 1
                    part_users.extend([o, f])
                    ops.extend([o, f])
                a.stat_acc = acc.data_ptr()
                a.gamma, a.beta = bn.weight.data_ptr(), bn.bias.data_ptr()
                a.running_mean, a.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
                a.count, a.momentum, a.eps = M, BN_MOMENTUM, BN_EPS
                ops.append(a)
            else:                                        # f32 training: slabs -> finalize (f64) -> table
                max_part[0] = max(max_part[0], tiles * 2 * cv.cout)
                o.tiles_m = tiles
                s, t = alloc((cv.cout,), torch.float32), alloc((cv.cout,), torch.float32)
                f = L.SatOp()
                f.kind, f.dtype = L.OP_BN_FINALIZE, dtype
                f.gamma, f.beta = bn.weight.data_ptr(), bn.bias.data_ptr()
                f.running_mean, f.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
                f.scale_out, f.shift_out = s.data_ptr(), t.data_ptr()
                f.Cout, f.count, f.tiles_m, f.training, f.momentum, f.eps = cv.cout, M, tiles, 1, BN_MOMENTUM, BN_EPS
                part_users.extend([o, f])
                a.scale0, a.shift0 = s.data_ptr(), t.data_ptr()
                ops.extend([o, f, a])
            return act, ho, wo

        def pool(kind, x, h, w, out=None, out_off=0, out_ld=0):
            c = x.shape[3]
            ho, wo = (h, w) if kind == L.OP_AVGPOOL3 else ((h - 3) // 2 + 1, (w - 3) // 2 + 1)
            o = L.SatOp()
            o.kind, o.dtype = kind, dtype
            o.in0 = x.data_ptr()
            o.N, o.Hin, o.Win, o.Cout, o.Hout, o.Wout = GN, h, w, c, ho, wo        # per image: the groups concatenate
            if out is None:
                res = alloc((GN, ho, wo, c))
                o.out = res.data_ptr()
            else:
                res = None
                o.out, o.ldc = out.data_ptr() + out_off * esz, out_ld
            ops.append(o)
            return res, ho, wo

        # ---- program ----
        cpad = ch
        self.img = alloc((GN, H, W, cpad), zero=True)
        for g_ in range(self.groups):                    # one image prep per batch, each into its slice
            o = L.SatOp()
            o.kind, o.dtype = L.OP_IMAGE_PREP, dtype
            o.out = self.img[g_ * Nb:].data_ptr()
            o.N, o.Hin, o.Win, o.Hout, o.Wout, o.pad, o.Cout = Nb, H, W, H, W, 0, cpad
            ops.append(o)
        x, h, w = self.img, H, W
        for name in ("Conv2d_1a_3x3", "Conv2d_2a_3x3", "Conv2d_2b_3x3"):
            x, h, w = basic(getattr(stack, name), x, h, w)
        x, h, w = pool(L.OP_MAXPOOL3S2, x, h, w)
        for name in ("Conv2d_3b_1x1", "Conv2d_4a_3x3"):
            x, h, w = basic(getattr(stack, name), x, h, w)
        x, h, w = pool(L.OP_MAXPOOL3S2, x, h, w)
        for name, kind, cin, arg, cout in BLOCKS:
            m = getattr(stack, name)
            if kind in ("B", "D"):
                ho, wo = (h - 3) // 2 + 1, (w - 3) // 2 + 1
            else:
                ho, wo = h, w
            y = alloc((GN, ho, wo, cout))
            off = 0

            def chain(names, src, hh, ww, width):
                nonlocal off
                t = src
                for n_ in names[:-1]:
                    t, hh, ww = basic(getattr(m, n_), t, hh, ww)
                basic(getattr(m, names[-1]), t, hh, ww, y, off, cout)
                off += width

            if kind == "A":
                chain(["branch1x1"], x, h, w, 64)
                chain(["branch5x5_1", "branch5x5_2"], x, h, w, 64)
                chain(["branch3x3dbl_1", "branch3x3dbl_2", "branch3x3dbl_3"], x, h, w, 96)
                ap, _, _ = pool(L.OP_AVGPOOL3, x, h, w)
                chain(["branch_pool"], ap, h, w, arg)
            elif kind == "B":
                chain(["branch3x3"], x, h, w, 384)
                chain(["branch3x3dbl_1", "branch3x3dbl_2", "branch3x3dbl_3"], x, h, w, 96)
                pool(L.OP_MAXPOOL3S2, x, h, w, y, off, cout)
                off += cin
            elif kind == "C":
                chain(["branch1x1"], x, h, w, 192)
                chain(["branch7x7_1", "branch7x7_2", "branch7x7_3"], x, h, w, 192)
                chain(["branch7x7dbl_%d" % i for i in range(1, 6)], x, h, w, 192)
                ap, _, _ = pool(L.OP_AVGPOOL3, x, h, w)
                chain(["branch_pool"], ap, h, w, 192)
            elif kind == "D":
                chain(["branch3x3_1", "branch3x3_2"], x, h, w, 320)
                chain(["branch7x7x3_%d" % i for i in range(1, 5)], x, h, w, 192)
                pool(L.OP_MAXPOOL3S2, x, h, w, y, off, cout)
                off += cin
            else:
                chain(["branch1x1"], x, h, w, 320)
                t, _, _ = basic(m.branch3x3_1, x, h, w)
                basic(m.branch3x3_2a, t, h, w, y, off, cout)
                basic(m.branch3x3_2b, t, h, w, y, off + 384, cout)
                off += 768
                t, _, _ = basic(m.branch3x3dbl_1, x, h, w)
                t, _, _ = basic(m.branch3x3dbl_2, t, h, w)
                basic(m.branch3x3dbl_3a, t, h, w, y, off, cout)
                basic(m.branch3x3dbl_3b, t, h, w, y, off + 384, cout)
                off += 768
                ap, _, _ = pool(L.OP_AVGPOOL3, x, h, w)
                chain(["branch_pool"], ap, h, w, 192)
            assert off == cout, (name, off, cout)
            x, h, w = y, ho, wo
        self.pooled = alloc((GN, stack.feature_dim), torch.float32)
        apo = L.SatOp()
        apo.kind, apo.dtype = L.OP_AVGPOOL, dtype
        apo.in0, apo.out = x.data_ptr(), self.pooled.data_ptr()
        apo.N, apo.Hin, apo.Win, apo.Cout = GN, h, w, stack.feature_dim
        ops.append(apo)
        self.final_map = (x, GN, h, w, stack.feature_dim)
        if max_part[0]:
            self.partial = alloc((max_part[0],), torch.float32)
            for o_ in part_users:
                o_.stat_partial = self.partial.data_ptr()
        self.ops = (L.SatOp * len(ops))(*ops)
        self.n_ops = len(ops)
        self._parity = 0
        self._use_graph = os.environ.get("SAT_GRAPH", "1") != "0" and torch.device(device).type == "cuda"
        self._runs, self._graphs = [0, 0], [None, None]
        self._running_items = None
        if self.groups > 1 and training:
            self.defer_running_stats()
        # per-geometry kernel selection, as on the ResNet path: the tuner's three fastest variants per geometry, the final choice by
        # timing whole-program passes (ConvStackProgram._autotune; no activation buffer needs re-randomising: the tuner only times)
        self._autotune(device, (), alloc)
