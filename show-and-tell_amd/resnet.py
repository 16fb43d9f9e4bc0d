"""The frozen ResNet conv stack of `EncoderCNN` (`/root/reference/models.py:13-15,27`) as an op program for
libsat_hip.so.

Host side only: this module owns the parameters (torchvision key names, so a `resnet152` state_dict loads
as is), lays the weights out for the kernels (OHWI, optional bf16 shadow), carves the activation
workspace and emits the `sat_op` array that `sat_run_ops` launches in ONE call per forward.  All arithmetic
is in the HIP kernels (csrc/sat_gemm.hip implicit-GEMM conv, csrc/sat_elementwise.hip batch-norm / pooling).

Layout: activations NHWC (channels contiguous -> the implicit-GEMM K axis is contiguous), dtype bf16
(throughput) or f32 (parity).  Per bottleneck (torchvision v1.5: stride on the 3x3):
    c1 = conv1x1(y)      stats -> (s1,t1)    a1 = relu(c1*s1+t1)
    c2 = conv3x3(a1)     stats -> (s2,t2)    a2 = relu(c2*s2+t2)
    c3 = conv1x1(a2)     stats -> (s3,t3)    [cd = conv1x1(y), stats -> (sd,td)]
    y' = relu(c3*s3+t3 + (cd*sd+td | y))
Training: batch statistics come out of the conv epilogue, as per-tile column sums reduced in fixed order by a finalize
launch, or (bf16, few M-tiles) as fixed-point integer atomics the consuming kernel turns into (scale, shift) itself;
conv3 applies bn2+ReLU to its operand in LDS.  Inference (bf16): every BatchNorm is a per-channel affine folded into
the producing conv's epilogue with the residual add and the ReLU.  The whole program replays as one hipGraph.
"""
import ctypes as C
import json
import os

import torch
import torch.nn as nn

from . import _lib as L

RESNET152 = dict(layers=(3, 8, 36, 3), width=64)
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


class _Conv(nn.Module):
    def __init__(self, cin, cout, k, stride, pad):
        super().__init__()
        self.cin, self.cout, self.k, self.stride, self.pad = cin, cout, k, stride, pad
        w = torch.empty(cout, cin, k, k)
        nn.init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu")     # torchvision resnet init
        self.weight = nn.Parameter(w, requires_grad=False)                  # models.py:14-15 (frozen)


class _BN(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(c), requires_grad=False)
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.zeros((), dtype=torch.long))


class _Bottleneck(nn.Module):
    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1, self.bn1 = _Conv(inplanes, planes, 1, 1, 0), _BN(planes)
        self.conv2, self.bn2 = _Conv(planes, planes, 3, stride, 1), _BN(planes)
        self.conv3, self.bn3 = _Conv(planes, planes * 4, 1, 1, 0), _BN(planes * 4)
        if downsample:
            self.downsample = nn.ModuleList([_Conv(inplanes, planes * 4, 1, stride, 0), _BN(planes * 4)])
        else:
            self.downsample = None


class _FC(nn.Module):
    def __init__(self, fin, fout):
        super().__init__()
        self.in_features, self.out_features = fin, fout
        self.weight = nn.Parameter(torch.empty(fout, fin).normal_(0.0, 0.02))   # models.py:22
        self.bias = nn.Parameter(torch.zeros(fout))                             # models.py:23


class ResNetStack(nn.Module):
    """Parameter tree with torchvision's names: conv1, bn1, layer{1..4}.{i}.{conv,bn}{1,2,3}, downsample.{0,1}, fc."""

    def __init__(self, embed_size, arch=RESNET152):
        super().__init__()
        self.arch = dict(arch)
        w = arch["width"]
        self.conv1, self.bn1 = _Conv(3, w, 7, 2, 3), _BN(w)
        inplanes = w
        for li, nblocks in enumerate(arch["layers"]):
            planes = w * (2 ** li)
            blocks = []
            for b in range(nblocks):
                stride = 2 if (li > 0 and b == 0) else 1
                ds = b == 0 and (stride != 1 or inplanes != planes * 4)
                blocks.append(_Bottleneck(inplanes, planes, stride, ds))
                inplanes = planes * 4
            setattr(self, "layer%d" % (li + 1), nn.ModuleList(blocks))
        self.feature_dim = inplanes
        self.fc = _FC(inplanes, embed_size)

    def blocks(self):
        for li in range(len(self.arch["layers"])):
            for blk in getattr(self, "layer%d" % (li + 1)):
                yield blk

    def bns(self):
        yield self.bn1
        for blk in self.blocks():
            yield blk.bn1
            yield blk.bn2
            yield blk.bn3
            if blk.downsample is not None:
                yield blk.downsample[1]


def weights_signature(stack):
    """Changes whenever a frozen conv weight is replaced or written in place through the Parameter (load_state_dict,
    copy_): the op program keeps permuted (bf16) copies of the conv weights and must be rebuilt then.  Writes through
    `.data` bypass the version counter -- call `EncoderCNN.refresh_weights()` after those."""
    # the walk over the module tree is cached (155 convs on ResNet-152, 3-4 signatures per step): `_sig_params` is dropped
    # by EncoderCNN._invalidate (load_state_dict, device / dtype moves, refresh_weights)
    ws = stack.__dict__.get("_sig_params")
    if ws is None:
        ws = [m.weight for m in stack.modules() if isinstance(m, _Conv)]
        stack.__dict__["_sig_params"] = ws
    sig = 0
    for w in ws:
        sig = (sig * 1000003 + w._version * 7 + (w.data_ptr() & 0xffffffff)) & ((1 << 61) - 1)
    return sig


# SAT_OP_CONV3_FUSED (conv3 + bn3 + residual add + ReLU in one launch, accumulators kept across a grid-wide statistics barrier),
# OPT-IN (SAT_FUSED_CONV3=1) where the device can hold the whole grid (layer 3 of ResNet-152 at batch 64: 196 workgroups on 256
# CUs).  Measured (profiles/r03_fused_conv3_ab.txt): the launch takes 28-31 us + a 4.8 us token-acquire launch against 24 + 16 us
# for conv3 + normalise+add -- the strictly sequential step gains 0.12 ms (6.30 vs 6.42) -- but with three stacks in flight the
# step LOSES 0.6 ms (5.41 vs 4.73 ms): the launch must own 196 CUs for its whole span (its workgroups idle at the barrier) and the
# residency token serialises the three stacks' fused launches, where conv3 and the normalise+add pass of different stacks used to
# overlap freely.  A barrier timeout reported by a run turns it off for the rest of the process.
_FUSED3 = {"enabled": os.environ.get("SAT_FUSED_CONV3", "0") == "1"}


def fused_conv3_enabled():
    return _FUSED3["enabled"]


def disable_fused_conv3():
    _FUSED3["enabled"] = False


def _tdtype(dtype):
    return torch.bfloat16 if dtype == L.SAT_BF16 else torch.float32


class ConvStackProgram:
    """Device buffers + sat_op array for one (batch, H, W, dtype, training) configuration."""
    n_fused3 = 0            # SAT_OP_CONV3_FUSED launches in the program (their status word is read back after every run)

    def __init__(self, stack, N, H, W, dtype, training, device):
        self.N, self.H, self.W, self.dtype, self.training = N, H, W, dtype, training
        self.keep = []      # tensors the op array points into
        td = _tdtype(dtype)
        ch = 8 if dtype == L.SAT_BF16 else 4
        width = stack.arch["width"]
        if width % ch:
            raise ValueError("conv stack width must be a multiple of %d for this dtype" % ch)
        ops = []

        def alloc(shape, dt=td, zero=False):
            t = (torch.zeros if zero else torch.empty)(shape, dtype=dt, device=device)
            self.keep.append(t)
            return t

        # ---- weights in kernel layout ([Cout][KH][KW][Cin]) ----
        def prep_w(conv):
            w = conv.weight.detach().to(device=device, dtype=torch.float32)
            return w.permute(0, 2, 3, 1).contiguous().to(td)

        # stem: 7x7/2 on a zero-bordered NHWC4 image; per kh one contiguous run of 8 pixels x 4 channels
        Hp, Wp = H + 6, (W + 8 + 1) // 2 * 2
        Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        w1 = stack.conv1.weight.detach().to(device=device, dtype=torch.float32)     # [w,3,7,7]
        wst = torch.zeros(width, 7, 8, 4, device=device, dtype=torch.float32)
        wst[:, :, :7, :3] = w1.permute(0, 2, 3, 1)
        wst = wst.reshape(width, 7 * 32).contiguous().to(td)
        self.keep.append(wst)
        self.images = None
        self.img_pad = alloc((N, Hp, Wp, 4), zero=True)
        self.bn_list = []
        self.stack = stack
        bns = list(stack.bns())
        nbn = len(bns)
        # one flat int64 counter tensor behind every bn.num_batches_tracked: one increment per forward
        flat = getattr(stack, "_nbt_flat", None)
        if flat is None or flat.device != torch.device(device) or \
                any(bn.num_batches_tracked.data_ptr() != flat[i].data_ptr() for i, bn in enumerate(bns)):
            flat = torch.stack([bn.num_batches_tracked.detach().to(device) for bn in bns])
            for i, bn in enumerate(bns):
                bn.num_batches_tracked = flat[i]
            object.__setattr__(stack, "_nbt_flat", flat)
        cmax = stack.feature_dim
        self.scale_shift = alloc((nbn, 2, cmax), torch.float32)
        bn_idx = [0]

        def new_scale_shift(c):
            i = bn_idx[0]
            bn_idx[0] += 1
            return self.scale_shift[i, 0, :c], self.scale_shift[i, 1, :c]

        # geometry pass to size scratch buffers
        geo = []
        h, w_ = (Ho + 2 - 3) // 2 + 1, (Wo + 2 - 3) // 2 + 1
        hp_, wp_ = h, w_
        inpl = width
        max_in = N * h * w_ * inpl
        max_c1 = max_c2 = max_c3 = 0
        for li, nblocks in enumerate(stack.arch["layers"]):
            planes = width * (2 ** li)
            for b in range(nblocks):
                stride = 2 if (li > 0 and b == 0) else 1
                h2, w2 = (h + 2 - 3) // stride + 1, (w_ + 2 - 3) // stride + 1
                geo.append((h, w_, h2, w2, inpl, planes, stride))
                max_c1 = max(max_c1, N * h * w_ * planes)
                max_c2 = max(max_c2, N * h2 * w2 * planes)
                max_c3 = max(max_c3, N * h2 * w2 * planes * 4)
                h, w_, inpl = h2, w2, planes * 4
        max_part = max([L.load().sat_conv_tiles_m(N * Ho * Wo) * 2 * width] +
                       [L.load().sat_conv_tiles_m(N * g[0] * g[1]) * 2 * g[5] for g in geo] +
                       [L.load().sat_conv_tiles_m(N * g[2] * g[3]) * 2 * g[5] * 4 for g in geo])
        self.partial = alloc((max_part,), torch.float32)
        self.c0 = alloc((N * Ho * Wo * width,))
        self.ybuf = [alloc((max(max_in, max_c3),)), alloc((max(max_in, max_c3),))]
        self.c1, self.a1 = alloc((max_c1,)), alloc((max_c1,))
        self.c2, self.a2 = alloc((max_c2,)), alloc((max_c2,))
        self.c3, self.cd = alloc((max_c3,)), alloc((max_c3,))
        self.pooled = alloc((N, stack.feature_dim), torch.float32)

        def conv_op(x, wt, out, n, hin, win, cin, hout, wout, cout, kh, kw, stride, pad, sN, sH, sW):
            o = L.SatOp()
            o.kind, o.dtype = L.OP_CONV, dtype
            o.in0, o.w, o.out = x.data_ptr(), wt.data_ptr(), out.data_ptr()
            o.N, o.Hin, o.Win, o.Cin, o.Hout, o.Wout, o.Cout = n, hin, win, cin, hout, wout, cout
            o.KH, o.KW, o.stride, o.pad = kh, kw, stride, pad
            o.sN, o.sH, o.sW = sN, sH, sW
            if training:
                o.stat_partial = self.partial.data_ptr()
                o.tiles_m = L.load().sat_conv_tiles_m(n * hout * wout)
            return o

        # BatchNorm statistics, two forms (both bitwise reproducible):
        #  * per-tile slabs + SAT_OP_BN_FINALIZE (f32 mode, eval mode, layers with many M-tiles);
        #  * bf16 training, <= ATOMIC_MAX_TILES M-tiles: the conv adds fixed-point sums with integer atomics into
        #    stat_acc[bn][parity][2][C] and the consuming BN_RELU / BN_ADD_RELU derives scale/shift itself
        #    (no finalize launch).  The parity alternates per run() so workgroup 0 of the consumer can clear the
        #    other half for the next step.
        ATOMIC_MAX_TILES = int(os.environ.get("SAT_ATOMIC_BN_MAX_TILES", "128"))
        atomic_stats = training and dtype == L.SAT_BF16 and ATOMIC_MAX_TILES > 0
        self._parity = 0
        bnref = {}
        fuse_in_bn = dtype == L.SAT_BF16 and os.environ.get("SAT_FUSE_INPUT_BN", "1") != "0"   # bn2+ReLU inside conv3
        # bn1+ReLU inside conv2 (3x3), OPT-IN: measured 7.33 vs 6.85 ms/step -- a 3x3 conv stages every input element 9 taps x
        # (N/128) tile columns = 18 times, so the in-LDS transform does 18x the work of the separate 5.6 us stream kernel and
        # doubles the LDS traffic of an LDS-bound K loop
        # SAT_FUSE_BN1=2 (round 3): only where conv_pr_kernel can run the conv (3x3 / stride 1, 128 <= planes <= 512, rows of <= 31
        # pixels): there the transform touches each 64-channel slice of the input patch ONCE per workgroup (LDS-resident patch,
        # sat_conv_pr.inc) instead of once per tap
        # Measured (bench.py, A/B on one box): 15.37 -> 15.95 k img/s with three stacks in flight, 11.1 -> 11.5 k strictly sequential:
        # 44 of the 50 normalise+ReLU launches of ResNet-152 disappear.  Default 2; 0 restores the separate launches, 1 fuses
        # everywhere (the in-ring transform of the older kernels where conv_pr_kernel cannot run: the measured loss above).
        fuse_bn1_mode = int(os.environ.get("SAT_FUSE_BN1", "2")) if (training and dtype == L.SAT_BF16) else 0
        fuse_bn1 = fuse_bn1_mode == 1
        slab_to_acc = os.environ.get("SAT_SLAB_TO_ACC", "1") != "0"
        fuse_out_bn = (not training) and dtype == L.SAT_BF16 and os.environ.get("SAT_FUSE_EVAL_BN", "1") != "0"
        # many-tile layers (ATOMIC_MAX_TILES < tiles <= SHARD_MAX_TILES): the same integer atomics into 8 SHARDS of the
        # accumulator (workgroup id % 8), summed by the consumer: no per-tile slabs, no reducer launch
        # OPT-IN (SAT_SHARDED_BN_MAX_TILES=1600): measured a wash at cfg 2 (6.861 vs 6.866 ms/step: the 36 reducer launches
        # it removes cost 0.18 ms, the contended atomics and the 8-shard table derivation in every consumer give it back)
        SHARD_MAX_TILES = int(os.environ.get("SAT_SHARDED_BN_MAX_TILES", "0"))
        self.stat_accs = []

        eval_items = []

        def fin_op(bn, c, count, tiles_m, consumer_can_derive=True):
            s, t = new_scale_shift(c)
            if atomic_stats and consumer_can_derive and tiles_m <= max(ATOMIC_MAX_TILES, SHARD_MAX_TILES):
                shards = 1 if tiles_m <= ATOMIC_MAX_TILES else 8
                acc = alloc((2, shards, 2, c), torch.int64, zero=True)
                self.stat_accs.append(acc)
                cv = ops[-1]                       # the conv that produces this BN's input
                assert cv.kind == L.OP_CONV and cv.Cout == c
                cv.stat_partial = None
                cv.stat_acc, cv.stat_shards = acc.data_ptr(), shards
                bnref[s.data_ptr()] = (acc.data_ptr(), bn, count, shards)
                self.bn_list.append(bn)
                return None, s, t
            if atomic_stats and consumer_can_derive and slab_to_acc:
                # very many M-tiles (the stem): the conv keeps writing per-tile slabs (no contended atomics), a wide reducer
                # launch folds them into the same integer accumulators, and the consumer derives (scale, shift) as above
                acc = alloc((2, 1, 2, c), torch.int64, zero=True)
                self.stat_accs.append(acc)
                o = L.SatOp()
                o.kind, o.dtype = L.OP_BN_FINALIZE, dtype
                o.stat_partial, o.stat_acc = self.partial.data_ptr(), acc.data_ptr()
                o.Cout, o.tiles_m, o.training = c, tiles_m, 1
                bnref[s.data_ptr()] = (acc.data_ptr(), bn, count, 1)
                self.bn_list.append(bn)
                return o, s, t
            if not training:
                # eval: (scale, shift) depend on parameters and running statistics only -> ONE batched launch for
                # all BatchNorms at the head of the program instead of a finalize launch per layer
                it = L.SatBnEvalItem()
                it.gamma, it.beta = bn.weight.data_ptr(), bn.bias.data_ptr()
                it.running_mean, it.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
                it.scale_out, it.shift_out, it.C = s.data_ptr(), t.data_ptr(), c
                eval_items.append(it)
                self.bn_list.append(bn)
                return None, s, t
            o = L.SatOp()
            o.kind, o.dtype = L.OP_BN_FINALIZE, dtype
            o.stat_partial = self.partial.data_ptr() if training else None
            o.gamma, o.beta = bn.weight.data_ptr(), bn.bias.data_ptr()
            o.running_mean, o.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
            o.scale_out, o.shift_out = s.data_ptr(), t.data_ptr()
            o.Cout, o.count, o.tiles_m, o.training = c, count, tiles_m, 1 if training else 0
            o.momentum, o.eps = BN_MOMENTUM, BN_EPS
            self.bn_list.append(bn)
            return o, s, t

        def fin_op_acc_only(bn, c, count):
            """integer accumulators of a BatchNorm whose producer AND consumer are one fused launch (SAT_OP_CONV3_FUSED)"""
            s, t = new_scale_shift(c)
            acc = alloc((2, 1, 2, c), torch.int64, zero=True)
            self.stat_accs.append(acc)
            bnref[s.data_ptr()] = (acc.data_ptr(), bn, count, 1)
            self.bn_list.append(bn)
            return None, s, t

        def act_op(kind, x, s, t, out, n, h_, w__, c, x1=None, s1=None, t1=None):
            o = L.SatOp()
            o.kind, o.dtype = kind, dtype
            o.in0, o.out = x.data_ptr(), out.data_ptr()
            ref = bnref.get(s.data_ptr())
            if ref is None:
                o.scale0, o.shift0 = s.data_ptr(), t.data_ptr()
            else:                                  # derive (scale, shift) from the conv's integer sums
                acc, bn, count, shards = ref
                o.stat_acc, o.stat_shards = acc, shards
                o.gamma, o.beta = bn.weight.data_ptr(), bn.bias.data_ptr()
                o.running_mean, o.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
                o.count, o.momentum, o.eps = count, BN_MOMENTUM, BN_EPS
            if x1 is not None:
                o.in1 = x1.data_ptr()
                if s1 is not None:
                    ref1 = bnref.get(s1.data_ptr())
                    if ref1 is None:
                        o.scale1, o.shift1 = s1.data_ptr(), t1.data_ptr()
                    else:
                        acc1, bn1_, count1, shards1 = ref1
                        o.stat_acc1, o.stat_shards1 = acc1, shards1
                        o.gamma1, o.beta1 = bn1_.weight.data_ptr(), bn1_.bias.data_ptr()
                        o.running_mean1, o.running_var1 = bn1_.running_mean.data_ptr(), bn1_.running_var.data_ptr()
                        o.count, o.momentum, o.eps = count1, BN_MOMENTUM, BN_EPS
            o.N, o.Hout, o.Wout, o.Cout = n, h_, w__, c
            return o

        def std_conv(conv, x, out, n, hin, win, hout, wout):
            wt = prep_w(conv).reshape(conv.cout, -1)
            self.keep.append(wt)
            cin = conv.cin
            return conv_op(x, wt, out, n, hin, win, cin, hout, wout, conv.cout, conv.k, conv.k, conv.stride, conv.pad,
                           hin * win * cin, win * cin, cin)

        def add(o):
            if o is not None:
                ops.append(o)

        # ---- program ----
        o = L.SatOp()
        o.kind, o.dtype = L.OP_IMAGE_PREP, dtype
        o.out = self.img_pad.data_ptr()
        o.N, o.Hin, o.Win, o.Hout, o.Wout, o.pad = N, H, W, Hp, Wp, 3
        ops.append(o)
        self._prep_index = 0
        ops.append(conv_op(self.img_pad, wst, self.c0, N, Hp, Wp, 32, Ho, Wo, width, 7, 1, 2, 0, Hp * Wp * 4, Wp * 4, 4))
        f, s, t = fin_op(stack.bn1, width, N * Ho * Wo, L.load().sat_conv_tiles_m(N * Ho * Wo))
        add(f)
        y, ynext = self.ybuf
        mp = L.SatOp()
        mp.kind, mp.dtype = L.OP_BN_RELU_MAXPOOL, dtype
        mp.in0, mp.out = self.c0.data_ptr(), y.data_ptr()
        ref = bnref.get(s.data_ptr())
        if ref is None:
            mp.scale0, mp.shift0 = s.data_ptr(), t.data_ptr()
        else:                                      # the pooling kernel derives (scale, shift) from the integer sums
            acc, bn_, count_, shards_ = ref
            mp.stat_acc, mp.stat_shards = acc, shards_
            mp.gamma, mp.beta = bn_.weight.data_ptr(), bn_.bias.data_ptr()
            mp.running_mean, mp.running_var = bn_.running_mean.data_ptr(), bn_.running_var.data_ptr()
            mp.count, mp.momentum, mp.eps = count_, BN_MOMENTUM, BN_EPS
        mp.N, mp.Hin, mp.Win, mp.Cout, mp.Hout, mp.Wout = N, Ho, Wo, width, hp_, wp_
        ops.append(mp)
        # bn3 + residual add + ReLU of an identity-residual bottleneck folded into the NEXT bottleneck's conv1 (bf16
        # training): that conv forms y = relu(c3*s3+t3 + y_prev) in LDS from two LDS-DMA sources and stores y once as
        # the next residual -- 45 of the 50 normalise+add launches of ResNet-152 and one re-read of y disappear.
        # OPT-IN (SAT_FUSE_RESIDUAL=1): measured at cfg 2 it is a wash (6.83 vs 6.81 ms/step): the 77 MB the
        # separate kernel streams at 5.5 TB/s from 2048 workgroups then has to come through 196 workgroups' LDS rings
        # (2.6 TB/s: bytes in flight per CU), which costs the conv what the removed launch saved.
        # SAT_FUSE_RESIDUAL=2 (round 3): only where conv_du_kernel can run the next conv1 (K = 4 * planes <= 1024 input channels, 256
        # output columns: the transitions inside layer 3 and into it) -- 64 rows x all columns per workgroup, 32-channel stages in a
        # six-slot ring, the transform and the y store on the loader waves (sat_conv_du.inc); y is written over the raw conv3
        # tensor IN PLACE, so the in-place passes of the other layers stay as they are.
        fuse_resid_mode = int(os.environ.get("SAT_FUSE_RESIDUAL", "0")) if (training and dtype == L.SAT_BF16) else 0
        fuse_resid = fuse_resid_mode == 1
        # TWO-PASS conv3 (bf16 training, identity-residual bottlenecks), OPT-IN (SAT_CONV3_TWOPASS=1): conv3 runs once for its
        # BatchNorm statistics only (no output), then again with bn3 + residual add + ReLU in its epilogue (scale / shift derived
        # from the sums of pass 1).  The raw conv3 tensor is never written or re-read and the normalise+add launch disappears
        # (a layer-3 bottleneck's memory traffic drops from ~167 to ~122 MB); bit-identical to the three-launch form (tested).
        # Built because with several stacks in flight the step is bound by memory traffic (tools/run_gpu_traffic_probe.sh: without
        # the normalise+add launches a step takes 3.84 instead of 4.78 ms), and MEASURED A LOSS: the statistics-only pass costs
        # 17.7 us of the conv's 22 (the store is the smaller part of a K = 256 conv with its fused input BatchNorm), the second
        # pass 25.3, together 43 us against 22 + 16 for conv + normalise+add: 4.95 vs 4.85 ms/step (profiles/r03_twopass_ab.txt).
        two_pass = training and dtype == L.SAT_BF16 and not fuse_resid and os.environ.get("SAT_CONV3_TWOPASS", "0") == "1"
        # SINGLE-PASS fused conv3 (SAT_OP_CONV3_FUSED, sat_conv3_fused.hip): conv3 + bn3 + residual add + ReLU in one launch, the
        # f32 accumulators held in registers across a grid-wide statistics barrier -- the raw conv3 tensor and the normalise+add
        # launch disappear without a second pass.  Needs the whole grid resident (sat_conv3_fused_ok) and integer-atomic sums.
        fused3 = training and dtype == L.SAT_BF16 and not fuse_resid and not two_pass and fused_conv3_enabled()
        self.fused_sync, self.fused_err, self.n_fused3 = None, None, 0
        if fused3:
            nblk = len(list(stack.blocks()))
            self.fused_sync = alloc((nblk, 2), torch.int32, zero=True)
            self.fused_err = alloc((4,), torch.int32, zero=True)
        # IN-PLACE BatchNorm-apply passes (bf16 training): relu(bn1(c1)) overwrites c1, and conv3 writes its raw output straight into
        # the next block-output buffer, which the normalise+add pass then transforms in place -- a bottleneck touches two large
        # buffers instead of three (c3 disappears), so a stack's live set in layer 3 drops from ~90 to ~65 MB and three stacks in
        # flight fit the 256 MB Infinity Cache (the look-ahead step is bound by memory traffic).  SAT_BN_INPLACE=0: off.
        inplace = (training and dtype == L.SAT_BF16 and not fuse_resid and not two_pass and
                   os.environ.get("SAT_BN_INPLACE", "1") != "0")
        pending = None          # (s3, t3, resid buffer) of the previous block when its bn_add is deferred to this conv1
        blocks_geo = list(zip(stack.blocks(), geo))
        for bi, (blk, (h, w_, h2, w2, inpl, planes, stride)) in enumerate(blocks_geo):
            tm1 = L.load().sat_conv_tiles_m(N * h * w_)
            tm2 = L.load().sat_conv_tiles_m(N * h2 * w2)
            if fuse_out_bn:
                # inference: every BatchNorm is a fixed per-channel affine -> it rides in the producing conv's epilogue
                # together with the residual add and the ReLU: 3-4 launches per bottleneck instead of 6-8
                def fused(conv, bn, x, out, hin, win, hout, wout, count, relu, resid=None):
                    cv = std_conv(conv, x, out, N, hin, win, hout, wout)
                    _, s_, t_ = fin_op(bn, conv.cout, count, 0)
                    cv.scale1, cv.shift1, cv.flags = s_.data_ptr(), t_.data_ptr(), 1 if relu else 0
                    if resid is not None:
                        cv.in1 = resid.data_ptr()
                    ops.append(cv)
                fused(blk.conv1, blk.bn1, y, self.a1, h, w_, h, w_, N * h * w_, True)
                fused(blk.conv2, blk.bn2, self.a1, self.a2, h, w_, h2, w2, N * h2 * w2, True)
                resid = y
                if blk.downsample is not None:
                    fused(blk.downsample[0], blk.downsample[1], y, self.cd, h, w_, h2, w2, N * h2 * w2, False)
                    resid = self.cd
                fused(blk.conv3, blk.bn3, self.a2, ynext, h2, w2, h2, w2, N * h2 * w2, True, resid)
                y, ynext = ynext, y
                continue
            if pending is not None:
                ps3, pt3, presid = pending
                if fuse_resid_mode == 2:
                    # the raw conv3 tensor sits in y itself (in-place mode): conv_du_kernel forms relu(bn3(y) + presid) stage by stage,
                    # multiplies it and writes it back over y
                    cv1 = std_conv(blk.conv1, y, self.c1, N, h, w_, h, w_)
                else:
                    cv1 = std_conv(blk.conv1, self.c3, self.c1, N, h, w_, h, w_)  # A = previous RAW conv3 output ...
                cv1.in1, cv1.out1 = presid.data_ptr(), y.data_ptr()              # ... + previous block input -> y (stored too)
                ref = bnref.get(ps3.data_ptr())
                if ref is None:
                    cv1.scale0, cv1.shift0 = ps3.data_ptr(), pt3.data_ptr()
                else:
                    acc3, bn3_, count3, shards3 = ref
                    cv1.stat_acc1, cv1.stat_shards1 = acc3, shards3
                    cv1.gamma1, cv1.beta1 = bn3_.weight.data_ptr(), bn3_.bias.data_ptr()
                    cv1.running_mean1, cv1.running_var1 = bn3_.running_mean.data_ptr(), bn3_.running_var.data_ptr()
                    cv1.count, cv1.momentum, cv1.eps = count3, BN_MOMENTUM, BN_EPS
                ops.append(cv1)
                pending = None
            else:
                ops.append(std_conv(blk.conv1, y, self.c1, N, h, w_, h, w_))
            f, s1, t1 = fin_op(blk.bn1, planes, N * h * w_, tm1)
            add(f)
            pr_geom = (h2 == h and w2 == w_ and planes % 64 == 0 and 128 <= planes <= 512 and w_ <= 31)
            if (fuse_bn1 or (fuse_bn1_mode == 2 and pr_geom)) and planes <= 512 and planes % 64 == 0:
                # conv2 (3x3) reads the RAW c1 and applies bn1 + ReLU to every landed A stage in LDS (pipelined one K-step
                # ahead of the MFMAs; a per-row tap mask keeps the zero padding zero): a1 never exists in HBM
                cv2 = std_conv(blk.conv2, self.c1, self.c2, N, h, w_, h2, w2)
                ref = bnref.get(s1.data_ptr())
                if ref is None:
                    cv2.scale0, cv2.shift0 = s1.data_ptr(), t1.data_ptr()
                else:
                    acc1_, bn1m, count1_, shards1_ = ref
                    cv2.stat_acc1, cv2.stat_shards1 = acc1_, shards1_
                    cv2.gamma1, cv2.beta1 = bn1m.weight.data_ptr(), bn1m.bias.data_ptr()
                    cv2.running_mean1, cv2.running_var1 = bn1m.running_mean.data_ptr(), bn1m.running_var.data_ptr()
                    cv2.count, cv2.momentum, cv2.eps = count1_, BN_MOMENTUM, BN_EPS
                ops.append(cv2)
            else:
                a1buf = self.c1 if inplace else self.a1
                ops.append(act_op(L.OP_BN_RELU, self.c1, s1, t1, a1buf, N, h, w_, planes))
                ops.append(std_conv(blk.conv2, a1buf, self.c2, N, h, w_, h2, w2))
            f, s2, t2 = fin_op(blk.bn2, planes, N * h2 * w2, tm2)
            add(f)
            ref2 = bnref.get(s2.data_ptr())
            if (fused3 and blk.downsample is None and ref2 is not None and ref2[3] == 1 and tm2 <= ATOMIC_MAX_TILES and
                    L.load().sat_conv3_fused_ok(N * h2 * w2, planes * 4, planes)):
                acc2, bn2_, count2, shards2 = ref2
                _, s3, t3 = fin_op_acc_only(blk.bn3, planes * 4, N * h2 * w2)
                acc3, bn3_, count3, _ = bnref[s3.data_ptr()]
                wt = prep_w(blk.conv3).reshape(blk.conv3.cout, -1)
                self.keep.append(wt)
                o = L.SatOp()
                o.kind, o.dtype = L.OP_CONV3_FUSED, dtype
                o.in0, o.w, o.in1, o.out = self.c2.data_ptr(), wt.data_ptr(), y.data_ptr(), ynext.data_ptr()
                o.N, o.Hin, o.Win, o.Cin, o.Hout, o.Wout, o.Cout = N, h2, w2, planes, h2, w2, planes * 4
                o.KH, o.KW, o.stride, o.pad = 1, 1, 1, 0
                o.stat_acc1, o.stat_shards1 = acc2, shards2
                o.gamma1, o.beta1 = bn2_.weight.data_ptr(), bn2_.bias.data_ptr()
                o.running_mean1, o.running_var1 = bn2_.running_mean.data_ptr(), bn2_.running_var.data_ptr()
                o.stat_acc, o.stat_shards = acc3, 1
                o.gamma, o.beta = bn3_.weight.data_ptr(), bn3_.bias.data_ptr()
                o.running_mean, o.running_var = bn3_.running_mean.data_ptr(), bn3_.running_var.data_ptr()
                o.count, o.momentum, o.eps = count3, BN_MOMENTUM, BN_EPS
                o.scale_out = self.fused_sync[bi].data_ptr()
                o.shift_out = self.fused_err.data_ptr()
                ops.append(o)
                self.n_fused3 += 1
                y, ynext = ynext, y
                continue
            if fuse_in_bn and planes <= 512 and planes % 64 == 0:
                # conv3 reads the RAW c2 and applies bn2 + ReLU to its A operand in LDS: a2 never exists in HBM
                c3buf = ynext if inplace else self.c3
                cv3 = std_conv(blk.conv3, self.c2, c3buf, N, h2, w2, h2, w2)
                ref = bnref.get(s2.data_ptr())
                if ref is None:
                    cv3.scale0, cv3.shift0 = s2.data_ptr(), t2.data_ptr()
                else:
                    acc2, bn2_, count2, shards2 = ref
                    cv3.stat_acc1, cv3.stat_shards1 = acc2, shards2
                    cv3.gamma1, cv3.beta1 = bn2_.weight.data_ptr(), bn2_.bias.data_ptr()
                    cv3.running_mean1, cv3.running_var1 = bn2_.running_mean.data_ptr(), bn2_.running_var.data_ptr()
                    cv3.count, cv3.momentum, cv3.eps = count2, BN_MOMENTUM, BN_EPS
                ops.append(cv3)
            else:
                c3buf = ynext if inplace else self.c3
                a2buf = self.c2 if inplace else self.a2
                ops.append(act_op(L.OP_BN_RELU, self.c2, s2, t2, a2buf, N, h2, w2, planes))
                ops.append(std_conv(blk.conv3, a2buf, c3buf, N, h2, w2, h2, w2))
            cv3_first = ops[-1]
            f, s3, t3 = fin_op(blk.bn3, planes * 4, N * h2 * w2, tm2)
            add(f)
            ref3 = bnref.get(s3.data_ptr())
            if (two_pass and blk.downsample is None and ref3 is not None and cv3_first.kind == L.OP_CONV and
                    (cv3_first.scale0 or cv3_first.stat_acc1) and fuse_in_bn):
                # pass 1 = the conv just emitted, statistics only; pass 2 = the same conv with the output-side BatchNorm
                cv3_first.flags |= L.CONV_STATS_ONLY
                acc3, bn3_, count3, shards3 = ref3
                cvb = std_conv(blk.conv3, self.c2, ynext, N, h2, w2, h2, w2)
                cvb.w = cv3_first.w                                  # same kernel-layout weights
                cvb.stat_partial = None
                cvb.scale0, cvb.shift0 = cv3_first.scale0, cv3_first.shift0
                cvb.stat_acc1, cvb.stat_shards1 = cv3_first.stat_acc1, cv3_first.stat_shards1
                cvb.gamma1, cvb.beta1 = cv3_first.gamma1, cv3_first.beta1
                cvb.running_mean1, cvb.running_var1 = None, None     # bn2's running statistics were updated by pass 1
                cvb.stat_acc, cvb.stat_shards = acc3, shards3
                cvb.gamma, cvb.beta = bn3_.weight.data_ptr(), bn3_.bias.data_ptr()
                cvb.running_mean, cvb.running_var = bn3_.running_mean.data_ptr(), bn3_.running_var.data_ptr()
                cvb.count, cvb.momentum, cvb.eps = count3, BN_MOMENTUM, BN_EPS
                cvb.in1 = y.data_ptr()
                cvb.flags |= 1 | L.CONV_OUT_BN
                ops.append(cvb)
                y, ynext = ynext, y
                continue
            if blk.downsample is not None:
                ops.append(std_conv(blk.downsample[0], y, self.cd, N, h, w_, h2, w2))
                f, sd, td_ = fin_op(blk.downsample[1], planes * 4, N * h2 * w2, tm2)
                add(f)
                ops.append(act_op(L.OP_BN_ADD_RELU, c3buf, s3, t3, ynext, N, h2, w2, planes * 4, self.cd, sd, td_))
            elif (fuse_resid and bi + 1 < len(blocks_geo) and (planes * 4) % 64 == 0 and planes * 4 <= 2048):
                pending = (s3, t3, y)            # the next block's conv1 forms relu(c3*s3+t3 + y) itself and writes it to ynext
            elif (fuse_resid_mode == 2 and inplace and bi + 1 < len(blocks_geo) and (planes * 4) % 64 == 0 and 320 <= planes * 4 <= 1024 and
                  blocks_geo[bi + 1][1][5] == 256 and ref3 is not None and tm2 <= ATOMIC_MAX_TILES):
                pending = (s3, t3, y)            # (conv3 wrote its raw output into ynext: the next conv1 transforms it in place)
            else:
                ops.append(act_op(L.OP_BN_ADD_RELU, c3buf, s3, t3, ynext, N, h2, w2, planes * 4, y))
            y, ynext = ynext, y
        ap = L.SatOp()
        ap.kind, ap.dtype = L.OP_AVGPOOL, dtype
        ap.in0, ap.out = y.data_ptr(), self.pooled.data_ptr()
        ap.N, ap.Hin, ap.Win, ap.Cout = N, geo[-1][2], geo[-1][3], stack.feature_dim
        ops.append(ap)
        self.final_map = (y, N, geo[-1][2], geo[-1][3], stack.feature_dim)
        if eval_items:
            arr = (L.SatBnEvalItem * len(eval_items))(*eval_items)
            self.eval_table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
            self.keep.append(self.eval_table)
            o = L.SatOp()
            o.kind, o.dtype = L.OP_BN_EVAL_BATCH, dtype
            o.in0, o.count, o.eps = self.eval_table.data_ptr(), len(eval_items), BN_EPS
            ops.insert(1, o)                     # right after image prep, before the first consumer
        self.ops = (L.SatOp * len(ops))(*ops)
        self.n_ops = len(ops)
        # replay as a hipGraph (SAT_GRAPH=0: eager launches).  Per step parity: first run eager, then captured.
        self._use_graph = os.environ.get("SAT_GRAPH", "1") != "0" and torch.device(device).type == "cuda"
        self._runs, self._graphs = [0, 0], [None, None]
        self._running_items = None              # defer_running_stats(): number of redirected BatchNorms
        # build-time kernel selection per conv geometry (bf16): time every variant on this program's own buffers.
        # The choice is timing dependent and the tile shape fixes the BatchNorm summation order, so bf16 results are
        # bit-reproducible across processes only with the same choices: SAT_TUNE_FILE=<json> saves them / loads them back.
        if dtype == L.SAT_BF16 and os.environ.get("SAT_AUTOTUNE", "1") != "0" and torch.device(device).type == "cuda":
            tune_file = os.environ.get("SAT_TUNE_FILE")
            table = {}
            if tune_file and os.path.exists(tune_file):
                with open(tune_file) as f:
                    table = json.load(f)
            missing = False
            for i in range(self.n_ops):
                if self.ops[i].kind == L.OP_CONV:
                    v = table.get(self._tune_key(self.ops[i]))
                    if v is None:
                        missing = True
                    else:
                        self.ops[i].variant = int(v)
            if missing:
                for t in (self.c0, self.c1, self.a1, self.c2, self.a2, self.c3, self.cd, *self.ybuf):
                    t.normal_()
                scratch = alloc((4096,), torch.float32)           # the tuner's neutral BatchNorm table lives in OUR memory
                L.check(L.load().sat_conv_autotune(self.ops, self.n_ops, 5, scratch.data_ptr(), scratch.numel() * 4,
                                                   L.stream()), "sat_conv_autotune")
                torch.cuda.synchronize()
                if tune_file:
                    for i in range(self.n_ops):
                        if self.ops[i].kind == L.OP_CONV:
                            table[self._tune_key(self.ops[i])] = int(self.ops[i].variant)
                    tmp = "%s.%d.tmp" % (tune_file, os.getpid())          # whole-file replace: other ranks may be reading it
                    with open(tmp, "w") as f:
                        json.dump(table, f, indent=0, sort_keys=True)
                    os.replace(tmp, tune_file)

    @staticmethod
    def _tune_key(o):
        fused = (1 if (o.stat_partial or o.stat_acc) else 0) + (2 if (o.scale0 or o.stat_acc1) else 0) + \
                (4 if o.scale1 else 0) + (8 if o.in1 else 0) + (16 if o.out1 else 0) + (32 if (o.flags & L.CONV_OUT_BN) else 0) + \
                (64 if (o.flags & L.CONV_STATS_ONLY) else 0)
        return "%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d" % (o.N, o.Hin, o.Win, o.Cin, o.Hout, o.Wout, o.Cout, o.KH, o.KW, o.stride, fused)

    def __del__(self):
        for g in getattr(self, "_graphs", ()):
            if g is not None:
                try:
                    L.load().sat_graph_destroy(g)
                except Exception:
                    pass

    def defer_running_stats(self):
        """Aim every running-statistics update of this (train-mode) program at private zeroed buffers with momentum 1, so that a
        run leaves each layer's batch (mean, unbiased var) there and touches NO model state; `apply_running_stats()` then does
        the real momentum update in one launch.  Lets several batches' frozen stacks be in flight at once while the model's
        running statistics still advance in batch order (TrainStep.prefetch_encoder).  Call before the first run."""
        if not self.training or self._running_items is not None:
            return
        if self._runs != [0, 0]:
            raise RuntimeError("defer_running_stats must precede the first run (the hipGraph captures the pointers)")
        dev = self.pooled.device
        by_ptr = {bn.running_mean.data_ptr(): bn for bn in self.stack.bns()}
        items, seen = [], set()
        for i in range(self.n_ops):
            o = self.ops[i]
            hit = False
            for fm, fv in (("running_mean", "running_var"), ("running_mean1", "running_var1")):
                ptr = getattr(o, fm)
                if not ptr:
                    continue
                bn = by_ptr.get(ptr)
                if bn is None or ptr in seen:
                    raise RuntimeError("op %d updates running statistics this program cannot attribute to one BatchNorm" % i)
                seen.add(ptr)
                log = torch.zeros(2, bn.running_mean.numel(), dtype=torch.float32, device=dev)
                self.keep.append(log)
                setattr(o, fm, log[0].data_ptr())
                setattr(o, fv, log[1].data_ptr())
                it = L.SatBnRunningItem()
                it.running_mean, it.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
                it.batch_mean, it.batch_var, it.C = log[0].data_ptr(), log[1].data_ptr(), bn.running_mean.numel()
                items.append(it)
                hit = True
            if hit:
                o.momentum = 1.0           # running' = 0 * running + 1 * f32(batch statistic): the log holds the statistic itself
        arr = (L.SatBnRunningItem * max(len(items), 1))(*items)
        self._running_table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        self._running_items = len(items)

    def apply_running_stats(self):
        """Momentum update of the model's running statistics from the last run's batch statistics (deferred programs only),
        on the current stream; the caller has ordered that stream behind the run."""
        if self._running_items:
            L.check(L.load().sat_bn_running_apply(self._running_table.data_ptr(), self._running_items, BN_MOMENTUM, L.stream()),
                    "sat_bn_running_apply")
            self.stack._nbt_flat += 1

    def run(self, images):
        """images f32 [N,3,H,W] NCHW on the device -> pooled f32 [N, feature_dim] (owned by the program)."""
        L.require_gpu(images, "images")
        if images.dtype != torch.float32 or tuple(images.shape) != (self.N, 3, self.H, self.W):
            raise ValueError("images must be float32 [%d,3,%d,%d]" % (self.N, self.H, self.W))
        images = images.contiguous()
        lib, p = L.load(), self._parity
        self.ops[0].in0 = images.data_ptr()
        if not self._use_graph or self._runs[p] == 0:
            L.check(lib.sat_run_ops_parity(self.ops, self.n_ops, p, L.stream()), "sat_run_ops")
        else:
            # image prep reads the caller's tensor (a new pointer every batch) -> eager; everything after it only
            # touches the program's own buffers -> one hipGraph per step parity, captured on this parity's 2nd run
            if self._graphs[p] is None:
                tail = (L.SatOp * (self.n_ops - 1))(*list(self.ops)[1:])
                g = C.c_void_p()
                L.check(lib.sat_graph_create(tail, self.n_ops - 1, p, C.byref(g)), "sat_graph_create")
                self._graphs[p] = g
            L.check(lib.sat_run_ops_parity(self.ops, 1, p, L.stream()), "sat_run_ops")
            L.check(lib.sat_graph_launch(self._graphs[p], L.stream()), "sat_graph_launch")
        self._runs[p] += 1
        self._parity ^= 1
        if self.training and self._running_items is None:
            self.stack._nbt_flat += 1
        if self.n_fused3:
            # the fused conv3 launches' sticky status word (a grid-barrier wait that ran out): read back behind the run, raised
            # at the latest on the next submit; the process then builds its programs without the fused launch
            from .watch import ResidencyWatch
            ResidencyWatch.get(self.pooled.device).submit(self.fused_err[0:1], "the fused conv3 + BatchNorm launch", disable_fused_conv3)
        return self.pooled


def _run_timed(self, images):
    """Diagnostics (bench.py's roofline figure): one eager, in-order run of the whole program -- same kernels, same
    statistics / parity bookkeeping as `run` -- that also returns every conv launch's own duration in microseconds
    (dispatch timestamps via `sat_run_ops_timed`).  Synchronises the stream."""
    L.require_gpu(images, "images")
    images = images.contiguous()
    lib, p = L.load(), self._parity
    self.ops[0].in0 = images.data_ptr()
    us = (C.c_float * self.n_ops)()
    L.check(lib.sat_run_ops_timed(self.ops, self.n_ops, p, L.stream(), us), "sat_run_ops_timed")
    self._runs[p] += 1
    self._parity ^= 1
    if self.training:
        self.stack._nbt_flat += 1
    return self.pooled, [float(us[i]) for i in range(self.n_ops) if self.ops[i].kind in (L.OP_CONV, L.OP_CONV3_FUSED)]


ConvStackProgram.run_timed = _run_timed


def conv_flops(arch, H=224, W=224):
    """Algorithmic FLOPs (2*MAC) per image of the conv stack (23.02 GFLOP at 224x224 for ResNet-152)."""
    width = arch["width"]
    Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    fl = 2.0 * Ho * Wo * 3 * 49 * width
    h, w_ = (Ho + 2 - 3) // 2 + 1, (Wo + 2 - 3) // 2 + 1
    inpl = width
    for li, nblocks in enumerate(arch["layers"]):
        planes = width * (2 ** li)
        for b in range(nblocks):
            stride = 2 if (li > 0 and b == 0) else 1
            h2, w2 = (h + 2 - 3) // stride + 1, (w_ + 2 - 3) // stride + 1
            fl += 2.0 * h * w_ * inpl * planes
            fl += 2.0 * h2 * w2 * planes * planes * 9
            fl += 2.0 * h2 * w2 * planes * planes * 4
            if b == 0 and (stride != 1 or inpl != planes * 4):
                fl += 2.0 * h2 * w2 * inpl * planes * 4
            h, w_, inpl = h2, w2, planes * 4
    return fl
