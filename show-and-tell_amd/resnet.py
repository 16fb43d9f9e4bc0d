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
import os

import torch
import torch.nn as nn

from . import _lib as L
from . import tune as T

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
    """Changes whenever a frozen conv weight is replaced (a new Parameter object on the module) or written in place through the
    Parameter (load_state_dict, copy_): the op program keeps permuted (bf16) copies of the conv weights and must be rebuilt
    then.  Writes through `.data` bypass the version counter -- call `EncoderCNN.refresh_weights()` after those."""
    # the walk over the module tree is cached (155 convs on ResNet-152, 3-4 signatures per step) as (module, weight) pairs; a
    # module whose `.weight` is no longer the cached object (assignment, module conversion) re-walks; `_sig_params` is also
    # dropped by EncoderCNN._invalidate (load_state_dict, device / dtype moves, refresh_weights)
    ws = stack.__dict__.get("_sig_params")
    if ws is None or any(m._parameters.get("weight") is not w for m, w in ws):
        ws = [(m, m.weight) for m in stack.modules() if isinstance(m, _Conv)]
        stack.__dict__["_sig_params"] = ws
    sig = 0
    for _, w in ws:
        sig = (sig * 1000003 + w._version * 7 + (w.data_ptr() & 0xffffffff)) & ((1 << 61) - 1)
    return sig


def _tdtype(dtype):
    return torch.bfloat16 if dtype == L.SAT_BF16 else torch.float32


# BatchNorm statistics as integer atomics straight from the conv epilogue up to this many 128-row tiles; beyond (the stem,
# layer 1) the conv writes per-tile slabs and a wide reducer launch folds them into the same accumulators
ATOMIC_MAX_TILES = 128      # (measured again with the round-4 kernels: 400 / 1600 slow the convs by 2 / 6 % for the 21 / 37 reducer launches they save)


_PROGRAM_PICKS = {}      # tune key -> variant chosen in a program of this process (ConvStackProgram._pick_in_program)


class ConvStackProgram:
    """Device buffers + sat_op array for one (batch, H, W, dtype, training) configuration.

    groups = G > 1 (bf16, training): the program runs G independent batches in every launch (`sat_op.groups`, grid.y =
    group): activations are [G][N]..., every BatchNorm keeps per-group batch statistics, weights are shared.  Each group is,
    instruction for instruction, the ungrouped program on its batch as long as both run kernel variants of the same statistics
    signature: the first program built for a model state tunes freely, every other one gets its `signatures()` as a constraint
    (`signatures=`; `EncoderCNN._program` builds the grouped one first), so a batch's pooled features and BatchNorm statistics
    are bit-identical whichever program runs it.  Grouped programs always run with deferred running statistics
    (`defer_running_stats`), one update per consumed batch.  groups > 1 in eval mode: BatchNorm is a fixed affine there, so the
    batches of a group simply concatenate into one program over groups * N images (no `sat_op.groups`)."""

    def __init__(self, stack, N, H, W, dtype, training, device, groups=1, signatures=None):
        self.N, self.H, self.W, self.dtype, self.training, self.groups = N, H, W, dtype, training, int(groups)
        # statistics signatures (sat_conv_variant_signature) per conv geometry that the tuner has to stay within: those of the
        # FIRST program built for this model state (`signatures()`), so that every program gives a batch the same bits
        self._want_sigs = dict(signatures or {})
        if self.groups > 1 and dtype != L.SAT_BF16:
            raise ValueError("grouped programs are for the bf16 stack")
        # eval mode: BatchNorm is a fixed affine, so the batches of a group simply CONCATENATE -- one program over groups * N images
        # (no per-group statistics, no sat_op.groups), bit-identical per image to the ungrouped program
        Nb = N                                   # images per batch (what one image-prep op converts)
        if self.groups > 1 and not training:
            N, G = self.groups * N, 1
        else:
            G = self.groups
        self.keep = []      # tensors the op array points into
        td = _tdtype(dtype)
        ch = 8 if dtype == L.SAT_BF16 else 4
        width = stack.arch["width"]
        if width % ch:
            raise ValueError("conv stack width must be a multiple of %d for this dtype" % ch)
        ops = []
        lib = L.load()

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
        self.img_pad = alloc((G * N, Hp, Wp, 4), zero=True)           # (eval: N is already groups * Nb)
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
        max_part = max([lib.sat_conv_tiles_m(N * Ho * Wo) * 2 * width] +
                       [lib.sat_conv_tiles_m(N * g[0] * g[1]) * 2 * g[5] for g in geo] +
                       [lib.sat_conv_tiles_m(N * g[2] * g[3]) * 2 * g[5] * 4 for g in geo])
        # every per-batch buffer of a grouped program is G consecutive copies of the ungrouped one: a group's slice of an
        # activation buffer starts at g * (elements of THAT tensor), so the ping-pong buffers are sized G x their largest tenant
        self.partial = alloc((G * max_part,), torch.float32)
        self.c0 = alloc((G * N * Ho * Wo * width,))
        self.ybuf = [alloc((G * max(max_in, max_c3),)), alloc((G * max(max_in, max_c3),))]
        self.c1, self.a1 = alloc((G * max_c1,)), alloc((G * max_c1,))
        self.c2, self.a2 = alloc((G * max_c2,)), alloc((G * max_c2,))
        self.c3, self.cd = alloc((G * max_c3,)), alloc((G * max_c3,))
        self.pooled = alloc((G * N, stack.feature_dim), torch.float32)

        def conv_op(x, wt, out, n, hin, win, cin, hout, wout, cout, kh, kw, stride, pad, sN, sH, sW):
            o = L.SatOp()
            o.kind, o.dtype, o.groups = L.OP_CONV, dtype, G
            o.in0, o.w, o.out = x.data_ptr(), wt.data_ptr(), out.data_ptr()
            o.N, o.Hin, o.Win, o.Cin, o.Hout, o.Wout, o.Cout = n, hin, win, cin, hout, wout, cout
            o.KH, o.KW, o.stride, o.pad = kh, kw, stride, pad
            o.sN, o.sH, o.sW = sN, sH, sW
            if training:
                o.stat_partial = self.partial.data_ptr()
                o.tiles_m = lib.sat_conv_tiles_m(n * hout * wout)
            return o

        # BatchNorm statistics, two forms (both bitwise reproducible):
        #  * per-tile slabs + SAT_OP_BN_FINALIZE (f32 mode, eval mode, layers with many M-tiles);
        #  * bf16 training, <= ATOMIC_MAX_TILES M-tiles: the conv adds fixed-point sums with integer atomics into
        #    stat_acc[group][parity][2][C] and the consuming BN_RELU / BN_ADD_RELU / conv derives scale/shift itself
        #    (no finalize launch).  The parity alternates per run() so workgroup 0 of the consumer can clear the
        #    other half for the next step.
        atomic_stats = training and dtype == L.SAT_BF16
        self._parity = 0
        bnref = {}
        # bf16: bn2 + ReLU inside conv3 (1x1: the operand transform of conv_xp_kernel / the ring kernel), bn1 + ReLU inside
        # conv2 where conv_pr_kernel can run it (3x3 / stride 1, 128 <= planes <= 512, rows of <= 31 pixels: the transform
        # touches each 64-channel slice of the LDS-resident input patch once per workgroup).  Measured (round 3, bench.py, A/B on
        # one box): 15.37 -> 15.95 k img/s with three stacks in flight: 44 of the 50 normalise+ReLU launches of ResNet-152 disappear
        # (round 4, with the weights-in-registers kernels: un-fusing bn1 / bn2 / both makes the convs 2 / 1.5 / 4 % faster in sequence and
        # the step 0.5 / 2 / 4 % slower -- the separate passes' bytes still cost more than the transforms)
        fuse_in_bn = dtype == L.SAT_BF16
        fuse_bn1 = training and dtype == L.SAT_BF16
        fuse_out_bn = (not training) and dtype == L.SAT_BF16
        self.stat_accs = []
        # bf16 training: bn3 + residual add + ReLU in conv3's EPILOGUE, with bn3's batch statistics taken from the Gram matrix of
        # conv3's input (csrc/sat_gram.hip: mean_c = w_c . mu, var_c = w_c^T cov(a2) w_c) -- the raw conv3 tensor and the
        # normalise+add launch (42 % of the stack's memory traffic in round 4, at the HBM roof) never exist.  The kernels run
        # non-projection bottlenecks with planes in {128, 256, 384, 512} (44 of ResNet-152's 50); MEASURED (round 5, interleaved on
        # one box, profiles/r05_gram_ab.txt) it pays only where the normalise+add pass is large against the chain's four small
        # launches: planes 128 (layer 2: +1 % on the step); at planes 256 (layer 3: 35 of the 44) the chain costs 30-45 us per
        # bottleneck against the 19-37 us launch it removes and the step LOSES 9 % -- so the default fuses planes <= 128 only
        # (SAT_GRAM_MAX_PLANES=512: every eligible bottleneck; SAT_GRAM_BN3=0: none, the three-launch form everywhere)
        gram_bn3 = training and dtype == L.SAT_BF16 and os.environ.get("SAT_GRAM_BN3", "1") != "0"
        gram_pmax = int(os.environ.get("SAT_GRAM_MAX_PLANES", "128"))
        self.gram_blocks = 0
        gram_geo = [(N * g_[2] * g_[3], g_[5]) for g_ in geo
                    if g_[6] == 1 and g_[4] == g_[5] * 4 and g_[5] % 128 == 0 and g_[5] <= gram_pmax] if gram_bn3 else []
        if gram_geo:
            self.gram_slabs = alloc((G * max(lib.sat_gram_slab_floats(m_, p_) for m_, p_ in gram_geo),), torch.float32)
            pmax = max(p_ for _, p_ in gram_geo)
            self.gram_cov3 = alloc((G * 3 * pmax * pmax,), torch.bfloat16)
            self.gram_mu = alloc((G * pmax,), torch.float64)
            self.gram_T = alloc((G * 3 * max(p_ * p_ * 4 for _, p_ in gram_geo),), torch.float32)

        eval_items = []

        def bn_fields(o, ref, first=True):
            """aim an op's (first / second) BatchNorm source at a conv's integer sums: the kernel derives (scale, shift) itself"""
            acc, bn, count = ref
            sfx = "" if first else "1"
            setattr(o, "stat_acc" + sfx, acc)
            setattr(o, "gamma" + sfx, bn.weight.data_ptr())
            setattr(o, "beta" + sfx, bn.bias.data_ptr())
            setattr(o, "running_mean" + sfx, bn.running_mean.data_ptr())
            setattr(o, "running_var" + sfx, bn.running_var.data_ptr())
            o.count, o.momentum, o.eps = count, BN_MOMENTUM, BN_EPS

        def fin_op(bn, c, count, tiles_m):
            s, t = new_scale_shift(c)
            if atomic_stats:
                acc = alloc((G, 2, 2, c), torch.int64, zero=True)
                self.stat_accs.append(acc)
                bnref[s.data_ptr()] = (acc.data_ptr(), bn, count)
                self.bn_list.append(bn)
                cv = ops[-1]                       # the conv that produces this BN's input
                assert cv.kind == L.OP_CONV and cv.Cout == c
                if tiles_m <= ATOMIC_MAX_TILES:
                    cv.stat_partial = None
                    cv.stat_acc = acc.data_ptr()
                    return None, s, t
                # very many M-tiles (the stem, layer 1): the conv keeps writing per-tile slabs (no contended atomics), a wide
                # reducer launch folds them into the same integer accumulators, and the consumer derives (scale, shift) as above
                o = L.SatOp()
                o.kind, o.dtype, o.groups = L.OP_BN_FINALIZE, dtype, G
                o.stat_partial, o.stat_acc = self.partial.data_ptr(), acc.data_ptr()
                o.Cout, o.tiles_m, o.training = c, tiles_m, 1
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
            o.stat_partial = self.partial.data_ptr()
            o.gamma, o.beta = bn.weight.data_ptr(), bn.bias.data_ptr()
            o.running_mean, o.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
            o.scale_out, o.shift_out = s.data_ptr(), t.data_ptr()
            o.Cout, o.count, o.tiles_m, o.training = c, count, tiles_m, 1
            o.momentum, o.eps = BN_MOMENTUM, BN_EPS
            self.bn_list.append(bn)
            return o, s, t

        def act_op(kind, x, s, t, out, n, h_, w__, c, x1=None, s1=None, t1=None):
            o = L.SatOp()
            o.kind, o.dtype, o.groups = kind, dtype, G
            o.in0, o.out = x.data_ptr(), out.data_ptr()
            ref = bnref.get(s.data_ptr())
            if ref is None:
                o.scale0, o.shift0 = s.data_ptr(), t.data_ptr()
            else:                                  # derive (scale, shift) from the conv's integer sums
                bn_fields(o, ref)
            if x1 is not None:
                o.in1 = x1.data_ptr()
                if s1 is not None:
                    ref1 = bnref.get(s1.data_ptr())
                    if ref1 is None:
                        o.scale1, o.shift1 = s1.data_ptr(), t1.data_ptr()
                    else:
                        bn_fields(o, ref1, first=False)
            o.N, o.Hout, o.Wout, o.Cout = n, h_, w__, c
            return o

        def std_conv(conv, x, out, n, hin, win, hout, wout):
            wt = prep_w(conv).reshape(conv.cout, -1)
            self.keep.append(wt)
            cin = conv.cin
            o = conv_op(x, wt, out, n, hin, win, cin, hout, wout, conv.cout, conv.k, conv.k, conv.stride, conv.pad,
                        hin * win * cin, win * cin, cin)
            pw_geom = conv.k == 3 and conv.stride == 1 and conv.pad == 1 and win <= 31
            aw_geom = conv.k == 1 and conv.pad == 0
            if dtype == L.SAT_BF16 and cin % 64 == 0 and conv.cout % 128 == 0 and (pw_geom or aw_geom):
                # frozen weights: a second copy in MFMA fragment order lets the tuner pick conv_pw_kernel (3x3) / conv_aw_kernel (1x1):
                # weights straight into registers, two workgroups per CU
                wp = torch.empty_like(wt)
                L.check(lib.sat_conv_pack_weights(wt.data_ptr(), wp.data_ptr(), conv.cout, cin, conv.k * conv.k, L.stream()),
                        "sat_conv_pack_weights")
                self.keep.append(wp)
                o.w_packed = wp.data_ptr()
            return o

        def fused_input_bn(cv, s, t):
            """the conv reads the RAW output of its producer and applies that BatchNorm + ReLU to its staged operand in LDS"""
            ref = bnref.get(s.data_ptr())
            if ref is None:
                cv.scale0, cv.shift0 = s.data_ptr(), t.data_ptr()
            else:
                bn_fields(cv, ref, first=False)
            return cv

        def add(o):
            if o is not None:
                ops.append(o)

        # ---- program ----
        # image prep reads the caller's tensors: one launch per group (G separate image batches), each into its slice
        self._n_prep = self.groups
        for g in range(self.groups):
            o = L.SatOp()
            o.kind, o.dtype = L.OP_IMAGE_PREP, dtype
            o.out = self.img_pad[g * Nb:].data_ptr()
            o.N, o.Hin, o.Win, o.Hout, o.Wout, o.pad = Nb, H, W, Hp, Wp, 3
            ops.append(o)
        ops.append(conv_op(self.img_pad, wst, self.c0, N, Hp, Wp, 32, Ho, Wo, width, 7, 1, 2, 0, Hp * Wp * 4, Wp * 4, 4))
        f, s, t = fin_op(stack.bn1, width, N * Ho * Wo, lib.sat_conv_tiles_m(N * Ho * Wo))
        add(f)
        y, ynext = self.ybuf
        mp = L.SatOp()
        mp.kind, mp.dtype, mp.groups = L.OP_BN_RELU_MAXPOOL, dtype, G
        mp.in0, mp.out = self.c0.data_ptr(), y.data_ptr()
        ref = bnref.get(s.data_ptr())
        if ref is None:
            mp.scale0, mp.shift0 = s.data_ptr(), t.data_ptr()
        else:                                      # the pooling kernel derives (scale, shift) from the integer sums
            bn_fields(mp, ref)
        mp.N, mp.Hin, mp.Win, mp.Cout, mp.Hout, mp.Wout = N, Ho, Wo, width, hp_, wp_
        ops.append(mp)
        # IN-PLACE BatchNorm-apply passes (bf16 training): relu(bn1(c1)) overwrites c1, and conv3 writes its raw output straight into
        # the next block-output buffer, which the normalise+add pass then transforms in place -- a bottleneck touches two large
        # buffers instead of three, so a stack's live set in layer 3 drops from ~90 to ~65 MB (+0.7 % under look-ahead)
        inplace = training and dtype == L.SAT_BF16
        # bf16 training, OPT-IN (SAT_DEFER_BN3=1): bn3 + residual add + ReLU of bottleneck k DEFERRED into conv1 of bottleneck k + 1
        # (conv_ay_kernel, SAT_CONV_IN_RESIDUAL): that conv builds its operand relu(bn3(c3_k) + y_{k-1}) on the way to LDS and writes
        # y_k out as it goes -- the normalise + add launch and conv1's re-read of the tensor it wrote disappear (bit-identical
        # results, tests/test_gpu_conv_ay.py).  Eligible: a bottleneck without projection followed by another one in the same layer
        # whose conv1 the kernel runs (planes a multiple of 128: layers 2-4), and not already fused through the Gram statistics.
        # MEASURED (round 5, interleaved on one box, profiles/r05_defer_ab.txt): per layer-3 bottleneck 24 + 38 us -> 39.5 us in
        # sequence (the pass's bytes now stream at the HBM rate under conv1's MFMAs), 8 % fewer bytes per pass, the encoder pipeline
        # ALONE 2.96 -> 2.87 ms per batch -- and the training step unchanged to 1 % slower (3.41 -> 3.42-3.46 ms): under the
        # look-ahead the light normalise + add launches already ran beside the other stacks' convs for free, while the fused conv1
        # holds a conv workgroup's registers and LDS for 15 us longer -- what the step pays for is conv workgroup-time.  Hence opt-in.
        # SAT_DEFER_INPLACE=1: y_k overwrites the raw conv3 tensor (two large buffers per bottleneck instead of three; only where
        # one column tile covers conv1's Cout)
        defer_bn3 = training and dtype == L.SAT_BF16 and os.environ.get("SAT_DEFER_BN3", "0") == "1"
        defer_inplace = os.environ.get("SAT_DEFER_INPLACE", "0") == "1"
        self.deferred_blocks = 0
        pending = None                                   # (raw conv3 tensor, bn3 scale / shift handles) of the bottleneck in front
        blocks_geo = list(zip(stack.blocks(), geo))
        for bi, (blk, (h, w_, h2, w2, inpl, planes, stride)) in enumerate(blocks_geo):
            tm1 = lib.sat_conv_tiles_m(N * h * w_)
            tm2 = lib.sat_conv_tiles_m(N * h2 * w2)
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
                # this conv1 also finishes the bottleneck in front: operand = relu(bn3(c3) + y), written to the other y buffer
                c3_prev, s3p, t3p = pending
                pending = None
                cv1 = fused_input_bn(std_conv(blk.conv1, c3_prev, self.c1, N, h, w_, h, w_), s3p, t3p)
                cv1.in1, cv1.out1, cv1.flags = y.data_ptr(), ynext.data_ptr(), cv1.flags | L.CONV_IN_RESIDUAL
                ops.append(cv1)
                y, ynext = ynext, y
                self.deferred_blocks += 1
            else:
                ops.append(std_conv(blk.conv1, y, self.c1, N, h, w_, h, w_))
            f, s1, t1 = fin_op(blk.bn1, planes, N * h * w_, tm1)
            add(f)
            pr_geom = (h2 == h and w2 == w_ and planes % 64 == 0 and 128 <= planes <= 512 and w_ <= 31)
            if fuse_bn1 and pr_geom:
                # conv2 (3x3) reads the RAW c1 and applies bn1 + ReLU to the LDS-resident patch (conv_pr_kernel; padded taps read a
                # row of zeros, so the zero padding stays zero): a1 never exists in HBM
                ops.append(fused_input_bn(std_conv(blk.conv2, self.c1, self.c2, N, h, w_, h2, w2), s1, t1))
            else:
                a1buf = self.c1 if inplace else self.a1
                ops.append(act_op(L.OP_BN_RELU, self.c1, s1, t1, a1buf, N, h, w_, planes))
                ops.append(std_conv(blk.conv2, a1buf, self.c2, N, h, w_, h2, w2))
            f, s2, t2 = fin_op(blk.bn2, planes, N * h2 * w2, tm2)
            add(f)
            c3buf = ynext if inplace else self.c3
            nxt = blocks_geo[bi + 1] if bi + 1 < len(blocks_geo) else None
            defer = (defer_bn3 and blk.downsample is None and nxt is not None and nxt[0].downsample is None and nxt[1][6] == 1 and
                     nxt[1][5] == planes and nxt[1][4] == planes * 4 and planes % 128 == 0 and planes * 4 <= 2048)
            if defer:
                # (in place only where ONE column tile of the consuming conv1 covers its Cout = planes: 128, or 256 with the
                # eight-wave variant -- the library refuses the aliasing otherwise)
                c3buf = ynext if (defer_inplace and planes <= 256) else self.c3
            if (gram_bn3 and blk.downsample is None and stride == 1 and inpl == planes * 4 and planes % 128 == 0 and planes <= gram_pmax
                    and bnref.get(s2.data_ptr()) is not None):
                # bn3's batch statistics from the Gram matrix of conv3's input, then conv3 with bn3 + residual + ReLU in its epilogue
                M3, P3, N3 = N * h2 * w2, planes, planes * 4
                cv3 = fused_input_bn(std_conv(blk.conv3, self.c2, ynext, N, h2, w2, h2, w2), s2, t2)
                gr = L.SatOp()
                gr.kind, gr.dtype, gr.groups = L.OP_GRAM, dtype, G
                gr.in0, gr.out = self.c2.data_ptr(), self.gram_slabs.data_ptr()
                gr.N, gr.Hout, gr.Wout, gr.Cout = N, h2, w2, P3
                acc2, bn2_, cnt2 = bnref[s2.data_ptr()]
                gr.stat_acc1, gr.gamma1, gr.beta1 = acc2, bn2_.weight.data_ptr(), bn2_.bias.data_ptr()
                gr.count, gr.eps = cnt2, BN_EPS
                co = L.SatOp()
                co.kind, co.dtype, co.groups = L.OP_GRAM_COV, dtype, G
                co.in0, co.out, co.scale_out = self.gram_slabs.data_ptr(), self.gram_cov3.data_ptr(), self.gram_mu.data_ptr()
                co.N, co.Hout, co.Wout, co.Cout = N, h2, w2, P3
                gm = L.SatOp()
                gm.kind, gm.dtype = L.OP_GEMM_BF16_NT, dtype
                gm.in0, gm.w, gm.out = self.gram_cov3.data_ptr(), cv3.w, self.gram_T.data_ptr()
                gm.N, gm.Hout, gm.Wout, gm.Cin, gm.Cout = G * 3 * P3, 1, 1, P3, N3
                tab = alloc((G, 2, N3), torch.float32)
                fb = L.SatOp()
                fb.kind, fb.dtype, fb.groups = L.OP_BN_FROM_GRAM, dtype, G
                fb.in0, fb.in1, fb.w, fb.scale_out = self.gram_T.data_ptr(), self.gram_mu.data_ptr(), cv3.w, tab.data_ptr()
                fb.gamma, fb.beta = blk.bn3.weight.data_ptr(), blk.bn3.bias.data_ptr()
                fb.running_mean, fb.running_var = blk.bn3.running_mean.data_ptr(), blk.bn3.running_var.data_ptr()
                fb.Cin, fb.Cout, fb.count, fb.momentum, fb.eps = P3, N3, M3, BN_MOMENTUM, BN_EPS
                cv3.scale1, cv3.shift1 = tab[0, 0].data_ptr(), tab[0, 1].data_ptr()
                cv3.in1 = y.data_ptr()
                cv3.flags = 1 | L.CONV_GROUP_TABLE
                cv3.stat_partial, cv3.stat_acc = None, None       # no statistics of its own: bn3's came from its input
                new_scale_shift(N3)                               # (keeps the BatchNorm numbering of the three-launch form)
                self.bn_list.append(blk.bn3)
                ops.extend([gr, co, gm, fb, cv3])
                self.gram_blocks += 1
                y, ynext = ynext, y
                continue
            if defer and bnref.get(s2.data_ptr()) is None:
                defer = False
            if fuse_in_bn and planes <= 512 and planes % 64 == 0:
                # conv3 reads the RAW c2 and applies bn2 + ReLU to its A operand in LDS: a2 never exists in HBM
                ops.append(fused_input_bn(std_conv(blk.conv3, self.c2, c3buf, N, h2, w2, h2, w2), s2, t2))
            else:
                a2buf = self.c2 if inplace else self.a2
                ops.append(act_op(L.OP_BN_RELU, self.c2, s2, t2, a2buf, N, h2, w2, planes))
                ops.append(std_conv(blk.conv3, a2buf, c3buf, N, h2, w2, h2, w2))
            f, s3, t3 = fin_op(blk.bn3, planes * 4, N * h2 * w2, tm2)
            add(f)
            if defer and bnref.get(s3.data_ptr()) is not None:
                pending = (c3buf, s3, t3)                # y stays y_{k-1}: the next conv1 adds it and writes y_k into ynext
                continue
            if blk.downsample is not None:
                ops.append(std_conv(blk.downsample[0], y, self.cd, N, h, w_, h2, w2))
                f, sd, td_ = fin_op(blk.downsample[1], planes * 4, N * h2 * w2, tm2)
                add(f)
                ops.append(act_op(L.OP_BN_ADD_RELU, c3buf, s3, t3, ynext, N, h2, w2, planes * 4, self.cd, sd, td_))
            else:
                ops.append(act_op(L.OP_BN_ADD_RELU, c3buf, s3, t3, ynext, N, h2, w2, planes * 4, y))
            y, ynext = ynext, y
        ap = L.SatOp()
        ap.kind, ap.dtype = L.OP_AVGPOOL, dtype
        ap.in0, ap.out = y.data_ptr(), self.pooled.data_ptr()
        ap.N, ap.Hin, ap.Win, ap.Cout = G * N, geo[-1][2], geo[-1][3], stack.feature_dim       # per image: groups concatenate
        ops.append(ap)
        self.final_map = (y, G * N, geo[-1][2], geo[-1][3], stack.feature_dim)
        if eval_items:
            arr = (L.SatBnEvalItem * len(eval_items))(*eval_items)
            self.eval_table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
            self.keep.append(self.eval_table)
            o = L.SatOp()
            o.kind, o.dtype = L.OP_BN_EVAL_BATCH, dtype
            o.in0, o.count, o.eps = self.eval_table.data_ptr(), len(eval_items), BN_EPS
            ops.insert(self._n_prep, o)          # right after image prep, before the first consumer
        self.ops = (L.SatOp * len(ops))(*ops)
        self.n_ops = len(ops)
        # replay as a hipGraph (SAT_GRAPH=0: eager launches).  Per step parity: first run eager, then captured.
        self._use_graph = os.environ.get("SAT_GRAPH", "1") != "0" and torch.device(device).type == "cuda"
        self._runs, self._graphs = [0, 0], [None, None]
        self._running_items = None              # defer_running_stats(): number of redirected BatchNorms
        if self.groups > 1 and training:
            self.defer_running_stats()
        self._autotune(device, (self.c0, self.c1, self.a1, self.c2, self.a2, self.c3, self.cd, *self.ybuf), alloc)

    def _autotune(self, device, buffers, alloc):
        """Kernel selection per conv geometry (bf16).  Default: the COMMITTED table (`tune.py`, `tune/gfx950.json`: the BASELINE
        geometries, measured once) and, for a geometry it does not name, the library's geometry-only default -- no stopwatch, so
        every process, rank and box runs the same kernels and the same seed gives the same bits (round 4: a timing-based choice
        moved the first-forward CE by 1.2e-3 between two processes).  SAT_AUTOTUNE=1 times the geometries the table does not name
        on this program's own buffers (the tuner's three fastest per geometry, the final choice IN the program); SAT_TUNE_FILE=<json>
        saves / reloads those."""
        if self.dtype != L.SAT_BF16 or torch.device(device).type != "cuda":
            return
        want_of = lambda o: self._want_sigs.get(self._layer_key(o))
        missing = T.assign(self.ops, self.n_ops, want_of)
        if not missing:
            return
        if T.mode() not in ("time", "force"):
            T.defaults(self.ops, missing, want_of)
            return
        chosen = {i: int(self.ops[i].variant) for i in range(self.n_ops) if self.ops[i].kind == L.OP_CONV}
        for t in buffers:
            t.normal_()
        scratch = alloc((4096,), torch.float32)           # the tuner's neutral BatchNorm table lives in OUR memory
        topk = max(1, int(os.environ.get("SAT_TUNE_TOPK", "3")))
        cand = (C.c_int32 * (self.n_ops * topk))()
        L.check(L.load().sat_conv_autotune_topk(self.ops, self.n_ops, 5, scratch.data_ptr(), scratch.numel() * 4,
                                                L.stream(), topk, cand), "sat_conv_autotune")
        torch.cuda.synchronize()
        for i, v in chosen.items():
            if v > 0:
                self.ops[i].variant = v                   # (entries the table already had stay as loaded)
        if topk > 1:
            self._pick_in_program(cand, topk, {i for i, v in chosen.items() if v > 0}, device)
        T.save(self.ops, self.n_ops, want_of)

    def _pick_in_program(self, cand, topk, fixed, device):
        """The final choice among the tuner's `topk` fastest variants per conv geometry, made IN the program: a replayed launch finds
        its operand warm, the same launch in the program finds what the previous kernel just wrote (a layer-3 1x1 conv: 13 us
        replayed, 18 in the program), and the two rankings differ by a few microseconds either way.  Pass k runs the whole program
        with every geometry on its k-th candidate and takes each conv launch's own duration (`sat_run_ops_timed`); a geometry keeps
        the candidate with the smallest summed duration.  Leaves no trace: statistics accumulators, parity, running statistics are
        put back."""
        lib = L.load()
        classes = {}
        for i in range(self.n_ops):
            o = self.ops[i]
            if o.kind == L.OP_CONV and i not in fixed and cand[i * topk]:
                key = self._tune_key(o, self._want_sigs.get(self._layer_key(o)))
                if key in _PROGRAM_PICKS:             # decided earlier in this process: the same choice for every model (like the
                    o.variant = _PROGRAM_PICKS[key]   # library's own per-geometry cache), so two models of one shape agree bit for bit
                else:
                    classes.setdefault(key, []).append(i)
        lists = {key: [int(cand[ix[0] * topk + k]) for k in range(topk) if cand[ix[0] * topk + k]] for key, ix in classes.items()}
        depth = max([len(v) for v in lists.values()] or [1])
        if depth < 2:
            return
        Nb = self.N
        ims = [torch.randn(Nb, 3, self.H, self.W, device=device) for _ in range(self.groups)]
        bns = list(self.stack.bns()) if self.training else []
        saved = [(bn.running_mean.clone(), bn.running_var.clone()) for bn in bns]
        for g, im in enumerate(ims):
            self.ops[g].in0 = im.data_ptr()
        us = (C.c_float * self.n_ops)()
        total = {key: [0.0] * len(v) for key, v in lists.items()}
        try:
            for k in range(depth):
                for key, ix in classes.items():
                    v = lists[key][min(k, len(lists[key]) - 1)]
                    for i in ix:
                        self.ops[i].variant = v
                for rep in range(6):                          # parity pairs; the first pair warms up
                    L.check(lib.sat_run_ops_timed(self.ops, self.n_ops, rep & 1, L.stream(), us), "sat_run_ops_timed")
                    if rep >= 2:
                        for key, ix in classes.items():
                            if k < len(lists[key]):
                                total[key][k] += sum(us[i] for i in ix)
        finally:
            # the passes ran with real momentum on the model's running statistics (ungrouped programs defer theirs only after the
            # build): put them back whatever happened, before anybody else can read them
            torch.cuda.synchronize()
            for acc in self.stat_accs:
                acc.zero_()
            for bn, (m, v) in zip(bns, saved):
                bn.running_mean.copy_(m)
                bn.running_var.copy_(v)
            self._parity, self._runs = 0, [0, 0]
        verbose = os.environ.get("SAT_TUNE_VERBOSE") is not None
        for key, ix in classes.items():
            best = min(range(len(lists[key])), key=lambda k: total[key][k])
            if verbose:
                import sys
                print("tune in program %s: %s -> v%d" % (key, ", ".join("v%d %.1f us" % (lists[key][k], total[key][k] / 4 / len(ix))
                                                                            for k in range(len(lists[key]))), lists[key][best]), file=sys.stderr)
            for i in ix:
                self.ops[i].variant = lists[key][best]
            _PROGRAM_PICKS[key] = lists[key][best]

    def signatures(self):
        """{conv layer: signature of the variant this program runs}: the BatchNorm statistics signature (training: tile shape and
        summation order fix the bits of the statistics) or the output family (inference: only the K order matters).  Hand it to the
        other programs of the same model state (`signatures=`) and a batch gets bit-identical features from all of them."""
        out = {}
        if self.dtype != L.SAT_BF16:
            return out
        lib = L.load()
        for i in range(self.n_ops):
            o = self.ops[i]
            if o.kind == L.OP_CONV and int(o.variant) > 0:
                if o.stat_partial or o.stat_acc:
                    out[self._layer_key(o)] = int(lib.sat_conv_variant_signature(int(o.variant)))
                else:
                    out[self._layer_key(o)] = int(lib.sat_conv_variant_family(int(o.variant)))
        return out

    def _matches(self, variant, want):
        return T.matches(variant, want)

    _layer_key = staticmethod(T.layer_key)
    _geom_key = staticmethod(T.geom_key)
    _tune_key = staticmethod(T.tune_key)

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
        running statistics still advance in batch order (TrainStep.prefetch_encoder).  Call before the first run.
        Grouped programs: the log of a BatchNorm is [G][2][C] (what sat_op.groups expects) and every group has its own table."""
        if not self.training or self._running_items is not None:
            return
        if self._runs != [0, 0]:
            raise RuntimeError("defer_running_stats must precede the first run (the hipGraph captures the pointers)")
        dev = self.pooled.device
        G = self.groups
        by_ptr = {bn.running_mean.data_ptr(): bn for bn in self.stack.bns()}
        items, seen = [[] for _ in range(G)], set()
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
                c = bn.running_mean.numel()
                log = torch.zeros(G, 2, c, dtype=torch.float32, device=dev)
                self.keep.append(log)
                setattr(o, fm, log[0, 0].data_ptr())
                setattr(o, fv, log[0, 1].data_ptr())
                for g in range(G):
                    it = L.SatBnRunningItem()
                    it.running_mean, it.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
                    it.batch_mean, it.batch_var, it.C = log[g, 0].data_ptr(), log[g, 1].data_ptr(), c
                    items[g].append(it)
                hit = True
            if hit:
                o.momentum = 1.0           # running' = 0 * running + 1 * f32(batch statistic): the log holds the statistic itself
        self._running_tables = []
        for g in range(G):
            arr = (L.SatBnRunningItem * max(len(items[g]), 1))(*items[g])
            self._running_tables.append(torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev))
        self._running_items = len(items[0])

    def apply_running_stats(self, group=0):
        """Momentum update of the model's running statistics from the last run's batch statistics of `group` (deferred programs
        only), on the current stream; the caller has ordered that stream behind the run."""
        if self._running_items:
            L.check(L.load().sat_bn_running_apply(self._running_tables[group].data_ptr(), self._running_items, BN_MOMENTUM, L.stream()),
                    "sat_bn_running_apply")
            L.counter_add(self.stack._nbt_flat)

    def pooled_of(self, group=0):
        """pooled features f32 [N, feature_dim] of one group's batch (a view of the program's output buffer)"""
        return self.pooled[group * self.N:(group + 1) * self.N]

    def _images_list(self, images):
        ims = list(images) if isinstance(images, (list, tuple)) else [images]
        if len(ims) != self.groups:
            raise ValueError("this program runs %d image batch(es) per launch, got %d" % (self.groups, len(ims)))
        out = []
        for im in ims:
            L.require_gpu(im, "images")
            if im.dtype != torch.float32 or tuple(im.shape) != (self.N, 3, self.H, self.W):
                raise ValueError("images must be float32 [%d,3,%d,%d]" % (self.N, self.H, self.W))
            out.append(im.contiguous())
        return out

    def run(self, images):
        """images f32 [N,3,H,W] NCHW on the device (grouped program: a list of `groups` such batches) -> pooled f32
        [groups * N, feature_dim] (owned by the program; `pooled_of(g)` = one batch's rows)."""
        ims = self._images_list(images)
        lib, p, npre = L.load(), self._parity, self._n_prep
        for g, im in enumerate(ims):
            self.ops[g].in0 = im.data_ptr()
        if not self._use_graph or self._runs[p] == 0:
            L.check(lib.sat_run_ops_parity(self.ops, self.n_ops, p, L.stream()), "sat_run_ops")
        else:
            # image prep reads the caller's tensors (new pointers every batch) -> eager; everything after it only
            # touches the program's own buffers -> one hipGraph per step parity, captured on this parity's 2nd run
            if self._graphs[p] is None:
                tail = (L.SatOp * (self.n_ops - npre))(*list(self.ops)[npre:])
                g = C.c_void_p()
                L.check(lib.sat_graph_create(tail, self.n_ops - npre, p, C.byref(g)), "sat_graph_create")
                self._graphs[p] = g
            L.check(lib.sat_run_ops_parity(self.ops, npre, p, L.stream()), "sat_run_ops")
            L.check(lib.sat_graph_launch(self._graphs[p], L.stream()), "sat_graph_launch")
        self._runs[p] += 1
        self._parity ^= 1
        if self.training and self._running_items is None:
            L.counter_add(self.stack._nbt_flat)
        return self.pooled


def _run_timed(self, images):
    """Diagnostics (bench.py's roofline figure): one eager, in-order run of the whole program -- same kernels, same
    statistics / parity bookkeeping as `run` -- that also returns every conv launch's own duration in microseconds
    (dispatch timestamps via `sat_run_ops_timed`).  Synchronises the stream."""
    ims = self._images_list(images)
    lib, p = L.load(), self._parity
    for g, im in enumerate(ims):
        self.ops[g].in0 = im.data_ptr()
    us = (C.c_float * self.n_ops)()
    L.check(lib.sat_run_ops_timed(self.ops, self.n_ops, p, L.stream(), us), "sat_run_ops_timed")
    self._runs[p] += 1
    self._parity ^= 1
    if self.training and self._running_items is None:
        L.counter_add(self.stack._nbt_flat)
    return self.pooled, [float(us[i]) for i in range(self.n_ops) if self.ops[i].kind == L.OP_CONV]


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
