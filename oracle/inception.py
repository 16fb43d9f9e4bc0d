"""CPU restatement of an Inception-v3 encoder (TEST INFRASTRUCTURE -- see oracle/__init__.py).

BASELINE.json configs[3] names an "Inception-v3 encoder (299x299)"; the reference itself has none (its encoders
are torchvision ResNet-152, models.py:13, and VGG16, model2.py:15), so this is the published architecture (Szegedy et al.
2015, as laid out by torchvision.models.inception_v3 with aux_logits and transform_input off) dropped into
`EncoderCNN`'s slot (models.py:9-29): conv stack -> global average pool -> fc(2048 -> embed) -> BatchNorm1d.  Every
BasicConv2d = conv(bias=False) -> BatchNorm2d(eps=1e-3) -> ReLU; as in the reference, nothing calls .eval() during training,
so the BatchNorms use batch statistics.  "parity unpinned": seeded weights, no fixture anywhere.
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-3
BN_MOMENTUM = 0.1
FEATURE_DIM = 2048


def _bc(name, cin, cout, k, stride=1, pad=0):
    kh, kw = (k, k) if isinstance(k, int) else k
    ph, pw = (pad, pad) if isinstance(pad, int) else pad
    return (name, cin, cout, kh, kw, stride, ph, pw)


def block_a(p, cin, pf):
    return [_bc(p + "branch1x1", cin, 64, 1), _bc(p + "branch5x5_1", cin, 48, 1), _bc(p + "branch5x5_2", 48, 64, 5, 1, 2),
            _bc(p + "branch3x3dbl_1", cin, 64, 1), _bc(p + "branch3x3dbl_2", 64, 96, 3, 1, 1), _bc(p + "branch3x3dbl_3", 96, 96, 3, 1, 1),
            _bc(p + "branch_pool", cin, pf, 1)]


def block_b(p, cin):
    return [_bc(p + "branch3x3", cin, 384, 3, 2), _bc(p + "branch3x3dbl_1", cin, 64, 1), _bc(p + "branch3x3dbl_2", 64, 96, 3, 1, 1),
            _bc(p + "branch3x3dbl_3", 96, 96, 3, 2)]


def block_c(p, cin, c7):
    return [_bc(p + "branch1x1", cin, 192, 1), _bc(p + "branch7x7_1", cin, c7, 1), _bc(p + "branch7x7_2", c7, c7, (1, 7), 1, (0, 3)),
            _bc(p + "branch7x7_3", c7, 192, (7, 1), 1, (3, 0)), _bc(p + "branch7x7dbl_1", cin, c7, 1),
            _bc(p + "branch7x7dbl_2", c7, c7, (7, 1), 1, (3, 0)), _bc(p + "branch7x7dbl_3", c7, c7, (1, 7), 1, (0, 3)),
            _bc(p + "branch7x7dbl_4", c7, c7, (7, 1), 1, (3, 0)), _bc(p + "branch7x7dbl_5", c7, 192, (1, 7), 1, (0, 3)),
            _bc(p + "branch_pool", cin, 192, 1)]


def block_d(p, cin):
    return [_bc(p + "branch3x3_1", cin, 192, 1), _bc(p + "branch3x3_2", 192, 320, 3, 2), _bc(p + "branch7x7x3_1", cin, 192, 1),
            _bc(p + "branch7x7x3_2", 192, 192, (1, 7), 1, (0, 3)), _bc(p + "branch7x7x3_3", 192, 192, (7, 1), 1, (3, 0)),
            _bc(p + "branch7x7x3_4", 192, 192, 3, 2)]


def block_e(p, cin):
    return [_bc(p + "branch1x1", cin, 320, 1), _bc(p + "branch3x3_1", cin, 384, 1), _bc(p + "branch3x3_2a", 384, 384, (1, 3), 1, (0, 1)),
            _bc(p + "branch3x3_2b", 384, 384, (3, 1), 1, (1, 0)), _bc(p + "branch3x3dbl_1", cin, 448, 1),
            _bc(p + "branch3x3dbl_2", 448, 384, 3, 1, 1), _bc(p + "branch3x3dbl_3a", 384, 384, (1, 3), 1, (0, 1)),
            _bc(p + "branch3x3dbl_3b", 384, 384, (3, 1), 1, (1, 0)), _bc(p + "branch_pool", cin, 192, 1)]


BLOCKS = [("Mixed_5b", "A", 192, 32, 256), ("Mixed_5c", "A", 256, 64, 288), ("Mixed_5d", "A", 288, 64, 288),
          ("Mixed_6a", "B", 288, None, 768), ("Mixed_6b", "C", 768, 128, 768), ("Mixed_6c", "C", 768, 160, 768),
          ("Mixed_6d", "C", 768, 160, 768), ("Mixed_6e", "C", 768, 192, 768), ("Mixed_7a", "D", 768, None, 1280),
          ("Mixed_7b", "E", 1280, None, 2048), ("Mixed_7c", "E", 2048, None, 2048)]
STEM = [_bc("Conv2d_1a_3x3", 3, 32, 3, 2), _bc("Conv2d_2a_3x3", 32, 32, 3), _bc("Conv2d_2b_3x3", 32, 64, 3, 1, 1),
        _bc("Conv2d_3b_1x1", 64, 80, 1), _bc("Conv2d_4a_3x3", 80, 192, 3)]


def conv_specs():
    """every BasicConv2d in forward order: (name, cin, cout, kh, kw, stride, pad_h, pad_w)"""
    specs = list(STEM)
    for name, kind, cin, arg, _ in BLOCKS:
        p = name + "."
        specs += {"A": lambda: block_a(p, cin, arg), "B": lambda: block_b(p, cin), "C": lambda: block_c(p, cin, arg),
                  "D": lambda: block_d(p, cin), "E": lambda: block_e(p, cin)}[kind]()
    return specs


def conv_macs(H=299, W=299):
    """multiply-accumulates per image of the conv stack (for throughput figures)"""
    import torch as _t
    params, buffers = init_inception_params(8, generator=_t.Generator().manual_seed(0))
    tot = [0]

    def hook(name, x, w, stride, pad):
        ho = (x.shape[2] + 2 * pad[0] - w.shape[2]) // stride + 1
        wo = (x.shape[3] + 2 * pad[1] - w.shape[3]) // stride + 1
        tot[0] += ho * wo * w.shape[0] * w.shape[1] * w.shape[2] * w.shape[3]
    inception_forward(params, buffers, _t.zeros(1, 3, H, W), training=False, conv_hook=hook, shapes_only=True)
    return tot[0]


def init_inception_params(embed_size, generator=None, randomize_bn=False, prefix="resnet."):
    """(params, buffers) keyed as `EncoderCNN.state_dict()` has them when the stack sits in the encoder's slot
    (`resnet.<torchvision name>.conv.weight`, `.bn.*`, `resnet.fc.*`, `bn.*`)."""
    g = generator
    params, buffers = {}, {}
    for name, cin, cout, kh, kw, _, _, _ in conv_specs():
        std = math.sqrt(2.0 / (cin * kh * kw))
        params[prefix + name + ".conv.weight"] = torch.empty(cout, cin, kh, kw).normal_(0, std, generator=g)
        if randomize_bn:
            params[prefix + name + ".bn.weight"] = torch.empty(cout).uniform_(0.5, 1.5, generator=g)
            params[prefix + name + ".bn.bias"] = torch.empty(cout).normal_(0, 0.1, generator=g)
        else:
            params[prefix + name + ".bn.weight"] = torch.ones(cout)
            params[prefix + name + ".bn.bias"] = torch.zeros(cout)
        buffers[prefix + name + ".bn.running_mean"] = torch.zeros(cout)
        buffers[prefix + name + ".bn.running_var"] = torch.ones(cout)
        buffers[prefix + name + ".bn.num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    params[prefix + "fc.weight"] = torch.empty(embed_size, FEATURE_DIM).normal_(0.0, 0.02, generator=g)     # models.py:22
    params[prefix + "fc.bias"] = torch.zeros(embed_size)
    params["bn.weight"], params["bn.bias"] = torch.ones(embed_size), torch.zeros(embed_size)
    buffers["bn.running_mean"], buffers["bn.running_var"] = torch.zeros(embed_size), torch.ones(embed_size)
    buffers["bn.num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    return params, buffers


def inception_forward(params, buffers, images, training=True, prefix="resnet.", bf16_storage=False, conv_hook=None,
                      shapes_only=False):
    """images [B,3,H,W] -> pooled [B, 2048].  bf16_storage: round every stored tensor (image, weights, raw conv outputs,
    activations, pool outputs) to bfloat16 where the HIP bf16 path stores one; statistics from the un-rounded f32 conv output."""
    def r(t):
        return t.to(torch.bfloat16).to(torch.float32) if bf16_storage else t

    spec = {s[0]: s for s in conv_specs()}

    def bc(x, name):
        _, cin, cout, kh, kw, stride, ph, pw = spec[name]
        w = params[prefix + name + ".conv.weight"]
        if conv_hook is not None:
            conv_hook(name, x, w, stride, (ph, pw))
        if shapes_only:
            ho, wo = (x.shape[2] + 2 * ph - kh) // stride + 1, (x.shape[3] + 2 * pw - kw) // stride + 1
            return torch.zeros(x.shape[0], cout, ho, wo)
        c = F.conv2d(x, r(w), None, stride, (ph, pw))
        pre = prefix + name + ".bn"
        if training:
            mean, var = c.mean((0, 2, 3)), c.var((0, 2, 3), unbiased=False)
            n = c.numel() / c.shape[1]
            if not bf16_storage:
                buffers[pre + ".running_mean"].mul_(1 - BN_MOMENTUM).add_(mean * BN_MOMENTUM)
                buffers[pre + ".running_var"].mul_(1 - BN_MOMENTUM).add_(var * (n / max(n - 1, 1)) * BN_MOMENTUM)
                buffers[pre + ".num_batches_tracked"] += 1
        else:
            mean, var = buffers[pre + ".running_mean"], buffers[pre + ".running_var"]
        scale = params[pre + ".weight"] / torch.sqrt(var + BN_EPS)
        shift = params[pre + ".bias"] - mean * scale
        return r(F.relu(r(c) * scale[None, :, None, None] + shift[None, :, None, None]))

    x = r(images)
    x = bc(x, "Conv2d_1a_3x3")
    x = bc(x, "Conv2d_2a_3x3")
    x = bc(x, "Conv2d_2b_3x3")
    x = F.max_pool2d(x, 3, 2)
    x = bc(x, "Conv2d_3b_1x1")
    x = bc(x, "Conv2d_4a_3x3")
    x = F.max_pool2d(x, 3, 2)
    for name, kind, cin, arg, cout in BLOCKS:
        p = name + "."
        if kind == "A":
            b1 = bc(x, p + "branch1x1")
            b5 = bc(bc(x, p + "branch5x5_1"), p + "branch5x5_2")
            b3 = bc(bc(bc(x, p + "branch3x3dbl_1"), p + "branch3x3dbl_2"), p + "branch3x3dbl_3")
            bp = bc(r(F.avg_pool2d(x, 3, 1, 1)), p + "branch_pool")
            x = torch.cat([b1, b5, b3, bp], 1)
        elif kind == "B":
            b3 = bc(x, p + "branch3x3")
            bd = bc(bc(bc(x, p + "branch3x3dbl_1"), p + "branch3x3dbl_2"), p + "branch3x3dbl_3")
            x = torch.cat([b3, bd, F.max_pool2d(x, 3, 2)], 1)
        elif kind == "C":
            b1 = bc(x, p + "branch1x1")
            b7 = bc(bc(bc(x, p + "branch7x7_1"), p + "branch7x7_2"), p + "branch7x7_3")
            bd = x
            for i in range(1, 6):
                bd = bc(bd, p + "branch7x7dbl_%d" % i)
            bp = bc(r(F.avg_pool2d(x, 3, 1, 1)), p + "branch_pool")
            x = torch.cat([b1, b7, bd, bp], 1)
        elif kind == "D":
            b3 = bc(bc(x, p + "branch3x3_1"), p + "branch3x3_2")
            b7 = x
            for i in range(1, 5):
                b7 = bc(b7, p + "branch7x7x3_%d" % i)
            x = torch.cat([b3, b7, F.max_pool2d(x, 3, 2)], 1)
        else:
            b1 = bc(x, p + "branch1x1")
            t = bc(x, p + "branch3x3_1")
            b3 = torch.cat([bc(t, p + "branch3x3_2a"), bc(t, p + "branch3x3_2b")], 1)
            t = bc(bc(x, p + "branch3x3dbl_1"), p + "branch3x3dbl_2")
            bd = torch.cat([bc(t, p + "branch3x3dbl_3a"), bc(t, p + "branch3x3dbl_3b")], 1)
            bp = bc(r(F.avg_pool2d(x, 3, 1, 1)), p + "branch_pool")
            x = torch.cat([b1, b3, bd, bp], 1)
        assert x.shape[1] == cout, (name, x.shape)
    return x.mean((2, 3))
