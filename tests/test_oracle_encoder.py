"""CPU: self-consistency of the encoder oracle ("parity unpinned" by the reference, see oracle/__init__.py):
explicit head backward vs torch autograd, layer inventory vs the published ResNet-152 numbers."""
import torch

from oracle import encoder as OE

TINY = dict(layers=(1, 1, 1, 1), width=8)


def test_resnet152_inventory():
    specs = OE.conv_specs(OE.RESNET152)
    assert len(specs) == 155                                  # 1 stem + 50*3 + 4 downsample
    nparam = sum(co * ci * k * k + 2 * co for _, _, ci, co, k, _, _ in specs)
    assert nparam == 58_143_808                               # torchvision resnet152 minus fc (60,192,808 - 2,049,000)
    # MACs at 224x224 (11.512 GMAC/img incl. fc 2048x1000 in the published figure)
    h = 224
    macs, hw = 0, {}
    size = 224
    for name, _, ci, co, k, s, p in specs:
        if name == "conv1":
            out = (224 + 2 * p - k) // s + 1
            cur = out // 2          # after maxpool
            macs += out * out * ci * co * k * k
            hw["in"] = cur
            continue
        if name.endswith("conv1") or name.endswith("downsample.0"):
            i = hw["in"]
        elif name.endswith("conv2"):
            i = hw["in"]
        else:
            i = hw["mid"]
        out = (i + 2 * p - k) // s + 1
        macs += out * out * ci * co * k * k
        if name.endswith("conv2"):
            hw["mid"] = out
        if name.endswith("conv3"):
            hw["next"] = out
        if name.endswith("conv3") and not any(n == name.replace("conv3", "downsample.0") for n, *_ in specs):
            hw["in"] = hw["next"]
        if name.endswith("downsample.0"):
            hw["in"] = out
    assert abs(macs / 1e9 - 11.51) < 0.03


def test_head_backward_matches_autograd():
    g = torch.Generator().manual_seed(0)
    params, buffers = OE.init_encoder_params(16, TINY, generator=g, randomize_bn=True)
    pooled = torch.randn(6, OE.feature_dim(TINY), generator=g)
    dy = torch.randn(6, 16, generator=g)
    y, tape = OE.head_forward(params, {k: v.clone() for k, v in buffers.items()}, pooled)
    grads = OE.head_backward(params, tape, dy)
    w = params["resnet.fc.weight"].clone().requires_grad_(True)
    b = params["resnet.fc.bias"].clone().requires_grad_(True)
    ga = params["bn.weight"].clone().requires_grad_(True)
    be = params["bn.bias"].clone().requires_grad_(True)
    z = torch.nn.functional.linear(pooled, w, b)
    y2 = torch.nn.functional.batch_norm(z, torch.zeros(16), torch.ones(16), ga, be, True, 0.01, 1e-5)
    assert torch.allclose(y, y2, atol=1e-5)
    (y2 * dy).sum().backward()
    for k, t in (("resnet.fc.weight", w), ("resnet.fc.bias", b), ("bn.weight", ga), ("bn.bias", be)):
        assert torch.allclose(grads[k], t.grad, rtol=1e-3, atol=1e-5), k


def test_tiny_resnet_forward_shapes_and_running_stats():
    g = torch.Generator().manual_seed(1)
    params, buffers = OE.init_encoder_params(16, TINY, generator=g, randomize_bn=True)
    x = torch.randn(2, 3, 64, 64, generator=g)
    taps = {}
    pooled, fmap = OE.resnet_forward(params, buffers, x, TINY, True, taps)
    assert pooled.shape == (2, 256) and fmap.shape == (2, 256, 2, 2)
    assert taps["conv1_raw"].shape == (2, 8, 32, 32) and taps["pool"].shape == (2, 8, 16, 16)
    assert int(buffers["resnet.bn1.num_batches_tracked"]) == 1
    assert not torch.allclose(buffers["resnet.bn1.running_mean"], torch.zeros(8))
    # eval mode uses running stats and leaves them untouched
    before = buffers["resnet.bn1.running_mean"].clone()
    OE.resnet_forward(params, buffers, x, TINY, False)
    assert torch.equal(before, buffers["resnet.bn1.running_mean"])
