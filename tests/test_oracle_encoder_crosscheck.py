"""The encoder oracle's WIRING against an independent implementation.  torchvision (the reference's `models.resnet152`,
`/root/reference/models.py:13`) is not installed here and the reference holds no encoder fixture, so `oracle/encoder.py` is
"parity unpinned" (DESIGN.md section 6).  `transformers.ResNetModel` IS installed: an independent implementation of the same
published network (HF's `microsoft/resnet-152` conversion loads torchvision's checkpoint into it), built from the same
torch-CPU ops.  With the oracle's seeded parameters copied over by name, both must produce the same stem, the same
2048-channel map and the same pooled features, in eval and in train mode (batch statistics + running-buffer updates): that
checks the stage layout [3,8,36,3], the v1.5 stride placement (stride on the 3x3 conv), the projection shortcuts, padding and
pooling -- everything the oracle could have restated wrongly.  It is a cross-check, not the reference's own fixture."""
import pytest
import torch

from oracle import encoder as OE

transformers = pytest.importorskip("transformers")


def _hf_resnet152():
    from transformers import ResNetConfig, ResNetModel
    cfg = ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[256, 512, 1024, 2048], depths=[3, 8, 36, 3],
                       layer_type="bottleneck", hidden_act="relu", downsample_in_first_stage=False, downsample_in_bottleneck=False)
    return ResNetModel(cfg)


def _copy_params(model, params, buffers):
    """oracle (torchvision) names -> transformers names"""
    sd = model.state_dict()

    def put(dst_conv, dst_bn, conv, bn):
        sd[dst_conv + ".weight"].copy_(params["resnet." + conv + ".weight"])
        sd[dst_bn + ".weight"].copy_(params["resnet." + bn + ".weight"])
        sd[dst_bn + ".bias"].copy_(params["resnet." + bn + ".bias"])
        sd[dst_bn + ".running_mean"].copy_(buffers["resnet." + bn + ".running_mean"])
        sd[dst_bn + ".running_var"].copy_(buffers["resnet." + bn + ".running_var"])

    put("embedder.embedder.convolution", "embedder.embedder.normalization", "conv1", "bn1")
    used = 2
    for s, nblocks in enumerate(OE.RESNET152["layers"]):
        for b in range(nblocks):
            src, dst = "layer%d.%d." % (s + 1, b), "encoder.stages.%d.layers.%d." % (s, b)
            for j in range(3):
                put(dst + "layer.%d.convolution" % j, dst + "layer.%d.normalization" % j, src + "conv%d" % (j + 1), src + "bn%d" % (j + 1))
                used += 2
            if b == 0:
                put(dst + "shortcut.convolution", dst + "shortcut.normalization", src + "downsample.0", src + "downsample.1")
                used += 2
    model.load_state_dict(sd)
    return used


def test_parameter_inventory_matches_the_independent_implementation():
    model = _hf_resnet152()
    n_hf = sum(p.numel() for p in model.parameters())
    params, _ = OE.init_encoder_params(256, generator=torch.Generator().manual_seed(1))
    n_or = sum(v.numel() for k, v in params.items() if k.startswith("resnet.") and not k.startswith("resnet.fc."))
    assert n_hf == n_or == 58143808                      # torchvision resnet152 minus its 1000-way fc
    assert len(OE.conv_specs()) == 155
    shapes_hf = sorted(tuple(p.shape) for n, p in model.named_parameters() if n.endswith("convolution.weight"))
    shapes_or = sorted(tuple(v.shape) for k, v in params.items() if v.dim() == 4)
    assert shapes_hf == shapes_or


@pytest.mark.parametrize("training", [False, True])
def test_oracle_resnet152_forward_equals_transformers_resnet(training):
    g = torch.Generator().manual_seed(7)
    params, buffers = OE.init_encoder_params(256, generator=g, randomize_bn=True)
    for k in buffers:                                    # non-trivial running statistics for the eval comparison
        if k.endswith("running_mean"):
            buffers[k].normal_(0, 0.05, generator=g)
        elif k.endswith("running_var"):
            buffers[k].uniform_(0.5, 1.5, generator=g)
    model = _hf_resnet152()
    assert _copy_params(model, params, buffers) == 2 * 155
    model.train(training)
    x = torch.randn(3, 3, 64, 64, generator=g)
    taps = {}
    b2 = {k: v.clone() for k, v in buffers.items()}
    with torch.no_grad():
        pooled, fmap = OE.resnet_forward(params, b2, x, training=training, taps=taps)
        out = model(x, output_hidden_states=True)
    hs = out.hidden_states                               # embedder output (after the max-pool), then one per stage
    assert hs[0].shape == taps["pool"].shape
    torch.testing.assert_close(hs[0], taps["pool"], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(hs[1], taps["layer1.2"], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(out.last_hidden_state, fmap, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(out.pooler_output.flatten(1), pooled, rtol=1e-3, atol=1e-4)
    if training:                                         # the running buffers moved the same way (momentum 0.1, unbiased variance)
        sd = model.state_dict()
        torch.testing.assert_close(sd["embedder.embedder.normalization.running_var"], b2["resnet.bn1.running_var"], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(sd["encoder.stages.2.layers.35.layer.2.normalization.running_mean"],
                                   b2["resnet.layer3.35.bn3.running_mean"], rtol=1e-4, atol=1e-5)
