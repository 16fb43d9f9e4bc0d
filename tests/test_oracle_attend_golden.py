"""CPU: the Show-Attend-Tell restatement (`oracle/attend.py`) against goldens produced by the reference class's own
methods (`tests/golden/make_goldens_attend.py`: model2.py's `init_lstm` / `attention_layer` / `output_layer` /
`lstmcell` / `embedding` driven through the loop of model2.py:54-62 and :98-109)."""
import os

import numpy as np
import pytest
import torch

from oracle import attend as OA


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    return {k: z[k] for k in z.files}


def setup(g):
    hidden, context, vocab, embed, B, T, P, feat = [int(x) for x in g["dims"]]
    params = OA.init_attend_params(hidden, context, vocab, embed, generator=torch.Generator().manual_seed(int(g["seed"])), feat=feat)
    return params, (hidden, context, vocab, embed, B, T, P, feat)


@pytest.mark.parametrize("name", ["G6_attend_small.npz", "G7_attend_vgg_dims.npz"])
def test_attend_forward_loss_grads_match_reference_methods(golden_dir, name):
    g = load(golden_dir, name)
    params, dims = setup(g)
    feats, caps = torch.from_numpy(g["features"]), torch.from_numpy(g["captions"])
    lengths = [int(x) for x in g["lengths"]]
    loss, grads, logits = OA.attend_loss_and_grads(params, feats, caps, lengths)
    assert abs(loss.item() - float(g["loss"])) < 2e-6
    if "argmax" in g:                    # summary fixture
        np.testing.assert_allclose(logits[:, :64].numpy(), g["logits"], rtol=0, atol=2e-6)
        assert np.array_equal(logits.argmax(1).numpy(), g["argmax"])
    else:
        np.testing.assert_allclose(logits.numpy(), g["logits"], rtol=0, atol=2e-6)
    for k in params:
        if "grad." + k in g:
            np.testing.assert_allclose(grads[k].numpy(), g["grad." + k], rtol=2e-4, atol=2e-8, err_msg=k)
        else:
            assert abs(grads[k].double().norm().item() - float(g["gradnorm." + k])) < 1e-4 * float(g["gradnorm." + k]) + 1e-9, k
            gk = grads[k].flatten()
            np.testing.assert_allclose(gk[::max(1, gk.numel() // 512)][:512].numpy(), g["gradsample." + k], rtol=2e-4, atol=2e-8, err_msg=k)


@pytest.mark.parametrize("name", ["G6_attend_small.npz", "G7_attend_vgg_dims.npz"])
def test_attend_greedy_sample_ids_bit_exact(golden_dir, name):
    g = load(golden_dir, name)
    params, (hidden, *_rest) = setup(g)
    feats = torch.from_numpy(g["features"])
    assert np.array_equal(OA.attend_sample(params, feats, None).numpy(), g["sample_ids_zero_state"])
    h0, c0 = OA.init_lstm(params, feats)
    assert np.array_equal(OA.attend_sample(params, feats, (h0, c0)).numpy(), g["sample_ids_init_state"])


def test_vgg_feature_stack_shape_and_indices():
    assert OA.vgg_conv_indices() == [0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26]       # torchvision vgg16.features[:-3]
    p = OA.init_vgg_params(torch.Generator().manual_seed(1), cfg=[8, "M", 16, "M", 16])
    f = OA.vgg_forward(p, torch.randn(2, 3, 16, 16), cfg=[8, "M", 16, "M", 16])
    assert f.shape == (2, 16, 16)                       # [B, (H/4)*(W/4), C]
    macs = 0
    c, hw = 3, 224 * 224
    for v in OA.VGG16_FEATURES:
        if v == "M":
            hw //= 4
        else:
            macs += hw * c * v * 9
            c = v
    assert abs(macs / 1e9 - 14.884) < 0.25             # 14.884 GMAC/img


@pytest.mark.parametrize("name", ["G6_attend_small.npz", "G7_attend_vgg_dims.npz"])
def test_attend_beam_width_one_is_the_pinned_greedy_decode(golden_dir, name):
    """the reference's `sample_beam` is a stub (model2.py:113-114): the oracle's beam search is pinned where it can be --
    width 1 reproduces the golden greedy ids of `sample` under both state conventions"""
    g = load(golden_dir, name)
    params, _ = setup(g)
    feats = torch.from_numpy(g["features"])
    ids, scores = OA.attend_beam_search(params, feats, beam_size=1)
    assert np.array_equal(ids[:, 0].numpy(), g["sample_ids_zero_state"]) and torch.isfinite(scores).all()
    h0, c0 = OA.init_lstm(params, feats)
    assert np.array_equal(OA.attend_beam_search(params, feats, 1, (h0, c0))[0][:, 0].numpy(), g["sample_ids_init_state"])
    ids3, sc3 = OA.attend_beam_search(params, feats, beam_size=3)
    assert (sc3[:, 0] >= scores[:, 0] - 1e-5).all()          # a wider beam never ends with a worse best hypothesis than greedy
    assert (sc3[:, :-1] >= sc3[:, 1:]).all()                 # best first
