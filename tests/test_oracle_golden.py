"""CPU: the oracle (oracle/decoder.py, oracle/train_step.py) against the golden vectors produced by the
imported reference `models.DecoderRNN` + train.py arithmetic (tests/golden/make_goldens.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import decoder as OD
from oracle import train_step as OT


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    return {k: z[k] for k in z.files}


def params_from_seed(g):
    E, H, V, L, B, T = [int(x) for x in g["dims"]]
    gen = torch.Generator().manual_seed(int(g["seed"]))
    return OD.init_decoder_params(E, H, V, L, generator=gen), (E, H, V, L, B, T)


@pytest.mark.parametrize("name", ["G1_dec_fwd_bwd_small.npz", "G2_dec_varlen_small.npz", "G5_dec_L2.npz"])
def test_forward_backward_matches_reference(golden_dir, name):
    g = load(golden_dir, name)
    params, (E, H, V, L, B, T) = params_from_seed(g)
    feats = torch.from_numpy(g["features"])
    caps = torch.from_numpy(g["captions"])
    lengths = [int(x) for x in g["lengths"]]
    targets, l1 = OT.pack_targets(caps, lengths)
    assert np.array_equal(targets.numpy(), g["targets"])            # packed time-major row order
    loss, grads, d_feat, logits = OT.decoder_loss_and_grads(params, feats, caps, lengths, L)
    np.testing.assert_allclose(logits.numpy(), g["logits"], rtol=0, atol=2e-6)
    assert abs(loss.item() - float(g["loss"])) < 2e-6
    np.testing.assert_allclose(d_feat.numpy(), g["d_features"], rtol=1e-4, atol=2e-8)
    for k in grads:
        np.testing.assert_allclose(grads[k].numpy(), g["grad." + k], rtol=1e-4, atol=2e-8, err_msg=k)


def test_clamp_adam_three_steps(golden_dir):
    g = load(golden_dir, "G1_dec_fwd_bwd_small.npz")
    params, (E, H, V, L, B, T) = params_from_seed(g)
    feats, caps = torch.from_numpy(g["features"]), torch.from_numpy(g["captions"])
    lengths = [int(x) for x in g["lengths"]]
    state = {}
    for it in range(3):
        loss, grads, _, _ = OT.decoder_loss_and_grads(params, feats, caps, lengths, L)
        assert abs(loss.item() - float(g["losses"][it])) < 5e-6
        OT.clamp_(grads, 0.1)
        OT.adam_step_(params, grads, state, lr=1e-3)
        if it + 1 in (1, 3):
            for k in params:
                np.testing.assert_allclose(params[k].numpy(), g["param_after%d.%s" % (it + 1, k)],
                                           rtol=0, atol=3e-7, err_msg="%s after %d" % (k, it + 1))


def test_cfg1_summary(golden_dir):
    g = load(golden_dir, "G3_dec_cfg1_summary.npz")
    params, (E, H, V, L, B, T) = params_from_seed(g)
    feats, caps = torch.from_numpy(g["features"]), torch.from_numpy(g["captions"])
    lengths = [int(x) for x in g["lengths"]]
    loss, grads, d_feat, logits = OT.decoder_loss_and_grads(params, feats, caps, lengths, L)
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    assert np.array_equal(logits.argmax(1).numpy(), g["argmax"])
    np.testing.assert_allclose(logits[:, :64].numpy(), g["logits_head"], rtol=0, atol=2e-6)
    for k in grads:
        n = float(np.sqrt((grads[k].double().numpy() ** 2).sum()))
        assert abs(n - float(g["gradnorm." + k])) <= 1e-5 * float(g["gradnorm." + k]) + 1e-9, k


@pytest.mark.parametrize("name", ["G1_dec_fwd_bwd_small.npz", "G3_dec_cfg1_summary.npz", "G5_dec_L2.npz"])
def test_greedy_ids_bit_exact(golden_dir, name):
    g = load(golden_dir, name)
    params, (E, H, V, L, B, T) = params_from_seed(g)
    ids = OD.greedy_sample(params, torch.from_numpy(g["features"]), L)
    assert ids.shape == (B, 20) and ids.dtype == torch.int64
    assert np.array_equal(ids.numpy(), g["greedy_ids"])


@pytest.mark.parametrize("name", ["G1_dec_fwd_bwd_small.npz", "G3_dec_cfg1_summary.npz", "G5_dec_L2.npz"])
def test_beam_width_one_is_the_golden_greedy_decode(golden_dir, name):
    """the beam restatement has no reference counterpart (stub), but beam_size=1 must be the pinned greedy ids"""
    g = load(golden_dir, name)
    params, (E, H, V, L, B, T) = params_from_seed(g)
    ids, scores = OD.beam_search(params, torch.from_numpy(g["features"]), 1, L)
    assert np.array_equal(ids[:, 0].numpy(), g["greedy_ids"])
    assert torch.isfinite(scores).all()


def test_beam_search_properties(golden_dir):
    g = load(golden_dir, "G5_dec_L2.npz")
    params, (E, H, V, L, B, T) = params_from_seed(g)
    feats = torch.from_numpy(g["features"])
    ids, scores = OD.beam_search(params, feats, 4, L)
    assert ids.shape == (B, 4, 20) and scores.shape == (B, 4)
    assert (scores[:, :-1] >= scores[:, 1:]).all()                      # best first
    assert len({tuple(r.tolist()) for r in ids[0]}) == 4               # hypotheses are distinct
    # the reported score is the sum of the token log-probabilities of the returned sequence (teacher forcing)
    b, k = 1, 2
    caps = torch.cat([ids[b, k, :19]]).unsqueeze(0)
    logits = OD.decoder_forward(params, feats[b:b + 1], caps, [20], L)
    lp = torch.log_softmax(logits, 1)
    total = sum(lp[t, ids[b, k, t]].item() for t in range(20))
    assert abs(total - scores[b, k].item()) < 1e-4
    # end_id: once emitted, a hypothesis only repeats it, at unchanged score
    end = int(ids[0, 0, 3])
    ids_e, scores_e = OD.beam_search(params, feats, 4, L, end_id=end)
    for b in range(B):
        for k in range(4):
            seq = ids_e[b, k].tolist()
            if end in seq:
                assert all(t == end for t in seq[seq.index(end):])
    assert (scores_e[:, :-1] >= scores_e[:, 1:]).all()


def test_lr_schedule():
    # train.py:101-107 with config.py defaults (decay_start 1, every 3, rate 0.8)
    assert OT.lr_for_epoch(1) == 1e-3
    assert OT.lr_for_epoch(2) == 1e-3
    assert abs(OT.lr_for_epoch(4) - 0.8e-3) < 1e-12
    assert abs(OT.lr_for_epoch(7) - 0.64e-3) < 1e-12


def test_batch_sizes_and_errors():
    assert OD.batch_sizes([5, 3, 3, 1]) == [4, 3, 3, 1, 1]
    with pytest.raises(AssertionError):
        OD.batch_sizes([3, 5])


def test_validation_forward_unshifted_captions_and_end_truncation(golden_dir):
    """eval.py:91-109 (G8, from the imported reference decoder): CE of `model(images, captions, lengths)` against
    `pack(captions, lengths)` -- captions unshifted, lengths full -- greedy ids, and the '<end>' truncation counts"""
    g = load(golden_dir, "G8_dec_eval_unshifted.npz")
    params, (E, H, V, L, B, T) = params_from_seed(g)
    feats, caps = torch.from_numpy(g["features"]), torch.from_numpy(g["captions"])
    lengths = [int(x) for x in g["lengths"]]
    loss, logits = OT.validation_loss(params, feats, caps, lengths, L)
    assert logits.shape[0] == sum(lengths)
    np.testing.assert_allclose(logits.numpy(), g["logits"], rtol=0, atol=2e-6)
    assert abs(loss.item() - float(g["loss"])) < 2e-6
    assert np.array_equal(OD.greedy_sample(params, feats, L).numpy(), g["greedy_ids"])
    assert OT.kept_tokens(torch.from_numpy(g["ids_planted"]), int(g["end_id"])) == [int(x) for x in g["kept_tokens"]]


@pytest.mark.parametrize("name", ["G9_dec_sample_states_L1.npz", "G9_dec_sample_states_L2.npz"])
def test_greedy_sample_from_a_nonzero_state(golden_dir, name):
    """models.py:56,61: `sample(features, states)` hands the state to nn.LSTM (G9: the reference decoder's own lstm / linear /
    embed driven from a seeded (h0, c0))"""
    g = load(golden_dir, name)
    params, (E, H, V, L, B, T) = params_from_seed(g)
    feats = torch.from_numpy(g["features"])
    states = (torch.from_numpy(g["h0"]), torch.from_numpy(g["c0"]))
    assert np.array_equal(OD.greedy_sample(params, feats, L, states=states).numpy(), g["greedy_ids"])
    assert np.array_equal(OD.greedy_sample(params, feats, L).numpy(), g["greedy_ids_zero_state"])
    assert not np.array_equal(g["greedy_ids"], g["greedy_ids_zero_state"])


@pytest.mark.skipif(not os.path.exists("/root/reference/models.py"), reason="the reference is only present in the build container")
def test_committed_goldens_regenerate_from_the_reference(golden_dir, tmp_path):
    """Every committed fixture is what the committed generator scripts produce from the imported reference TODAY: run both
    scripts into a scratch directory (child processes: they put the reference on sys.path) and compare array by array."""
    import subprocess
    import sys
    for script in ("make_goldens.py", "make_goldens_attend.py"):
        subprocess.check_call([sys.executable, os.path.join(golden_dir, script), str(tmp_path)], stdout=subprocess.DEVNULL)
    made = sorted(f for f in os.listdir(tmp_path) if f.endswith(".npz"))
    committed = sorted(f for f in os.listdir(golden_dir) if f.endswith(".npz"))
    assert made == committed                      # no fixture without a recipe, no recipe without its fixture
    for f in made:
        a, b = np.load(os.path.join(tmp_path, f)), np.load(os.path.join(golden_dir, f))
        assert sorted(a.files) == sorted(b.files), f
        for k in a.files:
            assert np.array_equal(a[k], b[k]), (f, k)
