#!/usr/bin/env python3
"""Generate tests/golden/G*.npz by running the REFERENCE's own `models.DecoderRNN` (imported from
/root/reference, build container only) plus train.py's loss / clamp / Adam arithmetic.

    python tests/golden/make_goldens.py [out_dir]  # rewrites G1, G2, G3, G5, G8, G9 (default: tests/golden)

The reference never travels: only these data files (inputs + expected outputs) are committed.
`torchvision` is absent offline; `models.py:3` imports it at top level only for the encoder, so an empty
placeholder module is registered before the import.  Weights are produced by
`oracle.decoder.init_decoder_params` (reference init distributions, models.py:41-45) and loaded INTO the
reference module, so tests can regenerate them from the seed without the reference.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import decoder as OD  # noqa: E402

REF = "/root/reference"


def import_reference_models():
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tv.models = tvm
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.models", tvm)
    sys.path.insert(0, REF)
    import models  # the reference's models.py
    return models


def synth_batch(B, T, V, E, lengths, seed):
    """SURVEY 8d synthetic batch: col0=<start>=1, last valid col=<end>=2, pad 0 (collate_fn invariant)."""
    g = torch.Generator().manual_seed(seed)
    caps = torch.zeros(B, T, dtype=torch.long)
    for b, l in enumerate(lengths):
        caps[b, 0] = 1
        caps[b, 1:l - 1] = torch.randint(4, V, (l - 2,), generator=g)
        caps[b, l - 1] = 2
    feats = torch.randn(B, E, generator=g)
    return feats, caps


def run_reference(models, E, H, V, L, B, T, lengths, seed, n_adam=0, lr=1e-3, grad_clip=0.1):
    from torch.nn.utils.rnn import pack_padded_sequence
    g = torch.Generator().manual_seed(seed)
    params = OD.init_decoder_params(E, H, V, L, generator=g)
    dec = models.DecoderRNN(E, H, V, L)
    dec.load_state_dict(params)
    feats, caps = synth_batch(B, T, V, E, lengths, seed + 1)
    crit = torch.nn.CrossEntropyLoss()                                        # train.py:53
    opt = torch.optim.Adam([p for p in dec.parameters() if p.requires_grad], lr=lr)   # train.py:55-56
    out = dict(seed=seed, dims=np.array([E, H, V, L, B, T]), lengths=np.array(lengths),
               features=feats.numpy(), captions=caps.numpy())
    l1 = [l - 1 for l in lengths]                                             # train.py:134
    targets = pack_padded_sequence(caps[:, 1:], l1, batch_first=True)[0]      # train.py:135
    out["targets"] = targets.numpy()
    feats_v = feats.clone().requires_grad_(True)
    dec.zero_grad()                                                           # train.py:137
    logits = dec(feats_v, caps[:, :-1], l1)                                   # train.py:139
    loss = crit(logits, targets)                                              # train.py:143
    loss.backward()                                                           # train.py:144
    out["logits"] = logits.detach().numpy()
    out["loss"] = np.float32(loss.item())
    out["d_features"] = feats_v.grad.numpy()
    for k, p in dec.named_parameters():
        out["grad." + k] = p.grad.detach().clone().numpy()
    losses = [loss.item()]
    for it in range(n_adam):
        if it > 0:
            dec.zero_grad()
            loss = crit(dec(feats, caps[:, :-1], l1), targets)
            loss.backward()
            losses.append(loss.item())
        for group in opt.param_groups:                                        # train.py:88-91
            for p in group["params"]:
                p.grad.data.clamp_(-grad_clip, grad_clip)
        opt.step()                                                            # train.py:146
        if it + 1 in (1, 3):
            for k, p in dec.named_parameters():
                out["param_after%d." % (it + 1) + k] = p.detach().clone().numpy()
    out["losses"] = np.array(losses, dtype=np.float32)
    return dec, params, feats, out


def greedy_reference(dec, feats, states=None):
    """models.py:56-67 driven through the reference module's own submodules, keepdim restatement
    (as written it raises on torch 2.x at iteration 2: SURVEY 3.3).  `states`: what models.py:56 takes and :61 hands to
    nn.LSTM -- None or (h0, c0), each [num_layers, B, H]."""
    with torch.no_grad():
        ids, inputs = [], feats.unsqueeze(1)
        for _ in range(20):
            hiddens, states = dec.lstm(inputs, states)
            outputs = dec.linear(hiddens.squeeze(1))
            predicted = outputs.max(1, keepdim=True)[1]
            ids.append(predicted)
            inputs = dec.embed(predicted)
        return torch.cat(ids, 1)


def states_case(models, out_dir, name, seed, B, E, H, V, L):
    """`sample(features, states)` with a NON-ZERO initial state (models.py:56,61): the greedy ids of the reference decoder's own
    lstm / linear / embed from seeded (h0, c0)."""
    dec = models_reload(models, E, H, V, L, seed)
    g = torch.Generator().manual_seed(seed + 7)
    feats = torch.randn(B, E, generator=g)
    h0 = torch.randn(L, B, H, generator=g) * 0.5
    c0 = torch.randn(L, B, H, generator=g) * 0.5
    ids = greedy_reference(dec, feats, (h0, c0))
    ids_zero = greedy_reference(dec, feats, None)
    assert not torch.equal(ids, ids_zero), "the state must matter for this case to pin anything"
    out = dict(seed=seed, dims=np.array([E, H, V, L, B, 20]), features=feats.numpy(), h0=h0.numpy(), c0=c0.numpy(),
               greedy_ids=ids.numpy(), greedy_ids_zero_state=ids_zero.numpy())
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), **out)
    print(name, "rows differing from the zero-state ids:", int((ids != ids_zero).any(1).sum()), "of", B)


def eval_case(models, out_dir, name, seed, B, T, E, H, V, L, lengths, end_id=2):
    """The validation half of `evaluation` (eval.py:91-93) on the reference decoder: UNSHIFTED captions and FULL lengths --
    `targets = pack(captions, lengths)`, `outputs = model(images, captions, lengths)`, `loss = crit(outputs, targets)` --
    plus the greedy ids of `model.sample` (eval.py:99) and, per row, the number of tokens the id->word loop of
    eval.py:103-109 keeps (everything before the first `<end>`)."""
    from torch.nn.utils.rnn import pack_padded_sequence
    dec = models_reload(models, E, H, V, L, seed)
    feats, caps = synth_batch(B, T, V, E, lengths, seed + 1)
    with torch.no_grad():
        logits = dec(feats, caps, lengths)                      # eval.py:93: captions NOT shifted, lengths NOT decremented
        targets = pack_padded_sequence(caps, lengths, batch_first=True)[0]      # eval.py:91
        loss = torch.nn.CrossEntropyLoss()(logits, targets)     # eval.py:95
    ids = greedy_reference(dec, feats)
    # rows with <end> at chosen places (the truncation rule does not care how the ids were produced): first column, middle,
    # last column, never
    ids_planted = ids.clone()
    ids_planted[ids_planted == end_id] = end_id + 1
    ids_planted[0, 5] = end_id
    ids_planted[0, 9] = end_id
    ids_planted[1, 0] = end_id
    ids_planted[2, 19] = end_id
    keep = []
    for row in ids_planted.tolist():
        n = 0
        for w in row:                                           # eval.py:103-109: break at '<end>'
            if w == end_id:
                break
            n += 1
        keep.append(n)
    out = dict(seed=seed, dims=np.array([E, H, V, L, B, T]), lengths=np.array(lengths), features=feats.numpy(),
               captions=caps.numpy(), logits=logits.numpy(), targets=targets.numpy(), loss=np.float32(loss.item()),
               greedy_ids=ids.numpy(), ids_planted=ids_planted.numpy(), kept_tokens=np.array(keep), end_id=np.array(end_id))
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), **out)
    print(name, "loss", loss.item(), "kept", keep)


def main(out_dir=HERE):
    HERE = out_dir
    os.makedirs(out_dir, exist_ok=True)
    torch.set_num_threads(4)
    models = import_reference_models()
    # G1: small, equal lengths, fwd/bwd + 3 Adam steps
    dec, _, feats, g1 = run_reference(models, 32, 64, 500, 1, 4, 20, [20] * 4, 123, n_adam=3)
    g1["greedy_ids"] = greedy_reference(models_reload(models, 32, 64, 500, 1, 123), feats).numpy()
    np.savez_compressed(os.path.join(HERE, "G1_dec_fwd_bwd_small.npz"), **g1)
    # G2: variable lengths (packed row order / batch_sizes)
    _, _, _, g2 = run_reference(models, 32, 64, 500, 1, 4, 20, [20, 17, 12, 8], 124)
    np.savez_compressed(os.path.join(HERE, "G2_dec_varlen_small.npz"), **g2)
    # G3: cfg1 dims -- summary only (weights regenerate from the seed)
    dec3, _, feats3, g3 = run_reference(models, 256, 512, 10000, 1, 4, 20, [20] * 4, 123)
    s3 = dict(seed=g3["seed"], dims=g3["dims"], lengths=g3["lengths"], features=g3["features"],
              captions=g3["captions"], targets=g3["targets"], loss=g3["loss"],
              argmax=g3["logits"].argmax(1), logits_head=g3["logits"][:, :64].copy(),
              d_features=g3["d_features"])
    for k in list(g3):
        if k.startswith("grad."):
            s3["gradnorm." + k[5:]] = np.float64(np.sqrt((g3[k].astype(np.float64) ** 2).sum()))
            s3["gradsum." + k[5:]] = np.float64(g3[k].astype(np.float64).sum())
    s3["greedy_ids"] = greedy_reference(dec3, feats3).numpy()
    np.savez_compressed(os.path.join(HERE, "G3_dec_cfg1_summary.npz"), **s3)
    # G5: two layers
    dec5, _, feats5, g5 = run_reference(models, 32, 48, 300, 2, 4, 12, [12, 12, 9, 5], 125, n_adam=1)
    g5["greedy_ids"] = greedy_reference(models_reload(models, 32, 48, 300, 2, 125), feats5).numpy()
    np.savez_compressed(os.path.join(HERE, "G5_dec_L2.npz"), **g5)
    # G8: the validation forward of eval.py:91-109 (unshifted captions, full lengths)
    eval_case(models, HERE, "G8_dec_eval_unshifted", 131, 4, 12, 32, 64, 500, 1, [12, 9, 9, 5])
    # G9: sample() from a non-zero LSTM state, one and two layers
    states_case(models, HERE, "G9_dec_sample_states_L1", 141, 5, 32, 64, 500, 1)
    states_case(models, HERE, "G9_dec_sample_states_L2", 142, 3, 32, 48, 300, 2)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


def models_reload(models, E, H, V, L, seed):
    """Fresh reference module with the pre-update weights (run_reference's module has been stepped)."""
    g = torch.Generator().manual_seed(seed)
    dec = models.DecoderRNN(E, H, V, L)
    dec.load_state_dict(OD.init_decoder_params(E, H, V, L, generator=g))
    return dec


if __name__ == "__main__":
    main(*sys.argv[1:2])
