#!/usr/bin/env python3
"""Generate tests/golden/G6_attend_*.npz from the REFERENCE's own `model2.ShowAttendTellModel` (imported from
/root/reference, build container only).

    python tests/golden/make_goldens_attend.py

`ShowAttendTellModel.__init__` downloads torchvision's VGG16 (model2.py:15) and `forward` unpacks a PackedSequence into
two names (model2.py:41, a torch-0.1 idiom that raises on torch 2.x), so neither can run here.  What CAN run, and does, is
every piece of arithmetic the class owns: an instance is built around `__init__` (the decoder-half submodules are created
with the very constructors of model2.py:19-36 and loaded with seeded weights), and the loop of model2.py:54-62 /
model2.py:98-109 is driven through the class's own `init_lstm`, `attention_layer`, `output_layer` methods and its
`embedding` / `lstmcell` modules on seeded encoder features.  Loss / backward are train.py:134-144's.
The reference never travels: only inputs + expected outputs are committed; weights regenerate from the seed
(`oracle.attend.init_attend_params`)."""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import attend as OA  # noqa: E402

REF = "/root/reference"


def import_model2():
    tv = sys.modules.get("torchvision") or types.ModuleType("torchvision")
    tvm = sys.modules.get("torchvision.models") or types.ModuleType("torchvision.models")
    tvm.vgg16 = None                       # model2.py:6 `from torchvision.models import vgg16` (only __init__ calls it)
    tv.models = tvm
    sys.modules["torchvision"], sys.modules["torchvision.models"] = tv, tvm
    sys.path.insert(0, REF)
    import model2
    return model2


def build(model2, hidden, context, vocab, embed, seed, feat=512):  # feat = feature_size[1]
    """the decoder half of ShowAttendTellModel.__init__ (model2.py:19-36), minus the VGG16 download"""
    m = model2.ShowAttendTellModel.__new__(model2.ShowAttendTellModel)
    nn.Module.__init__(m)
    m.opt = None
    m.image_att_w = nn.Parameter(torch.FloatTensor(feat, feat))
    m.init_hidden = nn.Linear(feat, hidden, bias=True)
    m.init_memory = nn.Linear(feat, hidden, bias=True)
    m.weight_hh = nn.Linear(hidden, context)
    m.weight_att = nn.Parameter(torch.FloatTensor(feat, 1))
    m.embedding = nn.Embedding(vocab, embed)
    m.lstmcell = nn.LSTMCell(hidden, hidden)
    m.context2out = nn.Linear(context, embed)
    m.hidden2tout = nn.Linear(hidden, embed)
    m.dropout = nn.Dropout(p=0.5)
    m.classifier = nn.Linear(embed, vocab)
    params = OA.init_attend_params(hidden, context, vocab, embed, generator=torch.Generator().manual_seed(seed), feat=feat)
    m.load_state_dict(params)
    return m, params


def reference_forward(m, features, captions, lengths):
    """model2.py:40-64 with `features` standing in for line 44's encoder call"""
    from torch.nn.utils.rnn import pack_padded_sequence
    embeddings = m.embedding(captions)
    batch_sizes = pack_padded_sequence(embeddings, lengths, batch_first=True).batch_sizes      # model2.py:41
    context_encode = torch.bmm(features, m.image_att_w.unsqueeze(0).expand(features.size(0), m.image_att_w.size(0), m.image_att_w.size(1)))
    hidden, c = m.init_lstm(features)
    outputs = []
    for t, batch_size in enumerate(batch_sizes.tolist()):
        embedding = embeddings[:batch_size, t, :]
        context, alpha = m.attention_layer(features[:batch_size], context_encode[:batch_size], hidden[:batch_size])
        rnn_input = torch.cat([embedding, context], dim=1)
        hidden, c = m.lstmcell(rnn_input, (hidden[:batch_size], c[:batch_size]))
        outputs.append(m.output_layer(context, hidden))
    return torch.cat(outputs, dim=0)


def reference_sample(m, features, states):
    """model2.py:93-111 with the encoder call replaced by `features`; torch-0.1 `max(1)` kept its dim (keepdim restatement)"""
    with torch.no_grad():
        embeddings = m.embedding(torch.ones(features.size(0)).long())
        sampled_ids = []
        context_encode = torch.bmm(features, m.image_att_w.unsqueeze(0).expand(features.size(0), m.image_att_w.size(0), m.image_att_w.size(1)))
        hidden, c = states
        for i in range(20):
            context, alpha = m.attention_layer(features, context_encode, hidden)
            if i == 0:
                rnn_input = torch.cat([embeddings, context], dim=1)
            hidden, c = m.lstmcell(rnn_input, (hidden, c))
            outputs = m.output_layer(context, hidden)
            predicted = outputs.max(1, keepdim=True)[1]
            sampled_ids.append(predicted)
            embedding = m.embedding(predicted).squeeze(1)
            rnn_input = torch.cat([embedding, context], dim=1)
        return torch.cat(sampled_ids, 1)


def make(model2, name, hidden, context, vocab, embed, B, T, P, lengths, seed, feat=512, summary=False):
    from torch.nn.utils.rnn import pack_padded_sequence
    import warnings
    warnings.simplefilter("ignore")
    m, params = build(model2, hidden, context, vocab, embed, seed, feat)
    g = torch.Generator().manual_seed(seed + 1)
    features = torch.randn(B, P, feat, generator=g).clamp(min=0)         # post-ReLU VGG features
    caps = torch.zeros(B, T, dtype=torch.long)
    for b, l in enumerate(lengths):
        caps[b, 0] = 1
        caps[b, 1:l - 1] = torch.randint(4, vocab, (l - 2,), generator=g)
        caps[b, l - 1] = 2
    l1 = [l - 1 for l in lengths]                                          # train.py:134
    targets = pack_padded_sequence(caps[:, 1:], l1, batch_first=True)[0]   # train.py:135
    m.zero_grad()
    logits = reference_forward(m, features, caps[:, :-1], l1)              # train.py:139
    loss = nn.CrossEntropyLoss()(logits, targets)                          # train.py:143
    loss.backward()                                                        # train.py:144
    out = dict(seed=seed, dims=np.array([hidden, context, vocab, embed, B, T, P, feat]), lengths=np.array(lengths),
               features=features.numpy(), captions=caps.numpy(), targets=targets.numpy(),
               logits=logits.detach().numpy(), loss=np.float32(loss.item()))
    for k, p in m.named_parameters():
        gk = p.grad.detach().clone()
        if summary and gk.numel() > 4096:         # full-size case: norms, sums and a strided sample keep the fixture small
            out["gradnorm." + k] = np.float64(gk.double().norm().item())
            out["gradsum." + k] = np.float64(gk.double().sum().item())
            out["gradsample." + k] = gk.flatten()[::max(1, gk.numel() // 512)][:512].numpy()
        else:
            out["grad." + k] = gk.numpy()
    if summary:
        out["argmax"] = out["logits"].argmax(1)
        out["logits"] = out["logits"][:, :64].copy()
        out["features_seed"] = np.int64(seed + 1)
    H = hidden
    out["sample_ids_zero_state"] = reference_sample(m, features, (torch.zeros(B, H), torch.zeros(B, H))).numpy()   # eval.py:82-83
    h0, c0 = m.init_lstm(features)
    out["sample_ids_init_state"] = reference_sample(m, features, (h0.detach(), c0.detach())).numpy()
    np.savez_compressed(os.path.join(OUT_DIR[0], name), **out)
    print(name, os.path.getsize(os.path.join(OUT_DIR[0], name)), "loss", loss.item())


OUT_DIR = [HERE]


def main(out_dir=HERE):
    OUT_DIR[0] = out_dir
    os.makedirs(out_dir, exist_ok=True)
    torch.set_num_threads(4)
    model2 = import_model2()
    # hidden = embed + 512 (the LSTMCell input is cat[embedding, context], model2.py:57-58); context = 512 (expand_as, model2.py:74)
    # (the class takes the feature width as a constructor argument, model2.py:11 `feature_size`: 64 keeps G6 small)
    make(model2, "G6_attend_small.npz", 96, 64, 300, 32, 4, 12, 16, [12, 12, 9, 5], 223, feat=64)
    make(model2, "G7_attend_vgg_dims.npz", 576, 512, 500, 64, 3, 10, 196, [10, 10, 7], 224, feat=512, summary=True)


if __name__ == "__main__":
    main(*sys.argv[1:2])
