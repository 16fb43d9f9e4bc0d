"""CPU restatement of the reference's Show-Attend-Tell model (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Follows `ShowAttendTellModel` `/root/reference/model2.py:9-111`, the model `train.py:37` constructs:
    encoder   = vgg16.features[:-3], frozen (model2.py:15-17)          -> [B,512,14,14] -> [B,196,512] (model2.py:44-45)
    context_encode = features @ image_att_w (model2.py:46)
    h, c      = init_hidden / init_memory of the mean feature (model2.py:49, 67-71)
    per step t over pack_padded_sequence's batch_sizes (model2.py:54-62):
        context, alpha = attention_layer(features[:bs], context_encode[:bs], h[:bs])   (model2.py:73-78)
        h, c = lstmcell(cat[embedding_t, context], (h[:bs], c[:bs]))                   (model2.py:57-58)
        output = classifier(context2out(context) + hidden2tout(h))                      (model2.py:59, 80-85)
    outputs = cat(outputs, 0): rows in time-major packed order (model2.py:64)
    sample (model2.py:91-111): 20 greedy steps from the `<start>` embedding (id 1) and the caller's states; as written,
    from the second step on the LSTM input carries the PREVIOUS step's context (model2.py:108) -- reproduced, not fixed.
`forward` as written cannot run on torch 2.x (model2.py:41 unpacks a 4-field PackedSequence into 2 names) and needs the
torchvision VGG16 download, so the goldens (tests/golden/G6) drive the reference class's OWN methods -- `init_lstm`,
`attention_layer`, `output_layer`, its `lstmcell` / `embedding` modules -- through the loop of model2.py:54-62 on seeded
features; this file must reproduce them.

Parity: decoder half PINNED by tests/golden/G6_attend_*.npz; the VGG16 encoder is "parity unpinned" (torchvision and
its weights are absent, the reference holds no fixture): published VGG16 configuration D, seeded weights.
"""
import math

import torch
import torch.nn.functional as F

VGG16_FEATURES = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512]   # features[:-3]: conv5_3, its ReLU and pool5 dropped
FEATURE_SIZE = (196, 512)


def vgg_conv_indices(cfg=VGG16_FEATURES):
    """nn.Sequential indices of the conv layers inside torchvision's `features` (conv, ReLU, [pool])."""
    idx, i = [], 0
    for v in cfg:
        if v == "M":
            i += 1
        else:
            idx.append(i)
            i += 2
    return idx


def init_vgg_params(generator=None, cfg=VGG16_FEATURES, cin=3):
    """`encoder.{i}.weight/bias` as ShowAttendTellModel.state_dict() names them; torchvision init (kaiming fan_out, bias 0;
    the bias is randomised slightly so a dropped bias shows in tests)."""
    params, c = {}, cin
    for i, v in zip(vgg_conv_indices(cfg), [v for v in cfg if v != "M"]):
        std = math.sqrt(2.0 / (v * 9))
        params["encoder.%d.weight" % i] = torch.empty(v, c, 3, 3).normal_(0, std, generator=generator)
        params["encoder.%d.bias" % i] = torch.empty(v).normal_(0, 0.05, generator=generator)
        c = v
    return params


def vgg_forward(params, images, cfg=VGG16_FEATURES, bf16_storage=False):
    """images [B,3,H,W] -> features [B, (H/16)*(W/16), C] (model2.py:44-45: view + transpose == NHWC flatten)."""
    def r(t):
        return t.to(torch.bfloat16).to(torch.float32) if bf16_storage else t
    x, i = r(images), 0
    for v in cfg:
        if v == "M":
            x = F.max_pool2d(x, 2, 2)
            i += 1
        else:
            x = r(F.relu(F.conv2d(x, r(params["encoder.%d.weight" % i]), params["encoder.%d.bias" % i], 1, 1)))
            i += 2
    B, C = x.shape[0], x.shape[1]
    return x.view(B, C, -1).transpose(2, 1).contiguous()


def init_attend_params(hidden_size, context_size, vocab_size, embed_size, generator=None, feat=FEATURE_SIZE[1]):
    """Decoder-half parameters with the reference's names (model2.py:19-36).  nn.Linear / nn.LSTMCell / nn.Embedding
    default inits; `image_att_w` and `weight_att` are uninitialised memory in the reference (model2.py:20,25): seeded
    N(0, 0.05) here."""
    g = generator

    def lin(out_f, in_f, name, p):
        k = 1.0 / math.sqrt(in_f)
        p[name + ".weight"] = torch.empty(out_f, in_f).uniform_(-k, k, generator=g)
        p[name + ".bias"] = torch.empty(out_f).uniform_(-k, k, generator=g)

    p = {}
    p["image_att_w"] = torch.empty(feat, feat).normal_(0, 0.05, generator=g)
    lin(hidden_size, feat, "init_hidden", p)
    lin(hidden_size, feat, "init_memory", p)
    lin(context_size, hidden_size, "weight_hh", p)
    p["weight_att"] = torch.empty(feat, 1).normal_(0, 0.05, generator=g)
    p["embedding.weight"] = torch.empty(vocab_size, embed_size).normal_(0, 1, generator=g)
    k = 1.0 / math.sqrt(hidden_size)
    for n, shape in (("weight_ih", (4 * hidden_size, hidden_size)), ("weight_hh", (4 * hidden_size, hidden_size)),
                     ("bias_ih", (4 * hidden_size,)), ("bias_hh", (4 * hidden_size,))):
        p["lstmcell." + n] = torch.empty(*shape).uniform_(-k, k, generator=g)
    lin(embed_size, context_size, "context2out", p)
    lin(embed_size, hidden_size, "hidden2tout", p)
    lin(vocab_size, embed_size, "classifier", p)
    return p


def batch_sizes(lengths):
    return [sum(1 for l in lengths if l > t) for t in range(int(lengths[0]))]


def attention_layer(p, features, context_encode, hidden):
    """model2.py:73-78."""
    proj = hidden @ p["weight_hh.weight"].t() + p["weight_hh.bias"]
    h_att = torch.tanh(context_encode + proj.unsqueeze(1))
    out_att = (h_att @ p["weight_att"]).squeeze(2)
    alpha = torch.softmax(out_att, dim=1)
    context = (features * alpha.unsqueeze(2)).mean(1)
    return context, alpha


def lstmcell(p, x, h, c):
    """nn.LSTMCell: gates i,f,g,o."""
    gates = x @ p["lstmcell.weight_ih"].t() + p["lstmcell.bias_ih"] + h @ p["lstmcell.weight_hh"].t() + p["lstmcell.bias_hh"]
    H = h.shape[1]
    i, f, g, o = gates[:, :H].sigmoid(), gates[:, H:2 * H].sigmoid(), gates[:, 2 * H:3 * H].tanh(), gates[:, 3 * H:].sigmoid()
    c2 = f * c + i * g
    return o * c2.tanh(), c2


def output_layer(p, context, hidden):
    """model2.py:80-85."""
    z = context @ p["context2out.weight"].t() + p["context2out.bias"] + hidden @ p["hidden2tout.weight"].t() + p["hidden2tout.bias"]
    return z @ p["classifier.weight"].t() + p["classifier.bias"]


def init_lstm(p, features):
    """model2.py:67-71."""
    m = features.mean(1)
    return m @ p["init_hidden.weight"].t() + p["init_hidden.bias"], m @ p["init_memory.weight"].t() + p["init_memory.bias"]


def attend_forward(p, features, captions, lengths):
    """model2.py:38-65 given the encoder features [B,P,C]: logits [sum(lengths), V], time-major packed."""
    emb = p["embedding.weight"][captions]
    context_encode = features @ p["image_att_w"]
    h, c = init_lstm(p, features)
    outs = []
    for t, bs in enumerate(batch_sizes(lengths)):
        context, _ = attention_layer(p, features[:bs], context_encode[:bs], h[:bs])
        h, c = lstmcell(p, torch.cat([emb[:bs, t], context], 1), h[:bs], c[:bs])
        outs.append(output_layer(p, context, h))
    return torch.cat(outs, 0)


def attend_sample(p, features, states=None, start_id=1, steps=20):
    """model2.py:91-111 (greedy; torch-0.1 keepdim semantics: [B,20] ids)."""
    B = features.shape[0]
    H = p["lstmcell.weight_hh"].shape[1]
    emb = p["embedding.weight"][torch.full((B,), start_id, dtype=torch.long)]
    context_encode = features @ p["image_att_w"]
    if states is None:
        h, c = torch.zeros(B, H), torch.zeros(B, H)         # eval.py:82-83 passes zeros
    else:
        h, c = states
    ids, rnn_input = [], None
    for i in range(steps):
        context, _ = attention_layer(p, features, context_encode, h)
        if i == 0:
            rnn_input = torch.cat([emb, context], 1)
        h, c = lstmcell(p, rnn_input, h, c)
        out = output_layer(p, context, h)
        pred = out.max(1)[1]
        ids.append(pred)
        rnn_input = torch.cat([p["embedding.weight"][pred], context], 1)      # model2.py:108: THIS step's context feeds the next LSTM input
    return torch.stack(ids, 1)


def attend_beam_search(p, features, beam_size=5, states=None, start_id=1, steps=20, end_id=None):
    """Beam decode for the Show-Attend-Tell model.  The reference has only a stub (`model2.py:113-114`: `def sample_beam: pass`),
    so this is the textbook algorithm laid over `sample`'s own loop (`model2.py:91-111`, including its habit of feeding THIS
    step's context into the NEXT step's LSTM input) -- PARITY UNPINNED by the reference, except that beam_size=1 with
    end_id=None must reproduce `attend_sample` (pinned by the G6/G7 greedy goldens).
    Candidate (k, v) scores score[k] + log_softmax(logits[k])[v]; the best K of the K*V per image survive (ties: lower k*V+v);
    h, c and the carried context follow their parent.  end_id: a finished hypothesis only repeats end_id at no cost.
    Returns (ids [B,K,steps] best-first, scores [B,K])."""
    B, K = features.shape[0], beam_size
    V = p["classifier.weight"].shape[0]
    H = p["lstmcell.weight_hh"].shape[1]
    rep = lambda t: t.repeat_interleave(K, 0)
    feats = rep(features)
    context_encode = feats @ p["image_att_w"]
    if states is None:
        h, c = torch.zeros(B * K, H), torch.zeros(B * K, H)
    else:
        h, c = rep(states[0]), rep(states[1])
    emb = p["embedding.weight"][torch.full((B * K,), start_id, dtype=torch.long)]
    scores = torch.full((B, K), float("-inf"))
    scores[:, 0] = 0.0
    seqs = torch.zeros(B, K, 0, dtype=torch.int64)
    last, rnn_input = None, None
    for i in range(steps):
        context, _ = attention_layer(p, feats, context_encode, h)
        if i == 0:
            rnn_input = torch.cat([emb, context], 1)
        h, c = lstmcell(p, rnn_input, h, c)
        logits = output_layer(p, context, h)
        logp = torch.log_softmax(logits, dim=1).view(B, K, V)
        cand = scores.unsqueeze(2) + logp
        if end_id is not None and last is not None:
            fin = last == end_id
            frozen = torch.full((B, K, V), float("-inf"))
            frozen[:, :, end_id] = scores
            cand = torch.where(fin.unsqueeze(2), frozen, cand)
        cand = cand.view(B, K * V)
        order = torch.sort(cand, dim=1, descending=True, stable=True)[1][:, :K]
        scores = torch.gather(cand, 1, order)
        parent, token = order // V, order % V
        rows = (torch.arange(B).unsqueeze(1) * K + parent).reshape(-1)
        h, c, context = h[rows], c[rows], context[rows]
        seqs = torch.cat([torch.gather(seqs, 1, parent.unsqueeze(2).expand(B, K, seqs.shape[2])), token.unsqueeze(2)], 2)
        last = token
        rnn_input = torch.cat([p["embedding.weight"][token.reshape(-1)], context], 1)     # model2.py:108
    return seqs, scores


def attend_loss_and_grads(p, features, captions, lengths, denom=None):
    """train.py:134-144 around the model: targets = pack(captions[:,1:], lengths-1), model(images, captions[:,:-1], lengths-1),
    mean CE, backward (torch autograd on this restatement -- test infrastructure).  Returns (loss, grads, logits)."""
    from . import decoder as D
    l1 = [int(l) - 1 for l in lengths]
    targets = D.pack_time_major(captions[:, 1:], l1)
    q = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    logits = attend_forward(q, features, captions[:, :-1], l1)
    n = logits.shape[0] if denom is None else denom
    loss = F.cross_entropy(logits, targets, reduction="sum") / n
    loss.backward()
    return loss.detach(), {k: v.grad for k, v in q.items()}, logits.detach()
