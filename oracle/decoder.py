"""CPU restatement of the reference decoder (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Follows, line by line:
  * `DecoderRNN.forward`   /root/reference/models.py:47-54  (embed -> cat feature -> pack -> LSTM -> linear)
  * `DecoderRNN.sample`    /root/reference/models.py:56-67  (greedy, torch-0.1 keepdim semantics)
  * `nn.CrossEntropyLoss`  /root/reference/train.py:53,143  (mean over packed rows)
  * `loss.backward()`      /root/reference/train.py:144     (explicit BPTT, no autograd)

Everything is explicit fp32 tensor arithmetic on the CPU: no nn.LSTM, no autograd.  Parameters are a
plain dict keyed exactly like the reference's `state_dict()` (`embed.weight`, `lstm.weight_ih_l{k}`,
`lstm.weight_hh_l{k}`, `lstm.bias_ih_l{k}`, `lstm.bias_hh_l{k}`, `linear.weight`, `linear.bias`).

Parity: PINNED by tests/golden/G1..G5 (generated from the imported reference `models.DecoderRNN`).
"""
import torch


def batch_sizes(lengths):
    """`pack_padded_sequence` batch_sizes for lengths sorted descending (models.py:51)."""
    lengths = [int(l) for l in lengths]
    assert all(lengths[i] >= lengths[i + 1] for i in range(len(lengths) - 1)), "lengths must be sorted desc"
    assert lengths[-1] >= 1
    return [sum(1 for l in lengths if l > t) for t in range(lengths[0])]


def pack_time_major(x, lengths):
    """x[B,T,...] -> packed rows [sum(lengths), ...] in time-major order (== pack_padded_sequence(...).data)."""
    bs = batch_sizes(lengths)
    return torch.cat([x[:bs[t], t] for t in range(len(bs))], 0)


def lstm_cell(x_gates, h, c, w_hh):
    """One step.  x_gates = x@W_ih^T + b_ih + b_hh already.  Gate order i,f,g,o (torch nn.LSTM)."""
    H = h.shape[1]
    gates = x_gates + h @ w_hh.t()
    i = torch.sigmoid(gates[:, 0:H])
    f = torch.sigmoid(gates[:, H:2 * H])
    g = torch.tanh(gates[:, 2 * H:3 * H])
    o = torch.sigmoid(gates[:, 3 * H:4 * H])
    c2 = f * c + i * g
    tc = torch.tanh(c2)
    h2 = o * tc
    return h2, c2, (i, f, g, o, tc)


def decoder_forward(params, features, captions, lengths, num_layers=1, keep=False):
    """models.py:47-54.  Returns logits [sum(lengths), V] (time-major packed); with keep=True also the tape."""
    emb = params["embed.weight"][captions]                       # models.py:49
    x = torch.cat((features.unsqueeze(1), emb), 1)               # models.py:50
    bs = batch_sizes(lengths)
    X = pack_time_major(x, lengths)                              # models.py:51
    B = features.shape[0]
    tape = {"bs": bs, "X": [X], "layers": []}
    for l in range(num_layers):                                  # models.py:52
        w_ih, w_hh = params["lstm.weight_ih_l%d" % l], params["lstm.weight_hh_l%d" % l]
        b = params["lstm.bias_ih_l%d" % l] + params["lstm.bias_hh_l%d" % l]
        H = w_hh.shape[1]
        xg = X @ w_ih.t() + b
        h = torch.zeros(B, H, dtype=X.dtype)
        c = torch.zeros(B, H, dtype=X.dtype)
        outs, steps, off = [], [], 0
        for t, n in enumerate(bs):
            h_prev, c_prev = h[:n], c[:n]
            h2, c2, (i, f, g, o, tc) = lstm_cell(xg[off:off + n], h_prev, c_prev, w_hh)
            steps.append(dict(i=i, f=f, g=g, o=o, tc=tc, c_prev=c_prev, h_prev=h_prev))
            h, c = h2, c2
            outs.append(h2)
            off += n
        X = torch.cat(outs, 0)
        tape["layers"].append(steps)
        tape["X"].append(X)
    logits = X @ params["linear.weight"].t() + params["linear.bias"]   # models.py:53
    return (logits, tape) if keep else logits


def cross_entropy(logits, targets):
    """nn.CrossEntropyLoss() default reduction='mean' (train.py:53,143)."""
    m = logits.max(1, keepdim=True)[0]
    lse = m.squeeze(1) + torch.log(torch.exp(logits - m).sum(1))
    return (lse - logits.gather(1, targets[:, None]).squeeze(1)).mean()


def cross_entropy_grad(logits, targets, denom=None):
    """d(mean CE)/d logits = (softmax - onehot)/N.  `denom` overrides N (data-parallel global token count)."""
    n = logits.shape[0] if denom is None else denom
    m = logits.max(1, keepdim=True)[0]
    e = torch.exp(logits - m)
    p = e / e.sum(1, keepdim=True)
    p[torch.arange(logits.shape[0]), targets] -= 1.0
    return p / n


def decoder_backward(params, tape, captions, lengths, dlogits, num_layers=1):
    """Explicit backward of decoder_forward.  Returns (grads dict keyed like params, d_features[B,E])."""
    bs = tape["bs"]
    grads = {}
    Xtop = tape["X"][-1]
    grads["linear.weight"] = dlogits.t() @ Xtop
    grads["linear.bias"] = dlogits.sum(0)
    dX = dlogits @ params["linear.weight"]
    for l in reversed(range(num_layers)):
        w_ih, w_hh = params["lstm.weight_ih_l%d" % l], params["lstm.weight_hh_l%d" % l]
        H = w_hh.shape[1]
        steps = tape["layers"][l]
        Xin = tape["X"][l]
        offs = [0]
        for n in bs:
            offs.append(offs[-1] + n)
        B = bs[0]
        dh_next = torch.zeros(B, H, dtype=dX.dtype)
        dc_next = torch.zeros(B, H, dtype=dX.dtype)
        DG = torch.zeros(offs[-1], 4 * H, dtype=dX.dtype)
        Hprev = torch.zeros(offs[-1], H, dtype=dX.dtype)
        for t in reversed(range(len(bs))):
            n = bs[t]
            s = steps[t]
            dh = dX[offs[t]:offs[t + 1]] + dh_next[:n]
            do = dh * s["tc"]
            dc = dh * s["o"] * (1 - s["tc"] ** 2) + dc_next[:n]
            di = dc * s["g"]
            df = dc * s["c_prev"]
            dg = dc * s["i"]
            dgates = torch.cat((di * s["i"] * (1 - s["i"]), df * s["f"] * (1 - s["f"]),
                                dg * (1 - s["g"] ** 2), do * s["o"] * (1 - s["o"])), 1)
            DG[offs[t]:offs[t + 1]] = dgates
            Hprev[offs[t]:offs[t + 1]] = s["h_prev"]
            dh_next = torch.zeros(B, H, dtype=dX.dtype)
            dc_next = torch.zeros(B, H, dtype=dX.dtype)
            dh_next[:n] = dgates @ w_hh
            dc_next[:n] = dc * s["f"]
        grads["lstm.weight_ih_l%d" % l] = DG.t() @ Xin
        grads["lstm.weight_hh_l%d" % l] = DG.t() @ Hprev
        grads["lstm.bias_ih_l%d" % l] = DG.sum(0)
        grads["lstm.bias_hh_l%d" % l] = DG.sum(0)
        dX = DG @ w_ih
    # layer-0 input = [feature rows (t=0) ; embedding rows (t>=1)]
    B = bs[0]
    d_features = dX[:B].clone()
    dE = torch.zeros_like(params["embed.weight"])
    off = B
    for t in range(1, len(bs)):
        n = bs[t]
        dE.index_add_(0, captions[:n, t - 1], dX[off:off + n])
        off += n
    grads["embed.weight"] = dE
    return grads, d_features


def greedy_sample(params, features, num_layers=1, steps=20, states=None):
    """models.py:56-67 with torch-0.1 `max(1)` keepdim semantics: ids [B,20] int64.
    `states` None == zeros (eval.py:82-83).  Ties: first maximal index (torch.max)."""
    B = features.shape[0]
    hs, cs = [], []
    for l in range(num_layers):
        H = params["lstm.weight_hh_l%d" % l].shape[1]
        hs.append(torch.zeros(B, H) if states is None else states[0][l].clone())
        cs.append(torch.zeros(B, H) if states is None else states[1][l].clone())
    x = features
    ids = []
    for _ in range(steps):
        inp = x
        for l in range(num_layers):
            xg = inp @ params["lstm.weight_ih_l%d" % l].t() + params["lstm.bias_ih_l%d" % l] + params["lstm.bias_hh_l%d" % l]
            hs[l], cs[l], _ = lstm_cell(xg, hs[l], cs[l], params["lstm.weight_hh_l%d" % l])
            inp = hs[l]
        logits = inp @ params["linear.weight"].t() + params["linear.bias"]
        pred = logits.max(1)[1]
        ids.append(pred)
        x = params["embed.weight"][pred]
    return torch.stack(ids, 1)


def beam_search(params, features, beam_size=5, num_layers=1, steps=20, end_id=None):
    """Beam decode.  The reference only has a stub (`model2.py:113-114` `sample_beam: pass`), so this is
    the textbook algorithm laid over `models.py:56-67`'s loop -- PARITY UNPINNED by the reference, except that
    beam_size=1 with end_id=None must reproduce `greedy_sample` (pinned by the G1/G3/G5 greedy goldens).

    Per image: `beam_size` hypotheses; step 0 feeds the image feature (all hypotheses identical, only hypothesis 0
    live); every step scores candidate (k, v) as score[k] + log_softmax(logits[k])[v] and keeps the best
    `beam_size` of the K*V candidates (ties: lower k*V+v first), re-ordering the LSTM state by parent.
    end_id: a hypothesis whose last token is end_id is finished -- it only extends with end_id at no cost.
    Returns (ids [B,K,steps] int64 sorted best-first, scores [B,K] f32)."""
    B, K = features.shape[0], beam_size
    V = params["linear.weight"].shape[0]
    rep = lambda t: t.repeat_interleave(K, 0)
    hs, cs = [], []
    for l in range(num_layers):
        H = params["lstm.weight_hh_l%d" % l].shape[1]
        hs.append(torch.zeros(B * K, H))
        cs.append(torch.zeros(B * K, H))
    scores = torch.full((B, K), float("-inf"))
    scores[:, 0] = 0.0
    x = rep(features)
    seqs = torch.zeros(B, K, 0, dtype=torch.int64)
    last = None
    for _ in range(steps):
        inp = x
        for l in range(num_layers):
            xg = inp @ params["lstm.weight_ih_l%d" % l].t() + params["lstm.bias_ih_l%d" % l] + params["lstm.bias_hh_l%d" % l]
            hs[l], cs[l], _ = lstm_cell(xg, hs[l], cs[l], params["lstm.weight_hh_l%d" % l])
            inp = hs[l]
        logits = inp @ params["linear.weight"].t() + params["linear.bias"]
        logp = torch.log_softmax(logits, dim=1).view(B, K, V)
        cand = scores.unsqueeze(2) + logp
        if end_id is not None and last is not None:
            fin = last == end_id                                     # [B,K]
            frozen = torch.full((B, K, V), float("-inf"))
            frozen[:, :, end_id] = scores
            cand = torch.where(fin.unsqueeze(2), frozen, cand)
        cand = cand.view(B, K * V)
        order = torch.sort(cand, dim=1, descending=True, stable=True)[1][:, :K]     # stable: lower flat index first
        scores = torch.gather(cand, 1, order)
        parent, token = order // V, order % V
        rows = (torch.arange(B).unsqueeze(1) * K + parent).reshape(-1)
        hs = [h[rows] for h in hs]
        cs = [c[rows] for c in cs]
        seqs = torch.cat([torch.gather(seqs, 1, parent.unsqueeze(2).expand(B, K, seqs.shape[2])), token.unsqueeze(2)], 2)
        last = token
        x = params["embed.weight"][token.reshape(-1)]
    return seqs, scores


def init_decoder_params(embed_size, hidden_size, vocab_size, num_layers, generator=None):
    """Reference init (models.py:35-45): embed U(-.1,.1); LSTM torch default U(+-1/sqrt(H)); linear W U(-.1,.1), b=0."""
    g = generator
    p = {}
    p["embed.weight"] = torch.empty(vocab_size, embed_size).uniform_(-0.1, 0.1, generator=g)
    k = 1.0 / (hidden_size ** 0.5)
    for l in range(num_layers):
        in_sz = embed_size if l == 0 else hidden_size
        p["lstm.weight_ih_l%d" % l] = torch.empty(4 * hidden_size, in_sz).uniform_(-k, k, generator=g)
        p["lstm.weight_hh_l%d" % l] = torch.empty(4 * hidden_size, hidden_size).uniform_(-k, k, generator=g)
        p["lstm.bias_ih_l%d" % l] = torch.empty(4 * hidden_size).uniform_(-k, k, generator=g)
        p["lstm.bias_hh_l%d" % l] = torch.empty(4 * hidden_size).uniform_(-k, k, generator=g)
    p["linear.weight"] = torch.empty(vocab_size, hidden_size).uniform_(-0.1, 0.1, generator=g)
    p["linear.bias"] = torch.zeros(vocab_size)
    return p
