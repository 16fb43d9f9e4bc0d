"""CPU restatement of the reference's per-iteration body (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Follows `/root/reference/train.py:126-146`:
    lengths = [l-1 for l in lengths]                                   (train.py:134)
    targets = pack_padded_sequence(captions[:,1:], lengths)[0]         (train.py:135)
    model.zero_grad(); outputs = model(images, captions[:,:-1], lengths)   (train.py:137-139)
    loss = CrossEntropyLoss()(outputs, targets); loss.backward()       (train.py:143-144)
    clip_gradient: param.grad.clamp_(-grad_clip, grad_clip)            (train.py:88-91,145)
    Adam(lr) step, torch defaults betas=(0.9,0.999) eps=1e-8 wd=0      (train.py:56,146)
and the epoch LR decay (train.py:101-107).

Parity: PINNED by tests/golden/G1 (grads, clamped grads, params after 1 and 3 Adam steps from the
imported reference module + torch.optim.Adam).
"""
import torch

from . import decoder as D
from . import encoder as E


def clamp_(grads, grad_clip):
    """train.py:88-91 -- ELEMENTWISE clamp, not a norm clip."""
    for g in grads.values():
        g.clamp_(-grad_clip, grad_clip)
    return grads


def adam_step_(params, grads, state, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam (no amsgrad, wd=0), same operation order as torch's single-tensor path.
    state: {'step': int, 'm': {k: tensor}, 'v': {k: tensor}} (created lazily)."""
    if "step" not in state:
        state["step"] = 0
        state["m"] = {k: torch.zeros_like(p) for k, p in params.items() if k in grads}
        state["v"] = {k: torch.zeros_like(p) for k, p in params.items() if k in grads}
    state["step"] += 1
    t = state["step"]
    bc1 = 1.0 - beta1 ** t
    bc2 = 1.0 - beta2 ** t
    step_size = lr / bc1
    bc2_sqrt = bc2 ** 0.5
    for k, g in grads.items():
        m, v, p = state["m"][k], state["v"][k], params[k]
        m.lerp_(g, 1.0 - beta1)                       # m += (g-m)*(1-b1)
        v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
        denom = (v.sqrt() / bc2_sqrt).add_(eps)
        p.addcdiv_(m, denom, value=-step_size)
    return params


def lr_for_epoch(epoch, learning_rate=1e-3, decay_start=1, decay_every=3, decay_rate=0.8):
    """train.py:101-107."""
    if epoch > decay_start and decay_start >= 1:
        fraction = (epoch - decay_start) // decay_every
        return learning_rate * decay_rate ** fraction
    return learning_rate


def pack_targets(captions, lengths):
    """train.py:134-135: returns (targets[N] int64, lengths-1)."""
    l1 = [int(l) - 1 for l in lengths]
    return D.pack_time_major(captions[:, 1:], l1), l1


def decoder_loss_and_grads(params, features, captions, lengths, num_layers=1, denom=None):
    """Decoder half of train.py:134-144 given encoder features.  `captions` is the FULL caption tensor
    [B,T]; `lengths` the full lengths.  Returns (loss_sum_over_rows/denom, grads, d_features, logits)."""
    targets, l1 = pack_targets(captions, lengths)
    logits, tape = D.decoder_forward(params, features, captions[:, :-1], l1, num_layers, keep=True)
    n = logits.shape[0] if denom is None else denom
    loss = D.cross_entropy(logits, targets) * (logits.shape[0] / n)
    dlogits = D.cross_entropy_grad(logits, targets, denom=n)
    grads, d_feat = D.decoder_backward(params, tape, captions[:, :-1], l1, dlogits, num_layers)
    return loss, grads, d_feat, logits


def full_step(enc_params, enc_buffers, dec_params, images, captions, lengths, opt_state,
              arch=E.RESNET152, num_layers=1, lr=1e-3, grad_clip=0.1, denom=None, do_update=True,
              encoder_storage="f32", out=None):
    """One whole train.py:126-146 iteration for the Show-and-Tell model (models.py:9-67).
    Trainable: encoder fc + bn (enc_params keys 'resnet.fc.*', 'bn.*') and all decoder params.
    `encoder_storage="bf16"`: the frozen conv stack with bf16 STORAGE emulated (`resnet_forward_bf16_storage`: the
    oracle of the library's bf16 mode); head, decoder, loss and optimizer stay f32 as they do there.
    `out`: optional dict that receives the step's `pooled` and `features` (head output).
    Returns (loss, grads dict) -- grads are post-clamp when do_update."""
    if encoder_storage == "bf16":
        pooled = E.resnet_forward_bf16_storage(enc_params, images, arch)
    else:
        pooled, _ = E.resnet_forward(enc_params, enc_buffers, images, arch, training=True)
    feats, tape = E.head_forward(enc_params, enc_buffers, pooled, training=True)
    if out is not None:
        out["pooled"], out["features"] = pooled, feats
    loss, grads, d_feat, _ = decoder_loss_and_grads(dec_params, feats, captions, lengths, num_layers, denom)
    hg = E.head_backward(enc_params, tape, d_feat)
    grads = dict(grads)
    grads.update(hg)
    if do_update:
        clamp_(grads, grad_clip)
        allp = {}
        allp.update({k: enc_params[k] for k in hg})
        allp.update(dec_params)
        adam_step_(allp, grads, opt_state, lr=lr)
    return loss, grads


def validation_loss(dec_params, features, captions, lengths, num_layers=1):
    """The validation forward of `evaluation` (eval.py:91-95): `targets = pack(captions, lengths)` with the captions NOT
    shifted and the lengths NOT decremented, `outputs = model(images, captions, lengths)`, mean CE.  (The decoder then
    consumes [feature, captions[:, :l-1]] -- models.py:49-51 -- so output row t is scored against caption token t.)
    Returns (loss, logits).  Pinned by tests/golden/G8 (the imported reference decoder)."""
    targets = D.pack_time_major(captions, [int(l) for l in lengths])
    logits = D.decoder_forward(dec_params, features, captions, [int(l) for l in lengths], num_layers)
    return D.cross_entropy(logits, targets), logits


def kept_tokens(ids, end_id):
    """eval.py:103-109: the id -> word loop breaks at '<end>': per row, the number of ids in front of the first end_id."""
    out = []
    for row in ids.tolist():
        n = 0
        for w in row:
            if w == end_id:
                break
            n += 1
        out.append(n)
    return out
