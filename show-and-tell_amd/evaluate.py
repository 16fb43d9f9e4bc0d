"""The validation half of the reference's `evaluation` (`/root/reference/eval.py:58-122`) for one mini-batch, on the device:

    targets = pack_padded_sequence(captions, lengths)[0]          eval.py:91   sat_pack_tokens (UNSHIFTED captions, FULL lengths)
    outputs = model(images, captions, lengths)                    eval.py:93   the model's own forward (eval mode: running statistics)
    loss = crit(outputs, targets)                                 eval.py:95   sat_ce_rows (no gradient)
    sampled_ids = model.sample(images, state)                     eval.py:99   greedy (or beam) decode
    for word_id in sentence_ids: if word == '<end>': break        eval.py:103-109   sat_kept_tokens: the count per row, on the device

`validation_step` returns device tensors (no host sync); `sentences` is the host-side id -> word join of eval.py:101-110 over
the already truncated rows.  `language_eval` (COCO scorers, eval.py:15-55) is out of scope."""
import torch

from . import _lib as L
from .pack import PackInfo


def pack_validation_targets(captions, lengths):
    """eval.py:91: `pack_padded_sequence(captions, lengths, batch_first=True)[0]` -- captions NOT shifted, lengths NOT
    decremented (unlike train.py:134-135).  Returns (targets i64 [sum(lengths)], PackInfo)."""
    L.require_gpu(captions, "captions")
    if captions.dtype != torch.int64 or captions.stride(1) != 1:
        captions = captions.long().contiguous()
    lengths = [int(l) for l in lengths]
    if len(lengths) != captions.shape[0] or lengths[0] > captions.shape[1] or lengths[-1] < 1:
        raise ValueError("need 1 <= every length <= captions.shape[1], one per row")
    pi = PackInfo.get(lengths, captions.device)
    targets = torch.empty(pi.N, dtype=torch.int64, device=captions.device)
    L.check(L.load().sat_pack_tokens(captions.data_ptr(), captions.stride(0), pi.prefix_dev.data_ptr(), pi.T, pi.N, 0,
                                     targets.data_ptr(), L.stream()), "sat_pack_tokens")
    return targets, pi


def mean_cross_entropy(logits, targets):
    """`nn.CrossEntropyLoss()(outputs, targets)` (eval.py:95) without a gradient: 1-element f32 device tensor."""
    L.require_gpu(logits, "logits")
    if logits.dim() != 2 or logits.dtype != torch.float32 or logits.stride(1) != 1:
        raise TypeError("logits must be a float32 matrix with contiguous rows")
    N, V = logits.shape
    if targets.shape[0] != N:
        raise ValueError("targets has %d rows, logits %d" % (targets.shape[0], N))
    row_loss = torch.empty(N, device=logits.device)
    loss = torch.zeros(1, device=logits.device)
    L.check(L.load().sat_ce_rows(logits.data_ptr(), logits.stride(0), targets.data_ptr(), N, V, 1.0 / N, 0, row_loss.data_ptr(),
                                 loss.data_ptr(), L.stream()), "sat_ce_rows")
    return loss


def kept_tokens(ids, end_id):
    """eval.py:103-109: per row, the number of ids in front of the first `end_id` (the row length when there is none):
    i32 [B] on the device."""
    L.require_gpu(ids, "ids")
    if ids.dim() != 2 or ids.dtype != torch.int64 or ids.stride(1) != 1:
        raise TypeError("ids must be an int64 matrix with contiguous rows")
    kept = torch.empty(ids.shape[0], dtype=torch.int32, device=ids.device)
    L.check(L.load().sat_kept_tokens(ids.data_ptr(), ids.stride(0), ids.shape[0], ids.shape[1], int(end_id), kept.data_ptr(),
                                     L.stream()), "sat_kept_tokens")
    return kept


@torch.no_grad()
def validation_step(model, images, captions, lengths, state=None, end_id=2, beam_size=1):
    """One iteration of the loop body eval.py:71-118 (the caller has put the model in eval mode, eval.py:65).
    Returns dict(loss f32[1], ids i64[B,20], kept i32[B]) -- all on the device, nothing synchronised."""
    if model.training:
        raise RuntimeError("validation_step expects model.eval() (eval.py:65)")
    targets, _ = pack_validation_targets(captions, lengths)
    outputs = model(images, captions, lengths)                          # eval.py:93
    if not outputs.is_contiguous():
        outputs = outputs.contiguous()
    loss = mean_cross_entropy(outputs, targets)                         # eval.py:95
    if beam_size > 1:
        ids = model.sample_beam(images, beam_size=beam_size, end_id=end_id)
    else:
        ids = model.sample(images, state)                               # eval.py:99
        if ids.dim() == 1:                                              # squeezed at batch 1 (models.py:67): one row
            ids = ids.view(1, -1)
    return {"loss": loss, "ids": ids, "kept": kept_tokens(ids, end_id)}


def sentences(ids, kept, idx2word):
    """eval.py:101-110 on the host: `' '.join(words in front of '<end>')` per row.  ids / kept: tensors (any device) or lists."""
    ids = ids.tolist() if hasattr(ids, "tolist") else ids
    kept = kept.tolist() if hasattr(kept, "tolist") else kept
    return [" ".join(idx2word[w] for w in row[:n]) for row, n in zip(ids, kept)]
