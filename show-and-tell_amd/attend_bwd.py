"""Backward of the Show-Attend-Tell decoder (`loss.backward()`, train.py:144, through model2.py:38-85), hand-written on
the forward's tapes: batched MFMA GEMMs for every weight gradient, one `sat_attention_bwd` + one `sat_lstmcell_bwd_point`
per packed step for the recurrence (BPTT).  The conv stack is frozen (model2.py:17): no gradient flows into it."""
import torch

from . import _lib as L
from .attend import _gemm


def _rows(t, r0):
    return t.data_ptr() + r0 * t.shape[1] * 4


def attend_backward(m, pi, captions, tp, dlogits, want_dfeat=False):
    """Returns (gradients in `attend.PARAM_ORDER`, d_features [B,P,C] or None, d_fmean [B,C] or None); the last two only when
    the conv stack is being fine-tuned (model2.py:87-89)."""
    lib, st = L.load(), L.stream()
    dev = dlogits.device
    f2, fmean, ctx_enc = tp["f2"], tp["fmean"], tp["ctx_enc"]
    CTX, HS, X, GATES, CS, ALPHA, Zin, Wz, Z = (tp[k] for k in ("CTX", "HS", "X", "GATES", "CS", "ALPHA", "Zin", "Wz", "Z"))
    h0, c0 = tp["h0"], tp["c0"]
    N, T, B = pi.N, pi.T, pi.B
    P = ALPHA.shape[1]
    C, H, E, V = CTX.shape[1], HS.shape[1], Z.shape[1], m.vocab_size
    Hin = X.shape[1]
    ldl = (V + 3) // 4 * 4
    if dlogits.shape[1] != ldl or not dlogits.is_contiguous():          # rows padded to 4 floats, zero pad
        pad = torch.zeros(N, ldl, device=dev)
        pad[:, :V] = dlogits
        dlogits = pad
    g = {}
    # ---- output_layer (model2.py:80-85), batched over all packed rows ----
    g["classifier.weight"] = torch.empty(V, E, device=dev)
    _gemm(lib, 2, 1, dlogits, ldl, Z, E, g["classifier.weight"], E, V, E, N)                # dW = dlogits^T Z
    g["classifier.bias"] = torch.empty(V, device=dev)
    L.check(lib.sat_colsum_f32(dlogits.data_ptr(), ldl, N, V, g["classifier.bias"].data_ptr(), st), "sat_colsum_f32")
    dZ = torch.empty(N, E, device=dev)
    _gemm(lib, 0, 1, dlogits, ldl, m.classifier.weight, E, dZ, E, N, E, V)                   # dZ = dlogits W_cls
    dWz = torch.empty(E, C + H, device=dev)
    _gemm(lib, 2, 1, dZ, E, Zin, C + H, dWz, C + H, E, C + H, N)
    g["context2out.weight"], g["hidden2tout.weight"] = torch.empty(E, C, device=dev), torch.empty(E, H, device=dev)
    L.check(lib.sat_rows_copy(dWz.data_ptr(), C + H, None, 0, E, E, C, g["context2out.weight"].data_ptr(), C, st), "sat_rows_copy")
    L.check(lib.sat_rows_copy(dWz.data_ptr() + C * 4, C + H, None, 0, E, E, H, g["hidden2tout.weight"].data_ptr(), H, st), "sat_rows_copy")
    g["context2out.bias"] = torch.empty(E, device=dev)
    L.check(lib.sat_colsum_f32(dZ.data_ptr(), E, N, E, g["context2out.bias"].data_ptr(), st), "sat_colsum_f32")
    g["hidden2tout.bias"] = g["context2out.bias"].clone()
    dZin = torch.empty(N, C + H, device=dev)
    _gemm(lib, 0, 1, dZ, E, Wz, C + H, dZin, C + H, N, C + H, E)                             # d[ctx | h] = dZ [W_c2o | W_h2o]
    DH = torch.empty(N, H, device=dev)
    L.check(lib.sat_rows_copy(dZin.data_ptr() + C * 4, C + H, None, 0, N, N, H, DH.data_ptr(), H, st), "sat_rows_copy")
    # ---- the recurrence, last step first (model2.py:54-62) ----
    DG, DPROJ = torch.empty(N, 4 * H, device=dev), torch.empty(N, C, device=dev)
    DX = torch.empty(B, Hin, device=dev)
    DEMB = torch.empty(N, E, device=dev)
    dctx = torch.empty(B, C, device=dev)
    dh_carry, dh_a, dh_b = torch.zeros(B, H, device=dev), torch.empty(B, H, device=dev), torch.empty(B, H, device=dev)
    dc_state = torch.zeros(B, H, device=dev)
    d_ctx_enc = torch.zeros_like(ctx_enc)
    d_feats = torch.zeros_like(ctx_enc) if want_dfeat else None      # [B*P, C]: what the weighted means send back (per step)
    dwatt_part = torch.empty(B, C, device=dev)
    att_ws = torch.empty(B * P, device=dev)
    g["weight_att"] = torch.zeros(C, device=dev)
    proj = torch.empty(B, C, device=dev)
    watt = m.weight_att.view(-1)
    for t in reversed(range(T)):
        bs, r0 = pi.batch_sizes[t], pi.prefix[t]
        n_carry = pi.batch_sizes[t + 1] if t + 1 < T else 0
        hprev = h0.data_ptr() if t == 0 else _rows(HS, pi.prefix[t - 1])
        cprev = c0.data_ptr() if t == 0 else _rows(CS, pi.prefix[t - 1])
        L.check(lib.sat_lstmcell_bwd_point(_rows(DH, r0), dh_carry.data_ptr() if n_carry else None, n_carry, _rows(GATES, r0),
                                           _rows(CS, r0), cprev, dc_state.data_ptr(), _rows(DG, r0), bs, H, st), "sat_lstmcell_bwd_point")
        _gemm(lib, 0, 1, _rows(DG, r0), 4 * H, m.lstmcell.weight_ih, Hin, DX, Hin, bs, Hin, 4 * H)       # d[emb | ctx]
        L.check(lib.sat_rows_copy(DX.data_ptr(), Hin, None, 0, bs, bs, E, _rows(DEMB, r0), E, st), "sat_rows_copy")
        L.check(lib.sat_rows_add(DX.data_ptr() + E * 4, Hin, _rows(dZin, r0), C + H, bs, C, dctx.data_ptr(), C, st), "sat_rows_add")
        _gemm(lib, 0, 0, hprev, H, m.weight_hh.weight, H, proj, C, bs, C, H, m.weight_hh.bias)           # recompute the projection
        L.check(lib.sat_attention_bwd(ctx_enc.data_ptr(), f2.data_ptr(), proj.data_ptr(), C, watt.data_ptr(), _rows(ALPHA, r0),
                                      dctx.data_ptr(), C, bs, P, C, d_ctx_enc.data_ptr(), _rows(DPROJ, r0), dwatt_part.data_ptr(),
                                      d_feats.data_ptr() if d_feats is not None else None, att_ws.data_ptr(), att_ws.numel() * 4, st),
                "sat_attention_bwd")
        L.check(lib.sat_rows_sum(dwatt_part.data_ptr(), C, bs, C, g["weight_att"].data_ptr(), 1, st), "sat_rows_sum")
        _gemm(lib, 0, 1, _rows(DG, r0), 4 * H, m.lstmcell.weight_hh, H, dh_a, H, bs, H, 4 * H)           # dh_{t-1} via the LSTM
        _gemm(lib, 0, 1, _rows(DPROJ, r0), C, m.weight_hh.weight, H, dh_b, H, bs, H, C)                  # ... and via the attention
        L.check(lib.sat_rows_add(dh_a.data_ptr(), H, dh_b.data_ptr(), H, bs, H, dh_carry.data_ptr(), H, st), "sat_rows_add")
    # ---- batched weight gradients of the recurrence ----
    HPREV = torch.empty(N, H, device=dev)              # h_{t-1} per packed row: h0 for step 0, HS rows of step t-1 after
    L.check(lib.sat_rows_copy(h0.data_ptr(), H, None, 0, B, pi.batch_sizes[0], H, HPREV.data_ptr(), H, st), "sat_rows_copy")
    for t in range(1, T):
        L.check(lib.sat_rows_copy(_rows(HS, pi.prefix[t - 1]), H, None, 0, pi.batch_sizes[t], pi.batch_sizes[t], H,
                                  _rows(HPREV, pi.prefix[t]), H, st), "sat_rows_copy")
    g["lstmcell.weight_ih"] = torch.empty(4 * H, Hin, device=dev)
    _gemm(lib, 2, 1, DG, 4 * H, X, Hin, g["lstmcell.weight_ih"], Hin, 4 * H, Hin, N)
    g["lstmcell.weight_hh"] = torch.empty(4 * H, H, device=dev)
    _gemm(lib, 2, 1, DG, 4 * H, HPREV, H, g["lstmcell.weight_hh"], H, 4 * H, H, N)
    g["lstmcell.bias_ih"] = torch.empty(4 * H, device=dev)
    L.check(lib.sat_colsum_f32(DG.data_ptr(), 4 * H, N, 4 * H, g["lstmcell.bias_ih"].data_ptr(), st), "sat_colsum_f32")
    g["lstmcell.bias_hh"] = g["lstmcell.bias_ih"].clone()
    g["weight_hh.weight"] = torch.empty(C, H, device=dev)
    _gemm(lib, 2, 1, DPROJ, C, HPREV, H, g["weight_hh.weight"], H, C, H, N)
    g["weight_hh.bias"] = torch.empty(C, device=dev)
    L.check(lib.sat_colsum_f32(DPROJ.data_ptr(), C, N, C, g["weight_hh.bias"].data_ptr(), st), "sat_colsum_f32")
    # ---- init_lstm (model2.py:67-71): dh0 = dh_carry, dc0 = dc_state ----
    for name, d in (("init_hidden", dh_carry), ("init_memory", dc_state)):
        g[name + ".weight"] = torch.empty(H, C, device=dev)
        _gemm(lib, 2, 1, d, H, fmean, C, g[name + ".weight"], C, H, C, B)
        g[name + ".bias"] = torch.empty(H, device=dev)
        L.check(lib.sat_colsum_f32(d.data_ptr(), H, B, H, g[name + ".bias"].data_ptr(), st), "sat_colsum_f32")
    # ---- image_att_w (model2.py:46): context_encode = features @ W  =>  dW = features^T d_ctx_enc ----
    g["image_att_w"] = torch.empty(C, C, device=dev)
    _gemm(lib, 2, 1, f2, C, d_ctx_enc, C, g["image_att_w"], C, C, C, f2.shape[0])
    g["weight_att"] = g["weight_att"].view(C, 1)
    # ---- embedding (dense gradient, nn.Embedding default): deterministic scatter of the per-row input gradients ----
    toks = torch.empty(N, dtype=torch.int64, device=dev)
    L.check(lib.sat_pack_tokens(captions.data_ptr(), captions.stride(0), pi.prefix_dev.data_ptr(), T, N, 0, toks.data_ptr(), st),
            "sat_pack_tokens")
    g["embedding.weight"] = torch.empty(V, E, device=dev)
    L.check(lib.sat_scatter_rows_add(DEMB.data_ptr(), toks.data_ptr(), N, E, V, g["embedding.weight"].data_ptr(), st),
            "sat_scatter_rows_add")
    d_fmean = None
    if want_dfeat:
        # context_encode = features @ W  =>  d features += d_ctx_enc @ W^T
        tmp = torch.empty_like(ctx_enc)
        _gemm(lib, 0, 0, d_ctx_enc, C, m.image_att_w, C, tmp, C, f2.shape[0], C, C)
        L.check(lib.sat_rows_add(d_feats.data_ptr(), C, tmp.data_ptr(), C, f2.shape[0], C, d_feats.data_ptr(), C, st), "sat_rows_add")
        # init_lstm: h0 = fmean Wh^T + b, c0 = fmean Wc^T + b  =>  d fmean = dh0 Wh + dc0 Wc
        da, db_ = torch.empty(B, C, device=dev), torch.empty(B, C, device=dev)
        _gemm(lib, 0, 1, dh_carry, H, m.init_hidden.weight, C, da, C, B, C, H)
        _gemm(lib, 0, 1, dc_state, H, m.init_memory.weight, C, db_, C, B, C, H)
        d_fmean = torch.empty(B, C, device=dev)
        L.check(lib.sat_rows_add(da.data_ptr(), C, db_.data_ptr(), C, B, C, d_fmean.data_ptr(), C, st), "sat_rows_add")
        d_feats = d_feats.view(B, P, C)
    from .attend import PARAM_ORDER
    return [g[k] for k in PARAM_ORDER], d_feats, d_fmean
