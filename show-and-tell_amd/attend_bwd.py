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
    PROJ, HS, X, GATES, CS, ALPHA, Zin, Wz, Z = (tp[k] for k in ("PROJ", "HS", "X", "GATES", "CS", "ALPHA", "Zin", "Wz", "Z"))
    h0, c0 = tp["h0"], tp["c0"]
    N, T, B = pi.N, pi.T, pi.B
    P = ALPHA.shape[1]
    C, H, E, V = PROJ.shape[1], HS.shape[1], Z.shape[1], m.vocab_size
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
    DX = torch.empty(N, Hin, device=dev)                # d[emb | ctx] of every packed row
    DWATT = torch.empty(N, C, device=dev)               # per-row partials of d weight_att, reduced once after the loop
    dh_carry = torch.zeros(B, H, device=dev)
    dc_state = torch.zeros(B, H, device=dev)
    d_ctx_enc = torch.zeros_like(ctx_enc)
    d_feats = torch.zeros_like(ctx_enc) if want_dfeat else None      # [B*P, C]: what the weighted means send back (per step)
    att_ws = torch.empty(B * P, device=dev)
    sk_ws = torch.empty(max(lib.sat_skinny_gemm_ws_bytes(B, H, 4 * H) // 4, 4), device=dev)
    watt = m.weight_att.view(-1)
    for t in reversed(range(T)):
        bs, r0 = pi.batch_sizes[t], pi.prefix[t]
        n_carry = pi.batch_sizes[t + 1] if t + 1 < T else 0
        cprev = c0.data_ptr() if t == 0 else _rows(CS, pi.prefix[t - 1])
        L.check(lib.sat_lstmcell_bwd_point(_rows(DH, r0), dh_carry.data_ptr() if n_carry else None, n_carry, _rows(GATES, r0),
                                           _rows(CS, r0), cprev, dc_state.data_ptr(), _rows(DG, r0), bs, H, st), "sat_lstmcell_bwd_point")
        _gemm(lib, 0, 1, _rows(DG, r0), 4 * H, m.lstmcell.weight_ih, Hin, _rows(DX, r0), Hin, bs, Hin, 4 * H)   # d[emb | ctx]
        # d context = its LSTMCell-input half + its output_layer half (summed where the attention backward reads it); the
        # projection weight_hh(h_{t-1}) comes from the forward's tape
        L.check(lib.sat_attention_bwd(ctx_enc.data_ptr(), f2.data_ptr(), _rows(PROJ, r0), C, watt.data_ptr(), _rows(ALPHA, r0),
                                      _rows(DX, r0) + E * 4, Hin, _rows(dZin, r0), C + H, bs, P, C, d_ctx_enc.data_ptr(),
                                      _rows(DPROJ, r0), _rows(DWATT, r0), d_feats.data_ptr() if d_feats is not None else None,
                                      att_ws.data_ptr(), att_ws.numel() * 4, st), "sat_attention_bwd")
        # dh_{t-1} = DG_t W_hh (through the LSTMCell) + DPROJ_t W_whh (through the attention projection): one launch
        L.check(lib.sat_skinny_gemm2_f32(_rows(DG, r0), 4 * H, m.lstmcell.weight_hh.data_ptr(), H, 4 * H,
                                         _rows(DPROJ, r0), C, m.weight_hh.weight.data_ptr(), H, C, 1, bs, H, None,
                                         dh_carry.data_ptr(), H, sk_ws.data_ptr(), sk_ws.numel() * 4, st), "sat_skinny_gemm2_f32")
    g["weight_att"] = torch.empty(C, device=dev)
    L.check(lib.sat_colsum_f32(DWATT.data_ptr(), C, N, C, g["weight_att"].data_ptr(), st), "sat_colsum_f32")
    DEMB = torch.empty(N, E, device=dev)
    L.check(lib.sat_rows_copy(DX.data_ptr(), Hin, None, 0, N, N, E, DEMB.data_ptr(), E, st), "sat_rows_copy")
    # ---- batched weight gradients of the recurrence ----
    HPREV = torch.empty(N, H, device=dev)              # h_{t-1} per packed row: h0 for step 0, HS rows of step t-1 after
    L.check(lib.sat_rows_copy(tp["HSX"].data_ptr(), H, pi.prev_rows().data_ptr(), 1, B + N, N, H, HPREV.data_ptr(), H, st),
            "sat_rows_copy")
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
    toks = tp["toks"]
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
