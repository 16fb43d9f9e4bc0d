"""Fused training step: the body of the reference's hot loop (`/root/reference/train.py:126-146`) for the
Show-and-Tell model, with every tensor op a libsat_hip.so kernel and no autograd:

    lengths-1, targets = pack(captions[:,1:])            train.py:134-135   sat_pack_targets
    outputs = model(images, captions[:,:-1], lengths)     train.py:139       sat_run_ops / sat_fc_bn1d_fwd /
                                                                             sat_embed_concat_fwd / sat_lstm_fwd / sat_vocab_logits_fwd
    loss = CrossEntropyLoss()(outputs, targets)           train.py:143       sat_ce_rows (gradient written in place)
    loss.backward()                                       train.py:144       sat_vocab_ce_bwd / sat_lstm_bwd / sat_embed_concat_bwd / sat_fc_bn1d_bwd
    clip_gradient (elementwise clamp) + Adam step         train.py:145-146   sat_clamp_adam_step (one launch)

MI355X-first layout: all trainable parameters live in ONE flat f32 buffer (module parameters are views of it),
with matching flat grad / Adam-m / Adam-v buffers.  The optimizer is one HBM-bound launch over 28 B/param and
the data-parallel gradient exchange is a handful of large RCCL all-reduces over contiguous buckets, launched
as each bucket's gradients become final and overlapped with the rest of the backward (`DataParallelStep`).
"""
import torch

from . import _lib as L
from .models import BN1D_MOMENTUM, BN_EPS, ShowAndTell, decoder_backward_tapes, decoder_forward_tapes
from .pack import PackInfo


import os as _os

# bf16 throughput mode (compute_dtype="bf16", BASELINE configs[1]): the vocab projection and its two gradient GEMMs run on the
# bf16 matrix pipe from per-step bf16 operand copies (sat_gemm_bf16.hip; f32 accumulate, f32 logits / CE / outputs / master
# weights).  SAT_DECODER_BF16=0 keeps them exact-f32 as in the parity mode (round 2's behaviour: 0.39 ms/step instead of ~0.1).
_DECODER_BF16 = _os.environ.get("SAT_DECODER_BF16", "1") != "0"


def lr_for_epoch(epoch, learning_rate=1e-3, decay_start=1, decay_every=3, decay_rate=0.8):
    """Epoch step decay of the reference trainer (train.py:101-107; defaults config.py:38-46)."""
    if epoch > decay_start and decay_start >= 1:
        return learning_rate * decay_rate ** ((epoch - decay_start) // decay_every)
    return learning_rate


def _pad4(n):
    return (n + 3) // 4 * 4


class FlatParams:
    """Trainable parameters re-homed into one flat f32 buffer, in gradient-completion order:
    bucket 0 = vocab projection, bucket 1 = LSTM (top layer first), bucket 2 = encoder head + embedding."""

    def __init__(self, model):
        dec, enc = model.decoder, model.encoder
        groups = [[("decoder.linear.weight", dec.linear.weight), ("decoder.linear.bias", dec.linear.bias)], [], []]
        for l in reversed(range(dec.num_layers)):
            for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                groups[1].append(("decoder.lstm.%s_l%d" % (n, l), getattr(dec.lstm, "%s_l%d" % (n, l))))
        groups[2] = [("encoder.resnet.fc.weight", enc.resnet.fc.weight), ("encoder.resnet.fc.bias", enc.resnet.fc.bias),
                     ("encoder.bn.weight", enc.bn.weight), ("encoder.bn.bias", enc.bn.bias),
                     ("decoder.embed.weight", dec.embed.weight)]
        dev = dec.linear.weight.device
        self.slices, self.buckets = {}, []
        off = 0
        for g in groups:
            start = off
            for name, p in g:
                self.slices[name] = (off, p.numel(), tuple(p.shape))
                off += _pad4(p.numel())
            self.buckets.append((start, off))
        self.n = off
        # 4 trailing floats: slot 0 carries this rank's loss term through the last bucket's all-reduce, slot 1 the step's fault
        # flag (TrainStep._fault_flag: non-zero on EVERY rank after the reduction when any rank's persistent LSTM launch gave up)
        self.loss_slot = off
        self.buckets[-1] = (self.buckets[-1][0], off + 4)
        self.params = torch.zeros(off, device=dev)
        self.grads = torch.zeros(off + 4, device=dev)
        self.m = torch.zeros(off, device=dev)
        self.v = torch.zeros(off, device=dev)
        for g in groups:
            for name, p in g:
                o, n, shape = self.slices[name]
                self.params[o:o + n].copy_(p.data.reshape(-1))
                p.data = self.params[o:o + n].view(shape)
                p.grad = self.grads[o:o + n].view(shape)

    def grad(self, name):
        o, n, shape = self.slices[name]
        return self.grads[o:o + n].view(shape)


class TrainStep:
    """One object = model + flat optimizer state; `step()` = one train.py:126-146 iteration on this GPU."""

    def __init__(self, model, lr=1e-3, grad_clip=0.1, betas=(0.9, 0.999), eps=1e-8):
        if not isinstance(model, ShowAndTell):
            raise TypeError("TrainStep drives a ShowAndTell model")
        L.require_gpu(model.decoder.linear.weight, "model")
        self.lib = L.load()
        self.model = model
        self.flat = FlatParams(model)
        self.lr, self.grad_clip, self.betas, self.eps = lr, grad_clip, betas, eps
        self.step_count = 0
        self.buckets = self.flat.buckets
        self.flat_grad = self.flat.grads
        self._bufs = {}
        # "bf16": vocab projection + its gradient GEMMs on the bf16 matrix pipe (the throughput mode); "f32": exact-f32 MFMA
        self.decoder_gemm_dtype = "bf16" if (_DECODER_BF16 and model.encoder.compute_dtype == "bf16") else "f32"
        self._params_list = [p for _, p in self._trainable()]
        # sticky device-side fault word of this engine (sat_step_fault_flag): set by a step whose persistent LSTM recurrence gave
        # up, cleared by the host once it has raised for it; while it is set every clamp + Adam launch drops its update
        self._fault_sticky = torch.zeros(1, device=self.flat.params.device)
        self.fault_slot = self.flat.grads[self.flat.loss_slot + 1:self.flat.loss_slot + 2]

    def __del__(self):
        try:                                     # the library remembers a backward workspace by address: tell it the memory is gone
            for _, bws in getattr(self, "_lstm_ws", []):
                self.lib.sat_lstm_ws_release(bws.data_ptr())
        except Exception:
            pass

    # -- encoder look-ahead ---------------------------------------------------------------------------
    def prefetch_encoder(self, images):
        """Start the frozen conv stack of a LATER batch on a side stream (`EncoderCNN.prefetch`: up to three batches' stacks in
        flight next to each other and under this batch's decoder work; bitwise identical results)."""
        return self.model.encoder.prefetch(images)

    def _encoder_pooled(self, images, out, next_images=None):
        """pooled features [B, F] of `images` into `out` (a buffer this step owns): from the look-ahead if this tensor was
        prefetched, computed now otherwise.  `next_images`: the following batches -- their stacks are started on side streams
        (`EncoderCNN.prefetch_many`): BEFORE this batch's own stack when that still has to run (the first step of a loop: the
        look-ahead then fills the chip next to it instead of waiting for it), behind the hand-over otherwise (the instance this
        batch frees is available to them)."""
        enc = self.model.encoder
        nxt = None
        last = getattr(next_images, "last", None)      # (DevicePrefetcher.upcoming_images() says whether the data ends inside the list)
        if next_images is not None:
            nxt = list(next_images) if isinstance(next_images, (list, tuple)) else [next_images]
        hit = enc._take_prefetched(images)
        if hit is not None:
            prog, g = hit
            out.copy_(prog.pooled_of(g))
            prog.apply_running_stats(g)         # batch order = consumption order
            if nxt:
                enc.prefetch_many(nxt, last=last)
            return out
        if nxt:
            enc.prefetch_many(nxt, last=last, own_stack=True)
        out.copy_(enc._pooled_raw(images))
        return out

    def drop_lookahead(self):
        self.model.encoder.drop_lookahead()

    # -- engine interface used by DataParallelStep ----------------------------------------------------
    def forward_backward(self, batch, inv_denom, on_bucket_ready=None, next_images=None):
        """batch = (images f32[B,3,H,W], captions i64[B,T], lengths list[int] desc).  Fills the flat grad buffer
        (gradients of sum-CE * inv_denom) and the loss slot; returns the loss slot tensor (device, 1 elem).
        next_images: the FOLLOWING batch's images, or a list of the next few in order; their conv stacks are started on side
        streams under this batch's decoder work (prefetch_encoder)."""
        images, captions, lengths = batch
        lib, model, flat = self.lib, self.model, self.flat
        enc, dec = model.encoder, model.decoder
        if not model.training:
            raise RuntimeError("TrainStep needs model.train() (batch-statistics BatchNorm, train.py never calls eval())")
        L.require_gpu(captions, "captions")
        dev = images.device
        st = L.stream()
        B = images.shape[0]
        if captions.dtype != torch.int64 or captions.stride(1) != 1:
            captions = captions.long().contiguous()
        lengths = [int(l) for l in lengths]
        # sat_pack_targets reads captions[b][t+1] up to column lengths[0]-1 and the embedding gather captions[b][t-1]:
        # every caption must hold at least <start> + one target, and no length may exceed the matrix width
        if len(lengths) != B or captions.dim() != 2 or captions.shape[0] != B:
            raise ValueError("captions must be [B, T] with one length per image")
        if lengths[-1] < 2 or lengths[0] > captions.shape[1]:
            raise ValueError("need 2 <= every length <= captions.shape[1] (got min %d, max %d, width %d)"
                             % (lengths[-1], lengths[0], captions.shape[1]))
        dec.id_guard().submit(captions, lengths[0], dec.vocab_size, "captions")   # raises for a bad EARLIER batch at the latest here
        l1 = [l - 1 for l in lengths]                                        # train.py:134
        pi = PackInfo.get(l1, dev)
        N, V, E = pi.N, dec.vocab_size, dec.embed_size
        key = (B, N)
        bufs = self._bufs.get(key)
        if bufs is None:
            self._bufs.clear()
            F = enc.resnet.feature_dim
            wsb = lib.sat_fc_bn1d_ws_bytes(B, F, E)
            bufs = self._bufs[key] = dict(
                targets=torch.empty(N, dtype=torch.int64, device=dev), logits=torch.zeros(N, (V + 3) // 4 * 4, device=dev),
                row_loss=torch.empty(N, device=dev), feats=torch.empty(B, E, device=dev),
                xhat=torch.empty(B, E, device=dev), rstd=torch.empty(E, device=dev),
                head_ws=torch.empty(max(wsb // 4, B * E), device=dev), d_feat=torch.empty(B, E, device=dev),
                pooled=torch.empty(B, F, device=dev))
        # LSTM workspaces this engine owns for good, keyed by (B, caption width) -- NOT by N, which changes with almost every batch
        # of real captions: sized for the longest batch of that width (sat_lstm_bwd_ws_bytes_max), the exchange region and the
        # status words at offsets that depend on (B, H) only; their status words are folded into the step's fault flag below
        wkey = (B, int(captions.shape[1]))
        if getattr(self, "_lstm_ws_key", None) != wkey:
            for _, old in getattr(self, "_lstm_ws", []):
                lib.sat_lstm_ws_release(old.data_ptr())          # (the library forgets the address: a later buffer there is cleared again)
            self._lstm_ws, words = [], []
            n_max = B * (int(captions.shape[1]) - 1)
            for l in range(dec.num_layers):
                In = E if l == 0 else dec.hidden_size
                fb, bb = lib.sat_lstm_fwd_ws_bytes(B, dec.hidden_size), lib.sat_lstm_bwd_ws_bytes_max(n_max, B, In, dec.hidden_size)
                fws = torch.zeros(max(fb, 16), dtype=torch.uint8, device=dev)
                bws = torch.empty(bb, dtype=torch.uint8, device=dev)
                self._lstm_ws.append((fws, bws))
                fo = lib.sat_lstm_fwd_status_offset(B, dec.hidden_size)
                if fo >= 0 and fb > 0:
                    words.append(fws.data_ptr() + fo)
                so = lib.sat_lstm_bwd_status_offset(N, B, In, dec.hidden_size)
                bws[so:so + 64].zero_()                          # (read by the fault flag even when the call below never runs persistently)
                words.append(bws.data_ptr() + so)
            if len(words) > 8:
                raise ValueError("at most 4 LSTM layers (8 status words per step)")
            import ctypes as _C
            self._fault_words = ((_C.c_void_p * len(words))(*words), len(words))
            self._lstm_ws_key = wkey
        bufs["lstm_ws"], bufs["fault_words"] = self._lstm_ws, self._fault_words
        # targets = pack(captions[:,1:], lengths-1)                           train.py:135
        L.check(lib.sat_pack_targets(captions.data_ptr(), captions.stride(0), L.ptr(pi.prefix_dev), pi.T, N,
                                     L.ptr(bufs["targets"]), st), "sat_pack_targets")
        # ---- forward (train.py:139) ----
        cached_features = images.dim() == 2      # [B,E] precomputed encoder features: decoder-only training
        if cached_features:
            feats_in = images.contiguous()
            pooled = None
        else:
            pooled = self._encoder_pooled(images, bufs["pooled"], next_images)
            F = pooled.shape[1]
            fc, bn = enc.resnet.fc, enc.bn
            L.check(lib.sat_fc_bn1d_fwd(L.ptr(pooled), L.ptr(fc.weight), L.ptr(fc.bias), L.ptr(bn.weight), L.ptr(bn.bias),
                                        L.ptr(bn.running_mean), L.ptr(bn.running_var), BN1D_MOMENTUM, BN_EPS, 1, B, F, E,
                                        L.ptr(bufs["feats"]), L.ptr(bufs["xhat"]), L.ptr(bufs["rstd"]),
                                        L.ptr(bufs["head_ws"]), bufs["head_ws"].numel() * 4, st), "sat_fc_bn1d_fwd")
            L.counter_add(bn.num_batches_tracked)
            feats_in = bufs["feats"]
        layers = [dec.lstm.layer(l) for l in range(dec.num_layers)]
        loss_slot = flat.grads[flat.loss_slot:flat.loss_slot + 1]
        ce = None
        if self.decoder_gemm_dtype == "bf16":
            wsb = lib.sat_vocab_bf16_ws_bytes(N, dec.hidden_size, V)
            if wsb > 0:
                if "vocab_bf16_ws" not in bufs:
                    bufs["vocab_bf16_ws"] = torch.empty(wsb, dtype=torch.uint8, device=dev)
                ce = dict(kind="bf16", targets=bufs["targets"], inv_denom=inv_denom, row_loss=bufs["row_loss"], loss_out=loss_slot,
                          ws=bufs["vocab_bf16_ws"])
        mixed_ws = None
        if self.decoder_gemm_dtype == "bf16":          # the LSTM layers' batched GEMMs on the bf16 matrix pipe too
            if "lstm_mixed_ws" not in bufs:
                need = max(lib.sat_lstm_mixed_ws_bytes(N, E if l == 0 else dec.hidden_size, dec.hidden_size) for l in range(dec.num_layers))
                bufs["lstm_mixed_ws"] = torch.empty(need, dtype=torch.uint8, device=dev)
            mixed_ws = bufs["lstm_mixed_ws"]
        logits, tapes = decoder_forward_tapes(lib, feats_in, dec.embed.weight, layers, dec.linear.weight,
                                              dec.linear.bias, captions[:, :-1], pi, logits=bufs["logits"], ce=ce, mixed_ws=mixed_ws,
                                              lstm_ws=bufs["lstm_ws"])
        if ce is None:
            # ---- loss + d(loss)/d(logits) in place (train.py:143) ----
            L.check(lib.sat_ce_rows(L.ptr(logits), logits.stride(0), L.ptr(bufs["targets"]), N, V, float(inv_denom), 1,
                                    L.ptr(bufs["row_loss"]), L.ptr(loss_slot), st), "sat_ce_rows")
        # ---- backward (train.py:144): gradients land directly in the flat buffer ----
        g = {"embed": flat.grad("decoder.embed.weight"), "lin_w": flat.grad("decoder.linear.weight"),
             "lin_b": flat.grad("decoder.linear.bias"), "features": bufs["d_feat"]}
        for l in range(dec.num_layers):
            for short, n in (("w_ih", "weight_ih"), ("w_hh", "weight_hh"), ("b_ih", "bias_ih"), ("b_hh", "bias_hh")):
                g[(short, l)] = flat.grad("decoder.lstm.%s_l%d" % (n, l))
        decoder_backward_tapes(lib, logits, tapes, dec.embed.weight, layers, dec.linear.weight, pi, g,
                               on_stage=on_bucket_ready, ce=ce, mixed_ws=mixed_ws, lstm_ws=bufs["lstm_ws"])
        if not cached_features:
            fc, bn = enc.resnet.fc, enc.bn
            L.check(lib.sat_fc_bn1d_bwd(L.ptr(bufs["d_feat"]), L.ptr(pooled), L.ptr(bufs["xhat"]), L.ptr(bufs["rstd"]),
                                        L.ptr(bn.weight), B, pooled.shape[1], E, L.ptr(flat.grad("encoder.resnet.fc.weight")),
                                        L.ptr(flat.grad("encoder.resnet.fc.bias")), L.ptr(flat.grad("encoder.bn.weight")),
                                        L.ptr(flat.grad("encoder.bn.bias")), L.ptr(bufs["head_ws"]),
                                        bufs["head_ws"].numel() * 4, st), "sat_fc_bn1d_bwd")
        self.last_d_features = bufs["d_feat"]
        # this step's encoder outputs (views of step-owned buffers, overwritten by the next step): the parity tests at the
        # benchmarked configuration read them
        self.last_pooled, self.last_features = pooled, feats_in
        # the step's fault flag (one 1-thread launch): OR of the status words of this step's persistent LSTM launches and of the
        # engine's sticky word, into slot 1 of the trailing floats -- it rides the last bucket's all-reduce, and clamp + Adam
        # read it on the device (optimizer_step): a step that any rank lost never reaches the parameters of any rank
        words, nw = bufs["fault_words"]
        L.check(lib.sat_step_fault_flag(words, nw, L.ptr(self._fault_sticky), L.ptr(self.fault_slot), st), "sat_step_fault_flag")
        if on_bucket_ready is not None:
            on_bucket_ready(2)   # encoder head + embedding gradients (and the loss slot) are final
        return loss_slot

    def check_ids(self):
        """Block until the device-side verdicts of the last submitted batch are known: raises IndexError on a bad caption id,
        RuntimeError when a persistent LSTM recurrence gave up (`watch.ResidencyWatch`; see `optimizer_step`)."""
        self.model.decoder.id_guard().poll(block=True)
        self._retire_fixed_lag(0)          # data-parallel steps: every flag still in flight, oldest first (same order on every rank)
        from .watch import ResidencyWatch
        ResidencyWatch.get(self.model.decoder.linear.weight.device).poll(block=True)

    FAULT_NOTE = ("the parameter update of that step and of every step submitted since was SKIPPED on the device (on every rank in "
                  "data-parallel training, where every rank raises this at the same step): parameters and Adam moments are exactly "
                  "those before the faulted step and the step count has been rolled back; repeat the batches.  What is NOT rolled "
                  "back: the BatchNorm running statistics and num_batches_tracked (the frozen stack's and the head's) have seen the "
                  "dropped batches once already")

    # data-parallel steps look at step k - DP_FAULT_LAG's (all-reduced, hence rank-identical) fault flag right after submitting step
    # k, BLOCKING: every rank then raises at the same step, with the same steps dropped.  A lag of 2 keeps the host up
    # to two steps ahead of the device, which is all the look-ahead needs.
    DP_FAULT_LAG = 2

    def _on_fault(self, step_before):
        """host side of a raised fault: per-step launches from now on, step count back to where the first dropped update found
        it (the updates in between were dropped on the device), sticky word cleared so that updates resume"""
        self.lib.sat_lstm_persist_enable(0)
        self.step_count = step_before
        self._fault_sticky.zero_()

    def _watch_fixed_lag(self, before, lag):
        """The host half of the fault path for DATA-PARALLEL steps.  The device half is rank-consistent by construction (the flag
        rides the all-reduce); the host half must be too: with the non-blocking poll of the single-rank path, ranks could see
        the flag one step apart, the early one would clear its sticky word, and the late one's next update -- no longer dropped --
        would mix a retried batch with a later one while both rolled their step counts back.  Here every rank copies
        the step's REDUCED flag to pinned memory and, after submitting step k, waits for the copy of step k - lag: same flag, same
        point in the loop, same decision on every rank.  While the faulted rank's sticky word is set, every rank's reduced flag is
        non-zero, so the steps between the fault and the raise are dropped everywhere; the raise rolls the step count back to the
        faulted step's on every rank and clears the sticky word."""
        if getattr(self, "_dp_host", None) is None:
            self._dp_host = torch.zeros(8, dtype=torch.int32).pin_memory()
            self._dp_pending, self._dp_slot = [], 0
        slot = self._dp_slot
        self._dp_slot = (slot + 1) % 8
        self._dp_host[slot:slot + 1].copy_(self.fault_slot.view(torch.int32), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._dp_pending.append((slot, ev, before))
        self._retire_fixed_lag(lag)

    def _retire_fixed_lag(self, keep):
        pend = getattr(self, "_dp_pending", None)
        while pend and len(pend) > keep:
            slot, ev, before = pend.pop(0)
            ev.synchronize()
            code = int(self._dp_host[slot])
            if code != 0:
                torch.cuda.synchronize()
                pend.clear()
                self._dp_host.zero_()
                self._on_fault(before)
                raise RuntimeError("show-and-tell_amd: a persistent LSTM recurrence of a training step timed out waiting for its "
                                   "workgroups to be resident together (reduced fault flag 0x%x: on this rank or another); %s.  "
                                   "Later calls use the form without a device-wide wait" % (code & 0xffffffff, self.FAULT_NOTE))

    def optimizer_step(self, lr=None, fixed_lag=None):
        """clip_gradient + Adam (train.py:145-146) as one launch over the flat buffers, GUARDED on the device by the step's
        fault flag: when a persistent LSTM launch of this step (any rank's, after the all-reduce) gave up waiting for its
        workgroups, the launch drops the whole update -- the garbage gradients never reach parameters or moments -- and so does
        every later one until the host has seen the flag (sticky), raised RuntimeError (at the latest on the next submit;
        `check_ids()` at once) and rolled the step count back.  fixed_lag (data-parallel steps): look at the flag of the step
        submitted `fixed_lag` steps ago, blocking, instead of polling (`_watch_fixed_lag`)."""
        before = self.step_count
        self.step_count += 1
        f = self.flat
        L.check(self.lib.sat_clamp_adam_step_guarded(L.ptr(f.params), L.ptr(f.grads), L.ptr(f.m), L.ptr(f.v), f.n,
                                                     float(self.lr if lr is None else lr), self.betas[0], self.betas[1],
                                                     self.eps, float(self.grad_clip), self.step_count, L.ptr(self.fault_slot),
                                                     L.stream()), "sat_clamp_adam_step_guarded")
        torch.autograd.graph.increment_version(self._params_list)     # written through raw pointers: bump `_version` like torch would
        if fixed_lag is not None:
            self._watch_fixed_lag(before, int(fixed_lag))
            return
        from .watch import ResidencyWatch
        ResidencyWatch.get(f.params.device).submit(self.fault_slot.view(torch.int32), "a persistent LSTM recurrence of a training step",
                                                   lambda: self._on_fault(before), note=self.FAULT_NOTE)

    # -- optimizer checkpoint interchange (the reference's load_optimizer is an empty stub, ------
    #    train.py:60-64, and only model.state_dict() is saved, train.py:191-193) ---------------------------
    def _trainable(self):
        """(name, param) in the order train.py:55 hands them to optim.Adam: model.parameters() with requires_grad"""
        return [(n, p) for n, p in self.model.named_parameters() if p.requires_grad]

    def optimizer_state_dict(self):
        """The flat Adam state in `torch.optim.Adam.state_dict()` layout (loads into a torch Adam built over
        `filter(requires_grad, model.parameters())` and back)."""
        f, state, names = self.flat, {}, []
        for i, (name, p) in enumerate(self._trainable()):
            o, n, shape = f.slices[name]
            names.append(name)
            if self.step_count > 0:
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": f.m[o:o + n].view(shape).clone(),
                            "exp_avg_sq": f.v[o:o + n].view(shape).clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": False, "params": list(range(len(names)))}
        return {"state": state, "param_groups": [group], "param_names": names, "grad_clip": self.grad_clip}

    def load_optimizer_state_dict(self, sd):
        f, trainable = self.flat, self._trainable()
        group = sd["param_groups"][0]
        if len(group["params"]) != len(trainable):
            raise ValueError("optimizer state has %d parameters, model has %d trainable"
                             % (len(group["params"]), len(trainable)))
        self.lr, self.betas, self.eps = float(group["lr"]), tuple(group["betas"]), float(group["eps"])
        steps = set()
        f.m.zero_()
        f.v.zero_()
        for i, (name, p) in enumerate(trainable):
            st = sd["state"].get(group["params"][i])
            if st is None:
                continue
            o, n, shape = f.slices[name]
            if tuple(st["exp_avg"].shape) != shape:
                raise ValueError("optimizer state of %s has shape %s, expected %s" % (name, tuple(st["exp_avg"].shape), shape))
            f.m[o:o + n].copy_(st["exp_avg"].reshape(-1))
            f.v[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(st["step"]))
        if len(steps) > 1:
            raise ValueError("per-parameter Adam step counts differ (%s): one flat step count is kept" % sorted(steps))
        self.step_count = steps.pop() if steps else 0

    # -- single-GPU convenience ----------------------------------------------------------------------
    def step(self, images, captions, lengths, lr=None, next_images=None):
        """One whole iteration; returns the mean-CE loss as a 1-element device tensor (no host sync).  next_images (the
        following batch's images, optional) starts that batch's frozen conv stack under this one's decoder work."""
        n_tokens = sum(int(l) - 1 for l in lengths)
        loss = self.forward_backward((images, captions, lengths), 1.0 / n_tokens, next_images=next_images).clone()
        self.optimizer_step(lr)
        return loss


def dp_shard(images, captions, lengths, rank, world):
    """Interleaved shard of a length-sorted global batch: rank r takes rows r, r+world, ...  Every shard stays
    sorted by decreasing length (pack_padded_sequence's requirement) and shards are balanced in tokens.
    Returns (images_r, captions_r, lengths_r, global_tokens) with global_tokens = sum(lengths-1) over ALL rows."""
    idx = list(range(rank, len(lengths), world))
    sel = torch.as_tensor(idx, device=captions.device)
    return (images.index_select(0, sel.to(images.device)), captions.index_select(0, sel), [lengths[i] for i in idx],
            sum(int(l) - 1 for l in lengths))


def decode_shard(n, rank, world):
    """Decode (eval.py:93-99 `model.sample`) shards by image with no exchange step: rank r owns the contiguous rows
    [lo, hi) of an n-image batch (the first n % world ranks take one row more)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_decoded(ids_local, n, process_group=None):
    """Collect every rank's decoded ids (i64 [rows_r, ...]) into the full [n, ...] tensor on every rank -- only the
    result collection for the host-side id->word loop (eval.py:99-118), not part of the decode itself."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return ids_local
    world = dist.get_world_size(process_group)
    rows = max(decode_shard(n, r, world)[1] - decode_shard(n, r, world)[0] for r in range(world))
    pad = ids_local.new_zeros((rows,) + tuple(ids_local.shape[1:]))
    pad[:ids_local.shape[0]] = ids_local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=process_group)
    return torch.cat([parts[r][:decode_shard(n, r, world)[1] - decode_shard(n, r, world)[0]] for r in range(world)], 0)


class DataParallelStep:
    """One process per GPU; replaces `nn.DataParallel` (train.py:43-44).  Semantics:
      * the minibatch is sharded along dim 0, parameters replicated, forward/backward rank-local;
      * gradients are those of the GLOBAL mean CE: every rank scales by 1/(global token count), then one
        all-reduce(sum) per bucket over RCCL/xGMI, launched as soon as the bucket is final and overlapped with
        the remaining backward kernels; the loss rides in the last bucket;
      * clamp + Adam run AFTER the reduction (train.py:144-146 order), identically on every rank;
      * BatchNorm uses per-rank batch statistics (what nn.DataParallel replicas do as well).
    `engine` needs: .flat_grad, .buckets, .forward_backward(batch, inv_denom, on_bucket_ready), .optimizer_step(lr).
    """

    def __init__(self, engine, process_group=None):
        import torch.distributed as dist
        self.dist = dist
        self.engine = engine
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # The bucket all-reduces overlap the rest of the backward: c10d `async_op=True` as each bucket's gradients become final,
        # `wait()` before clamp + Adam.  (Measured on ONE rank, where RCCL launches no kernel -- bench.py --force-dist, round 3: any
        # DEFERRED wait costs 0.4-0.5 ms of a 4.5 ms step, whatever the form -- a communication stream of our own, one bucket or
        # three -- while waiting at once would expose the whole exchange with N > 1; those variants were removed in round 4.)
        if self.world > 1:
            self.cap_lookahead()

    LOOKAHEAD_STREAMS_DP = 2

    def cap_lookahead(self):
        """With collectives in the step, RCCL's stream is one more concurrently active hardware queue: main + three look-ahead streams +
        RCCL = five, and the fifth costs 0.45-0.55 ms per step (bench.py --force-dist on one rank, end of round 3: 14.0-14.2 k img/s with
        three stacks on three streams against 15.9 k without collectives).  So the data-parallel step spreads the (still three)
        stacks in flight over TWO side streams -- the third queues behind the first on its hardware queue and starts the moment that
        one ends: 15.05-15.1 k img/s (depth 2 on two streams: 14.85 k; tools/run_gpu_dist_ab.sh).  SAT_LOOKAHEAD_STREAMS overrides."""
        enc = getattr(getattr(self.engine, "model", None), "encoder", None)
        if enc is not None and hasattr(enc, "lookahead_depth") and not _os.environ.get("SAT_LOOKAHEAD_STREAMS"):
            enc.lookahead_streams = min(enc.lookahead_depth, self.LOOKAHEAD_STREAMS_DP)

    def step(self, batch, global_tokens, lr=None, next_images=None):
        eng, dist = self.engine, self.dist
        works = []

        def ready(i):
            if self.world > 1:
                s_, e_ = eng.buckets[i]
                works.append(dist.all_reduce(eng.flat_grad[s_:e_], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

        if next_images is not None:
            loss = eng.forward_backward(batch, 1.0 / float(global_tokens), ready, next_images=next_images)
        else:
            loss = eng.forward_backward(batch, 1.0 / float(global_tokens), ready)
        for w in works:
            w.wait()                 # every bucket is reduced before clamp + Adam read the gradients
        loss = loss.clone()          # the slot in the flat gradient buffer is overwritten by the next step
        if self.world > 1 and hasattr(eng, "DP_FAULT_LAG"):
            eng.optimizer_step(lr, fixed_lag=eng.DP_FAULT_LAG)     # rank-consistent fault handling (TrainStep._watch_fixed_lag)
        else:
            eng.optimizer_step(lr)
        return loss
