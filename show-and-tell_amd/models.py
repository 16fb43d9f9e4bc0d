"""Drop-in `models` module: `EncoderCNN` / `DecoderRNN` with the reference's constructor and forward()/sample()
signatures (`/root/reference/models.py:9-67`), every tensor op behind them a libsat_hip.so kernel.

    EncoderCNN(embed_size).forward(images f32[B,3,H,W]) -> f32[B,embed_size]              (models.py:10,25)
    DecoderRNN(embed_size, hidden_size, vocab_size, num_layers)
        .forward(features, captions i64[B,T'], lengths list[int] desc) -> f32[sum(lengths), V]   (models.py:47-54)
        .sample(features, states=None) -> i64[B,20]                                        (models.py:56-67)
    ShowAndTell(...)(images, captions, lengths) / .sample(images, state): the single-module contract the
        reference trainer uses (train.py:139, eval.py:93,99).

`state_dict()` key names equal the reference's (`resnet.*`, `bn.*`, `embed.weight`, `lstm.weight_ih_l{k}`, ...,
`linear.weight`, `linear.bias`).  Parameters are ordinary nn.Parameters with `.grad`, so the reference's
`clip_gradient` + `optim.Adam` loop works unchanged; `trainer.TrainStep` is the fused HIP replacement.
The modules run on the GPU only: there is no CPU fallback.
"""
import os

import torch
import torch.nn as nn

from . import _lib as L
from .pack import PackInfo
from .resnet import RESNET152, ConvStackProgram, ResNetStack, weights_signature
from .watch import ResidencyWatch

BN1D_MOMENTUM = 0.01   # models.py:17
BN_EPS = 1e-5


def _f32c(t, name):
    L.require_gpu(t, name)
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32" % name)
    return t.contiguous()


# ------------------------------------------------------------------------------------------------------
# encoder
class _HeadFn(torch.autograd.Function):
    """resnet.fc + BatchNorm1d (models.py:16-17,27-28) -- sat_fc_bn1d_fwd / sat_fc_bn1d_bwd."""

    @staticmethod
    def forward(ctx, pooled, w_fc, b_fc, gamma, beta, running_mean, running_var, training):
        lib = L.load()
        B, F = pooled.shape
        E = w_fc.shape[0]
        dev = pooled.device
        feats = torch.empty(B, E, device=dev)
        xhat = torch.empty(B, E, device=dev)
        rstd = torch.empty(E, device=dev)
        wsb = lib.sat_fc_bn1d_ws_bytes(B, F, E)
        ws = torch.empty(wsb // 4, device=dev)
        L.check(lib.sat_fc_bn1d_fwd(L.ptr(pooled), L.ptr(w_fc), L.ptr(b_fc), L.ptr(gamma), L.ptr(beta),
                                    L.ptr(running_mean), L.ptr(running_var), BN1D_MOMENTUM, BN_EPS,
                                    1 if training else 0, B, F, E, L.ptr(feats), L.ptr(xhat), L.ptr(rstd),
                                    L.ptr(ws), wsb, L.stream()), "sat_fc_bn1d_fwd")
        ctx.save_for_backward(pooled, xhat, rstd, gamma)
        ctx.training = training
        return feats

    @staticmethod
    def backward(ctx, dy):
        if not ctx.training:
            raise RuntimeError("EncoderCNN backward is only defined in training mode (batch statistics)")
        lib = L.load()
        pooled, xhat, rstd, gamma = ctx.saved_tensors
        B, F = pooled.shape
        E = gamma.shape[0]
        dev = pooled.device
        dy = dy.contiguous()
        dw = torch.empty(E, F, device=dev)
        db = torch.empty(E, device=dev)
        dg = torch.empty(E, device=dev)
        dbe = torch.empty(E, device=dev)
        ws = torch.empty(B * E, device=dev)
        L.check(lib.sat_fc_bn1d_bwd(L.ptr(dy), L.ptr(pooled), L.ptr(xhat), L.ptr(rstd), L.ptr(gamma), B, F, E,
                                    L.ptr(dw), L.ptr(db), L.ptr(dg), L.ptr(dbe), L.ptr(ws), B * E * 4, L.stream()),
                "sat_fc_bn1d_bwd")
        return None, dw, db, dg, dbe, None, None, None


class _BN1d(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.zeros((), dtype=torch.long))


_LOOKAHEAD_STREAMS = {}


def lookahead_stream(device, i):
    """Side stream i of `device` for look-ahead work, ONE set per process: HIP multiplexes streams onto a few hardware queues
    (4 by default), so per-model streams end up sharing a queue with the main stream from the second model on -- measured: the
    second model of a process lost the whole look-ahead gain (12.0 k -> 9.8 k img/s) until the streams became process-wide."""
    key = (torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device(), i)
    st = _LOOKAHEAD_STREAMS.get(key)
    if st is None:
        st = _LOOKAHEAD_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


class EncoderCNN(nn.Module):
    """models.py:9-29.  `arch`/`compute_dtype` are build extensions (defaults = the reference: ResNet-152).
    compute_dtype: 'bf16' (MFMA bf16, f32 accumulate; throughput) or 'f32' (exact-f32 MFMA; parity)."""

    def __init__(self, embed_size, arch=RESNET152, compute_dtype="bf16"):
        super().__init__()
        if arch == "inception_v3":                          # BASELINE configs[3]; the attribute keeps the reference's name
            from .inception import InceptionStack
            self.resnet = InceptionStack(embed_size)
        else:
            self.resnet = ResNetStack(embed_size, arch)     # frozen stack + trainable fc (models.py:13-16)
        self.bn = _BN1d(embed_size)                         # models.py:17
        self.compute_dtype = compute_dtype
        # GROUPED look-ahead (ResNet stacks, bf16, train mode): `lookahead_groups` = G later batches run as ONE op program
        # (`ConvStackProgram(groups=G)`: every launch covers G batches, per-batch BatchNorm statistics, results bit-identical per
        # batch) -- half the launch boundaries and twice the workgroups per launch of the frozen stack.  1 = one program per batch.
        env_g = os.environ.get("SAT_LOOKAHEAD_GROUPS")
        arch_name = arch if isinstance(arch, str) else ""
        self.lookahead_groups = max(1, int(env_g)) if env_g else self.LOOKAHEAD_GROUPS_BY_ARCH.get(arch_name, self.LOOKAHEAD_GROUPS)
        env = os.environ.get("SAT_LOOKAHEAD_DEPTH")
        self.lookahead_depth = int(env) if env else self.LOOKAHEAD_DEPTH.get(arch if isinstance(arch, str) else "",
                                                                              3 * self.lookahead_groups)
        # side streams the stacks in flight are spread over (round robin in prefetch order); None = one per stack in flight.  Fewer
        # streams than stacks queue a later batch's stack BEHIND an earlier one's on the same hardware queue (it starts the moment
        # that one ends, without waiting for the host) -- for steps that need a hardware queue for something else (RCCL: trainer.py)
        env_s = os.environ.get("SAT_LOOKAHEAD_STREAMS")
        self.lookahead_streams = int(env_s) if env_s else None
        self._pf_seq = 0
        self._programs = {}      # insertion-ordered: least recently used first (`_program` re-inserts on a hit)
        self._lead_sigs = {}     # (shape, mode, weights): BatchNorm statistics signatures of the first program built (`_program`)
        self._inflight = []      # look-ahead (prefetch): [dict(images, taken, inst, ev, prog, sig, vers)], one per program run in flight
        self.register_load_state_dict_post_hook(lambda m, k: m._invalidate())
        # `encoder.resnet.load_state_dict(torchvision_sd)` -- the natural way to load the pretrained ResNet-152 the
        # reference uses (models.py:13) -- does not fire the parent's hook: hook the stack too
        self.resnet.register_load_state_dict_post_hook(lambda m, k: self._invalidate())

    def _invalidate(self):
        """cached op programs AND batches in flight belong to the old weights / device / mode"""
        self._programs.clear()
        self._lead_sigs.clear()
        self.resnet.__dict__.pop("_sig_params", None)       # weights_signature's cached parameter list
        if self._inflight:
            self.drop_lookahead()

    def train(self, mode=True):
        if bool(mode) != self.training and getattr(self, "_inflight", None):
            self.drop_lookahead()           # a stack prefetched with batch statistics must not feed an eval-mode forward
        return super().train(mode)

    def init_weights(self):
        """models.py:20-23."""
        self.resnet.fc.weight.data.normal_(0.0, 0.02)
        self.resnet.fc.bias.data.fill_(0)

    def _apply(self, fn, *a, **k):
        self._invalidate()          # device / dtype moves invalidate cached device pointers
        return super()._apply(fn, *a, **k)

    def _program(self, images, instance=None, groups=1):
        """instance None: the program `forward` runs (updates running statistics itself).  instance 0, 1, ...: independent
        copies (own activations, statistics accumulators, graphs) with DEFERRED running-statistics updates, for batches in
        flight next to each other on side streams (TrainStep.prefetch_encoder); groups > 1: a copy that runs that many batches
        per launch (`prefetch_many`)."""
        N, _, H, W = images.shape
        dt = L.SAT_BF16 if self.compute_dtype == "bf16" else L.SAT_F32
        key = (N, H, W, dt, self.training, str(images.device), weights_signature(self.resnet), instance, groups)
        prog = self._programs.pop(key, None)
        if prog is not None:
            self._programs[key] = prog                      # most recently used last
        else:
            # room for BOTH modes of one shape with every look-ahead instance (the reference alternates train and eval,
            # train.py:157-159: None + depth instances each), plus a partial last batch; evict the least recently used one
            # program at a time -- a program is ~0.9 GB of activations and its captured graphs at cfg 2
            cap = 2 * (2 * self._n_slots() + 2) + 2
            while len(self._programs) >= cap:
                old = next(iter(self._programs))
                if any(e["prog"] is self._programs[old] for e in self._inflight):
                    break                                   # never evict a program with a batch in flight
                del self._programs[old]
            make = getattr(self.resnet, "program", None)           # a stack with its own op program (Inception-v3)
            # Every program of one (shape, mode, weights) runs its convs on kernel variants of the SAME BatchNorm statistics
            # signature (tile shape / summation order), so a batch gets bit-identical features whichever program runs it.  The
            # first program built tunes freely and leads; with the grouped look-ahead on that should be a grouped one (the
            # tile shapes that win at `lookahead_groups` batches per launch are not the ones that win at one), so build it
            # first when an ungrouped program is asked for
            lead = key[:7]
            grouped_la = dt == L.SAT_BF16 and self.lookahead_depth > 0 and self.lookahead_groups > 1
            if lead not in self._lead_sigs and groups == 1 and grouped_la:
                self._program(images, instance="g0", groups=self.lookahead_groups)
            if make is not None:                         # a stack with its own op program (Inception-v3)
                prog = make(N, H, W, dt, self.training, images.device, groups=groups, signatures=self._lead_sigs.get(lead))
            else:
                prog = ConvStackProgram(self.resnet, N, H, W, dt, self.training, images.device, groups=groups,
                                        signatures=self._lead_sigs.get(lead))
            if lead not in self._lead_sigs:
                self._lead_sigs[lead] = prog.signatures()
            self._programs[key] = prog
            if instance is not None:
                prog.defer_running_stats()
        return prog

    # -- look-ahead: the frozen stack of LATER batches on side streams ---------------------------------------------------
    # stacks in flight.  Measured on MI355X at batch 64, per stack: 5.68 ms alone, 4.40 with two in flight, 4.33 with three (with
    # 8 hardware queues; 4.86 with HIP's default 4, where the third stream shares a queue)
    # whole ResNet-152 step: 5.25 ms at depth 2, 5.10 at depth 3, 5.38 at depth 4; Inception-v3 299x299: 5.68 at 2, 5.98 at 3
    # Inception-v3 (configs[3]): THREE batches per launch, two runs in flight -- its launches are smaller than ResNet-152's (94 convs, many
    # on 17 x 17 and 8 x 8 maps), so they gain more from a third batch per launch, and its step is bound by the decoder (two LSTM layers
    # of hidden 1024), which leaves the Infinity Cache room for six batches of activations (round 5, one box, interleaved,
    # tools/run_gpu_inception_g.sh: G = 2 / depth 4 14152 img/s, conv launches 0.132 of peak; G = 3 / depth 6 14368, 0.143; G = 4 / depth 8
    # 14097, 0.150; G = 4 / depth 4 12622)
    LOOKAHEAD_DEPTH = {"inception_v3": 6}
    LOOKAHEAD_GROUPS = 2
    LOOKAHEAD_GROUPS_BY_ARCH = {"inception_v3": 3}

    # Look-ahead RUN SLOTS: at most `lookahead_depth // lookahead_groups` op-program runs are in flight (3 by default: more than
    # three side streams beside the main one cost a hardware queue), each on its slot's side stream; a slot holds a
    # grouped run (instance "g<k>", `lookahead_groups` batches) or a single one (instance k)
    def _n_slots(self):
        return max(1, self.lookahead_depth // max(1, self.lookahead_groups))

    def _slot_stream(self, device, slot):
        """side stream of run slot `slot`: one per slot, or -- `lookahead_streams` fewer than slots -- round robin in prefetch
        order (a later run then queues behind an earlier one on its hardware queue)"""
        n_slots = self._n_slots()
        n_streams = self.lookahead_streams or n_slots
        idx = slot if (n_streams >= n_slots and slot < n_slots) else self._pf_seq % n_streams
        self._pf_seq += 1
        return lookahead_stream(device, idx)

    def _free_slot(self, extra=False):
        busy = {e["slot"] for e in self._inflight}
        for k in range(self._n_slots() + (1 if extra else 0)):
            if k not in busy:
                return k
        return None

    def _is_in_flight(self, images):
        # (a tensor already consumed from a group that is only partly taken is NOT in flight any more: a staging buffer refilled
        # and handed in again must start a new run instead of silently returning False -- ADVICE r4)
        return any(im is images and not e["taken"][g] for e in self._inflight for g, im in enumerate(e["images"]))

    def _launch(self, ims, slot, groups):
        dev = ims[0].device
        inst = ("g%d" % slot) if groups > 1 else slot
        stream = self._slot_stream(dev, slot)
        stream.wait_stream(torch.cuda.current_stream(dev))             # the images, and this instance's previous consumers
        for im in ims:
            ready = getattr(im, "_sat_ready_event", None)              # a DevicePrefetcher copy still in flight on its own stream
            if ready is not None:
                stream.wait_event(ready)
        with torch.cuda.stream(stream), torch.no_grad():
            prog = self._program(ims[0], instance=inst, groups=groups)
            prog.run(ims if groups > 1 else ims[0])
            for im in ims:
                im.record_stream(stream)                               # the side stream reads the tensor: the allocator must know
            ev = torch.cuda.Event()
            ev.record(stream)
        self._inflight.append(dict(images=list(ims), taken=[False] * len(ims), slot=slot, ev=ev, prog=prog,
                                   sig=weights_signature(self.resnet), vers=[im._version for im in ims]))

    def prefetch(self, images, _extra=False):
        """Start the conv stack (frozen, `no_grad`: models.py:14-15, 25-27) of a LATER batch on a side stream.  Its pooled
        features depend on the images and the frozen weights only, not on the optimizer steps in between, so computing them
        early changes nothing but the schedule: up to three stacks run next to each other (one's HBM-bound
        BatchNorm passes and under-filled launches under the other's convs) and under the current batch's head / decoder /
        backward / optimizer.  Each batch keeps its own BatchNorm batch statistics (separate program instances); the model's
        running statistics are updated when the batch is consumed, i.e. in batch order.  `forward(images)` /
        `pooled_features(images)` / `TrainStep.step(images, ...)` of the SAME tensor object later picks the result up.
        Returns False (and does nothing) when the tensor is already in flight or every run slot is taken."""
        if images is None or images.dim() != 4 or self._is_in_flight(images):
            return False
        slot = self._free_slot(extra=_extra)
        if slot is None:
            return False
        L.require_gpu(images, "images")
        self._launch([images], slot, 1)
        return True

    def prefetch_many(self, images_list, last=None, own_stack=False):
        """`prefetch` for the next few batches IN ORDER.  With `lookahead_groups` = G > 1 (bf16) G batches that are not in flight
        yet start together as ONE grouped program (every launch of the stack covers G batches; each batch's statistics and
        features are bit for bit those of its own ungrouped run); a batch left over is started alone when it is the very next
        one or at the END OF THE DATA (no partner will come).  `last`: True / False says explicitly whether the list reaches the
        end of the data (`DevicePrefetcher.upcoming_images()` sets it on the list it returns); None infers it from a list shorter
        than the look-ahead window -- right for callers that always hand in `lookahead_depth` batches when they have them
        (`bench.py`), wrong for a prefetcher of smaller depth, which would start a single per step and never a group.
        `own_stack`: the caller runs the CURRENT batch's stack itself right after this call (`TrainStep` when that batch was not
        prefetched: the first step of a loop).  Returns the number of batches started."""
        if last is None:
            last = getattr(images_list, "last", None)
        ims = [im for im in images_list if im is not None and im.dim() == 4]
        G = self.lookahead_groups if self.compute_dtype == "bf16" else 1
        started = 0
        new = [im for im in ims if not self._is_in_flight(im)]
        # COLD START (nothing in flight, and the caller is about to run the current batch's own stack beside what starts here): that
        # stack takes one of the pipeline's places, so one grouped run fewer starts now -- the next call starts it, behind the first
        # hand-over.  With all `n_slots` runs AND the own stack started together, four stacks share the chip, all finish at the same
        # moment (21 ms into a region, profiles/r05_step_timeline.txt) and the refills then trickle in one decoder step apart: 18.45-18.53
        # -> 18.72-18.80 k img/s at the contract's 20-step regions (interleaved on one box; one run fewer still: no better than before)
        cap = max(1, self._n_slots() - 1) if (own_stack and not self._inflight) else None
        if G > 1:
            while len(new) >= G:
                if cap is not None and started >= cap * G:
                    return started                                      # (the rest starts with the next call)
                grp = new[:G]
                if len({tuple(im.shape) for im in grp}) != 1 or len({id(im) for im in grp}) != G:
                    break                                               # ragged last batch / the same tensor twice: singles below
                slot = self._free_slot()
                if slot is None:
                    break
                for im in grp:
                    L.require_gpu(im, "images")
                self._launch(grp, slot, G)
                new = new[G:]
                started += G
            # a batch left without a partner starts alone when it is the very next one (it must not wait: one slot beyond the
            # regular ones is its), or when the caller's list is shorter than the look-ahead window -- the end of the data: no
            # partner will come, and started now its stack runs beside the last groups instead of alone behind them
            tail = bool(last) if last is not None else len(ims) < self.lookahead_depth
            for im in new:
                if im is ims[0]:
                    started += 1 if self.prefetch(im, _extra=True) else 0
                elif tail:
                    started += 1 if self.prefetch(im) else 0
            return started
        for im in new:
            started += 1 if self.prefetch(im) else 0
        return started

    def build_lookahead(self, images):
        """Build (autotune, capture the hipGraphs of) every op program the look-ahead of batches shaped like `images` will use,
        now instead of inside the first steps: each look-ahead instance runs four times on `images` (both statistics parities,
        eager then captured).  Touches no model state: look-ahead instances keep their running-statistics updates deferred, and
        nothing here applies them."""
        L.require_gpu(images, "images")
        G = self.lookahead_groups if self.compute_dtype == "bf16" else 1
        progs = []
        with torch.no_grad():
            if G > 1:
                progs += [(self._program(images, instance="g%d" % k, groups=G), [images] * G) for k in range(self._n_slots())]
                progs += [(self._program(images, instance=0), images)]      # the single a left-over batch runs on
            else:
                progs += [(self._program(images, instance=k), images) for k in range(self._n_slots())]
            for prog, arg in progs:
                for _ in range(4):
                    prog.run(arg)
        torch.cuda.current_stream(images.device).synchronize()

    def _take_prefetched(self, images):
        """(program instance, group index) of a prefetched `images` whose run has finished (the current stream now waits for
        it), or None.  The caller reads `prog.pooled_of(g)` and then calls `prog.apply_running_stats(g)` -- both on the current
        stream."""
        for k, e in enumerate(self._inflight):
            for g, im in enumerate(e["images"]):
                if im is images and not e["taken"][g]:
                    e["taken"][g] = True
                    if all(e["taken"]):
                        del self._inflight[k]
                    torch.cuda.current_stream(images.device).wait_event(e["ev"])
                    # conv weights rewritten since (version counters): the stack in flight used the old ones -> recompute;
                    # the same for the IMAGES: a staging buffer refilled in place (copy_, normal_, ...) between prefetch and
                    # forward is the same tensor object with other contents (its version counter moved)
                    ok = e["sig"] == weights_signature(self.resnet) and e["vers"][g] == images._version
                    return (e["prog"], g) if ok else None
        return None

    def drop_lookahead(self):
        """Forget batches in flight (their results are discarded; the model's running statistics never see them)."""
        for e in self._inflight:
            e["ev"].synchronize()
        self._inflight = []

    def refresh_weights(self):
        """Drop the cached op programs (and their kernel-layout weight copies); needed only after writing conv weights
        through `.data`, which no version counter sees."""
        self._invalidate()

    def _pooled_raw(self, images):
        """The program-owned pooled buffer (overwritten by the next forward of the same shape): internal use within
        one step only (`TrainStep`)."""
        L.require_gpu(images, "images")
        with torch.no_grad():
            return self._program(images).run(images)

    def pooled_features(self, images):
        """conv stack + global average pool: f32 [B, 2048] (no autograd: the stack is frozen, models.py:14-15).
        Returns a tensor the caller owns (a copy of the program's output buffer, B x 2048 f32)."""
        hit = self._take_prefetched(images)
        if hit is not None:
            prog, g = hit
            out = prog.pooled_of(g).clone()
            prog.apply_running_stats(g)             # batch order = consumption order
            return out
        return self._pooled_raw(images).clone()

    def forward(self, images):
        """Extract the image feature vectors (models.py:25-29)."""
        # the head's backward needs `pooled`: it must not alias the program's buffer, which a second forward (two
        # micro-batches before one backward, or model(images) followed by model.sample(images)) would overwrite
        pooled = self.pooled_features(images) if (torch.is_grad_enabled() or self._inflight) else self._pooled_raw(images)
        out = _HeadFn.apply(pooled, self.resnet.fc.weight, self.resnet.fc.bias, self.bn.weight, self.bn.bias,
                            self.bn.running_mean, self.bn.running_var, self.training)
        if self.training:
            L.counter_add(self.bn.num_batches_tracked)
        return out


# ------------------------------------------------------------------------------------------------------
# decoder
def _watch_lstm(dev, ws, offset):
    """The persistent LSTM recurrence (`sat_lstm_persist.hip`) needs all its workgroups resident at once and bounds every
    hand-off wait; when a wait runs out the kernel sets the STATUS WORD of its workspace and drains -- the tapes and `HS` of that
    call are garbage.  `watch.ResidencyWatch` reads the word back behind every call and raises RuntimeError (at the latest one
    call later) after switching the process to one launch per step (`sat_lstm_persist_enable(0)`), which needs no co-residency.
    ws: the uint8 workspace the call just ran with; offset: sat_lstm_fwd_status_offset / sat_lstm_bwd_status_offset."""
    ResidencyWatch.get(dev).submit(ws[offset:offset + 4].view(torch.int32), "the persistent LSTM recurrence",
                                   lambda: L.load().sat_lstm_persist_enable(0))


_WS_CACHE = {}


def _persistent_ws(dev, nbytes, tag):
    """A uint8 workspace that the same (device, stream, size, role) gets again on every call: the persistent backward recurrence tags
    its exchange granules per call instead of clearing 17 MB per step.  The invariant behind that -- no foreign bit pattern in the
    exchange region -- is the LIBRARY's (include/sat_hip.h, sat_lstm_bwd_ws_bytes_full: it clears a buffer the first time it sees
    its address); all this side owes it is `sat_lstm_ws_release` when a buffer goes away."""
    key = (str(dev), int(nbytes), tag, torch.cuda.current_stream(dev).cuda_stream)
    ws = _WS_CACHE.get(key)
    if ws is None:
        if len(_WS_CACHE) >= 16:
            old = _WS_CACHE.pop(next(iter(_WS_CACHE)))
            L.load().sat_lstm_ws_release(old.data_ptr())
        ws = _WS_CACHE[key] = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=dev)
    return ws


class IdGuard:
    """Out-of-range caption ids.  `nn.Embedding` (models.py:49) and `nn.CrossEntropyLoss` (train.py:143) raise on an id
    outside [0, V); the gather / CE kernels here clamp such ids for memory safety only, so every batch is range-checked
    on the device (`sat_validate_ids`, one tiny launch) and the verdict is read back WITHOUT stalling the stream: the
    status word is copied to pinned host memory behind the check and looked at when the next batch is submitted (or
    at once with `poll(block=True)`).  A corrupt caption therefore raises at the latest one step later."""

    DEPTH = 4     # verdicts in flight: the host may run this many submits ahead of the GPU without waiting

    def __init__(self, device):
        self.status = torch.zeros(1, dtype=torch.int32, device=device)       # sticky: only a raise clears it
        self.host = torch.zeros(self.DEPTH, dtype=torch.int32).pin_memory()
        self.pending = []           # (slot, event, description), oldest first
        self.slot = 0

    def submit(self, ids, ncols, V, what):
        self.poll(block=False)
        if ids.dim() != 2 or ids.dtype != torch.int64 or ids.stride(1) != 1:
            raise TypeError("%s must be an int64 matrix with contiguous rows" % what)
        while len(self.pending) >= self.DEPTH:
            self._retire(block=True)
        L.check(L.load().sat_validate_ids(ids.data_ptr(), ids.stride(0), ids.shape[0], min(int(ncols), ids.shape[1]), 0, int(V),
                                          self.status.data_ptr(), L.stream()), "sat_validate_ids")
        slot = self.slot
        self.slot = (slot + 1) % self.DEPTH
        self.host[slot:slot + 1].copy_(self.status, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.pending.append((slot, ev, "%s: id outside [0, %d)" % (what, V)))

    def _retire(self, block):
        slot, ev, what = self.pending[0]
        if block:
            ev.synchronize()
        elif not ev.query():
            return False
        self.pending.pop(0)
        if int(self.host[slot]) != 0:
            torch.cuda.current_stream().synchronize()
            self.status.zero_()
            self.host.zero_()
            self.pending.clear()
            raise IndexError("show-and-tell_amd: %s (nn.Embedding / CrossEntropyLoss would raise here: "
                             "models.py:49, train.py:143)" % what)
        return True

    def poll(self, block=False):
        while self.pending and self._retire(block):
            pass


class _LSTMParams(nn.Module):
    """Parameter holder with nn.LSTM's names and default init U(-1/sqrt(H), 1/sqrt(H)) (models.py:36)."""

    def __init__(self, input_size, hidden_size, num_layers):
        super().__init__()
        self.input_size, self.hidden_size, self.num_layers = input_size, hidden_size, num_layers
        k = 1.0 / (hidden_size ** 0.5)
        for l in range(num_layers):
            in_sz = input_size if l == 0 else hidden_size
            for name, shape in (("weight_ih", (4 * hidden_size, in_sz)), ("weight_hh", (4 * hidden_size, hidden_size)),
                                ("bias_ih", (4 * hidden_size,)), ("bias_hh", (4 * hidden_size,))):
                setattr(self, "%s_l%d" % (name, l), nn.Parameter(torch.empty(*shape).uniform_(-k, k)))

    def layer(self, l):
        return tuple(getattr(self, "%s_l%d" % (n, l)) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"))


class _Weight(nn.Module):
    def __init__(self, *shape, bias=None):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(*shape))
        if bias is not None:
            self.bias = nn.Parameter(torch.zeros(bias))


def decoder_forward_tapes(lib, features, embed_w, lstm_layers, lin_w, lin_b, captions, pi, logits=None, ce=None, mixed_ws=None,
                          lstm_ws=None):
    """embed+cat+pack -> L x LSTM -> vocab logits (models.py:49-53).  Returns (logits, tapes).
    `ce` = dict(kind="bf16", targets, inv_denom, row_loss, loss_out, ws): the projection on the bf16 matrix pipe and the cross
    entropy (train.py:143) as one call (`sat_vocab_ce_fwd_bf16`), d(loss)/d(logits) left in `ws` for the backward.
    `lstm_ws`: per layer (forward workspace, backward workspace) uint8 tensors the CALLER owns and watches (`TrainStep`: it
    folds their status words into its step's fault flag); None: workspaces of this module, each call's status word handed to
    `watch.ResidencyWatch`."""
    dev = features.device
    E = embed_w.shape[1]
    V = lin_w.shape[0]
    N, T, B = pi.N, pi.T, pi.B
    st = L.stream()
    X = torch.empty(N, E, device=dev)
    cap_ptr, cap_stride = (None, 0)
    if T > 1:
        if captions.dtype != torch.int64 or captions.stride(1) != 1:
            captions = captions.long().contiguous()
        if captions.shape[1] < T - 1:
            raise ValueError("captions has %d columns but lengths need %d" % (captions.shape[1], T - 1))
        cap_ptr, cap_stride = captions.data_ptr(), captions.stride(0)
    L.check(lib.sat_embed_concat_fwd(L.ptr(features), L.ptr(embed_w), cap_ptr, cap_stride, L.ptr(pi.prefix_dev),
                                     T, N, B, E, embed_w.shape[0], L.ptr(X), st), "sat_embed_concat_fwd")
    tapes = {"X": [X], "layers": [], "captions": captions}
    inp = X
    for (w_ih, w_hh, b_ih, b_hh) in lstm_layers:
        H = w_hh.shape[1]
        In = w_ih.shape[1]
        GA = torch.empty(N, 4 * H, device=dev)
        CS = torch.empty(N, H, device=dev)
        HS = torch.empty(N, H, device=dev)
        HP = torch.empty(N, H, device=dev)
        cst = torch.empty(B, H, device=dev)
        wsb = lib.sat_lstm_fwd_ws_bytes(B, H)            # hidden-state exchange of the persistent recurrence
        li = len(tapes["layers"])
        ws = lstm_ws[li][0] if lstm_ws is not None else torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
        if mixed_ws is not None:          # bf16 throughput mode: the x-gates GEMM on the bf16 matrix pipe
            L.check(lib.sat_lstm_fwd_bf16(L.ptr(inp), L.ptr(w_ih), L.ptr(w_hh), L.ptr(b_ih), L.ptr(b_hh), pi.bs_c, T, In, H,
                                          L.ptr(GA), L.ptr(CS), L.ptr(HS), L.ptr(HP), L.ptr(cst), L.ptr(ws), wsb, L.ptr(mixed_ws),
                                          mixed_ws.numel(), st), "sat_lstm_fwd_bf16")
        else:
            L.check(lib.sat_lstm_fwd(L.ptr(inp), L.ptr(w_ih), L.ptr(w_hh), L.ptr(b_ih), L.ptr(b_hh), pi.bs_c, T, In, H,
                                     L.ptr(GA), L.ptr(CS), L.ptr(HS), L.ptr(HP), L.ptr(cst), L.ptr(ws), wsb, st), "sat_lstm_fwd")
        soff = lib.sat_lstm_fwd_status_offset(B, H)
        if soff >= 0 and wsb > 0 and lstm_ws is None:
            _watch_lstm(dev, ws, soff)                   # the recurrence's status word: raises (at the latest one call later)
        tapes["layers"].append((GA, CS, HP))
        tapes["X"].append(HS)
        inp = HS
    if logits is None:
        logits = torch.zeros(N, (V + 3) // 4 * 4, device=dev) if V % 4 else torch.empty(N, V, device=dev)
    if ce is not None and ce.get("kind") == "bf16":
        # bf16 throughput mode: the projection on the bf16 matrix pipe (f32 accumulate, f32 logits), CE in f32 from them,
        # d(loss)/d(logits) left as bf16 in the workspace for decoder_backward_tapes
        L.check(lib.sat_vocab_ce_fwd_bf16(L.ptr(inp), L.ptr(lin_w), L.ptr(lin_b), L.ptr(ce["targets"]), N, lin_w.shape[1], V,
                                          float(ce["inv_denom"]), L.ptr(logits), logits.stride(0), L.ptr(ce["row_loss"]),
                                          L.ptr(ce["loss_out"]), L.ptr(ce["ws"]), ce["ws"].numel(), st), "sat_vocab_ce_fwd_bf16")
        return logits, tapes
    L.check(lib.sat_vocab_logits_fwd(L.ptr(inp), L.ptr(lin_w), L.ptr(lin_b), N, lin_w.shape[1], V, L.ptr(logits),
                                     logits.stride(0), st), "sat_vocab_logits_fwd")
    return logits, tapes


def decoder_backward_tapes(lib, dlogits, tapes, embed_w, lstm_layers, lin_w, pi, grads_out, on_stage=None, ce=None, mixed_ws=None,
                           lstm_ws=None):
    """Backward of decoder_forward_tapes.  `dlogits`: f32 [N, ld] with ld = V rounded up to 4 and zero pad columns.  grads_out: dict name -> preallocated f32 tensor to fill:
    'embed', ('w_ih',l), ('w_hh',l), ('b_ih',l), ('b_hh',l), 'lin_w', 'lin_b', 'features'.
    on_stage(i) is called when gradient group i is final (0 vocab projection, 1 LSTM) -- the data-parallel
    wrapper launches that bucket's all-reduce there, under the remaining backward kernels."""
    dev = dlogits.device
    st = L.stream()
    N, T, B = pi.N, pi.T, pi.B
    V, Hl = lin_w.shape
    Xtop = tapes["X"][-1]
    dH = torch.empty(N, Hl, device=dev)
    if ce is not None and ce.get("kind") == "bf16":
        L.check(lib.sat_vocab_ce_bwd_bf16(N, Hl, V, L.ptr(grads_out["lin_w"]), L.ptr(grads_out["lin_b"]), L.ptr(dH), L.ptr(ce["ws"]),
                                          ce["ws"].numel(), st), "sat_vocab_ce_bwd_bf16")
    else:
        vwsb = lib.sat_vocab_ce_bwd_ws_bytes(N, Hl, V)
        vws = torch.empty(max(vwsb // 4, 4), device=dev)
        L.check(lib.sat_vocab_ce_bwd(L.ptr(dlogits), dlogits.stride(0), L.ptr(Xtop), L.ptr(lin_w), N, Hl, V, L.ptr(grads_out["lin_w"]),
                                     L.ptr(grads_out["lin_b"]), L.ptr(dH), L.ptr(vws), vwsb, st), "sat_vocab_ce_bwd")
    if on_stage is not None:
        on_stage(0)
    for l in reversed(range(len(lstm_layers))):
        w_ih, w_hh, _, _ = lstm_layers[l]
        H, In = w_hh.shape[1], w_ih.shape[1]
        GA, CS, HP = tapes["layers"][l]
        DG = torch.empty(N, 4 * H, device=dev)
        dX = torch.empty(N, In, device=dev)
        # the FULL workspace (split-K weight-gradient GEMMs + the persistent backward recurrence), sized for any N <= B * T so that
        # batches of other lengths reuse it (its exchange region and status word sit at (B, H)-only offsets)
        ws = lstm_ws[l][1] if lstm_ws is not None else _persistent_ws(dev, lib.sat_lstm_bwd_ws_bytes_max(B * T, B, In, H), ("lstm_bwd", l))
        wsb = ws.numel()
        if mixed_ws is not None:
            L.check(lib.sat_lstm_bwd_bf16(L.ptr(dH), L.ptr(tapes["X"][l]), L.ptr(w_ih), L.ptr(w_hh), L.ptr(GA), L.ptr(CS),
                                          L.ptr(HP), pi.bs_c, T, In, H, L.ptr(DG), L.ptr(grads_out[("w_ih", l)]),
                                          L.ptr(grads_out[("w_hh", l)]), L.ptr(grads_out[("b_ih", l)]),
                                          L.ptr(grads_out[("b_hh", l)]), L.ptr(dX), L.ptr(ws), wsb, L.ptr(mixed_ws), mixed_ws.numel(),
                                          st), "sat_lstm_bwd_bf16")
        else:
            L.check(lib.sat_lstm_bwd(L.ptr(dH), L.ptr(tapes["X"][l]), L.ptr(w_ih), L.ptr(w_hh), L.ptr(GA), L.ptr(CS),
                                     L.ptr(HP), pi.bs_c, T, In, H, L.ptr(DG), L.ptr(grads_out[("w_ih", l)]),
                                     L.ptr(grads_out[("w_hh", l)]), L.ptr(grads_out[("b_ih", l)]),
                                     L.ptr(grads_out[("b_hh", l)]), L.ptr(dX), L.ptr(ws), wsb, st), "sat_lstm_bwd")
        if lstm_ws is None:                    # the backward recurrence may have run persistently: its status word
            _watch_lstm(dev, ws, lib.sat_lstm_bwd_status_offset(N, B, In, H))
        dH = dX
    if on_stage is not None:
        on_stage(1)
    E = embed_w.shape[1]
    caps = tapes["captions"]
    cap_ptr, cap_stride = (None, 0) if T <= 1 else (caps.data_ptr(), caps.stride(0))
    L.check(lib.sat_embed_concat_bwd(L.ptr(dH), cap_ptr, cap_stride, L.ptr(pi.prefix_dev), T, N, B, E,
                                     embed_w.shape[0], L.ptr(grads_out["embed"]), L.ptr(grads_out["features"]), st),
            "sat_embed_concat_bwd")


class _DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, captions, pi, num_layers, embed_w, lin_w, lin_b, *lstm_flat):
        lib = L.load()
        layers = [tuple(lstm_flat[4 * l:4 * l + 4]) for l in range(num_layers)]
        logits, tapes = decoder_forward_tapes(lib, features, embed_w, layers, lin_w, lin_b, captions, pi)
        ctx.tapes, ctx.pi, ctx.layers = tapes, pi, layers
        ctx.embed_w, ctx.lin_w = embed_w, lin_w
        V = lin_w.shape[0]
        return logits if logits.shape[1] == V else logits[:, :V]

    @staticmethod
    def backward(ctx, dlogits):
        lib = L.load()
        dev = dlogits.device
        V = ctx.lin_w.shape[0]
        if V % 4:                                  # rows padded to 4 floats, zero pad (sat_vocab_ce_bwd's layout)
            padded = torch.zeros(dlogits.shape[0], (V + 3) // 4 * 4, device=dev)
            padded[:, :V] = dlogits
            dlogits = padded
        else:
            dlogits = dlogits.contiguous()
        layers = ctx.layers
        g = {"embed": torch.empty_like(ctx.embed_w), "lin_w": torch.empty_like(ctx.lin_w),
             "lin_b": torch.empty(ctx.lin_w.shape[0], device=dev),
             "features": torch.empty(ctx.pi.B, ctx.embed_w.shape[1], device=dev)}
        for l, (w_ih, w_hh, b_ih, b_hh) in enumerate(layers):
            g[("w_ih", l)], g[("w_hh", l)] = torch.empty_like(w_ih), torch.empty_like(w_hh)
            g[("b_ih", l)], g[("b_hh", l)] = torch.empty_like(b_ih), torch.empty_like(b_hh)
        decoder_backward_tapes(lib, dlogits, ctx.tapes, ctx.embed_w, layers, ctx.lin_w, ctx.pi, g)
        flat = []
        for l in range(len(layers)):
            flat += [g[("w_ih", l)], g[("w_hh", l)], g[("b_ih", l)], g[("b_hh", l)]]
        return (g["features"], None, None, None, g["embed"], g["lin_w"], g["lin_b"], *flat)


class DecoderRNN(nn.Module):
    """models.py:31-67."""

    def __init__(self, embed_size, hidden_size, vocab_size, num_layers):
        super().__init__()
        self.embed = _Weight(vocab_size, embed_size)                     # nn.Embedding (models.py:35)
        self.lstm = _LSTMParams(embed_size, hidden_size, num_layers)     # nn.LSTM(batch_first) (models.py:36)
        self.linear = _Weight(vocab_size, hidden_size, bias=vocab_size)  # nn.Linear (models.py:37)
        self.embed_size, self.hidden_size, self.vocab_size, self.num_layers = embed_size, hidden_size, vocab_size, num_layers
        self.ss_prob = 0                                                 # inert in the reference too (models.py:38)
        self._id_guard = None
        self.init_weights()

    def id_guard(self):
        dev = self.embed.weight.device
        if self._id_guard is None or self._id_guard.status.device != dev:
            self._id_guard = IdGuard(dev)
        return self._id_guard

    def init_weights(self):
        """models.py:41-45."""
        self.embed.weight.data.uniform_(-0.1, 0.1)
        self.linear.weight.data.uniform_(-0.1, 0.1)
        self.linear.bias.data.fill_(0)

    def _lstm_flat(self):
        flat = []
        for l in range(self.num_layers):
            flat += list(self.lstm.layer(l))
        return flat

    def forward(self, features, captions, lengths):
        """Decode image feature vectors and generate caption logits (models.py:47-54): f32 [sum(lengths), V],
        rows in time-major packed order."""
        features = _f32c(features, "features")
        L.require_gpu(captions, "captions")
        if len(lengths) != features.shape[0]:
            raise ValueError("len(lengths) != batch size")
        pi = PackInfo.get(lengths, features.device)
        if pi.T > captions.shape[1] + 1:
            raise ValueError("a length exceeds captions.shape[1] + 1")
        if captions.dtype != torch.int64 or captions.stride(1) != 1:
            captions = captions.long().contiguous()
        if pi.T > 1:
            self.id_guard().submit(captions, pi.T - 1, self.vocab_size, "captions")
        return _DecoderFn.apply(features, captions, pi, self.num_layers, self.embed.weight, self.linear.weight,
                                self.linear.bias, *self._lstm_flat())

    @torch.no_grad()
    def sample(self, features, states=None):
        """Greedy search, 20 steps (models.py:56-67; torch-0.1 keepdim semantics): i64 [B,20].
        `states`: None (zeros), (h0, c0) each [num_layers, B, H], or eval.py:82-89's stacked [2, B, H] tensor
        (`_initial_states`); the result is squeezed like models.py:67 ([20] at batch 1)."""
        lib = L.load()
        features = _f32c(features, "features")
        dev = features.device
        B = features.shape[0]
        H, V, E = self.hidden_size, self.vocab_size, self.embed_size
        st = L.stream()
        h0, c0 = self._initial_states(states, B, dev)
        h, c = h0.clone().contiguous(), c0.clone().contiguous()         # [num_layers, B, H]: state in, state out
        h_tmp, xe = torch.empty_like(h), torch.empty(B, E, device=dev)
        ids = torch.empty(B, 20, dtype=torch.int64, device=dev)
        wsb = lib.sat_vocab_argmax_ws_bytes(B, V)
        ws = torch.empty(max(wsb // 4, 4), device=dev)
        # the 20 steps (LSTM step, vocab projection + arg-max, embedding row) are enqueued by ONE library call: the loop was
        # host-bound when every launch came through ctypes (round 4: 0.72 ms per 20 steps at batch 64)
        L.check(lib.sat_greedy_decode(L.ptr(features), L.ptr(self.embed.weight), self._lstm_ptrs(), self.num_layers,
                                      L.ptr(self.linear.weight), L.ptr(self.linear.bias), B, E, H, V, 20, L.ptr(h), L.ptr(c),
                                      L.ptr(h_tmp), L.ptr(xe), ids.data_ptr(), ids.stride(0), L.ptr(ws), wsb, st), "sat_greedy_decode")
        return ids.squeeze()                # models.py:67: [20] at batch 1

    def _lstm_ptrs(self):
        """HOST array of the LSTM's device pointers, (w_ih, w_hh, b_ih, b_hh) per layer: the `lstm_w` argument of the decode calls"""
        import ctypes as _C
        ptrs = []
        for l in range(self.num_layers):
            ptrs += [t.data_ptr() for t in self.lstm.layer(l)]
        return (_C.c_void_p * len(ptrs))(*ptrs)

    def _initial_states(self, states, B, dev):
        """The LSTM state `sample` starts from (models.py:56,61 hands `states` to nn.LSTM) as two f32 [num_layers, B, H]
        tensors.  Accepted: None (zeros); `(h0, c0)` each [num_layers, B, H] (nn.LSTM's own form) or, for one layer, [B, H];
        eval.py:82-89's stacked tensor [2, B, H] (= (h0, c0) of a one-layer LSTM) or [2, num_layers, B, H].  Anything else
        raises -- nothing is silently replaced by zeros."""
        Lh, H = self.num_layers, self.hidden_size
        if states is None:
            z = torch.zeros(2, Lh, B, H, device=dev)
            return z[0], z[1]
        if torch.is_tensor(states):
            if states.dim() not in (3, 4) or states.shape[0] != 2:
                raise ValueError("states tensor must be stacked (h0, c0): [2, B, H] or [2, num_layers, B, H], got %s"
                                 % (tuple(states.shape),))
            states = (states[0], states[1])
        if not isinstance(states, (tuple, list)) or len(states) != 2 or not all(torch.is_tensor(s) for s in states):
            raise TypeError("states must be None, a (h0, c0) pair of tensors or their stacked tensor")
        out = []
        for name, s in zip(("h0", "c0"), states):
            if s.dim() == 2 and Lh == 1:
                s = s.unsqueeze(0)
            if tuple(s.shape) != (Lh, B, H):
                raise ValueError("%s must be [num_layers=%d, B=%d, H=%d]%s, got %s"
                                 % (name, Lh, B, H, " (or [B, H])" if Lh == 1 else "", tuple(s.shape)))
            L.require_gpu(s, name)
            out.append(s.to(dtype=torch.float32).contiguous())
        return out[0], out[1]

    @torch.no_grad()
    def sample_beam(self, features, beam_size=5, end_id=None, steps=20, return_all=False):
        """Beam search over the same 20-step loop (`model2.py:113-114` is a stub in the reference).
        Returns the best hypothesis per image, i64 [B,steps]; with return_all also (ids [B,K,steps] best-first,
        scores [B,K] = sum of token log-probabilities).  beam_size=1, end_id=None is exactly `sample`.
        end_id (eval.py's `<end>`, id 2): a finished hypothesis only repeats end_id at no cost."""
        lib = L.load()
        features = _f32c(features, "features")
        dev = features.device
        B, K = features.shape[0], int(beam_size)
        if K < 1 or K > 8:
            raise ValueError("beam_size must be in 1..8")
        H, V, E = self.hidden_size, self.vocab_size, self.embed_size
        st = L.stream()
        eid = -1 if end_id is None else int(end_id)
        wsb = lib.sat_beam_decode_ws_bytes(B, K, E, H, V, self.num_layers, steps)
        ws = torch.empty(wsb + 256, dtype=torch.uint8, device=dev)
        off = (-ws.data_ptr()) % 256
        ids = torch.empty(B, K, steps, dtype=torch.int64, device=dev)
        scores = torch.empty(B, K, device=dev)
        # every step (LSTM step, exact-f32 vocab projection, per-row log-softmax + top-K, per-image merge that also gathers the next
        # input's embedding rows, one (h, c) re-ordering launch) is enqueued by ONE library call (sat_beam_decode)
        L.check(lib.sat_beam_decode(L.ptr(features), L.ptr(self.embed.weight), self._lstm_ptrs(), self.num_layers,
                                    L.ptr(self.linear.weight), L.ptr(self.linear.bias), B, K, E, H, V, int(steps), eid,
                                    ids.data_ptr(), L.ptr(scores), ws.data_ptr() + off, wsb, st), "sat_beam_decode")
        if return_all:
            return ids, scores
        return ids[:, 0].contiguous()


class ShowAndTell(nn.Module):
    """Encoder + decoder behind the reference trainer's single-module call contract (train.py:37,139;
    eval.py:93,99): `model(images, captions, lengths)` and `model.sample(images, state)`.
    state_dict keys: `encoder.*`, `decoder.*`."""

    def __init__(self, embed_size, hidden_size, vocab_size, num_layers=1, arch=RESNET152, compute_dtype="bf16"):
        super().__init__()
        self.encoder = EncoderCNN(embed_size, arch, compute_dtype)
        self.decoder = DecoderRNN(embed_size, hidden_size, vocab_size, num_layers)

    @classmethod
    def from_trainer_args(cls, hidden_size, context_size, vocab_size, embed_size, opt=None, **kw):
        """The argument order of the model the reference trainer constructs, train.py:37
        `ShowAttendTellModel(opt.hidden_size, opt.embed_size, len(vocab), opt.embed_size, opt)`: hidden FIRST, then a
        context size this architecture has no use for, vocab, embed, and the options namespace (its `num_layers`,
        config.py:30, is honoured).  Dropping `ShowAndTell.from_trainer_args` in at that call site keeps every
        dimension where the trainer put it."""
        num_layers = int(getattr(opt, "num_layers", 1)) if opt is not None else 1
        return cls(embed_size, hidden_size, vocab_size, num_layers, **kw)

    def prefetch(self, images):
        """Start the frozen conv stack of a LATER batch on a side stream (`EncoderCNN.prefetch`)."""
        return self.encoder.prefetch(images)

    def prefetch_many(self, images_list):
        """... of the next few batches, in order (`EncoderCNN.prefetch_many`: two batches per program run)."""
        return self.encoder.prefetch_many(images_list)

    def forward(self, images, captions, lengths):
        return self.decoder(self.encoder(images), captions, lengths)

    @torch.no_grad()
    def sample(self, images, state=None):
        """eval.py:93,99 `model.sample(images, state)`: the state is handed to the decoder's LSTM (models.py:56,61)."""
        return self.decoder.sample(self.encoder(images), state)

    @torch.no_grad()
    def sample_beam(self, images, beam_size=5, end_id=None, return_all=False):
        return self.decoder.sample_beam(self.encoder(images), beam_size, end_id, return_all=return_all)


Encoder = EncoderCNN      # names BASELINE.json uses
Decoder = DecoderRNN
CaptionModel = ShowAndTell
