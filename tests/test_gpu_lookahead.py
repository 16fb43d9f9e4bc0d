"""Encoder look-ahead (TrainStep.prefetch_encoder): running the frozen conv stack of batch i+1 on a side stream under batch i's
decoder work must not change a single bit of any step's loss or of the parameters (models.py:14-15, 25-27: the stack is frozen
and under no_grad, so it does not depend on the optimizer step in between)."""
import importlib

import pytest
import torch

sat = importlib.import_module("show-and-tell_amd")
pytestmark = pytest.mark.gpu


def _run(lookahead, groups=1, steps=9, B=8, img=64):
    torch.manual_seed(5)
    model = sat.ShowAndTell(32, 64, 120, 1, compute_dtype="bf16").cuda().train()
    assert model.encoder.lookahead_groups == 2 and model.encoder.lookahead_depth == 6     # ResNet default: 3 grouped runs of 2 batches
    model.encoder.lookahead_groups = groups
    model.encoder.lookahead_depth = max(lookahead, 1)
    ts = sat.TrainStep(model, lr=1e-3, grad_clip=0.1)
    g = torch.Generator().manual_seed(11)
    nb = 8
    batches = [torch.rand(B, 3, img, img, generator=g).cuda() for _ in range(nb)]
    caps = torch.randint(1, 120, (B, 9), generator=g).cuda()
    lengths = [9, 9, 8, 7, 6, 5, 4, 3]
    losses = []
    grouped_runs = 0
    for i in range(steps):
        nxt = [batches[j % nb] for j in range(i + 1, i + 1 + lookahead) if j < steps] or None
        losses.append(ts.step(batches[i % nb], caps, lengths, next_images=nxt))
        grouped_runs += sum(1 for e in model.encoder._inflight if len(e["images"]) > 1 and not any(e["taken"]) and e.get("_seen") is None)
        for e in model.encoder._inflight:
            e["_seen"] = True
    torch.cuda.synchronize()
    ts.check_ids()
    rs = torch.cat([torch.cat([bn.running_mean, bn.running_var]) for bn in model.encoder.resnet.bns()])
    nbt = next(iter(model.encoder.resnet.bns())).num_batches_tracked
    return (torch.cat(losses).cpu(), ts.flat.params.clone().cpu(), model.encoder.bn.running_mean.clone().cpu(), rs.cpu(), nbt.cpu()), grouped_runs


@pytest.mark.parametrize("groups,depth", [(1, 1), (1, 2), (1, 3), (2, 2), (2, 3), (2, 4), (2, 6), (3, 6)])
def test_lookahead_is_bitwise_identical_to_sequential(groups, depth):
    """losses, parameters, the head's and EVERY conv-stack BatchNorm's running statistics and num_batches_tracked -- with one
    program per batch in flight (groups 1) and with GROUPED programs (`ConvStackProgram(groups=G)`: every launch of the frozen
    stack covers G look-ahead batches, per-batch statistics; every program of a model runs kernel variants of the statistics
    signatures its first -- grouped -- program chose): each batch gets, bit for bit, what the same model gives it when nothing
    runs ahead"""
    a, _ = _run(0, groups)
    b, grouped_runs = _run(depth, groups)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert int(a[4]) == 9
    if groups > 1:
        assert grouped_runs >= 2, grouped_runs          # the grouped path really ran (not only single prefetches)


def test_a_prefetcher_shallower_than_the_window_still_runs_grouped_programs():
    """`prefetch_many` inferred "end of the data" from a list shorter than the look-ahead window, so a `DevicePrefetcher`
    of depth 3 under a window of 6 started one single per step and never a group.  The prefetcher now says explicitly whether the data
    ends inside the list it hands over (`upcoming_images().last`): grouped runs in steady state, singles only at the real end -- and
    the losses equal the sequential run bit for bit."""
    def run(lookahead):
        torch.manual_seed(5)
        model = sat.ShowAndTell(32, 64, 120, 1, compute_dtype="bf16").cuda().train()
        if not lookahead:
            model.encoder.lookahead_depth = 0
        ts = sat.TrainStep(model, lr=1e-3, grad_clip=0.1)
        g = torch.Generator().manual_seed(11)
        caps = torch.randint(1, 120, (8, 9), generator=g)
        lengths = [9, 9, 8, 7, 6, 5, 4, 3]
        host = [(torch.rand(8, 3, 64, 64, generator=g), caps, lengths) for _ in range(9)]
        pf = sat.DevicePrefetcher(host, "cuda", depth=3)
        losses, grouped, singles = [], 0, 0
        for im, cp, ln in pf:
            up = pf.upcoming_images() if lookahead else None
            before = {id(e) for e in model.encoder._inflight}
            losses.append(ts.step(im, cp, ln, next_images=up or None))
            for e in model.encoder._inflight:
                if id(e) not in before:
                    grouped += len(e["images"]) > 1
                    singles += len(e["images"]) == 1
        torch.cuda.synchronize()
        ts.check_ids()
        return torch.cat(losses).cpu(), grouped, singles
    seq, _, _ = run(False)
    la, grouped, singles = run(True)
    assert torch.equal(seq, la)
    assert grouped >= 3 and singles <= 3, (grouped, singles)


def test_a_cold_pipeline_starts_one_grouped_run_fewer_beside_the_own_stack():
    """the first step of a loop runs its batch's stack itself, beside the look-ahead it starts: that stack takes one of the pipeline's
    places (`prefetch_many(own_stack=True)`: n_slots - 1 grouped runs), the next step starts the rest; losses equal the sequential run"""
    def run(lookahead):
        torch.manual_seed(6)
        model = sat.ShowAndTell(32, 64, 120, 1, compute_dtype="bf16").cuda().train()
        if not lookahead:
            model.encoder.lookahead_depth = 0
        ts = sat.TrainStep(model, lr=1e-3, grad_clip=0.1)
        g = torch.Generator().manual_seed(12)
        caps = torch.randint(1, 120, (8, 9), generator=g).cuda()
        lengths = [9, 9, 8, 7, 6, 5, 4, 3]
        ims = [torch.rand(8, 3, 64, 64, generator=g).cuda() for _ in range(10)]
        depth, losses, runs_after = model.encoder.lookahead_depth, [], []
        for i, im in enumerate(ims):
            nxt = ims[i + 1:i + 1 + depth] if lookahead else None
            losses.append(ts.step(im, caps, lengths, next_images=nxt or None))
            runs_after.append(len(model.encoder._inflight))
        torch.cuda.synchronize()
        return torch.cat(losses).cpu(), runs_after
    seq, _ = run(False)
    la, runs = run(True)
    assert torch.equal(seq, la)
    assert runs[0] == 2 and runs[1] == 3, runs          # (depth 6, two batches per run: three run slots)


def test_lookahead_of_a_different_tensor_is_discarded():
    torch.manual_seed(5)
    model = sat.ShowAndTell(32, 64, 120, 1, compute_dtype="bf16").cuda().train()
    ts = sat.TrainStep(model)
    g = torch.Generator().manual_seed(11)
    x0, x1, x2 = (torch.rand(4, 3, 64, 64, generator=g).cuda() for _ in range(3))
    caps = torch.randint(1, 120, (4, 6), generator=g).cuda()
    lengths = [6, 5, 4, 3]
    ts.step(x0, caps, lengths, next_images=x1)
    got = ts.step(x2, caps, lengths)                     # NOT the prefetched batch: must be computed for x2
    torch.manual_seed(5)
    model2 = sat.ShowAndTell(32, 64, 120, 1, compute_dtype="bf16").cuda().train()
    ts2 = sat.TrainStep(model2)
    ts2.step(x0, caps, lengths)
    want = ts2.step(x2, caps, lengths)
    assert torch.equal(got.cpu(), want.cpu())


def test_dropin_encoder_prefetch_is_bitwise_identical():
    """EncoderCNN.prefetch on the drop-in path (`features = encoder(images)`, models.py:25-29): features, head gradients and every
    running statistic equal the un-prefetched sequence bit for bit, in train mode, with two batches in flight"""
    def run(prefetch):
        torch.manual_seed(9)
        enc = sat.EncoderCNN(32).cuda().train()
        g = torch.Generator().manual_seed(4)
        xs = [torch.rand(6, 3, 64, 64, generator=g).cuda() for _ in range(4)]
        outs = []
        for i, x in enumerate(xs):
            if prefetch:
                for j in (i + 1, i + 2):
                    if j < len(xs):
                        enc.prefetch(xs[j])
            f = enc(x)
            f.sum().backward()
            outs.append(f.detach().clone())
        torch.cuda.synchronize()
        rs = torch.cat([torch.cat([bn.running_mean, bn.running_var]) for bn in enc.resnet.bns()])
        return torch.stack(outs).cpu(), enc.resnet.fc.weight.grad.clone().cpu(), rs.cpu(), enc.bn.running_var.clone().cpu()
    a, b = run(False), run(True)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_mode_or_weight_change_drops_batches_in_flight():
    torch.manual_seed(9)
    enc = sat.EncoderCNN(32).cuda().train()
    g = torch.Generator().manual_seed(4)
    x = torch.rand(4, 3, 64, 64, generator=g).cuda()
    assert enc.prefetch(x) and not enc.prefetch(x)          # the same tensor is not started twice
    enc.eval()                                              # batch-statistics features must not feed an eval forward
    assert not enc._inflight
    with torch.no_grad():
        want = enc(x).clone()
        enc.train(); enc.prefetch(x); enc.eval()
        assert torch.equal(enc(x), want)
    enc.train()
    assert enc.prefetch(x)
    sd = {k: v.clone() for k, v in enc.state_dict().items()}
    enc.load_state_dict(sd)                                 # new weights: the stack in flight belongs to the old ones
    assert not enc._inflight and not enc._programs


def test_eval_mode_decode_with_prefetch_gives_the_same_ids():
    """eval.py:93-99 as a loop over batches with the next batches' stacks prefetched: greedy ids identical"""
    torch.manual_seed(5)
    model = sat.ShowAndTell(32, 64, 120, 1, compute_dtype="bf16").cuda().eval()
    g = torch.Generator().manual_seed(12)
    xs = [torch.rand(4, 3, 64, 64, generator=g).cuda() for _ in range(3)]
    with torch.no_grad():
        want = [model.sample(x).clone() for x in xs]
        got = []
        for i, x in enumerate(xs):
            for j in (i + 1, i + 2):
                if j < len(xs):
                    model.prefetch(xs[j])
            got.append(model.sample(x).clone())
    torch.cuda.synchronize()
    for a, b in zip(want, got):
        assert torch.equal(a, b)


def test_eval_mode_grouped_prefetch_concatenates_batches_and_gives_the_same_ids():
    """eval.py:93-99 with `prefetch_many`: in eval mode BatchNorm is a fixed affine, so the two batches of a program run simply
    concatenate (one program over 2 x B images): features bitwise equal to the one-batch program's, greedy ids identical"""
    torch.manual_seed(5)
    model = sat.ShowAndTell(32, 64, 120, 1, compute_dtype="bf16").cuda().eval()
    g = torch.Generator().manual_seed(12)
    xs = [torch.rand(4, 3, 64, 64, generator=g).cuda() for _ in range(5)]
    with torch.no_grad():
        want_f = [model.encoder(x).clone() for x in xs]
        want = [model.sample(x).clone() for x in xs]
        got, got_f, grouped = [], [], 0
        for i, x in enumerate(xs):
            model.prefetch_many(xs[i + 1:i + 1 + model.encoder.lookahead_depth])
            grouped = max(grouped, max([len(e["images"]) for e in model.encoder._inflight] or [0]))
            got_f.append(model.encoder(x).clone())
            got.append(model.sample(x).clone())
    torch.cuda.synchronize()
    assert grouped == 2                                  # a two-batch run really was in flight
    for a, b in zip(want_f, got_f):
        assert torch.equal(a, b)
    for a, b in zip(want, got):
        assert torch.equal(a, b)


def test_prefetched_stack_is_recomputed_after_an_in_place_weight_write():
    torch.manual_seed(9)
    enc = sat.EncoderCNN(32).cuda().eval()
    g = torch.Generator().manual_seed(4)
    x = torch.rand(4, 3, 64, 64, generator=g).cuda()
    with torch.no_grad():
        assert enc.prefetch(x)
        enc.resnet.conv1.weight.mul_(0.5)               # version counter moves: the stack in flight used the old weights
        got = enc(x).clone()
        want = enc(x).clone()                           # nothing in flight: computed with the current weights
    assert torch.equal(got, want)


def test_a_replaced_weight_parameter_rebuilds_the_program():
    """`conv.weight = nn.Parameter(...)` (no load_state_dict, no in-place write) must be seen too: the cached parameter
    list of `weights_signature` is checked against the modules' current Parameter objects"""
    torch.manual_seed(9)
    enc = sat.EncoderCNN(32).cuda().eval()
    g = torch.Generator().manual_seed(4)
    x = torch.rand(4, 3, 64, 64, generator=g).cuda()
    with torch.no_grad():
        before = enc(x).clone()
        w = enc.resnet.conv1.weight
        enc.resnet.conv1.weight = torch.nn.Parameter((w * 0.5).detach(), requires_grad=False)
        after = enc(x).clone()
        enc.refresh_weights()                           # the reference point: everything rebuilt from the current weights
        want = enc(x).clone()
    assert torch.equal(after, want) and not torch.equal(after, before)


def test_images_refilled_in_place_after_prefetch_are_recomputed():
    """VERDICT r2 (robustness 11): the look-ahead is keyed on the tensor OBJECT; a staging buffer refilled in place between
    `prefetch` and `forward` is the same object with other contents -- its version counter moved, so the stack in flight is
    discarded and the forward computes the new images (models.py:25-29)."""
    arch = dict(layers=(1, 1, 1, 1), width=8)
    torch.manual_seed(3)
    enc = sat.EncoderCNN(32, arch=arch, compute_dtype="bf16").cuda().eval()
    g = torch.Generator().manual_seed(5)
    a, b = torch.randn(4, 3, 64, 64, generator=g).cuda(), torch.randn(4, 3, 64, 64, generator=g).cuda()
    with torch.no_grad():
        want_b = enc(b).clone()
        staging = a.clone()
        assert enc.prefetch(staging)
        staging.copy_(b)                        # the loader reuses its buffer
        got = enc(staging).clone()
        assert not enc._inflight
        assert torch.equal(got, want_b)
        assert enc.prefetch(staging)            # untouched: the prefetched result is used (bitwise the same)
        assert torch.equal(enc(staging), want_b)


def test_train_eval_alternation_keeps_every_program():
    """VERDICT r2 (robustness 12): the reference alternates training and validation (train.py:157-159); with three look-ahead run slots
    each mode owns 3 + 1 op programs.  The cache holds both sets: switching modes rebuilds nothing."""
    arch = dict(layers=(1, 1, 1, 1), width=8)
    torch.manual_seed(4)
    enc = sat.EncoderCNN(32, arch=arch, compute_dtype="bf16").cuda()
    depth = enc.lookahead_depth
    g = torch.Generator().manual_seed(6)
    xs = [torch.randn(4, 3, 64, 64, generator=g).cuda() for _ in range(depth + 1)]

    def sweep():
        with torch.no_grad():
            for i in range(len(xs)):
                for j in range(i + 1, min(len(xs), i + 1 + depth)):
                    enc.prefetch(xs[j])
                enc(xs[i])
        enc.drop_lookahead()

    enc.train()
    sweep()
    enc.eval()
    sweep()
    ids = {k: id(v) for k, v in enc._programs.items()}
    # per mode: the forward's own program + one instance per run slot + the grouped program that leads the kernel choice
    assert len(ids) == 2 * (enc._n_slots() + 2)
    for _ in range(2):
        enc.train()
        sweep()
        enc.eval()
        sweep()
    assert {k: id(v) for k, v in enc._programs.items()} == ids          # same objects: nothing was evicted and rebuilt




def test_program_builds_and_their_tuning_passes_leave_the_model_untouched(tmp_path, monkeypatch):
    """Building the op programs (the grouped leader, the forward's own, the look-ahead instances) times kernel variants on random data and,
    for the final choice, runs the WHOLE program several times (`ConvStackProgram._pick_in_program`): no running statistic, counter or
    parameter of the model may move, the statistics accumulators are handed over zeroed, and the choices round-trip through
    SAT_TUNE_FILE: a second model loads them instead of tuning and computes the same bits (models.py:14-15: the stack is frozen)."""
    import json
    tune = tmp_path / "tune.json"
    monkeypatch.setenv("SAT_TUNE_FILE", str(tune))
    monkeypatch.setenv("SAT_AUTOTUNE", "1")               # timing-based tuning is opt-in since round 5 (tune.py): this test is about it
    torch.manual_seed(3)
    arch = dict(layers=(1, 2, 1, 1), width=64)            # wide enough for the weights-in-registers kernels (Cin % 64, Cout % 128)
    enc = sat.EncoderCNN(32, arch=arch, compute_dtype="bf16").cuda().train()
    before = {k: v.clone() for k, v in enc.state_dict().items()}
    x = torch.rand(4, 3, 64, 64, generator=torch.Generator().manual_seed(1)).cuda()
    enc.build_lookahead(x)
    prog = enc._program(x)
    torch.cuda.synchronize()
    after = enc.state_dict()
    for k, v in before.items():
        assert torch.equal(v, after[k]), k
    assert prog._parity == 0 and prog._runs == [0, 0]        # (built and tuned, never run by a caller)
    for acc in prog.stat_accs:
        assert int(acc.abs().sum()) == 0
    table = json.loads(tune.read_text())
    assert any(",g2" in k for k in table) and any(",g1,s" in k for k in table)      # the grouped leader's entries and the constrained followers'
    with torch.no_grad():
        want = enc(x).clone()
    variants = [int(prog.ops[i].variant) for i in range(prog.n_ops)]
    torch.manual_seed(3)
    enc2 = sat.EncoderCNN(32, arch=arch, compute_dtype="bf16").cuda().train()
    with torch.no_grad():
        got = enc2(x).clone()
    prog2 = enc2._program(x)
    assert [int(prog2.ops[i].variant) for i in range(prog2.n_ops)] == variants
    assert torch.equal(got, want)
