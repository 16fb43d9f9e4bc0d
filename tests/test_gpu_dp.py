"""GPU: the data-parallel wrapper driving the real HIP engine (`TrainStep`) with 2 ranks.  The GPU box has ONE
MI355X, so both ranks share it and the collective runs over gloo (RCCL refuses two ranks on one device); what is
exercised is everything but the transport: sharding, 1/N_global scaling, bucket order, all-reduce on views of the
flat gradient buffer while later backward kernels are queued, loss slot, clamp+Adam after the reduction.
Result must equal the single-process step on the concatenated batch (decoder-only: cached features, so BatchNorm
batch statistics do not enter)."""
import importlib
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

DIMS = dict(E=32, H=64, V=500, L=1)
LENGTHS = [14, 14, 13, 11, 11, 9, 6, 4]


def make_batch():
    g = torch.Generator().manual_seed(7)
    B, T = len(LENGTHS), LENGTHS[0]
    caps = torch.zeros(B, T, dtype=torch.long)
    for b, l in enumerate(LENGTHS):
        caps[b, 0] = 1
        caps[b, 1:l - 1] = torch.randint(4, DIMS["V"], (l - 2,), generator=g)
        caps[b, l - 1] = 2
    return torch.randn(B, DIMS["E"], generator=g), caps


def make_model(sat):
    from oracle import decoder as OD
    model = sat.ShowAndTell(DIMS["E"], DIMS["H"], DIMS["V"], DIMS["L"], arch=dict(layers=(1, 1, 1, 1), width=8), compute_dtype="f32")
    model.decoder.load_state_dict(OD.init_decoder_params(DIMS["E"], DIMS["H"], DIMS["V"], DIMS["L"],
                                                         generator=torch.Generator().manual_seed(5)))
    return model.cuda().train()


def worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sat = importlib.import_module("show-and-tell_amd")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    feats, caps = make_batch()
    model = make_model(sat)
    dp = sat.DataParallelStep(sat.TrainStep(model))
    losses = []
    for _ in range(2):
        f, c, ln, tokens = sat.dp_shard(feats.cuda(), caps.cuda(), LENGTHS, rank, world)
        losses.append(float(dp.step((f, c, ln), tokens).item()))
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({"params": {k: v.detach().cpu() for k, v in model.decoder.named_parameters()}, "losses": losses}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_single_process(tmp_path):
    sat = importlib.import_module("show-and-tell_amd")
    out = str(tmp_path / "r0.pt")
    port = 29600 + os.getpid() % 2000
    try:
        mp.spawn(worker, args=(2, port, out), nprocs=2, join=True)
    except Exception as e:                                  # gloo built without device-tensor support
        if "gloo" in str(e).lower() and "cuda" in str(e).lower():
            pytest.skip("gloo cannot reduce device tensors in this build: %s" % str(e)[:200])
        raise
    got = torch.load(out)
    feats, caps = make_batch()
    model = make_model(sat)
    ts = sat.TrainStep(model)
    ref_losses = [float(ts.step(feats.cuda(), caps.cuda(), LENGTHS).item()) for _ in range(2)]
    for a, b in zip(got["losses"], ref_losses):
        assert abs(a - b) < 1e-5
    for k, p in model.decoder.named_parameters():
        np.testing.assert_allclose(got["params"][k].numpy(), p.detach().cpu().numpy(), rtol=0, atol=2e-6, err_msg=k)


# ------------------------------------------------------------------------------------------------------
# a persistent LSTM launch gives up on ONE rank: every rank must drop the same steps and raise at the same step
HOOKS_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libsat_hip_testhooks.so")


def stall_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["SAT_LIB"] = HOOKS_LIB              # the -DSAT_TESTHOOKS build: the product library has no fault injection
    sat = importlib.import_module("show-and-tell_amd")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    feats, caps = make_batch()
    torch.manual_seed(77)                          # (the encoder head is random-initialised: the same on both ranks, like a broadcast)
    model = make_model(sat)
    ts = sat.TrainStep(model)
    dp = sat.DataParallelStep(ts)
    f, c, ln, tokens = sat.dp_shard(feats.cuda(), caps.cuda(), LENGTHS, rank, world)
    log = {"raised_at": None, "msg": ""}
    for i in range(2):
        dp.step((f, c, ln), tokens)
    ts.check_ids()
    before = [t.clone() for t in (ts.flat.params, ts.flat.m, ts.flat.v)]
    count = ts.step_count
    for i in range(2, 8):
        if i == 2 and rank == 1:                   # the third step's forward recurrence gives up on rank 1 ONLY
            os.environ["SAT_LSTM_DEBUG_STALL"], os.environ["SAT_LSTM_SPIN_LIMIT"] = "1", "128"
        try:
            dp.step((f, c, ln), tokens)
        except RuntimeError as e:
            log["raised_at"], log["msg"] = i, str(e)
            break
        finally:
            os.environ.pop("SAT_LSTM_DEBUG_STALL", None)
            os.environ.pop("SAT_LSTM_SPIN_LIMIT", None)
    torch.cuda.synchronize()
    log["unchanged"] = all(torch.equal(a, b) for a, b in zip(before, (ts.flat.params, ts.flat.m, ts.flat.v)))
    log["count_ok"] = ts.step_count == count
    # training goes on (per-step launches now), the replicas stay bit-identical
    log["before"] = before[0].cpu()
    log["losses_after"] = [float(dp.step((f, c, ln), tokens).item()) for _ in range(2)]
    ts.check_ids()
    log["count_after"] = ts.step_count - count
    log["params"] = ts.flat.params.detach().cpu()
    log["slices"] = dict(ts.flat.slices)
    torch.save(log, out + ".%d" % rank)
    dist.barrier()
    dist.destroy_process_group()


def test_a_fault_on_one_rank_is_handled_identically_on_every_rank(tmp_path):
    """ADVICE r4 (trainer.py): the device half of the fault path was rank-consistent (the flag rides the all-reduce), the host half
    was not -- ranks polled without blocking, could raise one step apart, and then disagreed about which updates were dropped and
    about Adam's step count.  Now data-parallel steps look at the REDUCED flag of the step submitted `DP_FAULT_LAG` steps ago,
    blocking: rank 1's forward recurrence is made to give up in step 3 (test build of the library, SAT_LSTM_DEBUG_STALL on rank 1
    only); BOTH ranks must raise in the same step, with parameters, moments and step count exactly those after step 2, and stay
    bit-identical replicas afterwards.  train.py:43-44 (nn.DataParallel) / train.py:144-146"""
    out = str(tmp_path / "stall.pt")
    port = 31600 + os.getpid() % 2000
    try:
        mp.spawn(stall_worker, args=(2, port, out), nprocs=2, join=True)
    except Exception as e:
        if "gloo" in str(e).lower() and "cuda" in str(e).lower():
            pytest.skip("gloo cannot reduce device tensors in this build: %s" % str(e)[:200])
        raise
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    sat = importlib.import_module("show-and-tell_amd")
    lag = sat.TrainStep.DP_FAULT_LAG
    assert r0["raised_at"] == r1["raised_at"] == 2 + lag, (r0["raised_at"], r1["raised_at"])
    for r in (r0, r1):
        assert "SKIPPED on the device" in r["msg"] and "running statistics" in r["msg"]
        assert r["unchanged"] and r["count_ok"] and r["count_after"] == 2
    assert torch.equal(r0["before"], r1["before"])                # replicas were bit-identical going in ...
    assert r0["losses_after"] == r1["losses_after"]
    diff = {k: (r0["params"][o:o + n] - r1["params"][o:o + n]).abs().max().item() for k, (o, n, _) in r0["slices"].items()}
    assert torch.equal(r0["params"], r1["params"]), diff         # ... and still are


# ------------------------------------------------------------------------------------------------------
# full model under DP: encoder (per-rank BatchNorm batch statistics, as nn.DataParallel replicas have them,
# train.py:43-44), head gradients in bucket 2, checked against the CPU oracle run PER SHARD
ARCH = dict(layers=(1, 1, 1, 1), width=8)
FULL = dict(E=32, H=64, V=300, L=1, B=8, T=12, HW=64)


def full_setup():
    from oracle import decoder as OD
    from oracle import encoder as OE
    gen = torch.Generator().manual_seed(11)
    ep, eb = OE.init_encoder_params(FULL["E"], ARCH, generator=gen, randomize_bn=True)
    dp = OD.init_decoder_params(FULL["E"], FULL["H"], FULL["V"], FULL["L"], generator=gen)
    B, T = FULL["B"], FULL["T"]
    lengths = sorted([int(x) for x in torch.randint(3, T + 1, (B,), generator=gen)], reverse=True)
    lengths[0] = T
    caps = torch.zeros(B, T, dtype=torch.long)
    for b, l in enumerate(lengths):
        caps[b, 0] = 1
        caps[b, 1:l - 1] = torch.randint(4, FULL["V"], (l - 2,), generator=gen)
        caps[b, l - 1] = 2
    images = torch.randn(B, 3, FULL["HW"], FULL["HW"], generator=gen)
    return ep, eb, dp, images, caps, lengths


def full_model(sat, ep, eb, dp, device):
    model = sat.ShowAndTell(FULL["E"], FULL["H"], FULL["V"], FULL["L"], arch=ARCH, compute_dtype="f32")
    sd = dict(ep)
    sd.update(eb)
    model.encoder.load_state_dict(sd)
    model.decoder.load_state_dict(dp)
    return model.to(device).train()


def full_worker(rank, world, port, out, backend):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sat = importlib.import_module("show-and-tell_amd")
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    ep, eb, dp, images, caps, lengths = full_setup()
    model = full_model(sat, ep, eb, dp, dev)
    step = sat.DataParallelStep(sat.TrainStep(model))
    losses = []
    shards = [sat.dp_shard(images.to(dev), caps.to(dev), lengths, rank, world) for _ in range(2)]
    for i, (im, c, ln, tokens) in enumerate(shards):
        # the second step's conv stack runs ahead on a side stream, under the first step's decoder work and all-reduces
        nxt = shards[i + 1][0] if i + 1 < len(shards) else None
        losses.append(float(step.step((im, c, ln), tokens, next_images=nxt).item()))
    torch.cuda.synchronize()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    torch.save({"sd": sd, "losses": losses}, out + ".%d" % rank)
    dist.barrier()
    dist.destroy_process_group()


def oracle_dp_reference(world):
    """what N ranks must compute: per-shard forward/backward (per-shard BatchNorm statistics), gradients scaled by
    1/global tokens and summed, THEN clamp + Adam (train.py:144-146 order)"""
    from oracle import train_step as OT
    ep, eb, dp, images, caps, lengths = full_setup()
    bufs = [{k: v.clone() for k, v in eb.items()} for _ in range(world)]
    state, losses = {}, []
    tokens = sum(l - 1 for l in lengths)
    for _ in range(2):
        total, loss = None, 0.0
        for r in range(world):
            idx = list(range(r, len(lengths), world))
            l_r, g_r = OT.full_step(ep, bufs[r], dp, images[idx], caps[idx], [lengths[i] for i in idx], {}, arch=ARCH,
                                    num_layers=FULL["L"], denom=tokens, do_update=False)
            loss += l_r.item()
            total = g_r if total is None else {k: total[k] + g_r[k] for k in total}
        losses.append(loss)
        OT.clamp_(total, 0.1)
        allp = {k: ep[k] for k in total if k in ep}
        allp.update(dp)
        OT.adam_step_(allp, total, state, lr=1e-3)
    return ep, dp, bufs, losses


def _check_full(tmp_path, backend):
    out = str(tmp_path / "full.pt")
    port = 30600 + os.getpid() % 2000
    try:
        mp.spawn(full_worker, args=(2, port, out, backend), nprocs=2, join=True)
    except Exception as e:
        if backend == "gloo" and "gloo" in str(e).lower() and "cuda" in str(e).lower():
            pytest.skip("gloo cannot reduce device tensors in this build: %s" % str(e)[:200])
        raise
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    ep, dp, bufs, ref_losses = oracle_dp_reference(2)
    for a, b in zip(r0["losses"], ref_losses):
        assert abs(a - b) < 1e-4, (r0["losses"], ref_losses)
    assert r0["losses"] == r1["losses"]                         # the loss rides the last bucket: identical on every rank
    for k, v in dp.items():
        np.testing.assert_allclose(r0["sd"]["decoder." + k].numpy(), v.numpy(), rtol=0, atol=2e-5, err_msg=k)
        assert torch.equal(r0["sd"]["decoder." + k], r1["sd"]["decoder." + k]), k      # replicas stay bit-identical
    for k in ("resnet.fc.weight", "bn.weight", "bn.bias"):      # bucket 2: encoder head (fc.bias: zero-gradient noise)
        np.testing.assert_allclose(r0["sd"]["encoder." + k].numpy(), ep[k].numpy(), rtol=0, atol=2e-4, err_msg=k)
        assert torch.equal(r0["sd"]["encoder." + k], r1["sd"]["encoder." + k]), k
    # BatchNorm running statistics are PER RANK (each replica saw its own shard), like nn.DataParallel replicas
    for r, got in enumerate((r0, r1)):
        for k in ("resnet.bn1.running_mean", "resnet.layer3.0.bn2.running_var", "bn.running_mean"):
            np.testing.assert_allclose(got["sd"]["encoder." + k].numpy(), bufs[r][k].numpy(), rtol=2e-3, atol=2e-5, err_msg="rank%d %s" % (r, k))


def test_full_model_two_ranks_gloo_equals_oracle_per_shard(tmp_path):
    _check_full(tmp_path, "gloo")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank: >= 2 visible devices")
def test_full_model_two_ranks_nccl_equals_oracle_per_shard(tmp_path):
    """the same check over RCCL (backend 'nccl'), one rank per GPU: bucketed async all-reduce on views of the flat
    gradient buffer, the loss slot riding in bucket 2, rank-local autotune"""
    _check_full(tmp_path, "nccl")


def _bench(args, env_extra, timeout=900):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "SAT_TUNE_FILE"):
        env.pop(k, None)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]
    return json.loads(lines[0])


@pytest.mark.timeout(1200)
def test_bench_py_gpus2_runs_its_real_multi_rank_branch_on_one_device():
    """The driver's SCALE run is `bench.py --gpus N` on a node this build never sees: run the REAL N = 2 child tree here (spawned
    ranks, process group, kernel variants from the committed table on every rank, DataParallelStep with its bucket all-reduces and the look-ahead spread over two
    streams, barrier + MAX-over-ranks timing, one JSON line) -- both ranks on the one device, transport gloo instead of RCCL
    (SAT_BENCH_BACKEND / SAT_BENCH_SHARE_DEVICE; RCCL refuses two ranks on one device).  Everything but the transport is the
    code the 8-GPU run executes (train.py:43-44's nn.DataParallel replaced by one process per GPU)."""
    import math
    out = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--repeats", "1", "--no-cpu-baseline"],
                 {"SAT_BENCH_BACKEND": "gloo", "SAT_BENCH_SHARE_DEVICE": "1"})
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1
    cfg = out["config"]
    assert cfg["global_batch"] == 128 and cfg["parallelism"] == "dp2" and cfg["backend"] == "gloo"
    assert cfg["lookahead_depth"] == 6 and cfg["lookahead_streams"] == 2          # DataParallelStep.cap_lookahead took effect
    assert "on 2 side streams" in cfg["schedule"]
    assert math.isfinite(cfg["final_loss"]) and 8.5 < cfg["final_loss"] < 9.6      # ~ ln(10000) after three steps
    assert out["value"] > 0 and abs(out["value"] - 128 * 2 / (out["ms_per_step"] * 2e-3)) / out["value"] < 1e-2
    assert out["scaling"] == "weak" and out["metric"].startswith("images/sec")
    assert "f32_parity_mode" not in out and "cpu_baseline" not in out              # rank-0-only extras stay off at N > 1
    assert 0 < out["roofline"]["frac"] < 1


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("workload", ["train", "decode"])
def test_bench_py_gpus4_rehearsal_on_one_device(workload):
    """Towards the shape the driver's 8-GPU node runs first: `bench.py --gpus 4` -- ranks 2 and 3 exist, `decode_shard` / `dp_shard`
    see a world beyond two, several processes build their op programs from the committed variant table at once -- on the one
    device over gloo.  FOUR ranks, not eight: this pool kills a job with more than six processes on its GPU (the test session is
    one of them), so N = 8 itself can only run on the driver's node.  One run, no loops.  train.py:43-44"""
    import math
    out = _bench(["--gpus", "4", "--workload", workload, "--steps", "2", "--warmup", "1", "--repeats", "1", "--no-cpu-baseline"],
                 {"SAT_BENCH_BACKEND": "gloo", "SAT_BENCH_SHARE_DEVICE": "1"}, timeout=1100)
    assert out["n_gpus"] == 4 and out["config"]["global_batch"] == 256 and out["config"]["parallelism"].startswith("dp4")
    assert out["config"]["backend"] == "gloo" and out["value"] > 0
    if workload == "train":
        assert math.isfinite(out["config"]["final_loss"]) and 8.5 < out["config"]["final_loss"] < 9.6
        assert "committed table" in out["roofline"]["kernel"]       # every rank took its kernel variants from the same file
    else:
        assert out["config"]["features_finite"] is True and out["config"]["ids_shape"] == [64, 20]


@pytest.mark.timeout(1200)
def test_bench_py_decode_workload_shards_by_image_over_two_ranks():
    """configs[4] at N = 2 on the one device: every rank decodes its own 64 images (no collective inside the decode), captions/sec
    is the whole job's"""
    out = _bench(["--gpus", "2", "--workload", "decode", "--steps", "2", "--warmup", "1", "--repeats", "1", "--no-cpu-baseline"],
                 {"SAT_BENCH_BACKEND": "gloo", "SAT_BENCH_SHARE_DEVICE": "1"})
    assert out["n_gpus"] == 2 and out["unit"] == "captions/sec" and out["config"]["global_batch"] == 128
    assert out["config"]["beam_size"] == 5 and out["config"]["features_finite"] is True
    assert out["config"]["ids_shape"] == [64, 20] and out["value"] > 0


@pytest.mark.timeout(1200)
def test_bench_py_single_gpu_line_carries_every_contract_key():
    """the line the driver parses at N = 1 (`python bench.py --gpus 1 --steps K --warmup W`): contract keys, the `roofline` object of
    the dominant kernel (achieved / peak / frac consistent, traffic either null or measured on this very library build) and the
    `cpu_baseline` object (oracle port on the host's cores, rank 0, bounded sample)"""
    out = _bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--repeats", "1"], {})
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1 and out["higher_is_better"] is True
    assert out["unit"] == "images/sec" and out["dtype"] == "bf16" and out["data"] == "synthetic" and out["vs_baseline"] is None
    assert "configs[1]" in out["config"]["workload"] and out["config"]["global_batch"] == 64
    assert abs(out["value"] - 64 / (out["ms_per_step"] * 1e-3)) / out["value"] < 1e-2
    r = out["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.05 < r["frac"] < 1.0
    assert r["traffic"] is None or r["traffic"] > 1e6
    c = out["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "images/sec" and c["value"] > 0 and c["cores"] >= 1 and c["threads"] >= 1 and c["sample"]
    assert out["roofline_lstm"]["frac_of_f32_mfma_peak"] > 0 and out["f32_parity_mode"]["value"] > 0 and out["sequential_schedule"]["value"] > 0
