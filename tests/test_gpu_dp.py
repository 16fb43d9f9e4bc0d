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
