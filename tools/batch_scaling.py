"""Train-step throughput vs per-GPU batch (BASELINE fixes 64): how much of the gap to the conv kernel's roofline is granularity
(tiles per launch), sequential schedule and with the encoder look-ahead.  python tools/batch_scaling.py [32 64 128 256]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
sat = importlib.import_module("show-and-tell_amd")
CFG = bench.CFG
dev = torch.device("cuda", 0)
for B in [int(a) for a in sys.argv[1:]] or [32, 64, 128, 256]:
    torch.manual_seed(123)
    model = sat.ShowAndTell(CFG["embed"], CFG["hidden"], CFG["vocab"], CFG["layers"], compute_dtype="bf16").to(dev).train()
    ts = sat.TrainStep(model)
    depth = model.encoder.lookahead_depth
    nb = depth + 1
    bt = [bench.synth_batch(torch, B, CFG["vocab"], CFG["cap_len"], CFG["image"], dev, 100 + k) for k in range(nb)]
    caps, lengths = bt[0][1], bt[0][2]
    res = []
    for la in (False, True):
        def run(n):
            for i in range(n):
                nxt = [bt[j % nb][0] for j in range(i + 1, i + 1 + depth) if j < n] if la else None
                ts.step(bt[i % nb][0], caps, lengths, next_images=nxt or None)
        run(4); torch.cuda.synchronize()
        n = 16
        t0 = time.perf_counter(); run(n); torch.cuda.synchronize()
        res.append(B * n / (time.perf_counter() - t0))
    prog = model.encoder._program(bt[0][0])
    prog.run_timed(bt[0][0])
    _, us = prog.run_timed(bt[0][0])
    fl = sat.conv_flops(sat.RESNET152, CFG["image"], CFG["image"]) * B
    print("batch %4d: sequential %7.0f img/s, look-ahead %7.0f img/s; conv kernels %.2f ms in sequence = %.0f TFLOP/s = %.3f of the bf16 peak"
          % (B, res[0], res[1], sum(us) * 1e-3, fl / (sum(us) * 1e-6) / 1e12, fl / (sum(us) * 1e-6) / 1e12 / 2500), flush=True)
    del model, ts, bt, prog
    torch.cuda.empty_cache()
