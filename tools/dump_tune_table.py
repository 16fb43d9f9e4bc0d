#!/usr/bin/env python3
"""Measure, ONCE, on an MI355X, which kernel variant every conv geometry of the BASELINE configurations runs, and write the
committed table `show-and-tell_amd/tune/gfx950.json` (tune.py: loaded by default, so that no run times anything at start-up and
two runs on one seed compute the same bits).

    python tools/dump_tune_table.py [out.json]

Builds, in the order `bench.py` / the tests build them (the first program of a model state leads, the others stay within its
statistics signatures -- the keys carry that constraint), with SAT_AUTOTUNE=force (the timing tuner: three fastest per geometry
replayed, the final choice IN the program):
  * ResNet-152, batch 64, 224x224, bf16, train mode: the grouped look-ahead programs (2 batches per launch), the ungrouped one;
  * the same in eval mode (decode: the two batches of a look-ahead run concatenate);
  * Inception-v3, batch 64, 299x299, train and eval mode (BASELINE configs[3]);
  * VGG16 features[:-3], batch 64, 224x224 (Show-Attend-Tell, model2.py:15).
"""
import hashlib
import importlib
import json
import os
import sys
import tempfile

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
tmp = os.path.join(tempfile.gettempdir(), "sat_dump_tune_%d.json" % os.getpid())
os.environ["SAT_AUTOTUNE"] = "force"
os.environ["SAT_TUNE_FILE"] = tmp

import torch  # noqa: E402

sat = importlib.import_module("show-and-tell_amd")
L, T = sat._lib, importlib.import_module("show-and-tell_amd.tune")
out_path = sys.argv[1] if len(sys.argv) > 1 else T.TABLE_PATH
torch.manual_seed(123)
dev = torch.device("cuda", 0)


def note(msg):
    print(msg, file=sys.stderr, flush=True)


with torch.no_grad():
    images = torch.randn(64, 3, 224, 224, device=dev)
    # the default program form, then the opt-in one (SAT_DEFER_BN3=1: bn3 + add + ReLU inside the next conv1, conv_ay_kernel): its
    # conv1 geometries carry their own keys
    for env in ({}, {"SAT_DEFER_BN3": "1"}):
        os.environ.update(env)
        model = sat.ShowAndTell(256, 512, 10000, 1, compute_dtype="bf16").to(dev).train()
        model.encoder._program(images)                   # grouped lead first, then the ungrouped program within its signatures
        model.encoder.build_lookahead(images)
        note("resnet152 train %s: done" % (env or ""))
        for k in env:
            del os.environ[k]
        if env:
            del model
            torch.cuda.empty_cache()
            continue
        for _ in range(8):                               # running statistics that match the data before the eval-mode programs run
            model.encoder(images)
        model.eval()
        model.encoder._program(images)
        model.encoder.build_lookahead(images)
        note("resnet152 eval: done")
        del model
        torch.cuda.empty_cache()

    im299 = torch.randn(64, 3, 299, 299, device=dev)
    model = sat.ShowAndTell(512, 1024, 10000, 2, arch="inception_v3", compute_dtype="bf16").to(dev).train()
    model.encoder._program(im299)                        # grouped lead, the ungrouped program within its signatures ...
    model.encoder.build_lookahead(im299)                 # ... and the look-ahead instances (grouped FOLLOWERS: keys with ",s<signature>")
    for _ in range(8):
        model.encoder(im299)
    model.eval()
    model.encoder._program(im299)
    model.encoder.build_lookahead(im299)
    note("inception_v3 train + eval: done")
    del model
    torch.cuda.empty_cache()

    model = sat.ShowAttendTellModel(1024, 512, 10000, 512, None, compute_dtype="bf16").to(dev)
    model._encode(images)
    note("vgg16: done")
    del model
    torch.cuda.synchronize()

lib = L.load()
with open(tmp) as f:
    table = json.load(f)
os.remove(tmp)
with open(L.LIB_PATH, "rb") as f:
    sha = hashlib.sha256(f.read()).hexdigest()[:16]
doc = {"what": "kernel variant (sat_op.variant, 1-based) per conv geometry key of show-and-tell_amd/tune.py; measured by tools/dump_tune_table.py",
       "device": torch.cuda.get_device_name(0), "arch": "gfx950",
       "abi": L.ABI_VERSION, "variants": int(lib.sat_conv_num_variants()), "libsat_hip_sha16": sha,
       "table": {k: int(table[k]) for k in sorted(table)}}
os.makedirs(os.path.dirname(out_path), exist_ok=True)
with open(out_path, "w") as f:
    json.dump(doc, f, indent=1, sort_keys=False)
    f.write("\n")
note("%d geometries -> %s" % (len(table), out_path))
