"""CPU: the C-ABI library loads and exports every symbol include/sat_hip.h declares (no compute calls without a
GPU), the ctypes struct mirrors the C struct, and the host-side logic (packing, sharding, LR schedule, flat
parameter layout errors) behaves like the reference's (`train.py:101-107,134-135`, `data_loader.py:48-62`)."""
import ctypes as C
import importlib
import os
import re
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sat = importlib.import_module("show-and-tell_amd")
L = sat._lib


def header_symbols():
    src = open(os.path.join(ROOT, "include", "sat_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sat_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = L.load()
    names = header_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libsat_hip.so does not export %s" % n
        assert n in L.SIGNATURES, "no ctypes signature for %s" % n
    assert set(L.SIGNATURES) == set(names)
    assert lib.sat_version() == L.ABI_VERSION == int(re.search(r"#define SAT_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "sat_hip.h")).read()).group(1))
    assert b"workspace" in lib.sat_error_string(1002)


def test_sat_op_struct_layout_matches_c():
    """compile a 3-line C program against the header and compare sizeof/offsetof with the ctypes mirror"""
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "sat_hip.h"
int main(){printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(sat_op), offsetof(sat_op,N), offsetof(sat_op,sN), offsetof(sat_op,count), offsetof(sat_op,eps),
 sizeof(sat_bn_eval_item), offsetof(sat_bn_eval_item,C));return 0;}
'''
    d = os.path.join(ROOT, "tests", "_build")
    os.makedirs(d, exist_ok=True)
    open(os.path.join(d, "layout.c"), "w").write(prog)
    exe = os.path.join(d, "layout")
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "layout.c"), "-o", exe])
    out = subprocess.check_output([exe]).decode().split()
    S = L.SatOp
    E = L.SatBnEvalItem
    assert [int(x) for x in out] == [C.sizeof(S), S.N.offset, S.sN.offset, S.count.offset, S.eps.offset, C.sizeof(E), E.C.offset]


def test_argument_errors_are_reported_not_computed():
    lib = L.load()
    # null pointers / bad shapes are rejected before any launch (works without a GPU)
    assert lib.sat_gemm_f32(0, 0, None, 4, None, 4, None, 4, None, None, 4, 4, 4, None) == 1001
    assert lib.sat_clamp_adam_step(None, None, None, None, 10, 1e-3, 0.9, 0.999, 1e-8, 0.1, 1, None) == 1001
    assert lib.sat_run_ops(None, 1, None) == 1001
    with pytest.raises(RuntimeError):
        L.check(1003, "x")


def test_modules_refuse_cpu_tensors():
    dec = sat.DecoderRNN(8, 16, 50, 1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        dec(torch.zeros(2, 8), torch.zeros(2, 5, dtype=torch.long), [6, 4])
    enc = sat.EncoderCNN(8, arch=dict(layers=(1, 1, 1, 1), width=8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        enc(torch.zeros(1, 3, 32, 32))


def test_packinfo_matches_pack_padded_sequence():
    from torch.nn.utils.rnn import pack_padded_sequence
    lengths = [7, 5, 5, 2, 1]
    pi = sat.PackInfo(lengths, "cpu")
    x = torch.arange(5 * 7).view(5, 7)
    ref = pack_padded_sequence(x, lengths, batch_first=True)
    assert pi.batch_sizes == ref.batch_sizes.tolist() and pi.N == ref.data.numel()
    caps = torch.randint(0, 9, (5, 8))
    tg, l1 = sat.pack_targets(caps, [8, 6, 6, 3, 2])
    ref_t = pack_padded_sequence(caps[:, 1:], l1, batch_first=True)[0]
    assert torch.equal(tg, ref_t)
    with pytest.raises(ValueError):
        sat.PackInfo([3, 5], "cpu")
    with pytest.raises(ValueError):
        sat.PackInfo([3, 0], "cpu")


def test_dp_shard_keeps_sorted_and_covers_batch():
    lengths = [20, 19, 19, 15, 12, 9, 9, 4]
    imgs, caps = torch.arange(8).float().view(8, 1), torch.arange(8).view(8, 1)
    seen = []
    for r in range(2):
        im, cp, ln, tok = sat.dp_shard(imgs, caps, lengths, r, 2)
        assert ln == sorted(ln, reverse=True) and tok == sum(l - 1 for l in lengths)
        seen += cp.flatten().tolist()
    assert sorted(seen) == list(range(8))


def test_lr_schedule_matches_train_py():
    assert sat.lr_for_epoch(1) == 1e-3 and sat.lr_for_epoch(3) == 1e-3
    assert abs(sat.lr_for_epoch(4) - 8e-4) < 1e-15 and abs(sat.lr_for_epoch(10) - 1e-3 * 0.8 ** 3) < 1e-15


def test_state_dict_keys_equal_reference_names():
    dec = sat.DecoderRNN(8, 16, 50, 2)
    assert list(dec.state_dict()) == ["embed.weight", "lstm.weight_ih_l0", "lstm.weight_hh_l0", "lstm.bias_ih_l0",
                                      "lstm.bias_hh_l0", "lstm.weight_ih_l1", "lstm.weight_hh_l1", "lstm.bias_ih_l1",
                                      "lstm.bias_hh_l1", "linear.weight", "linear.bias"]
    enc = sat.EncoderCNN(8)
    keys = list(enc.state_dict())
    assert keys[0] == "resnet.conv1.weight" and "resnet.layer3.35.bn3.running_var" in keys
    assert "resnet.layer1.0.downsample.0.weight" in keys and keys[-5:] == ["bn.weight", "bn.bias", "bn.running_mean",
                                                                           "bn.running_var", "bn.num_batches_tracked"]
    trainable = [k for k, p in enc.named_parameters() if p.requires_grad]
    assert trainable == ["resnet.fc.weight", "resnet.fc.bias", "bn.weight", "bn.bias"]      # models.py:14-17
    assert abs(sat.conv_flops(sat.RESNET152) / 1e9 - 23.02) < 0.1


def test_plain_c_program_links_and_calls_the_abi():
    """the boundary is a C ABI: a C translation unit that only includes sat_hip.h links against the library and gets
    the documented error codes back (no GPU needed: arguments are rejected before any launch)"""
    prog = r'''
#include <stdio.h>
#include <string.h>
#include "sat_hip.h"
int main(void) {
    sat_op op; memset(&op, 0, sizeof op);
    op.kind = SAT_OP_CONV;                       /* no buffers -> SAT_ERR_ARG */
    int a = sat_run_ops(&op, 1, NULL);
    int b = sat_gemm_f32(0, 0, NULL, 4, NULL, 4, NULL, 4, NULL, NULL, 4, 4, 4, NULL);
    int c = sat_beam_step(NULL, 0, NULL, NULL, -1, 1, 1, 1, NULL, NULL, NULL, NULL, 0, NULL);
    long long w = (long long)sat_lstm_bwd_ws_bytes_full(1216, 64, 256, 512);
    printf("%d %d %d %d %lld %s\n", sat_version(), a, b, c, w, sat_error_string(SAT_ERR_WORKSPACE));
    return 0;
}
'''
    d = os.path.join(ROOT, "tests", "_build")
    os.makedirs(d, exist_ok=True)
    open(os.path.join(d, "cabi.c"), "w").write(prog)
    exe = os.path.join(d, "cabi")
    libdir = os.path.join(ROOT, "show-and-tell_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(d, "cabi.c"),
                           "-o", exe, "-L", libdir, "-lsat_hip", "-L/opt/rocm/lib", "-Wl,-rpath," + libdir,
                           "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib"])
    out = subprocess.check_output([exe]).decode().split(None, 5)
    assert [int(x) for x in out[:4]] == [L.load().sat_version(), 1001, 1001, 1001]
    assert int(out[4]) >= L.load().sat_lstm_bwd_ws_bytes(64, 512) and "workspace" in out[5]


def test_graft_entry_build_runs_clean():
    """the driver's build check: compiles (incrementally) for gfx950, loads the library, versions agree"""
    ge = importlib.import_module("__graft_entry__")
    assert ge.build() is None
    hdr = open(os.path.join(ROOT, "include", "sat_hip.h")).read()
    assert L.load().sat_version() == int(re.search(r"#define SAT_ABI_VERSION (\d+)", hdr).group(1))


def test_collate_batch_keeps_the_reference_invariant():
    """data_loader.py:48-62: longest caption first (ties keep their order), zero padding, int64, lengths as a list --
    and the result feeds pack_padded_sequence / PackInfo without re-sorting"""
    g = torch.Generator().manual_seed(3)
    lens = [5, 9, 5, 12, 9, 3]
    samples = [(torch.full((3, 4, 4), float(i)), torch.randint(1, 50, (n,), generator=g).int(), 100 + i)
               for i, n in enumerate(lens)]
    images, caps, lengths, ids = sat.collate_batch(list(samples))
    assert lengths == [12, 9, 9, 5, 5, 3] and ids == (103, 101, 104, 100, 102, 105)      # stable among equal lengths
    assert caps.dtype == torch.int64 and caps.shape == (6, 12) and images.shape == (6, 3, 4, 4)
    for row, i in enumerate([3, 1, 4, 0, 2, 5]):
        n = lens[i]
        assert torch.equal(caps[row, :n], samples[i][1].long()) and int(caps[row, n:].abs().sum()) == 0
        assert float(images[row, 0, 0, 0]) == float(i)
    packed = torch.nn.utils.rnn.pack_padded_sequence(caps, lengths, batch_first=True)      # accepts it unsorted-check on
    assert packed.batch_sizes.tolist() == sat.PackInfo(lengths, "cpu").batch_sizes
    with pytest.raises(ValueError):
        sat.collate_batch([])
    with pytest.raises(RuntimeError):
        sat.DevicePrefetcher([], "cpu")


def test_inception_mac_constant_of_bench_py_matches_the_oracle():
    """bench.py --workload inception prices its roofline with a constant (the timed path may not import the oracle): keep it equal
    to the oracle's count"""
    import re
    from oracle import inception as OI
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert int(re.search(r"INCEPTION_V3_CONV_MACS = (\d+)", src).group(1)) == OI.conv_macs()


def test_packinfo_prev_rows_index_the_h_prev_table():
    """PackInfo.prev_rows(): row of [h_0 (B rows) ; packed h (N rows)] holding h_{t-1} for every packed row (one gather builds
    h_prev for the batched weight-gradient GEMMs of the Show-Attend-Tell backward, model2.py:54-62)"""
    lengths = [5, 4, 4, 2, 1]
    pi = sat.PackInfo(lengths, "cpu")
    B, N = pi.B, pi.N
    idx = pi.prev_rows().tolist()
    assert len(idx) == N == sum(lengths)
    # packed row (t, b) sits at prefix[t] + b; its predecessor is h_0[b] for t = 0, packed row (t-1, b) after
    want = []
    for t in range(pi.T):
        for b in range(pi.batch_sizes[t]):
            want.append(b if t == 0 else B + pi.prefix[t - 1] + b)
    assert idx == want
    assert pi.prev_rows() is pi.prev_rows()              # cached


def test_write_through_store_keeps_its_wait_states(tmp_path):
    """Static half of the `store16_wt` regression pin (robustness 13): the inline-asm write-through store
    (`global_store_dwordx4 ... sc0 sc1`, sat_common.h) must carry its own `s_nop 1` -- a VMEM store of more than 64 bits needs
    wait states before its data VGPRs may be overwritten and the compiler's hazard recognizer does not look inside asm.  A tiny
    kernel that overwrites the stored registers at once is compiled to gfx950 assembly; every such store is followed by the nop."""
    src = tmp_path / "wt.hip"
    src.write_text('#include "sat_common.h"\n'
                   '__global__ void k(u32x4* dst, const u32x4* src, int n) {\n'
                   '    u32x4 v = src[threadIdx.x];\n'
                   '    for (int i = 0; i < n; ++i) { store16_wt(dst + i * 64 + threadIdx.x, v); v = v * 3u + 1u; }\n'
                   '}\n')
    asm = tmp_path / "wt.s"
    inc = os.path.join(ROOT, "show-and-tell_amd", "csrc")
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-I", inc,
                        "-I", os.path.join(ROOT, "include"), str(src), "-o", str(asm)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l.strip() for l in asm.read_text().splitlines() if l.strip() and not l.strip().startswith((";", ".", "//"))]
    stores = [i for i, l in enumerate(lines) if l.startswith("global_store_dwordx4") and "sc0 sc1" in l]
    assert stores, "the write-through store disappeared from the generated code"
    for i in stores:
        assert lines[i + 1].startswith("s_nop 1"), (lines[i], lines[i + 1])
