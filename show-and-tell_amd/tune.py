"""Which kernel variant a bf16 conv of an op program runs -- decided WITHOUT a stopwatch by default.

A kernel variant fixes the order in which a conv sums its K axis and its BatchNorm column sums, so the choice is part of the
arithmetic: a run whose choice came from timing differs from the next run in the last bits of every activation (round 4: the
same seed gave a first-forward CE of 9.211105 in one process and 9.212307 in another; `/root/reference/config.py:15`
`random_seed` promises the opposite).  So:

* `show-and-tell_amd/tune/gfx950.json` is a COMMITTED table {geometry key -> variant number} for the BASELINE geometries
  (ResNet-152 at batch 64: grouped look-ahead / ungrouped / eval; Inception-v3; VGG16), measured once on an MI355X by
  `tools/dump_tune_table.py` and stamped with the library ABI and its number of variants (a variant number means something for
  the build it was measured on only);
* a geometry the table does not name runs `sat_conv_default_variant`: a function of the geometry alone;
* timing-based tuning is opt-in: `SAT_AUTOTUNE=1` times the geometries the table does not name (`SAT_TUNE_FILE=<json>` saves /
  reloads those choices), `SAT_AUTOTUNE=force` ignores the committed table and times everything (what the dump tool runs),
  `SAT_AUTOTUNE=0` ignores table and tuner (defaults only).

Every process, rank and box therefore runs the same kernels on the same geometry: eight data-parallel ranks need no shared
temp file, and two runs on one seed print the same loss (`tests/test_gpu_reproducible.py`).
"""
import json
import os

from . import _lib as L

TABLE_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tune", "gfx950.json")

_committed = None


def mode():
    """'table' (default), 'time' (SAT_AUTOTUNE=1), 'force' (SAT_AUTOTUNE=force), 'off' (SAT_AUTOTUNE=0)"""
    v = os.environ.get("SAT_AUTOTUNE", "").strip().lower()
    if v in ("0", "off"):
        return "off"
    if v in ("1", "time"):
        return "time"
    if v == "force":
        return "force"
    return "table"


def committed():
    """the committed table, {} when it is missing or was measured on another build of the library"""
    global _committed
    if _committed is None:
        tab = {}
        try:
            with open(os.environ.get("SAT_TUNE_TABLE") or TABLE_PATH) as f:
                doc = json.load(f)
            lib = L.load()
            if int(doc.get("abi", -1)) == L.ABI_VERSION and int(doc.get("variants", -1)) == int(lib.sat_conv_num_variants()):
                tab = {str(k): int(v) for k, v in doc.get("table", {}).items()}
        except (OSError, ValueError):
            tab = {}
        _committed = tab
    return _committed


def _fused_bits(o):
    return (1 if (o.stat_partial or o.stat_acc) else 0) + (2 if (o.scale0 or o.stat_acc1) else 0) + \
           (4 if o.scale1 else 0) + (8 if o.in1 else 0) + (16 if o.w_packed else 0)


def layer_key(o):
    """a conv layer whatever the batch: eval-mode look-ahead programs run the same layers on a concatenated batch"""
    return "%d,%d,%d,%d,%d,%d,%d,%d,%d,%d" % (o.Hin, o.Win, o.Cin, o.Hout, o.Wout, o.Cout, o.KH, o.KW, o.stride, _fused_bits(o))


def geom_key(o):
    return "%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d" % (o.N, o.Hin, o.Win, o.Cin, o.Hout, o.Wout, o.Cout, o.KH, o.KW, o.stride,
                                                 _fused_bits(o))


def tune_key(o, want_sig=None):
    return "%s,g%d%s" % (geom_key(o), max(int(o.groups), 1), "" if want_sig is None else ",s%d" % want_sig)


def matches(variant, want):
    lib = L.load()
    return (lib.sat_conv_variant_family(variant) if want >= 100000 else lib.sat_conv_variant_signature(variant)) == want


def load_tables():
    """committed table (unless SAT_AUTOTUNE=force / 0) overlaid with SAT_TUNE_FILE's entries (time / force modes)"""
    m = mode()
    table = dict(committed()) if m in ("table", "time") else {}
    tune_file = os.environ.get("SAT_TUNE_FILE")
    if m in ("time", "force") and tune_file and os.path.exists(tune_file):
        try:
            with open(tune_file) as f:
                table.update({str(k): int(v) for k, v in json.load(f).items()})
        except ValueError:
            pass
    return table


def assign(ops, n_ops, want_of=None):
    """Give every bf16 conv of `ops` its variant from the tables; returns the indexes of the convs no table names.  Those keep
    `variant = -want` (the tuner's constraint) or 0 and are the caller's to time (`time` / `force` modes) or to hand to
    `defaults`.  want_of(op) -> the statistics signature / output family the op must stay within, or None."""
    table = load_tables() if mode() != "off" else {}
    missing = []
    for i in range(n_ops):
        o = ops[i]
        if o.kind != L.OP_CONV or o.dtype != L.SAT_BF16:
            continue
        want = want_of(o) if want_of else None
        v = table.get(tune_key(o, want))
        if v is None or v < 1 or (want is not None and not matches(int(v), want)):
            missing.append(i)
            o.variant = -want if want is not None else 0
        else:
            o.variant = int(v)
    return missing


def defaults(ops, idx, want_of=None):
    """the geometry-only default (`sat_conv_default_variant`) for the convs `idx` of `ops`"""
    lib = L.load()
    for i in idx:
        o = ops[i]
        want = want_of(o) if want_of else None
        o.variant = 0
        o.variant = int(lib.sat_conv_default_variant(o, -1 if want is None else int(want)))


def save(ops, n_ops, want_of=None):
    """time / force modes: merge this program's choices into SAT_TUNE_FILE (whole-file replace: other ranks may be reading it)"""
    tune_file = os.environ.get("SAT_TUNE_FILE")
    if not tune_file or mode() not in ("time", "force"):
        return
    table = {}
    if os.path.exists(tune_file):
        try:
            with open(tune_file) as f:
                table.update(json.load(f))
        except ValueError:
            pass
    for i in range(n_ops):
        o = ops[i]
        if o.kind == L.OP_CONV and o.dtype == L.SAT_BF16 and int(o.variant) > 0:
            table[tune_key(o, want_of(o) if want_of else None)] = int(o.variant)
    tmp = "%s.%d.tmp" % (tune_file, os.getpid())
    with open(tmp, "w") as f:
        json.dump(table, f, indent=0, sort_keys=True)
    os.replace(tmp, tune_file)
