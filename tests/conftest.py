import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle is many small torch ops: on a 256-thread host the default thread count makes it ~10x slower
    import torch
    torch.set_num_threads(min(16, os.cpu_count() or 1))


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


_HOOKS_LIB = {}


@pytest.fixture
def hooks_lib(monkeypatch):
    """The -DSAT_TESTHOOKS build of the library (tests/_build/libsat_hip_testhooks.so, made by `__graft_entry__.build()`), bound in
    place of the product library for ONE test: fault injection (SAT_LSTM_DEBUG_STALL: a persistent-LSTM workgroup withholds its
    hand-off) is compiled into this build only -- the shipped libsat_hip.so has no such switch."""
    import importlib
    L = importlib.import_module("show-and-tell_amd._lib")
    path = os.path.join(ROOT, "tests", "_build", "libsat_hip_testhooks.so")
    if path not in _HOOKS_LIB:
        _HOOKS_LIB[path] = L.open_library(path)          # raises when the test build is missing: these tests never skip
    monkeypatch.setattr(L, "_lib", _HOOKS_LIB[path])
    return _HOOKS_LIB[path]
