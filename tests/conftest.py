import os
import sys

import numpy as np
import pytest

# the oracle's OpenMP team: a 1-GPU box has a 16-core CPU share although it reports 256 logical CPUs
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 8)))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "piper-swift_amd", "python"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Stated tolerances (SURVEY.md §8c — the reference states none):
#   fp32 op level:        max|Δ| ≤ 1e-4 · max(1, ‖ref‖∞)
#   fp32 full waveform:   max|Δ| ≤ 1e-3
OP_TOL = 1e-4
WAVE_TOL = 1e-3


def assert_close(got, ref, tol=OP_TOL, what=""):
    got = np.asarray(got, np.float32).reshape(-1)
    ref = np.asarray(ref, np.float32).reshape(-1)
    assert got.shape == ref.shape, f"{what}: shape {got.shape} vs {ref.shape}"
    if ref.size == 0:
        return
    assert np.all(np.isfinite(got)), f"{what}: non-finite output"
    bound = tol * max(1.0, float(np.max(np.abs(ref))))
    err = float(np.max(np.abs(got - ref)))
    assert err <= bound, f"{what}: max|Δ|={err:.3e} > {bound:.3e}"


def subsample_like(a, max_store=20000):
    """The same sub-sampling tools/gen_golden.py applied before storing a golden output."""
    a = np.ascontiguousarray(a, np.float32).reshape(-1)
    if a.size <= max_store:
        return a
    step = -(-a.size // max_store)
    return a[::step]


@pytest.fixture(scope="session")
def golden_ops():
    return np.load(os.path.join(ROOT, "tests", "golden", "ops_kat.npz"))


@pytest.fixture(scope="session")
def golden_mods():
    return np.load(os.path.join(ROOT, "tests", "golden", "modules.npz"))


@pytest.fixture(scope="session")
def voices():
    """Host-side synthetic voices (no GPU needed): {quality: (cfg, blob)}."""
    import piper_hip as ph
    out = {}
    for q in ("medium", "high"):
        cfg = ph.voice_config(q)
        out[q] = (cfg, ph.synthetic_blob(cfg, 1234))
    return out


@pytest.fixture(scope="session")
def backend():
    """The HIP backend on cuda:0 — fails loudly (no fallback) if the library or the device is missing."""
    import piper_hip as ph
    b = ph.HipBackend(0)
    yield b
    b.close()
