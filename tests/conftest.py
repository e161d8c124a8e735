import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def tiny_cfg():
    from vibevoice_rocm_amd.config import VVConfig
    return VVConfig.preset("tiny")


@pytest.fixture(scope="session")
def tiny_weights_np(tiny_cfg):
    from vibevoice_rocm_amd.synth import synth_state_dict
    return synth_state_dict(tiny_cfg, 1234)


@pytest.fixture(scope="session")
def tiny_weights(tiny_weights_np):
    return {k: torch.from_numpy(v) for k, v in tiny_weights_np.items()}


# ---- parity record: every relative-RMS error a test measures is kept, pass or fail, and written at session end ---------------------
# ("the measured errors live only in assert messages that print on failure").  On the GPU box the file lands in
# gpurun_out/ (the only directory that travels back); the copy judged is profiles/r03_parity_errors.json.
_PARITY = []
_CURRENT = {"id": None}


def pytest_deselected(items):
    if items:
        cfg = items[0].config
        cfg._vv_deselected = getattr(cfg, "_vv_deselected", 0) + len(items)


@pytest.hookimpl(tryfirst=True)
def pytest_runtest_setup(item):
    _CURRENT["id"] = item.nodeid


def pytest_sessionfinish(session, exitstatus):
    if not _PARITY:
        return
    import json
    # a partial run (-k / one file) must not overwrite the record of a whole-suite run
    whole = getattr(session, "testscollected", 0) - getattr(session.config, "_vv_deselected", 0) >= 200
    out = os.environ.get("VV_PARITY_OUT") or os.path.join(ROOT, "gpurun_out", "r03_parity_errors.json" if whole else "r03_parity_errors.partial.json")
    try:
        os.makedirs(os.path.dirname(out), exist_ok=True)
        by_test = {}
        for r in _PARITY:
            by_test.setdefault(r["test"], []).append({k: v for k, v in r.items() if k != "test"})
        dev = torch.cuda.get_device_name(0) if torch.cuda.is_available() else "cpu"
        with open(out, "w") as f:
            json.dump({"device": dev, "exitstatus": int(exitstatus), "n_measurements": len(_PARITY),
                       "note": "rel_rms = sqrt(mean((got - ref)^2)) / sqrt(mean(ref^2)); one entry per comparison a test made, in call order",
                       "tests": by_test}, f, indent=1)
    except OSError:
        pass


def rel_rms(a, b, what=None):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    e = float(np.sqrt(np.mean((a - b) ** 2)) / (np.sqrt(np.mean(b ** 2)) + 1e-30))
    rec = {"test": _CURRENT["id"], "rel_rms": e, "n": int(a.size)}
    if what:
        rec["what"] = str(what)
    _PARITY.append(rec)
    return e


def vt_tiles(v):
    """vv_kv.vt layout of a key-major value cache v[..., s_max, d]: 32-key tiles [..., s_max / 32, d, 32] (include/vv_hip.h)."""
    *lead, s_max, d = v.shape
    return v.reshape(*lead, s_max // 32, 32, d).transpose(-1, -2).contiguous()
