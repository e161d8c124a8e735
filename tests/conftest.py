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


def rel_rms(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / (np.sqrt(np.mean(b ** 2)) + 1e-30))
