import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "3m-asr-inference_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    import numpy as np
    from m3asr.config import EncoderConfig
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    cfg = EncoderConfig.from_json(str(z["config"]))
    return cfg, z


@pytest.fixture(scope="session")
def golden():
    return load_golden
