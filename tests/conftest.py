import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def env():
    """(oracle module, SPMM, tiny_config, SPMMConfig, BertConfig) for the GPU parity tests."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import spmm_oracle as O
    from spmm_amd.config import SPMMConfig, BertConfig, tiny_config
    from spmm_amd.model import SPMM
    return O, SPMM, tiny_config, SPMMConfig, BertConfig
