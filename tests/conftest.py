import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: a GPU test that takes minutes by construction (ranks oversubscribing one device)")


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
