"""Shared pytest configuration.

Markers: ``gpu`` = needs a real MI355X (run with ``-m gpu`` on the GPU box).
Everything else must pass on CPU (``-m "not gpu"``).
"""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X GPU")


def load_golden(name):
    with open(os.path.join(GOLDEN_DIR, name + ".json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def has_gpu():
    import torch
    return torch.cuda.is_available()
