"""pytest configuration: marker registration and shared fixtures.

`-m "not gpu"`: oracle vs golden vectors, CPU emulation of the kernel templates
vs oracle, C-ABI export checks (no compute), multi-process sharding logic (gloo).
`-m gpu`: parity of the HIP library against the oracle through the C ABI.
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def kat():
    with open(os.path.join(ROOT, "tests", "golden", "kat.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def lazy_words():
    with open(os.path.join(ROOT, "tests", "golden", "lazy_words.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def case0_vectors():
    with open(os.path.join(ROOT, "tests", "golden", "case0_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    from oracle_binding import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def lib():
    import ontt
    return ontt.load()
