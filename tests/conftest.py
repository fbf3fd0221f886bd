import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _load(name, path):
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="session")
def oracle():
    """the numpy oracle (test infrastructure only)"""
    return _load("gl_oracle", os.path.join(ROOT, "oracle", "oracle.py"))


@pytest.fixture(scope="session")
def synth():
    return _load("gl_synth", os.path.join(ROOT, "gan-leaks_amd", "synth.py"))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
