import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """libgdpt.so (HIP, cross-compiled here) + liboracle.so; builds them when missing."""
    pkg = os.path.join(ROOT, "gradient-based-path-tracing_amd")
    if not os.path.exists(os.path.join(pkg, "libgdpt.so")):
        subprocess.check_call(["make", "-C", os.path.join(pkg, "csrc"), "-j", "8", "all"], stdout=subprocess.DEVNULL)
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"], stdout=subprocess.DEVNULL)
    import gdpt_amd
    return gdpt_amd


@pytest.fixture(scope="session")
def G(built):
    return built


@pytest.fixture(scope="session")
def O(built):
    import oracle_py
    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "ref_kat.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def scene_tmp(tmp_path_factory):
    return tmp_path_factory.mktemp("scenes")
