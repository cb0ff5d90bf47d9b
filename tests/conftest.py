import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.dirname(os.path.abspath(__file__)) not in sys.path:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import __graft_entry__ as ge  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return ge.load_package()


@pytest.fixture(scope="session")
def orc():
    o = ge.load_oracle()
    o.build()
    return o


@pytest.fixture(scope="session")
def scene_c1(pkg):
    """BASELINE config C1: 20 views x ~100 obs/view (PTZRay)."""
    return pkg.synth.make_scene(0, 20, 100)


@pytest.fixture(scope="session")
def scene_c1_dist(pkg):
    return pkg.synth.make_scene(3, 20, 100, factor_type=1)
