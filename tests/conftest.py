import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

PKG = "3_orb_slam3_selfnote_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module(PKG)


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module(PKG + ".synth")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_py
    oracle_py.build()
    return oracle_py


_frame_cache = {}


@pytest.fixture(scope="session")
def frame(synth):
    def get(seed, H=480, W=752):
        key = (seed, H, W)
        if key not in _frame_cache:
            _frame_cache[key] = synth.make_frame(seed, H, W)
        return _frame_cache[key]
    return get


EUROC = dict(nfeatures=1000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7)   # Examples/Monocular/EuRoC.yaml:34-47
TUMVI = dict(nfeatures=1500, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7)   # Examples/Monocular/TUM_512.yaml:36-51
