import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def load_golden():
    return golden


# the parity rules live beside the oracle (bench.py's parity legs use the same ones); the tests import them from here
from oracle.parity_rules import (order_insensitive_topk_match, ranked_lists_match, beam_cut_explains_absence,   # noqa: E402,F401
                                 hypothesis_lists_match)
