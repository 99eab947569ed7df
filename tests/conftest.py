import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class HP(object):
    """Plain attribute bag standing in for the reference's Hyperparams in tests."""
    pass


def hp_from_snapshot(cfg, **over):
    snap = json.load(open(os.path.join(GOLDEN, "config_snapshot.json")))[cfg]
    hp = HP()
    for k, v in snap.items():
        setattr(hp, k, v)
    for k, v in over.items():
        setattr(hp, k, v)
    return hp


def load_wiring_case(tag):
    meta = json.load(open(os.path.join(GOLDEN, "wiring_%s.json" % tag)))
    g = dict(np.load(os.path.join(GOLDEN, "wiring_%s.npz" % tag)))
    hp = hp_from_snapshot(meta["cfg"], max_N=meta["max_N"], max_T=meta["max_T"], **meta.get("override", {}))
    if getattr(hp, "turn_off_monotonic_for_synthesis", False):
        hp.text_lengths = g["ends"] + 1                  # what the host sets before building the graph (synthesize.py:505-507)
    return hp, meta, g


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
