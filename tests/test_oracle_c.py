"""Pins the C restatement (oracle/oph_cpu.c, the CPU-baseline port) against the goldens
generated from the reference's own graph code.  CPU only."""
import numpy as np
import pytest

from conftest import load_wiring_case
from oracle import ophelia_oracle as O
from oracle import cpu_oracle

TOL = 2e-5


@pytest.mark.parametrize("tag", ["lj_free", "lj_stop", "vctk_spk"])
def test_c_port_matches_reference_goldens(tag):
    hp, meta, g = load_wiring_case(tag)
    W = O.random_weights(hp, meta["weight_seed"])
    m = cpu_oracle.CpuModel(hp, W)
    K, V = m.encode_text(g["L"])
    assert np.abs(K - g["K"]).max() < TOL and np.abs(V - g["V"]).max() < TOL
    Y, t_ends, al, steps = m.text2mel(g["K"], g["V"], g["ends"], speakers=g.get("speakers"), stop=meta["stop"])
    assert steps == int(g["steps_run"]) and t_ends == g["t_ends"].tolist()
    assert np.abs(Y - g["Y"]).max() < TOL and np.abs(al - g["alignments"]).max() < TOL
    Z = m.ssrn(g["Y"])
    assert np.abs(Z - g["Z"]).max() < TOL
