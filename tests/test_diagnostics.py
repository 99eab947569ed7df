"""CDP / Ain / Aout attention diagnostics (reference calculate_CDP_Ain_Aout.py:9-57) on hand-computed cases."""
import os

import numpy as np

from ophelia_amd.calculate_CDP_Ain_Aout import get_att_per_input, getAP, getCDP, getEnt


def test_perfect_diagonal_alignment():
    A = np.zeros((6, 4)); A[np.arange(4), np.arange(4)] = 1.0          # 4 attended inputs, 2 trailing unattended
    per, n = get_att_per_input(A)
    assert n == 4 and np.array_equal(per, np.ones(4))
    assert getCDP(A) == 0.0
    ain, aout = getAP(A)
    assert ain == 0.0 and aout == 0.0


def test_known_values():
    A = np.array([[0.5, 0.5, 0.0, 0.0],
                  [0.0, 0.0, 1.0, 1.0],
                  [0.0, 0.0, 0.0, 0.0]])
    per, n = get_att_per_input(A)
    assert n == 2 and np.allclose(per, [1.0, 2.0])
    assert np.isclose(getCDP(A), (np.log(1.0) + np.log(2.0)) / 2)
    ain, aout = getAP(A)
    assert np.isclose(ain, np.log(2) / np.log(4))                      # both rows spread evenly over 2 of 4 outputs
    assert np.isclose(aout, 0.0)                                       # every output attends to exactly one input
    assert np.isclose(getEnt(np.full((3, 5), 0.2)), 1.0)               # uniform rows -> normalised entropy 1


def test_interior_zero_rows_are_kept():
    A = np.array([[1.0, 0.0], [0.0, 0.0], [0.0, 1.0], [0.0, 0.0]])
    per, n = get_att_per_input(A)
    assert n == 3                                                      # only trailing zeros are trimmed
    assert np.isclose(getCDP(A), np.log(2.0) / 3)


def test_plot_alignment_writes_png(tmp_path):
    """utils.plot_alignment naming rules (utils.py:119-153)"""
    pytest = __import__("pytest")
    pytest.importorskip("matplotlib")
    from types import SimpleNamespace
    from ophelia_amd.utils import plot_alignment
    hp = SimpleNamespace(config_name="cfgX", logdir=str(tmp_path / "log"))
    A = np.random.default_rng(0).random((7, 11)).astype(np.float32)
    plot_alignment(hp, A, utt_idx=3, t2m_epoch=12)
    assert open(tmp_path / "log" / "alignment_cfgX_utt3_epoch12.png", "rb").read(8) == b"\x89PNG\r\n\x1a\n"
    out = str(tmp_path / "utt")
    plot_alignment(hp, A, utt_idx=1, t2m_epoch=0, outfile=out, savematrix=True)
    assert os.path.exists(out + ".png") and np.array_equal(np.load(out + "_attention.npy"), A)
