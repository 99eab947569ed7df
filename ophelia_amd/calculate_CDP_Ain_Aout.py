"""Attention-quality numbers printed per utterance by synthesize() ("File | CDP | Ain"), from the alignments the decode
loop returns.  Same names and values as the reference's helper module (calculate_CDP_Ain_Aout.py:9-57) so callers can
switch imports; written array-at-a-time.  A: (inputs, outputs) attention weights.

  CDP   coverage deviation: mean over the attended inputs of log(1 + (1 - coverage)^2), coverage = row sum
  Ain   mean row entropy of A (rows renormalised), divided by log(n_outputs)
  Aout  the same on A transposed
Inputs after the last attended one (all-zero tail rows) do not count.
"""
import numpy as np


def _attended_inputs(A):
    """row sums up to and including the last non-zero one"""
    coverage = np.asarray(A, np.float64).sum(axis=1)
    nz = np.flatnonzero(coverage)
    return coverage[: nz[-1] + 1] if len(nz) else coverage[:0]


def get_att_per_input(A):
    coverage = _attended_inputs(A)
    return coverage, len(coverage)


def getCDP(A):
    coverage = _attended_inputs(A)
    return float(np.log1p((1.0 - coverage) ** 2).sum() / len(coverage))


def getEnt(A):
    P = np.asarray(A, np.float64)
    mass = P.sum(axis=1, keepdims=True)
    P = np.divide(P, mass, out=P.copy(), where=mass != 0)          # rows without mass stay as they are
    plogp = np.zeros_like(P)
    np.multiply(P, np.log(P, out=np.zeros_like(P), where=P != 0), out=plogp, where=P != 0)
    return float(-plogp.sum() / P.shape[0] / np.log(P.shape[1]))


def getAP(A):
    n = len(_attended_inputs(A))
    A = np.asarray(A)[:n]
    return getEnt(A), getEnt(A.T)
