"""Attention-quality diagnostics the reference's synthesize() prints per utterance (synthesize.py:592-598), computed on
the host from the alignments the decode loop returns.  Counterpart of calculate_CDP_Ain_Aout.py:9-57 (getCDP 18-25,
getEnt 27-42, getAP 46-57); A has axis 0 = input symbols, axis 1 = output frames.

  CDP  coverage deviation penalty: mean over attended inputs of log(1 + (1 - total attention on the input)^2)
  Ain  entropy of each input's attention over the outputs, averaged and normalised by log(n_outputs)
  Aout the same for the transposed matrix
"""
import numpy as np


def get_att_per_input(A):
    """total attention per input symbol with trailing never-attended inputs dropped, and how many remain"""
    per_input = np.trim_zeros(np.sum(A, axis=1), "b")
    return per_input, len(per_input)


def getCDP(A):
    per_input, n = get_att_per_input(A)
    return np.sum(np.log(1.0 + (1.0 - per_input) ** 2)) / n


def getEnt(A):
    A = np.asarray(A, np.float64)
    total = 0.0
    for row in A:
        s = row.sum()
        p = row / s if s != 0.0 else row
        nz = p[p != 0.0]
        total += float(np.sum(nz * np.log(nz)))
    return (-total / A.shape[0]) / np.log(A.shape[1])


def getAP(A):
    _, n = get_att_per_input(A)
    A = A[:n, :]
    return getEnt(A), getEnt(np.transpose(A))
