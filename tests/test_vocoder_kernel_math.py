"""CPU checks of the mathematics the fused Griffin-Lim kernel (ophelia_amd/csrc/oph_vocoder.hip, gl_fused) is built on,
written with the kernel's own index formulas so that a change there has a place to be re-derived first:

  * radix-4 Stockham 1024-point complex FFT (five passes, twiddles W_2048^(k*r*512/Ns)) == np.fft.fft / ifft
  * the in-place pass over bin pairs (k, 1024-k) that splits the packed real transform and re-packs for the inverse
  * the LDS index swizzle PX(i) = i ^ (((i >> 4) * 5) & 15) is a bijection and makes every ds_read_b64 / ds_write_b64
    of the five passes bank-conflict free under the CDNA4 rules (write: 4 groups of 16 lanes over 16 8-byte slots of a
    128-byte row; read: 2 groups of 32 lanes over 32 slots of a 256-byte row)
"""
from collections import Counter

import numpy as np

N2, NF = 1024, 2048


def W(e):
    return np.exp(-2j * np.pi * (np.asarray(e) % NF) / NF)


def fft1024(z, inverse=False):
    a = np.asarray(z, np.complex128).copy()
    Ns = 1
    while Ns < N2:
        b = np.empty_like(a)
        for tid in range(N2 // 4):
            k = tid & (Ns - 1)
            v = [a[tid + 256 * r] for r in range(4)]
            if Ns > 1:
                for r in (1, 2, 3):
                    t = W(k * r * (512 // Ns))
                    v[r] = v[r] * (np.conj(t) if inverse else t)
            a0, a1, a2, d = v[0] + v[2], v[0] - v[2], v[1] + v[3], v[1] - v[3]
            a3 = d * (1j if inverse else -1j)
            j0 = ((tid - k) << 2) + k
            b[j0], b[j0 + Ns], b[j0 + 2 * Ns], b[j0 + 3 * Ns] = a0 + a2, a1 + a3, a0 - a2, a1 - a3
        a = b
        Ns *= 4
    return a


def test_radix4_stockham_matches_numpy():
    rng = np.random.default_rng(0)
    z = rng.standard_normal(N2) + 1j * rng.standard_normal(N2)
    assert np.abs(fft1024(z) - np.fft.fft(z)).max() < 1e-11
    assert np.abs(fft1024(z, True) - np.fft.ifft(z) * N2).max() < 1e-11


def test_pair_pass_split_and_repack():
    rng = np.random.default_rng(1)
    x = rng.standard_normal(NF)
    Z = fft1024(x[0::2] + 1j * x[1::2])
    X = np.empty(N2 + 1, complex)
    X[0], X[N2] = Z[0].real + Z[0].imag, Z[0].real - Z[0].imag
    for k in range(1, 513):
        a, b = Z[k], np.conj(Z[N2 - k])
        e, o = 0.5 * (a + b), -0.5j * (a - b)
        wo = W(k) * o
        X[k], X[N2 - k] = e + wo, np.conj(e - wo)
    assert np.abs(X - np.fft.rfft(x)).max() < 1e-11
    # re-pack a Hermitian half spectrum P for the inverse transform (unscaled, like a C2R transform)
    P = np.fft.rfft(rng.standard_normal(NF))
    Zp = np.empty(N2, complex)
    Zp[0] = (P[0].real + P[N2].real) + 1j * (P[0].real - P[N2].real)
    for k in range(1, 513):
        a, b = P[k], np.conj(P[N2 - k])
        e, o = a + b, 1j * np.conj(W(k)) * (a - b)
        Zp[k] = e + o
        if k != 512:
            Zp[N2 - k] = np.conj(e - o)
    zt = fft1024(Zp, True)
    y = np.empty(NF)
    y[0::2], y[1::2] = zt.real, zt.imag
    assert np.abs(y - np.fft.irfft(P) * NF).max() < 1e-9
    # the kernel derives W^k for k in 257..512 from the table of k <= 256
    for k in (257, 300, 511, 512):
        wr = W(512 - k)
        assert abs(complex(-wr.imag, -wr.real) - W(k)) < 1e-15


def PX(i):
    return i ^ (((i >> 4) * 5) & 15)


def _ways(slots, nslots):
    return max(Counter(s % nslots for s in set(slots)).values())


def test_lds_swizzle_is_a_conflict_free_bijection():
    assert sorted(PX(i) for i in range(N2)) == list(range(N2))
    assert all(PX(i) >> 4 == i >> 4 for i in range(N2))                  # stays inside its block of 16
    assert all(PX(t + 256 * r) == PX(t) + 256 * r for t in range(256) for r in range(4))
    for Ns in (1, 4, 16, 64, 256):
        for r in range(4):
            for wave in range(4):
                for grp in range(4):                                      # ds_write_b64: 16 contiguous lanes
                    lanes = [wave * 64 + grp * 16 + l for l in range(16)]
                    idx = [((t - (t & (Ns - 1))) << 2) + (t & (Ns - 1)) + r * Ns for t in lanes]
                    assert _ways([PX(i) for i in idx], 16) == 1, (Ns, r, wave, grp)
                for grp in range(2):                                      # ds_read_b64: 32 lanes
                    lanes = [wave * 64 + grp * 32 + l for l in range(32)]
                    assert _ways([PX(t + 256 * r) for t in lanes], 32) == 1


def test_twiddle_table_layout():
    """entry (r-1)*Ns + k of the run of pass Ns holds W_2048^(k*r*512/Ns): runs are contiguous and total 1020"""
    off, total = {}, 0
    for Ns in (4, 16, 64, 256):
        off[Ns] = total
        total += 3 * Ns
    assert off == {4: 0, 16: 12, 64: 60, 256: 252} and total == 1020
    assert 1020 + 257 <= 1280
