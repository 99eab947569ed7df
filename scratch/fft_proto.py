# numpy prototype of the in-LDS transforms of the fused Griffin-Lim kernel: radix-4 Stockham 1024-point complex FFT
# + real-FFT post/pre-processing for n_fft = 2048.  Index formulas here are the ones transcribed into the HIP kernel.
import numpy as np
N2, NF = 1024, 2048
W = np.exp(-2j * np.pi * np.arange(1024) / 2048)      # W2048^k, k < 1024 (half circle)

def tw(e):            # W2048^e for any e >= 0 using the half-circle table
    e = e % 2048
    return np.where(e >= 1024, -W[e % 1024], W[e % 1024])

def fft1024(z, inverse=False):
    a = z.astype(np.complex128).copy()
    Ns = 1
    while Ns < N2:
        b = np.empty_like(a)
        for j in range(N2 // 4):
            k = j % Ns
            v = []
            for r in range(4):
                e = (k * r * (N2 // (Ns * 4))) * 2          # exponent in 2048ths
                t = tw(np.array(e))
                if inverse: t = np.conj(t)
                v.append(a[j + r * (N2 // 4)] * t)
            # radix-4 butterfly
            s = 1j if inverse else -1j
            t0, t1 = v[0] + v[2], v[0] - v[2]
            t2, t3 = v[1] + v[3], (v[1] - v[3]) * s
            o = [t0 + t2, t1 + t3, t0 - t2, t1 - t3]
            j0 = (j // Ns) * Ns * 4 + k
            for r in range(4):
                b[j0 + r * Ns] = o[r]
        a = b
        Ns *= 4
    return a

rng = np.random.default_rng(0)
z = rng.standard_normal(N2) + 1j * rng.standard_normal(N2)
print("fwd", np.abs(fft1024(z) - np.fft.fft(z)).max(), "inv", np.abs(fft1024(z, True) - np.fft.ifft(z) * N2).max())

# real forward: x[2048] -> X[0..1024]
x = rng.standard_normal(NF)
zz = x[0::2] + 1j * x[1::2]
Z = fft1024(zz)
X = np.empty(1025, complex)
X[0] = Z[0].real + Z[0].imag
X[1024] = Z[0].real - Z[0].imag
for k in range(1, 513):
    a, b = Z[k], np.conj(Z[N2 - k])
    e, o = 0.5 * (a + b), -0.5j * (a - b)
    X[k] = e + tw(np.array(k)) * o
    # partner bin N2-k: X[N2-k] = conj(e) - conj(W^k) * conj(o) ... derive: E[N2-k] = conj(E[k]), O[N2-k] = conj(O[k]), W^(N2-k) = -conj(W^k)
    X[N2 - k] = np.conj(e) - np.conj(tw(np.array(k))) * np.conj(o)
print("rfft", np.abs(X - np.fft.rfft(x)).max())

# real inverse: P[0..1024] (Hermitian half, P[0], P[1024] real) -> y[2048] = irfft(P) * 2048 (unscaled like C2R)
P = np.fft.rfft(rng.standard_normal(NF))
Zp = np.empty(N2, complex)
Zp[0] = (P[0].real + P[1024].real) + 1j * (P[0].real - P[1024].real)
for k in range(1, 513):
    a, b = P[k], np.conj(P[N2 - k])
    e, o = (a + b), (a - b) * np.conj(tw(np.array(k))) * 1j
    Zp[k] = e + o
    if k != 512:
        Zp[N2 - k] = np.conj(e) - np.conj(o) if False else np.conj(e - o)
    # k == 512: a = P[512], b = conj(P[512]); Zp[512] = e + o covers it
zt = fft1024(Zp, True)
y = np.empty(NF); y[0::2] = zt.real; y[1::2] = zt.imag
print("irfft", np.abs(y - np.fft.irfft(P) * NF).max())
