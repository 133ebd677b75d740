"""Fit the single-branch GELU used by the HIP kernels (tools only; prints coefficients + measured error).
    Phi(x) = 0.5 erfc(-x / sqrt2).  With t = min(|x|/sqrt2, T1) and h = 0.5 exp(-t Q(t)),  Phi = h (x < 0), 1 - h (x >= 0).
Q = polynomial fit of -log(erfc(t))/t on [0, T1]. Evaluated in emulated fp32-FMA arithmetic against float64."""
import numpy as np
from numpy.polynomial import chebyshev as Ch, polynomial as P
from scipy.special import erfc, erf

f32 = np.float32
def fma(a, b, c): return f32(np.float64(a) * np.float64(b) + np.float64(c))
T1 = 4.0
def fit(deg):
    t = np.cos(np.linspace(0, np.pi, 8000)) * 0.5 * T1 + 0.5 * T1
    t = np.maximum(t, 1e-9)
    y = -np.log(erfc(t)) / t
    c = Ch.chebfit(2 * t / T1 - 1, y, deg)
    p = Ch.cheb2poly(c)
    u = np.array([-1.0, 2.0 / T1]); out = np.zeros(1); pw = np.ones(1)
    for ck in p:
        out = P.polyadd(out, ck * pw); pw = P.polymul(pw, u)
    return out * (-1.4426950408889634)
def gelu_f32(x, q):
    x = x.astype(f32)
    t = np.minimum(np.abs(x) * f32(0.70710678118654752440), f32(T1)).astype(f32)
    r = np.full_like(x, f32(q[-1]))
    for c in q[-2::-1]:
        r = fma(r, t, f32(c))
    # coefficients already carry the factor -log2(e); the 0.5 is the "-1" in the exponent
    e = fma(r, t, f32(-1.0))
    h = np.exp2(e.astype(np.float64)).astype(f32)
    phi = (f32(0.5) + np.copysign((f32(0.5) - h).astype(f32), x)).astype(f32)
    return (x * phi).astype(f32)
x = np.concatenate([np.linspace(-8, 8, 4000001), np.random.default_rng(0).normal(size=1000000) * 1.5])
want = 0.5 * x.astype(f32).astype(np.float64) * (1.0 + erf(x.astype(f32).astype(np.float64) / np.sqrt(2.0)))
for deg in (7, 8, 9):
    q = fit(deg)
    err = np.abs(gelu_f32(x, q).astype(np.float64) - want)
    small = np.abs(x) < 3
    print(f"deg {deg}: max |gelu err| = {err.max():.3e} at x = {x[err.argmax()]:.4f};  |x|<3: {err[small].max():.3e}")
    if deg in (8, 9, 10):
        print("   ", ", ".join(f"{c:.9e}f" for c in q))
