"""Fit the two-branch polynomial erf used by the HIP GELU (tools only; prints C coefficients + measured error).
small |x| <= X0 : erf(x) = x + x*P(x^2)
large |x| >  X0 : erf(x) = sign(x) * (1 - exp(-(t*Q(t))))   with t = |x|, Q fitted to -log(1-erf(t))/t
Evaluated in emulated fp32 FMA arithmetic against scipy.special.erf (float64)."""
import numpy as np
from numpy.polynomial import chebyshev as Ch
from scipy.special import erf, erfc

f32 = np.float32
def fma(a, b, c):
    return f32(np.float64(a) * np.float64(b) + np.float64(c))

X0 = 1.0
def fit_small(deg):
    s = np.cos(np.linspace(0, np.pi, 4000)) * 0.5 * X0**2 + 0.5 * X0**2      # chebyshev nodes on [0, X0^2]
    x = np.sqrt(np.maximum(s, 1e-30))
    y = erf(x) / x - 1.0
    c = Ch.chebfit(2 * s / X0**2 - 1, y, deg)
    p = Ch.cheb2poly(c)
    # convert from u = 2s/X0^2 - 1 to s
    from numpy.polynomial import polynomial as P
    u = np.array([-1.0, 2.0 / X0**2])
    out = np.zeros(1)
    pw = np.ones(1)
    for k, ck in enumerate(p):
        out = P.polyadd(out, ck * pw)
        pw = P.polymul(pw, u)
    return out  # ascending in s

def fit_large(deg, T1=4.0):
    t = np.cos(np.linspace(0, np.pi, 6000)) * 0.5 * (T1 - X0) + 0.5 * (T1 + X0)
    y = -np.log(erfc(t)) / t
    c = Ch.chebfit((2 * t - (T1 + X0)) / (T1 - X0), y, deg)
    p = Ch.cheb2poly(c)
    from numpy.polynomial import polynomial as P
    u = np.array([-(T1 + X0) / (T1 - X0), 2.0 / (T1 - X0)])
    out = np.zeros(1); pw = np.ones(1)
    for ck in p:
        out = P.polyadd(out, ck * pw); pw = P.polymul(pw, u)
    return out  # ascending in t

def eval_f32(x, ps, ql):
    x = x.astype(f32); t = np.abs(x); s = (x * x).astype(f32)
    r = np.full_like(x, f32(ps[-1]))
    for c in ps[-2::-1]:
        r = fma(r, s, f32(c))
    small = fma(r * x if False else r, x, x)   # x + x*P(s)
    q = np.full_like(x, f32(ql[-1]))
    for c in ql[-2::-1]:
        q = fma(q, t, f32(c))
    e = (q * t).astype(f32)
    e = np.minimum(e, f32(30.0))
    big = (f32(1.0) - np.exp(-e.astype(np.float64)).astype(f32)).astype(f32)
    big = np.copysign(big, x)
    return np.where(t > f32(X0), big, small).astype(f32)

best = None
for ds in (5, 6):
    for dl in (6, 7, 8):
        ps, ql = fit_small(ds), fit_large(dl)
        x = np.concatenate([np.linspace(-6, 6, 2000001), np.random.default_rng(0).normal(size=1000000) * 1.5])
        got = eval_f32(x, ps, ql).astype(np.float64)
        want = erf(x.astype(f32).astype(np.float64))
        err = np.abs(got - want)
        g_err = np.abs(0.5 * x * (got - want))
        print(f"small deg {ds}, large deg {dl}: max |erf err| = {err.max():.3e} at x={x[err.argmax()]:.4f}; max gelu err = {g_err.max():.3e}")
        if best is None or err.max() < best[0]:
            best = (err.max(), ps, ql, ds, dl)
print("BEST", best[3], best[4], best[0])
print("small (ascending in s):", ", ".join(f"{c:.9e}f" for c in best[1]))
print("large (ascending in t):", ", ".join(f"{c:.9e}f" for c in best[2]))
