#!/usr/bin/env python3
"""Rational fit of g(w) = (erf(z)/z - 2/sqrt(pi) exp(-z^2)) / z^2, w = z^2, used by the
force-only nonbonded kernel: F_coul/r = qq (1/r^3 - beta^3 g(beta^2 r^2)).
Sanathanan-Koerner iteration for P(w)/Q(w), Q(0) = 1, on w in [0, 16]; prints float32 constants."""
import numpy as np
from scipy.special import erf

def g(w):
    z = np.sqrt(np.maximum(w, 1e-300))
    small = w < 1e-2
    ser = 2 / np.sqrt(np.pi) * (2 / 3 - 2 * w / 5 + w ** 2 / 7 - w ** 3 / 27 + w ** 4 / 132)
    with np.errstate(divide="ignore", invalid="ignore"):
        full = (erf(z) / z - 2 / np.sqrt(np.pi) * np.exp(-w)) / w
    return np.where(small, ser, full)

def fit(n, m, wmax=16.0, npts=6000):
    # Chebyshev-like node density at both ends
    t = np.cos(np.linspace(0, np.pi, npts)); w = 0.5 * wmax * (1 - t)
    y = g(w)
    wt = np.ones_like(w)
    x = w / wmax                       # scaled variable for conditioning
    for it in range(40):
        A = np.concatenate([np.stack([x ** k for k in range(n + 1)], 1), -y[:, None] * np.stack([x ** k for k in range(1, m + 1)], 1)], 1)
        coef, *_ = np.linalg.lstsq(A * wt[:, None], y * wt, rcond=None)
        p, q = coef[:n + 1], np.concatenate([[1.0], coef[n + 1:]])
        Q = sum(q[k] * x ** k for k in range(m + 1))
        wt = 1.0 / np.abs(Q)
    P = sum(p[k] * x ** k for k in range(n + 1))
    err = np.abs(P / Q - y).max()
    # back to the unscaled variable
    p = p / wmax ** np.arange(n + 1); q = q / wmax ** np.arange(m + 1)
    return p, q, err, Q.min()

if __name__ == "__main__":
    for n, m in ((4, 4), (5, 5), (6, 5), (6, 6)):
        p, q, err, qmin = fit(n, m)
        print(n, m, "max abs err", err, "min Q", qmin)
    p, q, err, qmin = fit(6, 5)
    w = np.linspace(0, 16, 20001).astype(np.float32)
    P = np.zeros_like(w); Q = np.zeros_like(w)
    for c in p[::-1].astype(np.float32): P = P * w + c
    for c in q[::-1].astype(np.float32): Q = Q * w + c
    print("float32 eval max abs err", np.abs(P / Q - g(w.astype(np.float64))).max())
    print("P:", ", ".join(f"{c:.9e}f" for c in p))
    print("Q:", ", ".join(f"{c:.9e}f" for c in q))
