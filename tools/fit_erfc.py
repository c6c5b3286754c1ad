#!/usr/bin/env python3
"""Least-squares fit of the scaled complementary error function used by the nonbonded
kernels: erfc(z) = exp(-z^2) * t * P(t), t = 1/(1 + p z), P of degree 7 (8 terms),
weighted for relative error on z in [0, 4.4] (covers beta*rlist for every .mdp in scope).
Prints the constants pasted into csrc/mdx_math.h."""
import numpy as np
from scipy.special import erfcx
p, deg = 0.41, 8
z = np.linspace(0, 4.4, 8001)
t = 1 / (1 + p * z)
A = np.stack([t ** k for k in range(1, deg + 1)], 1)
y = erfcx(z)
a, *_ = np.linalg.lstsq(A / y[:, None], np.ones_like(y), rcond=None)
a32 = a.astype(np.float32)
P = np.zeros_like(z, dtype=np.float32)
t32 = (np.float32(1) / (np.float32(1) + np.float32(p) * z.astype(np.float32)))
for c in a32[::-1]:
    P = P * t32 + c
approx = P * t32
print("max rel err f64 coeffs:", np.abs((A @ a - y) / y).max())
print("max rel err f32 eval :", np.abs((approx - y) / y).max())
print("p =", p)
print("coeffs a1..a8 =", ", ".join(f"{c:.9e}f" for c in a))
