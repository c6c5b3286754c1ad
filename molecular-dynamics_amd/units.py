"""Physical constants in GROMACS units (nm, ps, amu, kJ/mol, e, K, bar).

The .mdp contract of the pipeline (e.g. /root/reference/mdp/production.mdp:24-40) is
expressed in these units, so the engine uses them unchanged.
"""

ONE_4PI_EPS0 = 138.935458      # kJ mol^-1 nm e^-2
BOLTZ = 0.0083144626           # kJ mol^-1 K^-1
PRESFAC = 16.6054              # bar per (kJ mol^-1 nm^-3)

# TIP3P geometry / parameters (AMBER port as shipped with GROMACS tip3p.itp)
TIP3P_DOH = 0.09572
TIP3P_DHH = 0.15139
TIP3P_QO = -0.834
TIP3P_QH = 0.417
TIP3P_SIGMA_O = 0.315061
TIP3P_EPS_O = 0.636386
MASS_O = 15.9994
MASS_H = 1.008


def ewald_beta(rc: float, rtol: float = 1e-5) -> float:
    """Ewald splitting parameter with erfc(beta*rc) = rtol (bisection, as the
    `ewald-rtol` mdp default 1e-5 demands; gives beta*rc = 3.12341)."""
    from math import erfc
    beta = 5.0
    i = 0
    while erfc(beta * rc) > rtol:
        beta *= 2.0
        i += 1
    lo, hi = 0.0, beta
    for _ in range(i + 60):
        mid = 0.5 * (lo + hi)
        if erfc(mid * rc) > rtol:
            lo = mid
        else:
            hi = mid
    return 0.5 * (lo + hi)


_NICE = None


def fft_size(nmin: int) -> int:
    """Smallest 2^a 3^b 5^c 7^d >= nmin (rocFFT-friendly radices)."""
    global _NICE
    if _NICE is None:
        s = set()
        a = 1
        while a <= 4096:
            b = a
            while b <= 4096:
                c = b
                while c <= 4096:
                    d = c
                    while d <= 4096:
                        s.add(d)
                        d *= 7
                    c *= 5
                b *= 3
            a *= 2
        _NICE = sorted(s)
    for v in _NICE:
        if v >= nmin and v >= 8:
            return v
    raise ValueError("grid too large")


def pme_grid(box, spacing: float):
    """Grid dims from the box diagonal and `fourierspacing`
    (/root/reference/mdp/production.mdp:29), rounded up to FFT-friendly sizes."""
    import math
    return tuple(fft_size(int(math.ceil(float(box[d][d]) / spacing - 1e-9))) for d in range(3))
