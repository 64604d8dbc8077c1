#!/usr/bin/env python3
"""Derive the polynomial constants of the p3d arithmetic contract (include/p3d_numerics.h).

The HIP kernels and the CPU oracle implement the SAME sequence of IEEE-754 binary32
operations (explicit fma, no contraction, no hardware transcendentals), so that they agree
bit-for-bit.  exp() and log1p() are therefore fixed polynomials; this script derives their
coefficients (near-minimax Chebyshev interpolants) and prints them as C hex-float literals.
Run:  python tools/gen_numerics.py   (prints the block pasted into include/p3d_numerics.h)
"""
import numpy as np
from numpy.polynomial import chebyshev as C, Polynomial

f32 = np.float32


def chebfit_fn(fn, lo, hi, deg, npts=4000):
    k = np.arange(npts)
    x = np.cos(np.pi * (k + 0.5) / npts)
    xs = (hi + lo) / 2 + (hi - lo) / 2 * x
    c = C.chebfit(x, fn(xs), deg)
    p = C.cheb2poly(c)
    m, h = (hi + lo) / 2, (hi - lo) / 2
    return Polynomial(p)(Polynomial([-m / h, 1 / h])).coef


def g(x):  # (exp(x) - 1 - x) / x^2
    x = np.where(np.abs(x) < 1e-8, 1e-8, x)
    return (np.exp(x) - 1 - x) / x ** 2


def q(z):  # log1p(z)/z
    z = np.where(z < 1e-9, 1e-9, z)
    return np.log1p(z) / z


def main():
    a = np.log(2) / 2 * 1.0001
    ce = np.concatenate([[1.0, 1.0], chebfit_fn(g, -a, a, 4)]).astype(f32)
    cl = chebfit_fn(q, 0.0, 1.0, 8).astype(f32)
    print("/* exp(r), |r| <= ln2/2 : sum_k P3D_EXP_C[k] r^k (Horner, fma) */")
    for k, c in enumerate(ce):
        print(f"#define P3D_EXP_C{k} {float(c).hex()}f  /* {c:.9g} */")
    print("/* log1p(z), 0 <= z <= 1 : z * sum_k P3D_L1P_C[k] z^k (Horner, fma) */")
    for k, c in enumerate(cl):
        print(f"#define P3D_L1P_C{k} {float(c).hex()}f  /* {c:.9g} */")


if __name__ == "__main__":
    main()
