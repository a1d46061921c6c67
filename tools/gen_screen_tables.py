#!/usr/bin/env python3
"""Generates jpeg-encoder-opencl_amd/csrc/jpeg_screen_tables.h: the data the screened
(integer-MFMA) transform needs.

The reference's in-place chain (utils.cpp:314-348) is, in exact real arithmetic on its
double constants, a fixed linear map  c = L p  of the 64 level-shifted samples.  The
fp64 chain the reference actually runs differs from L p by rounding only; a rigorous
a-priori bound eps[k] on that difference over all p in [-128,127]^64 is derived here by
forward error analysis.  The kernel evaluates a fixed-point copy of L
exactly with int8 MFMAs and accepts a quantised coefficient only when the distance of
c/Q to the nearest rounding boundary exceeds eps plus the fixed-point error; everything
else is recomputed with the exact ordered fp64 chain.

Outputs (all in ZIG-ZAG row order, so the kernel's accumulator rows are zig-zag
positions):
  kScreenLimb[5][64][64]  int8   balanced base-256 digits of round(L * 2^39), digit 0 least significant
  kScreenEps[64]          double rigorous bound on |chain_fp64 - L p| per zig-zag position (incl. 1 % slack)
"""
import json
import os
from fractions import Fraction

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
T = json.load(open(os.path.join(ROOT, "tests", "golden", "tables.json")))
COS = [[float.fromhex(h) for h in row] for row in T["cos"]]
S00, S0X, SXX = (float.fromhex(T["scale"]["00"]), float.fromhex(T["scale"]["0x"]), float.fromhex(T["scale"]["xx"]))
FRAC_BITS = 39
NLIMB = 5
ZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20,
          13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52,
          45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]


def scale(u, v):
    return S00 if (u == 0 and v == 0) else (S0X if (u == 0 or v == 0) else SXX)


def steps():
    for u in range(8):
        for v in range(8):
            coef = [Fraction(COS[i % 8][u]) * Fraction(COS[i // 8][v]) * Fraction(scale(u, v)) for i in range(64)]
            yield v * 8 + u, coef


def exact_map():
    """M as exact rationals: final state = M p (natural index v*8+u)."""
    M = [[Fraction(int(i == j)) for j in range(64)] for i in range(64)]
    for pos, coef in steps():
        M[pos] = [sum(coef[i] * M[i][j] for i in range(64)) for j in range(64)]
    return M


def error_bound(M_exact_float):
    """Rigorous bound on |fp64 chain - exact chain| per natural index.

    Step k computes fl-sum of fl(fl(P_i*cx_i)*cy_i) and one scale multiply:
       computed = sum_i row_i P_i (1+theta_i), |theta_i| <= gamma_67   (2 products, <=64 adds, 1 scale)
    so the local error eta_k obeys |eta_k| <= gamma_67 * sum_i |row_i| |Phat_i|, with
    |Phat_i| <= B_i + E_i (exact-state bound + accumulated error bound).  By linearity the
    state error is sum_k' T_{k'->now} e_pos(k') eta_k' exactly, T = product of later exact
    step maps.  Evaluated in float64 with 1 % slack per step (these are bounds)."""
    u_ = 2.0 ** -53
    g = 67 * u_ / (1 - 67 * u_)
    St = np.eye(64)
    W, Hs = [], []
    E = np.zeros(64)
    for pos, coef in steps():
        c = np.array([float(x) for x in coef])
        B = 128.0 * np.abs(St).sum(1)
        Hk = g * float(np.sum(np.abs(c) * (B + E))) * 1.01
        St[pos] = c @ St
        for w in W:
            w[pos] = float(c @ w)
        e = np.zeros(64)
        e[pos] = 1.0
        W.append(e)
        Hs.append(Hk)
        E = np.zeros(64)
        for w, h in zip(W, Hs):
            E += np.abs(w) * h
        E *= 1.01
    assert np.abs(St - M_exact_float).max() < 1e-12
    return E


def limbs_of(x):
    """Balanced base-256 digits (each in [-128,127]) of integer x, least significant first."""
    d = []
    for _ in range(NLIMB):
        r = ((x + 128) % 256) - 128
        d.append(r)
        x = (x - r) // 256
    assert x == 0
    return d


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "jpeg-encoder-opencl_amd", "csrc", "jpeg_screen_tables.h"))
    ap.add_argument("--std-out", default=os.path.join(ROOT, "tests", "golden", "std_dct_q39.i64"))
    args = ap.parse_args()
    M = exact_map()
    Mf = np.array([[float(x) for x in r] for r in M])
    eps_nat = error_bound(Mf)
    limb = np.zeros((NLIMB, 64, 64), np.int8)
    max_fix_err = Fraction(0)
    for R in range(64):
        nat = ZIGZAG[R]
        for k in range(64):
            x = M[nat][k] * (1 << FRAC_BITS)
            xi = int(round(x))          # Fraction.__round__: exact, ties to even
            max_fix_err = max(max_fix_err, abs(x - xi))
            for l, d in enumerate(limbs_of(xi)):
                limb[l, R, k] = d
    assert max_fix_err <= Fraction(1, 2)
    # row 0 is exactly SCALE_00 * ones: the kernel recovers sum(p) from it
    l0 = sum(int(limb[l, 0, 0]) << (8 * l) for l in range(NLIMB))
    assert all(sum(int(limb[l, 0, k]) << (8 * l) for l in range(NLIMB)) == l0 for k in range(64))
    eps = np.array([eps_nat[ZIGZAG[R]] for R in range(64)])

    # empirical sanity: the fp64 chain vs the fixed-point map on random + extreme inputs
    rng = np.random.default_rng(0)
    Li = np.zeros((64, 64), object)
    for R in range(64):
        for k in range(64):
            Li[R, k] = sum(int(limb[l, R, k]) << (8 * l) for l in range(NLIMB))
    worst = 0.0
    cosm = np.array(COS)
    for it in range(300):
        if it % 3 == 0:
            p = rng.integers(-128, 128, 64)
        elif it % 3 == 1:
            p = rng.choice([-128, 127], 64)
        else:
            p = np.full(64, rng.integers(-128, 128))
        P = p.astype(np.float64).copy()
        for u in range(8):
            for v in range(8):
                s = 0.0
                for y in range(8):
                    for x in range(8):
                        s += P[y * 8 + x] * cosm[x][u] * cosm[y][v]
                s *= scale(u, v)
                P[v * 8 + u] = s
        for R in range(64):
            Y = sum(Li[R, k] * int(p[k]) for k in range(64))
            d = abs(Fraction(P[ZIGZAG[R]]) - Fraction(Y, 1 << FRAC_BITS))
            worst = max(worst, float(d))
    fix_bound = 64 * 128 * 2.0 ** -(FRAC_BITS + 1)
    print("eps max %.3e, fixed-point bound %.3e, worst observed |chain - fixed| %.3e" % (eps.max(), fix_bound, worst))
    assert worst <= eps.max() + fix_bound

    # ---- standard mode: the TRUE 8x8 DCT-II as the same kind of fixed-point map,
    # L_std[(v,u)][(y,x)] = 1/4 a(u) a(v) cos((2x+1)u pi/16) cos((2y+1)v pi/16), rounded to 2^-39.
    # Standard mode is DEFINED by this integer table (q = round-half-away(sum / (Q*2^39)) in exact
    # integer arithmetic), so GPU and test oracle agree bit for bit by construction.
    import mpmath
    mpmath.mp.dps = 60
    std = np.zeros((64, 64), np.int64)
    for R in range(64):
        nat = ZIGZAG[R]
        v, u = nat // 8, nat % 8
        for k in range(64):
            y, x = k // 8, k % 8
            au = mpmath.sqrt(mpmath.mpf(1) / 2) if u == 0 else mpmath.mpf(1)
            av = mpmath.sqrt(mpmath.mpf(1) / 2) if v == 0 else mpmath.mpf(1)
            val = au * av / 4 * mpmath.cos((2 * x + 1) * u * mpmath.pi / 16) * mpmath.cos((2 * y + 1) * v * mpmath.pi / 16)
            std[R, k] = int(mpmath.nint(val * (1 << FRAC_BITS)))
    assert (std[0] == 1 << (FRAC_BITS - 3)).all()  # row 0 is exactly 1/8: DC = sum/8
    std_limb = np.zeros((NLIMB, 64, 64), np.int8)
    for R in range(64):
        for k in range(64):
            for l, d in enumerate(limbs_of(int(std[R, k]))):
                std_limb[l, R, k] = d
    std.astype("<i8").tofile(args.std_out)

    out = args.out
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_screen_tables.py -- do not edit.\n")
        f.write("// Fixed-point copy of the exact linear map of the reference's in-place chain\n")
        f.write("// (utils.cpp:314-348), rows in zig-zag order, and the rigorous rounding-error bound\n")
        f.write("// of the fp64 chain against that map.  See DESIGN.md section 4.3.\n")
        f.write("#pragma once\n#include <stdint.h>\n\nnamespace mi355 {\n\n")
        f.write("constexpr int kScreenFracBits = %d;\nconstexpr int kScreenLimbs = %d;\n" % (FRAC_BITS, NLIMB))
        f.write("// |fixed-point map - exact map| <= 64*128*2^-(frac+1) per output\n")
        f.write("constexpr double kScreenFixErr = %s;\n" % float.hex(fix_bound))
        f.write("// round(SCALE_00 * 2^frac): every entry of row 0\n")
        f.write("constexpr long long kScreenRow0 = %dLL;\n\n" % l0)
        f.write("static const double kScreenEps[64] = {\n")
        for R in range(0, 64, 4):
            f.write("    " + ", ".join(float.hex(float(e)) for e in eps[R:R + 4]) + ",\n")
        f.write("};\n\n// [limb][zig-zag row][input sample k = y*8+x], balanced base-256 digits, limb 0 least significant\n")
        f.write("static const int8_t kScreenLimb[%d][64][64] = {\n" % NLIMB)
        for l in range(NLIMB):
            f.write("  {\n")
            for R in range(64):
                f.write("    {" + ",".join(str(int(v)) for v in limb[l, R]) + "},\n")
            f.write("  },\n")
        f.write("};\n\n// standard mode: the true DCT-II, same format (rows in zig-zag order)\n")
        f.write("static const int8_t kStdLimb[%d][64][64] = {\n" % NLIMB)
        for l in range(NLIMB):
            f.write("  {\n")
            for R in range(64):
                f.write("    {" + ",".join(str(int(v)) for v in std_limb[l, R]) + "},\n")
            f.write("  },\n")
        f.write("};\n\n}  // namespace mi355\n")
    print("wrote", out)


if __name__ == "__main__":
    main()
