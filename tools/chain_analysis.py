#!/usr/bin/env python3
"""Offline analysis of the reference's in-place transform chain (utils.cpp:314-348).

Computes, with exact dyadic-rational arithmetic (Python ints), the linear map L that
the chain realises in exact real arithmetic on the given double constants, and a
rigorous a-priori bound eps[k] on |chain_fp64(p)[k] - (L p)[k]| over all inputs
p in [-128,127]^64 (standard forward error analysis)."""
import json
import sys
from fractions import Fraction

import numpy as np

COS = [[float.fromhex(h) for h in row] for row in json.load(open(
    __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)),
                               "..", "tests", "golden", "tables.json")))["cos"]]
S00, S0X, SXX = float.fromhex("0x1.ffffffffffffep-4"), float.fromhex("0x1.6a09e667f3bccp-3"), 0.25


def scale(u, v):
    return S00 if (u == 0 and v == 0) else (S0X if (u == 0 or v == 0) else SXX)


def exact_map():
    """M[i][j] as Fractions: final state = M p; also the per-step rows."""
    M = [[Fraction(int(i == j)) for j in range(64)] for i in range(64)]
    rows = []  # (pos, [coef_i as Fraction])
    for u in range(8):
        for v in range(8):
            coef = [Fraction(COS[i % 8][u]) * Fraction(COS[i // 8][v]) * Fraction(scale(u, v)) for i in range(64)]
            new = [sum(coef[i] * M[i][j] for i in range(64)) for j in range(64)]
            pos = v * 8 + u
            M[pos] = new
            rows.append((pos, coef))
    return M, rows


def main():
    M, rows = exact_map()
    L = np.array([[float(x) for x in r] for r in M])
    print("max |L|", np.abs(L).max(), "max row 1-norm", np.abs(L).sum(1).max())
    # --- forward error bound (float64 arithmetic + slack; these are bounds, not results)
    u_ = 2.0 ** -53
    g67 = 67 * u_ / (1 - 67 * u_)
    # exact-arithmetic state maps per step, in float (bounds only)
    St = np.eye(64)
    W = []       # W[k'] = current image of a unit error injected at step k' (64-vector)
    H = []       # bound on |eta_k'|
    E = np.zeros(64)  # bound on accumulated error per state entry
    for k, (pos, coef) in enumerate(rows):
        c = np.array([float(x) for x in coef])
        B = 128.0 * np.abs(St).sum(1)            # max |exact state entry| over inputs
        Hk = g67 * float(np.sum(np.abs(c) * (B + E))) * 1.01
        # advance exact state map and all influence vectors through this step
        St[pos] = c @ St
        for w in W:
            w[pos] = float(c @ w)
        e = np.zeros(64); e[pos] = 1.0
        W.append(e); H.append(Hk)
        E = np.zeros(64)
        for w, h in zip(W, H):
            E += np.abs(w) * h
        E *= 1.01
    eps = E.copy()
    print("eps max %.3e  min %.3e" % (eps.max(), eps.min()))
    print("B final max", (128 * np.abs(L).sum(1)).max())
    np.save("/tmp/chain_L.npy", L)
    np.save("/tmp/chain_eps.npy", eps)
    return M, eps


if __name__ == "__main__":
    main()
