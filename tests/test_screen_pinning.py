"""Pins the screened transform, the piece that decides whether the GPU path is bit-exact:

* the generated table header is reproducible from its generator, byte for byte;
* the bound the accept thresholds rest on -- |fp64 chain - fixed-point map| <= eps_k + 2^-27 for every output k --
  holds on more than a million blocks: random, two-level, per-row worst cases and their neighbourhoods;
* on the GPU: blocks BUILT to sit on a rounding boundary (z_k within ~1e-10 of a half-integer, by a
  meet-in-the-middle search over the fixed-point rows) must take the second look and the exact fp64 chain
  (the counters of mi355_jpeg_screen_stats say so) and still come out equal to the oracle.  A regression of the
  thresholds towards "accept everything" flips about half of those coefficients and zeroes the counters.
"""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as ol
from conftest import GOLD, ROOT

HDR = os.path.join(ROOT, "jpeg-encoder-opencl_amd", "csrc", "jpeg_screen_tables.h")
FRAC = 39


def parse_tables():
    """(eps[64] float, Lt[64][64] int64, digits[5][64][64]) of the strict map from the committed header; rows in zig-zag order."""
    text = open(HDR).read()
    eps_txt = re.search(r"kScreenEps\[64\] = \{(.*?)\};", text, re.S).group(1)
    eps = np.array([float.fromhex(t) for t in re.findall(r"-?0x[0-9a-fA-F.]+p[+-]?\d+", eps_txt)])
    limb_txt = re.search(r"kScreenLimb\[5\]\[64\]\[64\] = \{(.*?)\n\};", text, re.S).group(1)
    vals = np.array([int(t) for t in re.findall(r"-?\d+", limb_txt)], np.int64)
    assert eps.shape == (64,) and vals.size == 5 * 64 * 64
    d = vals.reshape(5, 64, 64)
    Lt = sum(d[l] << (8 * l) for l in range(5))
    return eps, Lt, d


def chain(p):
    """The reference's in-place fp64 chain (oracle, utils.cpp:314-348) on blocks p [n][64] (level-shifted samples);
    returns the coefficients in ZIG-ZAG row order like the tables."""
    P = np.ascontiguousarray(p, np.float64).copy()
    ol.oracle().orc_dct_blocks(P.ctypes.data, P.shape[0])
    return P[:, ol.zigzag_order()]


def test_tables_header_regenerates_byte_for_byte(tmp_path):
    out, std = str(tmp_path / "tables.h"), str(tmp_path / "std.i64")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_screen_tables.py"), "--out", out, "--std-out", std],
                          stdout=subprocess.DEVNULL)
    assert open(out, "rb").read() == open(HDR, "rb").read(), "csrc/jpeg_screen_tables.h is not what tools/gen_screen_tables.py writes"
    assert open(std, "rb").read() == open(os.path.join(GOLD, "std_dct_q39.i64"), "rb").read()


def test_chain_stays_within_the_bound_of_the_fixed_point_map():
    eps, Lt, d = parse_tables()
    rng = np.random.default_rng(20261004)
    sets = [rng.integers(-128, 128, (400_000, 64)),                              # uniform noise
            rng.choice(np.array([-128, 127]), (300_000, 64)),                    # two-level
            rng.integers(-3, 4, (50_000, 64)) + rng.integers(-120, 120, (50_000, 1)),  # flat + small noise
            np.repeat(np.arange(-128, 128)[:, None], 64, 1)]                     # constant blocks
    # per-row worst cases of |L p| (the sign pattern of the row) and their neighbourhoods: random flips / random shrinks
    worst = np.where(Lt >= 0, 127, -128)
    near = []
    for R in range(64):
        base = np.repeat(worst[R][None], 2000, 0)
        flip = rng.random((2000, 64)) < rng.random((2000, 1)) * 0.3
        base = np.where(flip, -1 - base, base)                                   # 127 <-> -128
        near.append(base)
        near.append(-1 - base)
        near.append((base * rng.random((2000, 1))).astype(np.int64))
    sets += [worst, -1 - worst, np.concatenate(near)]
    p = np.concatenate(sets).astype(np.int64)
    assert p.shape[0] > 1_000_000 and p.min() >= -128 and p.max() <= 127
    tol = eps + 2.0 ** -27
    worst_ratio = 0.0
    for lo in range(0, p.shape[0], 200_000):
        blk = p[lo:lo + 200_000]
        c = chain(blk)                                    # [n][64] doubles, zig-zag order
        y = blk @ Lt.T                                    # exact in int64 (< 2^53)
        assert np.abs(y).max() < 2 ** 52
        diff = np.abs(c - y.astype(np.float64) * 2.0 ** -FRAC)   # both terms exact doubles; the subtraction errs by < 1e-13
        worst_ratio = max(worst_ratio, float((diff / tol).max()))
        assert (diff <= tol + 1e-13).all()
        # the dropped-digit bound of the three-digit first look: |digit1 p * 256 + digit0 p| <= 128 (256 S1 + S0)
        drop = np.abs((blk @ d[1].T) * 256 + blk @ d[0].T)
        bound = 128 * (256 * np.abs(d[1]).sum(1) + np.abs(d[0]).sum(1))
        assert (drop <= bound).all()
    assert 0.01 < worst_ratio <= 1.0   # the bound is respected, and not vacuous by orders of magnitude
    # coefficient 0 is exact: row 0 of the map is SCALE_00 * ones, the chain's first step adds integers exactly
    assert (Lt[0] == Lt[0, 0]).all()


def near_tie_blocks(Lt, R, Q, rng, count):
    """Blocks p in [-128,127]^64 with (Lt[R] . p) / 2^39 / Q within ~1e-10 of a half-integer: random base block, then four
    samples re-chosen by a meet-in-the-middle search (65536 x 65536 sums of two, sorted + searchsorted)."""
    out = []
    vals = np.arange(-128, 128, dtype=np.int64)
    while len(out) < count:
        p = rng.integers(-100, 101, 64)
        idx = rng.choice(64, 4, replace=False)
        a = Lt[R, idx]
        if (np.abs(a) < 2 ** 20).any():
            continue
        base = int(Lt[R] @ p - a @ p[idx])
        m = int(rng.integers(-3, 4))
        target = int(round((m + 0.5) * Q * 2 ** FRAC)) - base
        A = (a[0] * vals[:, None] + a[1] * vals[None, :]).ravel()
        B = (a[2] * vals[:, None] + a[3] * vals[None, :]).ravel()
        order = np.argsort(B)
        Bs = B[order]
        j = np.clip(np.searchsorted(Bs, target - A), 1, Bs.size - 1)
        cand = np.stack([Bs[j - 1], Bs[j]])
        err = np.abs(A[None, :] + cand - target)
        k = np.unravel_index(np.argmin(err), err.shape)
        ia = int(k[1])
        ib = int(order[j[ia] - 1 + k[0]])
        p[idx[0]], p[idx[1]] = vals[ia // 256], vals[ia % 256]
        p[idx[2]], p[idx[3]] = vals[ib // 256], vals[ib % 256]
        z = int(Lt[R] @ p) / 2.0 ** FRAC / Q
        if abs(abs(z - np.floor(z)) - 0.5) < 2e-9 / Q:
            out.append(p.copy())
    return np.array(out)


@pytest.mark.gpu
@pytest.mark.parametrize("quality", [50, 90, 100])
def test_blocks_built_on_rounding_boundaries(jpeg, quality):
    eps, Lt, _ = parse_tables()
    rng = np.random.default_rng(7 + quality)
    ql, qc = ol.quant_tables(quality)
    zz = ol.zigzag_order()
    # near-grey pixels for every luma value: the reference's truncating fp64 colour conversion (oracle) of a small set of
    # candidates, inverted (plain grey r = g = b = v skips a few values of Y)
    cand = np.array([[min(255, max(0, v + dr)), v, min(255, max(0, v + db))] for v in range(256) for dr in (0, 1, -1, 2) for db in (0, 1, -1, 2)],
                    np.uint8)
    ycc = cand.copy()
    ol.oracle().orc_csc(ycc.ctypes.data, ycc.shape[0])
    v_for = {}
    for px, y in zip(cand, ycc[:, 0]):
        v_for.setdefault(int(y), px)
    assert len(v_for) == 256
    rows = [1, 2, 3, 5, 9, 14, 20, 27, 35, 44, 54, 63]
    blocks = []
    for R in rows:
        blocks.append(near_tie_blocks(Lt, R, int(ql[zz[R]]), rng, 12))
    blocks = np.concatenate(blocks) + 128                   # samples 0..255
    n = blocks.shape[0]
    bw = 16
    img = np.zeros((8 * ((n + bw - 1) // bw), 8 * bw, 3), np.uint8)
    img[...] = 128
    for i, b in enumerate(blocks):
        by, bx = divmod(i, bw)
        img[8 * by:8 * by + 8, 8 * bx:8 * bx + 8, :] = np.array([v_for[int(s)] for s in b], np.uint8).reshape(8, 8, 3)
    e2 = jpeg.Encoder(0)
    e2.set_quant(ql, qc)
    e2.screen_stats(reset=True)
    o = ol.oracle_encode(img, ql, qc, False, ol.KEEP_ZIGZAG)
    cf = e2.probe_coefficients(img, 0)
    assert np.array_equal(cf.astype(np.int32), o.zigzag)
    bits, nb = e2.encode_scan(img, 0)
    assert nb[0] == o.n_bits and np.array_equal(bits[0], o.bits)
    looks, exact = e2.screen_stats()
    # every built block has a coefficient closer to a boundary than the first look can decide; most are closer than
    # the second look's margin too and go through the exact chain
    assert looks >= len(rows) and exact >= n // 2, (looks, exact, n)
    # and on plain noise the exact path is (almost) never taken
    e2.screen_stats(reset=True)
    e2.encode_scan(ol.lcg_frame(1920, 1080, 3), 0)
    looks2, exact2 = e2.screen_stats()
    assert exact2 <= 3 and looks2 > 0
    e2.close()
