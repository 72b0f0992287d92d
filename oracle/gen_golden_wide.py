"""Golden LPs with MORE THAN 64 candidate columns (run in the build container only; TEST INFRASTRUCTURE).

Same model and same third-party pin as gen_golden.py (SciPy's bundled HiGHS on
  min (1/n) sum_v y_v,  y_v >= +-(sum_{k in mask_v} x_k - a_v),  0 <= x_k <= ub_k,
profile.rs:1312-1460 / 2754-2822), for species whose first filter keeps 65 .. 256 strains: the
reference's coefficient matrix is dense nvert x npaths (profile.rs:1333-1342) and has no column cap.
Membership is stored as (n, ceil(p / 64)) uint64 words: bit k & 63 of word k >> 6 = node on path k.
Outputs tests/golden/lp_wide_cases.npz.
"""
import os
import sys

import numpy as np
from scipy import sparse
from scipy.optimize import linprog


def pack(M):
    n, p = M.shape
    nw = (p + 63) // 64
    out = np.zeros((n, nw), dtype=np.uint64)
    for k in range(p):
        out[:, k >> 6] |= M[:, k].astype(np.uint64) << np.uint64(k & 63)
    return out


def highs_lad(M, a, ub):
    rows = np.nonzero(a > 0)[0]
    n, p = len(rows), M.shape[1]
    As = sparse.csr_matrix(M[rows].astype(float))
    I = sparse.identity(n, format="csr")
    c = np.concatenate([np.zeros(p), np.ones(n) / n])
    Aub = sparse.vstack([sparse.hstack([As, -I]), sparse.hstack([-As, -I])]).tocsr()
    bub = np.concatenate([a[rows], -a[rows]])
    bounds = [(0, float(u)) for u in ub] + [(0, None)] * n
    r = linprog(c, A_ub=Aub, b_ub=bub, bounds=bounds, method="highs")
    assert r.status == 0
    return r.x[:p], r.fun


def make_case(rng, n, p, integer, present, fix=None):
    """Strain-like membership: core nodes on every path, clade nodes on a random subset of neighbouring strains, private
    nodes on one strain; `present` strains have coverage."""
    truth = np.zeros(p)
    truth[rng.choice(p, present, replace=False)] = rng.lognormal(np.log(8), 0.8, present)
    M = np.zeros((n, p), dtype=bool)
    kind = rng.random(n)
    for i in range(n):
        if kind[i] < 0.35:
            M[i] = True
        elif kind[i] < 0.8:
            lo = int(rng.integers(0, p)); w = int(rng.integers(2, max(3, p // 3)))
            M[i, np.arange(lo, lo + w) % p] = True
            M[i] &= rng.random(p) < 0.9
        else:
            M[i, int(rng.integers(0, p))] = True
    lam = M.astype(float) @ truth
    a = rng.poisson(lam).astype(float) if integer else rng.poisson(lam * 30) / 30.0 + (rng.random(n) < 0.05) * rng.random(n)
    a[rng.random(n) < 0.1] = 0.0
    M[rng.random(n) < 0.03] = False          # covered nodes on no candidate path: constant objective terms
    ub = np.full(p, 1.05 * a.max())
    if fix is not None:
        ub[fix] = 0.0
    return M, a, ub


def main(out):
    rng = np.random.default_rng(20261002)
    cases = {}
    specs = [(1500, 65, False, 4, None), (2500, 100, True, 6, None), (3000, 128, False, 8, [3, 64, 127]),
             (3000, 129, False, 5, None), (4000, 200, True, 10, [0, 70, 150, 199]), (5000, 256, False, 12, None)]
    for i, (n, p, integer, present, fix) in enumerate(specs):
        M, a, ub = make_case(rng, n, p, integer, present, fix)
        x, obj = highs_lad(M, a, ub)
        cases["mask_%d" % i] = pack(M); cases["a_%d" % i] = a; cases["ub_%d" % i] = ub
        cases["x_%d" % i] = x; cases["obj_%d" % i] = obj
        print("case %d: n=%d p=%d rows=%d obj=%.12g nnz(x)=%d" % (i, n, p, int((a > 0).sum()), obj, int((x > 1e-9).sum())))
    cases["n_cases"] = len(specs)
    np.savez_compressed(out, **cases)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "lp_wide_cases.npz"))
