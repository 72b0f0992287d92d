"""Generate LP golden vectors with SciPy-HiGHS (run in the build container only).

The reference hands the PAO model to Gurobi/HiGHS (profile.rs:1312-1460, 2754-2822);
neither the `grb` nor the `highs` crate can be built here, so the pinned third-party
answer is SciPy 1.15.3's bundled HiGHS 1.8.0 on the identical LP
  min (1/n) sum_v y_v,  y_v >= +-(sum_{k in mask_v} x_k - a_v),  0 <= x_k <= ub_k.
Outputs tests/golden/lp_cases.npz (inputs + HiGHS x / objective).  Fixtures are data;
SciPy itself does not travel with the tests.
"""
import os
import sys

import numpy as np
from scipy import sparse
from scipy.optimize import linprog


def highs_lad(mask, a, p, ub):
    rows = np.nonzero(a > 0)[0]
    n = len(rows)
    A = np.stack([((mask[rows] >> np.uint64(k)) & np.uint64(1)).astype(float) for k in range(p)], 1)
    b = a[rows]
    c = np.concatenate([np.zeros(p), np.ones(n) / n])
    As = sparse.csr_matrix(A)
    I = sparse.identity(n, format="csr")
    Aub = sparse.vstack([sparse.hstack([As, -I]), sparse.hstack([-As, -I])]).tocsr()
    bub = np.concatenate([b, -b])
    bounds = [(0, float(u)) for u in ub] + [(0, None)] * n
    r = linprog(c, A_ub=Aub, b_ub=bub, bounds=bounds, method="highs")
    assert r.status == 0
    # uniqueness probe: extremise a random direction over the optimal face
    rng = np.random.default_rng(p * 1000 + n)
    d = rng.normal(size=p)
    A2 = sparse.vstack([Aub, sparse.csr_matrix(c[None, :])]).tocsr()
    b2 = np.concatenate([bub, [r.fun * (1 + 1e-10) + 1e-12]])
    lo = linprog(np.concatenate([d, np.zeros(n)]), A_ub=A2, b_ub=b2, bounds=bounds, method="highs")
    hi = linprog(np.concatenate([-d, np.zeros(n)]), A_ub=A2, b_ub=b2, bounds=bounds, method="highs")
    unique = lo.status == 0 and hi.status == 0 and np.abs(lo.x[:p] - hi.x[:p]).sum() < 1e-6
    return r.x[:p], r.fun, unique


def make_case(rng, n, p, integer, frac_zero_rows=0.1, fix=None):
    truth = np.where(rng.random(p) < 0.5, rng.lognormal(np.log(8), 1, p), 0)
    # phylogenetically nested-ish patterns: core rows (all ones), clade rows, singletons
    mask = np.zeros(n, dtype=np.uint64)
    kind = rng.random(n)
    full = np.uint64((1 << p) - 1)
    clades = [np.uint64(int(rng.integers(1, 1 << p))) for _ in range(max(2, p))]
    for i in range(n):
        if kind[i] < 0.5:
            mask[i] = full
        elif kind[i] < 0.85:
            mask[i] = clades[int(rng.integers(0, len(clades)))]
        else:
            mask[i] = np.uint64(1 << int(rng.integers(0, p)))
    A = np.stack([((mask >> np.uint64(k)) & np.uint64(1)).astype(float) for k in range(p)], 1)
    lam = A @ truth
    if integer:
        a = rng.poisson(lam).astype(float)
    else:
        a = rng.poisson(lam * 30) / 30.0 + (rng.random(n) < 0.05) * rng.random(n)
    a[rng.random(n) < frac_zero_rows] = 0.0
    # some covered nodes on no candidate path (mask 0): constant objective terms
    off = rng.random(n) < 0.03
    mask[off] = 0
    ub = np.full(p, 1.05 * a.max())
    if fix is not None:
        ub[fix] = 0.0
    return mask, a, ub


def main(out):
    rng = np.random.default_rng(20260501)
    cases = {}
    specs = [(60, 2, True, None), (300, 3, False, None), (800, 4, True, None), (1500, 6, False, None),
             (1500, 6, False, [1, 4]), (2500, 10, False, None), (2500, 10, True, [0, 2, 5]), (400, 1, False, None),
             (1200, 8, True, None),
             # more than 16 candidates: the solver's wide (64-variable) instantiation
             (3000, 24, False, None), (4000, 40, True, [3, 7, 20])]
    for i, (n, p, integer, fix) in enumerate(specs):
        mask, a, ub = make_case(rng, n, p, integer, fix=fix)
        x, obj, unique = highs_lad(mask, a, p, ub)
        cases["unique_%d" % i] = np.bool_(unique)
        cases["mask_%d" % i] = mask
        cases["a_%d" % i] = a
        cases["ub_%d" % i] = ub
        cases["x_%d" % i] = x
        cases["obj_%d" % i] = np.float64(obj)
    cases["n_cases"] = np.int64(len(specs))
    np.savez_compressed(out, **cases)
    print("wrote", out)


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "tests", "golden", "lp_cases.npz"))
