"""Golden LPs with more than 64 candidate columns whose optimum is a POINT, so that the solution vector itself -- first_sol per
strain -- can be compared on the wide path, not only the objective (run in the build container only; TEST INFRASTRUCTURE).

Same model and third-party pin as gen_golden_wide.py (SciPy's bundled HiGHS; profile.rs:1312-1460 / 2754-2822).  A case is
kept as it is; per column k the generator solves min x_k and max x_k over the optimal face {objective <= obj* (1 + 1e-12)} and
records `determined[k]` = the two agree to 1e-8: the test compares x on exactly those columns (all of them when the optimum
is a point).  Outputs tests/golden/lp_wide_unique_cases.npz.
"""
import os
import sys

import numpy as np
from scipy import sparse
from scipy.optimize import linprog

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden_wide import pack   # noqa: E402


def system(M, a, ub):
    rows = np.nonzero(a > 0)[0]
    n, p = len(rows), M.shape[1]
    As = sparse.csr_matrix(M[rows].astype(float))
    I = sparse.identity(n, format="csr")
    Aub = sparse.vstack([sparse.hstack([As, -I]), sparse.hstack([-As, -I])]).tocsr()
    bub = np.concatenate([a[rows], -a[rows]])
    bounds = [(0, float(u)) for u in ub] + [(0, None)] * n
    return n, p, Aub, bub, bounds


def make_case(rng, n, p, present):
    """Every strain has private nodes (so its column is pinned by its own rows), clade nodes overlap neighbours, and the
    coverages are generic reals: ties -- the source of optimal faces -- do not occur."""
    truth = np.zeros(p)
    truth[rng.choice(p, present, replace=False)] = rng.lognormal(np.log(8), 0.8, present)
    M = np.zeros((n, p), dtype=bool)
    kind = rng.random(n)
    for i in range(n):
        if i < 3 * p:
            M[i, i % p] = True                                   # three private nodes per strain
        elif kind[i] < 0.3:
            M[i] = True
        else:
            lo = int(rng.integers(0, p)); w = int(rng.integers(2, max(3, p // 4)))
            M[i, np.arange(lo, lo + w) % p] = True
    lam = M.astype(float) @ truth
    a = lam * np.exp(rng.normal(0, 0.15, n)) + rng.random(n) * 0.3 + 0.01
    ub = np.full(p, 1.05 * a.max())
    return M, a, ub


def main(out):
    rng = np.random.default_rng(20261003)
    cases = {}
    specs = [(700, 65, 5), (900, 96, 8), (1100, 130, 10)]
    for i, (n, p, present) in enumerate(specs):
        M, a, ub = make_case(rng, n, p, present)
        nr, p, Aub, bub, bounds = system(M, a, ub)
        c = np.concatenate([np.zeros(p), np.ones(nr) / nr])
        r = linprog(c, A_ub=Aub, b_ub=bub, bounds=bounds, method="highs")
        assert r.status == 0
        obj = r.fun
        # the optimal face: objective row appended
        A2 = sparse.vstack([Aub, sparse.csr_matrix(c)]).tocsr()
        b2 = np.concatenate([bub, [obj * (1 + 1e-12) + 1e-15]])
        lo, hi = np.zeros(p), np.zeros(p)
        for k in range(p):
            e = np.zeros(p + nr); e[k] = 1.0
            r1 = linprog(e, A_ub=A2, b_ub=b2, bounds=bounds, method="highs")
            r2 = linprog(-e, A_ub=A2, b_ub=b2, bounds=bounds, method="highs")
            assert r1.status == 0 and r2.status == 0
            lo[k], hi[k] = r1.fun, -r2.fun
        det = (hi - lo) <= 1e-8 * np.maximum(1.0, np.abs(hi))
        cases["mask_%d" % i] = pack(M); cases["a_%d" % i] = a; cases["ub_%d" % i] = ub
        cases["x_%d" % i] = 0.5 * (lo + hi); cases["obj_%d" % i] = obj; cases["determined_%d" % i] = det
        print("case %d: n=%d p=%d obj=%.12g determined %d / %d columns, nnz(x)=%d" % (i, n, p, obj, int(det.sum()), p, int((r.x[:p] > 1e-9).sum())), flush=True)
    cases["n_cases"] = len(specs)
    np.savez_compressed(out, **cases)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "lp_wide_unique_cases.npz"))
