"""Golden vectors from the reference's ACTUAL model (run in the build container only; SciPy does not travel).

gurobi_opt (profile.rs:1297-1511) does not hand the solver a plain LP: next to x_j in [0, 1.05 max a] and y_v >= +-(sum_j A_vj x_j - a_v)
it adds one BINARY indicator z_j per path with z_j >= (x_j - minimization_min_cov) / (2 max a) (profile.rs:1363-1375) and
sum z <= npaths (:1377); the second solve adds x_j == 0 rows for the paths the second filter dropped (:1484-1488).  This
script builds exactly that mixed-integer model with scipy.optimize.milp (HiGHS 1.8 branch-and-cut) and the LP without the
indicators with linprog, checks that both give the same objective (the indicators are inert: they are not in the objective
and z = 1 always satisfies them), and commits inputs + both answers as tests/golden/lp_milp_cases.npz:
  * the BASELINE.md section 2 shape: 20 000 rows x 10 paths, 60 % dense, Poisson-noised coverage  (first solve)
  * the same with three paths pinned to zero                                                        (second solve)
  * integer-tied coverages (a_v in a handful of integer values: the degenerate regime of real short-read data)
  * a small nested-clade case with minimization_min_cov > 0 (the indicator's offset; still inert)
The GPU solver seam and the oracle's LAD solver are compared with these objectives to 1e-9 (tests/test_gpu_parity.py,
tests/test_oracle.py)."""
import os
import sys
import time

import numpy as np
from scipy import sparse
from scipy.optimize import Bounds, LinearConstraint, linprog, milp


def build(mask, a, p, fixed, min_cov=0.0):
    rows = np.nonzero(a > 0)[0]                     # valid_nodes (profile.rs:1380-1385)
    n = len(rows)
    amax = float(a.max())                           # max over ALL nodes (profile.rs:1316-1319)
    A = sparse.csr_matrix(np.stack([((mask[rows] >> np.uint64(k)) & np.uint64(1)).astype(float) for k in range(p)], 1))
    b = a[rows]
    I = sparse.identity(n, format="csr")
    Zp = sparse.csr_matrix((n, p))
    # variable order: x (p), z (p), y (n)
    cons = [LinearConstraint(sparse.hstack([A, Zp, -I]).tocsr(), -np.inf, b),          # sum x - y <= a
            LinearConstraint(sparse.hstack([-A, Zp, -I]).tocsr(), -np.inf, -b)]        # -sum x - y <= -a
    s = 1.0 / (2.0 * amax)
    Ip = sparse.identity(p, format="csr")
    cons.append(LinearConstraint(sparse.hstack([s * Ip, -Ip, sparse.csr_matrix((p, n))]).tocsr(), -np.inf, s * min_cov))   # z_j >= (x_j - min_cov) s
    cons.append(LinearConstraint(sparse.hstack([sparse.csr_matrix((1, p)), sparse.csr_matrix(np.ones((1, p))), sparse.csr_matrix((1, n))]).tocsr(), -np.inf, p))
    ub_x = np.where(fixed, 0.0, 1.05 * amax)        # x_j == 0 rows of the second solve (profile.rs:1484-1488)
    lb = np.zeros(2 * p + n)
    ub = np.concatenate([ub_x, np.ones(p), np.full(n, np.inf)])
    c = np.concatenate([np.zeros(2 * p), np.ones(n) / n])
    integrality = np.concatenate([np.zeros(p), np.ones(p), np.zeros(n)])
    return c, cons, Bounds(lb, ub), integrality, (A, b, ub_x, n)


def solve_both(mask, a, p, fixed, min_cov=0.0):
    c, cons, bounds, integrality, (A, b, ub_x, n) = build(mask, a, p, fixed, min_cov)
    t0 = time.perf_counter()
    r = milp(c, constraints=cons, bounds=bounds, integrality=integrality)
    t_milp = time.perf_counter() - t0
    assert r.status == 0, r.message
    I = sparse.identity(n, format="csr")
    Aub = sparse.vstack([sparse.hstack([A, -I]), sparse.hstack([-A, -I])]).tocsr()
    t0 = time.perf_counter()
    l = linprog(np.concatenate([np.zeros(p), np.ones(n) / n]), A_ub=Aub, b_ub=np.concatenate([b, -b]),
                bounds=[(0, float(u)) for u in ub_x] + [(0, None)] * n, method="highs")
    t_lp = time.perf_counter() - t0
    assert l.status == 0
    assert abs(r.fun - l.fun) <= 1e-9 * max(1.0, abs(l.fun)), (r.fun, l.fun)       # the indicators are inert
    return r.x[:p], float(r.fun), l.x[:p], float(l.fun), t_milp, t_lp


def dense_case(rng, n, p, integer):
    """0/1 matrix 60 % dense, Poisson-noised coverage"""
    M = rng.random((n, p)) < 0.6
    mask = np.zeros(n, dtype=np.uint64)
    for k in range(p):
        mask |= M[:, k].astype(np.uint64) << np.uint64(k)
    truth = np.where(rng.random(p) < 0.6, rng.lognormal(np.log(8), 0.7, p), 0.0)
    lam = M.astype(float) @ truth
    a = rng.poisson(lam).astype(float) if integer else rng.poisson(lam * 20) / 20.0
    a[rng.random(n) < 0.05] = 0.0
    return mask, a


def main(out):
    rng = np.random.default_rng(20260502)
    cases = []
    mask, a = dense_case(rng, 20000, 10, False)
    cases.append(("baseline_20000x10", mask, a, 10, np.zeros(10, bool), 0.0))
    cases.append(("baseline_20000x10_second_solve", mask, a, 10, np.isin(np.arange(10), [1, 4, 7]), 0.0))
    mask, a = dense_case(rng, 6000, 8, True)
    cases.append(("integer_ties_6000x8", mask, a, 8, np.zeros(8, bool), 0.0))
    cases.append(("integer_ties_6000x8_second_solve", mask, a, 8, np.isin(np.arange(8), [0, 5]), 0.0))
    mask, a = dense_case(rng, 1500, 6, False)
    cases.append(("min_cov_1500x6", mask, a, 6, np.zeros(6, bool), 2.0))
    z = {"n_cases": np.int64(len(cases))}
    for i, (name, mask, a, p, fixed, min_cov) in enumerate(cases):
        xm, fm, xl, fl, tm, tl = solve_both(mask, a, p, fixed, min_cov)
        print("%-36s rows %6d  milp %.12g (%.1f s)  lp %.12g (%.1f s)  |x_milp - x_lp|_1 = %.3g" % (name, int((a > 0).sum()), fm, tm, fl, tl, np.abs(xm - xl).sum()))
        z["name_%d" % i] = np.array(name)
        z["mask_%d" % i] = mask; z["a_%d" % i] = a; z["fixed_%d" % i] = fixed.astype(np.uint8)
        z["x_milp_%d" % i] = xm; z["obj_milp_%d" % i] = np.float64(fm); z["x_lp_%d" % i] = xl; z["obj_lp_%d" % i] = np.float64(fl)
        z["seconds_milp_%d" % i] = np.float64(tm); z["seconds_lp_%d" % i] = np.float64(tl)
    np.savez_compressed(out, **z)
    print("wrote", out)


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "tests", "golden", "lp_milp_cases.npz"))
