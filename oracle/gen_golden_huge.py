"""Golden LPs with MORE THAN 256 candidate columns (run in the build container only; TEST INFRASTRUCTURE).

Same model, same third-party pin (SciPy's bundled HiGHS) and same case generator as gen_golden_wide.py, for species whose
first filter keeps hundreds of strains: the reference's coefficient matrix is dense nvert x npaths (profile.rs:1333-1342)
and has no column cap.  Membership words as there: (n, ceil(p / 64)) uint64.
Outputs tests/golden/lp_huge_cases.npz.
"""
import os
import sys
import time

import numpy as np

from gen_golden_wide import highs_lad, make_case, pack


def main(out):
    rng = np.random.default_rng(20261003)
    cases = {}
    specs = [(4000, 257, False, 6, None), (5000, 320, True, 9, [0, 100, 256, 319]), (6000, 513, False, 12, None),
             (8000, 700, True, 15, [5, 300, 699]), (9000, 1100, False, 20, None)]
    for i, (n, p, integer, present, fix) in enumerate(specs):
        M, a, ub = make_case(rng, n, p, integer, present, fix)
        t0 = time.time()
        x, obj = highs_lad(M, a, ub)
        cases["mask_%d" % i] = pack(M); cases["a_%d" % i] = a; cases["ub_%d" % i] = ub
        cases["x_%d" % i] = x; cases["obj_%d" % i] = obj
        print("case %d: n=%d p=%d rows=%d obj=%.12g nnz(x)=%d  (HiGHS %.1f s)" % (i, n, p, int((a > 0).sum()), obj, int((x > 1e-9).sum()), time.time() - t0))
    cases["n_cases"] = len(specs)
    np.savez_compressed(out, **cases)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "lp_huge_cases.npz"))
