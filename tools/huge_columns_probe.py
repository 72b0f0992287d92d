"""Wall time of pantax_hip_pao_solve_batch on the committed LPs of more than 256 columns (tests/golden/lp_huge_cases.npz) and
of the 65..256-column ones (lp_wide_cases.npz): columns, LP rows, pivots, ms per solve (second call: buffers warm), ms per pivot."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from pantax_amd.engine import Engine  # noqa: E402


def paths_from_masks(mask, p):
    offs, nodes = [0], []
    m2 = mask.reshape(len(mask), -1)
    for k in range(p):
        sel = np.nonzero((m2[:, k >> 6] >> np.uint64(k & 63)) & np.uint64(1))[0]
        nodes.append(sel.astype(np.uint32)); offs.append(offs[-1] + len(sel))
    return np.array(offs, dtype=np.uint64), np.concatenate(nodes)


def main():
    eng = Engine()
    gd = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
    for name in ("lp_wide_cases.npz", "lp_huge_cases.npz"):
        z = np.load(os.path.join(gd, name))
        for i in range(int(z["n_cases"])):
            mask, a, ub, objh = z["mask_%d" % i], z["a_%d" % i], z["ub_%d" % i], float(z["obj_%d" % i])
            p = len(ub)
            po, pn = paths_from_masks(mask, p)
            sp = [(np.ones(len(a), dtype=np.int64), a, None, po, pn, np.arange(p))]
            fx = [(ub == 0).astype(np.uint8)]
            best = None
            for rep in range(2):
                t0 = time.perf_counter()
                (x, r, obj, st, it), = eng.pao_solve_batch(sp, fx)
                dt = (time.perf_counter() - t0) * 1e3
                best = dt if best is None else min(best, dt)
            rows = int(((a > 0) & (mask.reshape(len(mask), -1) != 0).any(1)).sum())
            print("%-20s case %d: %4d columns %6d rows  status %d  %5d pivots  %9.2f ms  (%.3f ms / pivot)  obj rel err %.1e"
                  % (name, i, p, rows, st, it, best, best / max(it, 1), abs(obj - objh) / max(abs(objh), 1e-300)))


if __name__ == "__main__":
    main()
