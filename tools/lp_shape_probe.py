#!/usr/bin/env python3
"""Times the solver seam on the committed LP fixtures (lp_cases / lp_milp_cases / lp_wide_cases): rows, columns, distinct
membership patterns, pivots and the lad kernel's time per case."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pantax_amd.engine import Engine
eng = Engine(0)
def paths(mask, p):
    m2 = mask.reshape(len(mask), -1); offs = [0]; nodes = []
    for k in range(p):
        sel = np.nonzero((m2[:, k >> 6] >> np.uint64(k & 63)) & np.uint64(1))[0]; nodes.append(sel.astype(np.uint32)); offs.append(offs[-1] + len(sel))
    return np.array(offs, dtype=np.uint64), np.concatenate(nodes)
for fn, key in (("lp_cases.npz", "ub"), ("lp_milp_cases.npz", "fixed"), ("lp_wide_cases.npz", "ub")):
    z = np.load(os.path.join(ROOT, "tests", "golden", fn))
    for i in range(int(z["n_cases"])):
        mask, a = z["mask_%d" % i], z["a_%d" % i]
        fz = (z["ub_%d" % i] == 0).astype(np.uint8) if key == "ub" else z["fixed_%d" % i]
        p = len(fz); po, pn = paths(mask, p)
        rows = (a > 0) & (mask.reshape(len(mask), -1).any(1))
        K = len(np.unique(mask.reshape(len(mask), -1)[rows], axis=0))
        args = (np.ones(len(a), dtype=np.int64), a, np.zeros(len(a), dtype=np.uint64), po, pn, np.arange(p))
        eng.pao_solve(*args, fixed_zero=fz)
        eng.timing_enable(True); eng.timing_reset()
        sp = [(args[0], a, None, po, pn, np.arange(p))]
        out = eng.pao_solve_batch(sp, [fz])
        t = eng.timing_get(); eng.timing_enable(False)
        it = out[0][4]
        ms = t.get("lad_solve_kernel", (0, 0.0))[1]
        print("%-18s case %2d: rows %6d cols %3d patterns %5d pivots %4d  lad %8.3f ms  (%.1f us/pivot)" % (fn, i, int(rows.sum()), p, K, it, ms, 1e3 * ms / max(it, 1)))
