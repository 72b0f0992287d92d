#!/usr/bin/env python3
"""Regression fixtures for the two load-time operators beside the hot path: the row sampler (a11) and the long-read
GAF filter.  The expected values are the ORACLE's (oracle/pantax_oracle.c) -- neither rand 0.9.2 nor the
reference binary can run here, so these pin the restatement against itself over time, not against the reference.
usage: python oracle/gen_golden_aux.py   (writes tests/golden/sampler_positions.json and tests/golden/gaf_filter.json)"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path = [q for q in sys.path if os.path.abspath(q or ".") != HERE]   # `oracle` must resolve to the package, not oracle/oracle.py
sys.path.insert(0, os.path.join(HERE, ".."))
from oracle import oracle as orc  # noqa: E402
from tests.helpers import make_longread_gaf  # noqa: E402


def main():
    out = os.path.join(HERE, "..", "tests", "golden")
    cases = []
    for n, k in [(600000, 500000), (1000, 500), (200000, 500), (700000, 500), (50, 10), (1000, 20), (5000, 100), (501, 500), (3000, 162), (3000, 163)]:
        pos = orc.sample_sorted_positions(n, k)
        cases.append(dict(n=n, amount=k, seed=42, first=pos[:8].tolist(), last=pos[-4:].tolist(), sum=int(pos.astype("uint64").sum()),
                          xor=int(__import__("numpy").bitwise_xor.reduce(pos.astype("uint32")))))
    json.dump(dict(comment="sample_sorted (profile.rs:1287-1295) as restated by the oracle; parity with rand 0.9.2 unpinned", cases=cases),
              open(os.path.join(out, "sampler_positions.json"), "w"), indent=1)
    txt = make_longread_gaf(9, 120, path_ids=6)
    keep, nrec = orc.gaf_filter(txt)
    json.dump(dict(comment="filter_max_alignment_mt (gaf_filter.rs:44-97) as restated by the oracle on a generated GAF", text=txt.decode("latin-1"),
                   n_records=int(nrec), kept_lines=[int(i) for i in keep.nonzero()[0]]),
              open(os.path.join(out, "gaf_filter.json"), "w"), indent=1)
    print("wrote sampler_positions.json (%d cases), gaf_filter.json (%d lines, %d kept)" % (len(cases), len(keep), int(keep.sum())))


if __name__ == "__main__":
    main()
