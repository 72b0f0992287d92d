#!/usr/bin/env python3
"""oracle/check_sampler_vectors.py -- TEST INFRASTRUCTURE.  Compares the output of the ten-line Rust program of INTEGRATION.md
(section "Pinning the row sampler against rand 0.9.2": real `sample_sorted`, profile.rs:1287-1295, on the real crates) with
tests/golden/sampler_positions.json, the vectors this repo's restated sampler produces.  One `cargo run` on any machine with a
Rust toolchain turns row a11 / f4 from "restated" into "pinned":

    cargo run --release | python3 oracle/check_sampler_vectors.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
want = {(c["n"], c["amount"]): c for c in json.load(open(os.path.join(ROOT, "tests", "golden", "sampler_positions.json")))["cases"]}
bad = seen = 0
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    got = json.loads(line)
    w = want.get((got["n"], got["amount"]))
    if w is None:
        print("no golden case for", got["n"], got["amount"])
        continue
    seen += 1
    for k in ("first", "last", "sum", "xor"):
        if got[k] != w[k]:
            bad += 1
            print("MISMATCH n=%d amount=%d %s: rand 0.9.2 %r, restated %r" % (got["n"], got["amount"], k, got[k], w[k]))
print("%d cases compared, %d mismatching fields, %d golden cases not seen" % (seen, bad, len(want) - seen))
sys.exit(1 if bad or seen != len(want) else 0)
