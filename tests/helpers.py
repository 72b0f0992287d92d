"""Shared helpers for the parity tests (inputs from golden JSON / synthetic sets)."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_micro_cov():
    with open(os.path.join(ROOT, "tests", "golden", "micro_cov.json")) as f:
        j = json.load(f)
    names = sorted(j["paths"].keys())  # BTreeMap byte order
    node_len = np.array(j["node_len"], dtype=np.int64)
    path_off = np.zeros(len(names) + 1, dtype=np.uint64)
    path_off[1:] = np.cumsum([len(j["paths"][n]) for n in names])
    path_nodes = np.concatenate([np.array(j["paths"][n], dtype=np.uint32) for n in names])
    rs = j["range_start"]
    step_off = np.zeros(len(j["reads"]) + 1, dtype=np.uint64)
    step_off[1:] = np.cumsum([len(r["walk"]) for r in j["reads"]])
    node_id = np.array([w + rs for r in j["reads"] for w in r["walk"]], dtype=np.uint32)
    pstart = np.array([r["pstart"] for r in j["reads"]], dtype=np.int64)
    pend = np.array([r["pend"] for r in j["reads"]], dtype=np.int64)
    return j, names, node_len, path_off, path_nodes, rs, step_off, node_id, pstart, pend


def load_micro_bin():
    with open(os.path.join(ROOT, "tests", "golden", "micro_bin.json")) as f:
        j = json.load(f)
    step_off = np.zeros(len(j["reads"]) + 1, dtype=np.uint64)
    step_off[1:] = np.cumsum([len(r["walk"]) for r in j["reads"]])
    node_id = np.array([w for r in j["reads"] for w in r["walk"]], dtype=np.uint32)
    qlen = np.array([r["qlen"] for r in j["reads"]], dtype=np.int64)
    mapq = np.array([r["mapq"] for r in j["reads"]], dtype=np.int64)
    species = np.array([r["species"] for r in j["reads"]], dtype=np.int32)
    rs = np.array([r[1] for r in j["ranges"]], dtype=np.int64)
    re = np.array([r[2] for r in j["ranges"]], dtype=np.int64)
    return j, step_off, node_id, qlen, mapq, species, rs, re


def select_reads(reads, sel):
    """Sub-select reads `sel` (indices) from a PackedReads -> (step_off, node_id, pstart, pend)."""
    so = reads.step_off.astype(np.int64)
    ns = (so[1:] - so[:-1])[sel]
    new_off = np.zeros(len(sel) + 1, dtype=np.uint64)
    new_off[1:] = np.cumsum(ns)
    starts = so[:-1][sel]
    tot = int(ns.sum())
    idx = np.repeat(starts, ns) + (np.arange(tot) - np.repeat(new_off[:-1].astype(np.int64), ns))
    return new_off, reads.node_id[idx], reads.pstart[sel], reads.pend[sel]
