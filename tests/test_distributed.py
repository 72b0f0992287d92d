"""CPU test of the N>1 path: two gloo processes run finalize_stage (the one all-reduce + row gather of
pantax_amd.pipeline) on per-rank shards; the result must equal a single process holding both shards."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests.conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_finalize_matches_single_process(tmp_path):
    from pantax_amd.pipeline import LocalComm, StepConfig, finalize_stage
    from tests.dist_worker import fake_local, names
    out = tmp_path / "dist.json"
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(out)], env=env))
    for p in procs:
        assert p.wait(timeout=180) == 0
    got = json.load(open(out))
    # single process with both shards concatenated
    n_species = [4, 6]
    loc = [fake_local(r, n) for r, n in enumerate(n_species)]
    sn, hn = [], []
    rows = []
    s_off = h_off = 0
    for r, n in enumerate(n_species):
        a, b = names(r, n)
        sn += a
        hn += b
        rows += [(s + s_off, h + h_off) + tuple(rest) for (s, h, *rest) in loc[r]["rows"]]
        s_off += n
        h_off += 3 * n
    merged = dict(keep=np.concatenate([l["keep"] for l in loc]), absolute=np.concatenate([l["absolute"] for l in loc]),
                  s_all=np.concatenate([l["s_all"] for l in loc]), s_pass=np.concatenate([l["s_pass"] for l in loc]), rows=rows)
    exp_species, exp_strain, _ = finalize_stage(merged, sn, hn, StepConfig(), LocalComm())
    assert [r[0] for r in got["species"]] == [r[0] for r in exp_species]
    assert [(r[0], r[1]) for r in got["strain"]] == [(r[0], r[1]) for r in exp_strain]
    for g, e in zip(got["species"], exp_species):
        assert g[1] == pytest.approx(e[1], rel=1e-12) and g[2] == pytest.approx(e[2], rel=1e-12)
    for g, e in zip(got["strain"], exp_strain):
        assert g[2] == pytest.approx(e[2], rel=1e-12) and g[3] == pytest.approx(e[3], rel=1e-12)
    assert sum(r[3] for r in got["strain"]) == pytest.approx(1.0, rel=1e-12)
    # the species under the -a cut contributes no strain rows on either side
    assert not any(r[0] == "sp1_0" for r in got["strain"])


@pytest.mark.parametrize("world", [2, 3])
def test_read_routing_exchange_over_gloo(tmp_path, world):
    """every rank's messages reach their owners in source-rank order (sizes by one all-to-all, payload by one
    all-to-all(v)); pairs that exchange nothing are fine."""
    from tests.dist_route_worker import slice_messages
    out = tmp_path / "route.json"
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_route_worker.py"), str(out)], env=env))
    for p in procs:
        assert p.wait(timeout=180) == 0
    sent = [slice_messages(r, world) for r in range(world)]
    for d in range(world):
        got = json.load(open("%s.%d" % (out, d)))
        exp = np.concatenate([sent[r][0][d] for r in range(world)])
        assert got["recv"] == exp.tolist()
        assert got["nr"] == [int(sent[r][1][d]) for r in range(world)] and got["nt"] == [int(sent[r][2][d]) for r in range(world)]
        assert got["stats"]["recv_words"] == len(exp)
        assert got["tot"][0] == got["tot"][1] and got["tot"][2] == world       # every read sent is received by somebody


def test_partition_species_is_balanced_and_deterministic():
    from pantax_amd.pipeline import partition_species
    rng = np.random.default_rng(4)
    w = rng.lognormal(10, 1.5, size=200).tolist()
    for world in (1, 2, 4, 8):
        owner = partition_species(w, world)
        assert owner == partition_species(w, world) and set(owner) == set(range(world))
        load = [sum(x for x, o in zip(w, owner) if o == r) for r in range(world)]
        assert max(load) <= sum(w) / world + max(w)          # the LPT guarantee
    assert partition_species([5.0, 5.0, 1.0], 2) == [0, 1, 0]


def test_several_dbs_on_one_gpu_finalize_like_ranks():
    """A database of more than 2^32 path positions is cut by species into several dbs that share the GPU (pipeline.finalize_many /
    profile_steps_many; BASELINE configs[4] at full size): their slabs meet like those of ranks -- the tables equal those of ONE db holding
    every species -- and the species are cut into contiguous groups under the position limit."""
    from pantax_amd.pipeline import LocalComm, StepConfig, finalize_many, finalize_stage, split_species_by_path_steps
    from tests.dist_worker import fake_local, names
    ps = [5, 5, 5, 9, 1, 10]
    gr = split_species_by_path_steps(ps, limit=10)
    assert gr[0][0] == 0 and gr[-1][1] == len(ps) and all(a[1] == b[0] for a, b in zip(gr, gr[1:])) and all(0 < sum(ps[a:b]) <= 10 for a, b in gr)
    assert split_species_by_path_steps([3] * 10, limit=16) == [(0, 5), (5, 10)]          # as few groups as the limit allows, about equally heavy
    assert split_species_by_path_steps([11_300_000] * 1000) == [(0, 250), (250, 500), (500, 750), (750, 1000)]   # BASELINE configs[4]: four dbs
    assert split_species_by_path_steps([1, 2, 3]) == [(0, 3)]
    with pytest.raises(ValueError):
        split_species_by_path_steps([11], limit=10)
    n_species = [4, 6, 3]
    loc = [dict(fake_local(r, n), stats=dict(obj=[(0.0, 0.0)] * n)) for r, n in enumerate(n_species)]
    sn_l, hn_l = zip(*[names(r, n) for r, n in enumerate(n_species)])
    got_species, got_strain, stats = finalize_many(loc, sn_l, hn_l, StepConfig())
    sn, hn, rows = [], [], []
    s_off = h_off = 0
    for r, n in enumerate(n_species):
        sn += sn_l[r]
        hn += hn_l[r]
        rows += [(s + s_off, h + h_off) + tuple(rest) for (s, h, *rest) in loc[r]["rows"]]
        s_off += n
        h_off += 3 * n
    merged = dict(keep=np.concatenate([l["keep"] for l in loc]), absolute=np.concatenate([l["absolute"] for l in loc]),
                  s_all=np.concatenate([l["s_all"] for l in loc]), s_pass=np.concatenate([l["s_pass"] for l in loc]), rows=rows)
    exp_species, exp_strain, _ = finalize_stage(merged, sn, hn, StepConfig(), LocalComm())
    assert [r[0] for r in got_species] == [r[0] for r in exp_species] and [(r[0], r[1]) for r in got_strain] == [(r[0], r[1]) for r in exp_strain]
    for g, e in zip(got_species, exp_species):
        assert g[1] == pytest.approx(e[1], rel=1e-12) and g[2] == pytest.approx(e[2], rel=1e-12)
    for g, e in zip(got_strain, exp_strain):
        assert g[2] == pytest.approx(e[2], rel=1e-12) and g[3] == pytest.approx(e[3], rel=1e-12)
    assert len(stats["obj"]) == sum(n_species)


def test_tables_built_column_wise_equal_the_row_by_row_reading():
    """finalize_end builds the tables from numpy columns (round 6); the row-by-row reading of the same slab -- profile.rs:341-344 (species
    normaliser, descending), :602 (-a cut), :3243-3248 (strain normaliser over the passing strains, descending) -- is restated here."""
    from pantax_amd.pipeline import LocalComm, StepConfig, finalize_stage, rows_to_array
    rng = np.random.default_rng(7)
    S, per = 40, 5
    keep = (rng.random(S) < 0.8).astype(np.uint8)
    absolute = np.where(rng.random(S) < 0.9, rng.random(S) * 30, 1e-9)
    absolute[3] = absolute[4]                                         # a tie: both sorts are stable
    s_all, s_pass = rng.random(S) * 10, rng.random(S) * 5
    rows = []
    for s in range(S):
        for h in range(per):
            if keep[s] and rng.random() < 0.6:
                rows.append((s, s * per + h, float(rng.random() * 9)) + tuple(None if rng.random() < 0.3 else float(rng.random()) for _ in range(6)))
    rows[5] = rows[5][:2] + (rows[6][2],) + rows[5][3:]               # equal coverages next to each other
    sn = ["sp%d" % s for s in range(S)]
    hn = ["h%d" % h for h in range(S * per)]
    cfg = StepConfig()
    base = dict(keep=keep, absolute=absolute, s_all=s_all, s_pass=s_pass)
    got_a = finalize_stage(dict(base, rows=rows), sn, hn, cfg, LocalComm())
    got_b = finalize_stage(dict(base, rows_np=rows_to_array(rows)), sn, hn, cfg, LocalComm())
    assert got_a == got_b
    total_abs = float(np.where(keep == 1, absolute, 0.0).sum())
    act = [bool(keep[s] == 1 and absolute[s] > 0 and absolute[s] / total_abs > cfg.min_species_abundance) for s in range(S)]
    g_pass = float(sum(s_pass[s] for s in range(S) if act[s]))
    exp_species = [(sn[s], float(absolute[s] / total_abs), float(absolute[s])) for s in range(S) if keep[s] == 1]
    exp_species.sort(key=lambda t: -t[1])
    exp_strain = [(sn[s], hn[h], cov, cov / g_pass) + tuple(opt) for (s, h, cov, *opt) in rows if act[s]]
    exp_strain.sort(key=lambda t: -t[3])
    assert got_a[0] == exp_species
    assert [r[:2] for r in got_a[1]] == [r[:2] for r in exp_strain]
    for g, e in zip(got_a[1], exp_strain):
        assert g[4:] == e[4:] and g[2] == e[2] and g[3] == pytest.approx(e[3], rel=1e-15)
    assert any(v is None for r in got_a[1] for v in r[4:]) and len(got_a[1]) > 20
