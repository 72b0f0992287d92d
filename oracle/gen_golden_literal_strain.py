"""Fixtures from the literal Python restatement of the species / strain level (oracle/ref_literal_strain.py + ref_literal.py),
run in the build container: tests/golden/literal_strain_<k>.json = a small multi-species DB (graphs, ranges, genome
lengths), a few thousand reads (walk, read_start, read_end, read_len, mapq) and what the literal reading of
rcls.rs:237-258 / profile.rs:208-349, 658-1285, 1297-1511 (LP by SciPy-HiGHS), 2884-3070, 3167-3248 makes of them: species
of every read, the species table, every HapMetrics field of every haplotype, the final strain rows.  The C oracle (CPU
test) and the HIP path (GPU test) are compared with these.

A case is kept only if every LP optimum in it is unique (checked here by solving each LP a second time with a tilted
objective: an optimal FACE would make first_sol / second_sol a free choice of the solver, and the comparison meaningless)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import ref_literal_strain as ls          # noqa: E402
import synthdata as synth             # noqa: E402


def lp_is_unique(coeff, rows, abund, ub, fixed, x_ref, obj_ref):
    """the optimal FACE {objective <= optimum} is a point: every x_j is minimised and maximised over it (2p small LPs)"""
    from scipy import sparse
    from scipy.optimize import linprog
    n, p = len(rows), coeff.shape[1]
    A = sparse.csr_matrix(coeff[rows].astype(np.float64))
    I = sparse.identity(n, format="csr")
    a = np.array([abund[v] for v in rows])
    face = sparse.csr_matrix(np.concatenate([np.zeros(p), np.ones(n) / n])[None, :])
    Aub = sparse.vstack([sparse.hstack([A, -I]), sparse.hstack([-A, -I]), face]).tocsr()
    bub = np.concatenate([a, -a, [obj_ref * (1 + 1e-9) + 1e-12]])
    bounds = [((0.0, 0.0) if fixed[j] else (0.0, ub)) for j in range(p)] + [(0, None)] * n
    scale = max(1.0, float(np.abs(x_ref).max()))
    for j in range(p):
        if fixed[j]:
            continue
        for sign in (1.0, -1.0):
            c = np.zeros(p + n)
            c[j] = sign
            r = linprog(c, A_ub=Aub, b_ub=bub, bounds=bounds, method="highs")
            if r.status != 0 or abs(r.x[j] - x_ref[j]) > 1e-6 * scale:
                return False
    return True


def make_case(seed, S, H, genome_len, n_reads, args, single_every=0, present_frac=0.5, mapq_null_every=0, dup_ids=0, null_start_every=0):
    sset = synth.make_set(seed, S, H, n_reads, genome_len, adversarial_frac=0.01, single_strain_every=single_every, present_frac=present_frac)
    rd = sset.reads
    so = rd.step_off.astype(np.int64)
    species_info = [(g.name, int(g.range_start), int(g.range_end)) for g in sset.species]        # species_range.txt rows, file order
    species_len = {g.name: float(g.genome_len.mean()) for g in sset.species}                      # species_genomes_stats.txt
    reads = []
    for r in range(rd.n_reads):
        ids = [int(x) for x in rd.node_id[so[r]:so[r + 1]]]
        st = [int(x) for x in rd.strand[so[r]:so[r + 1]]]
        mq = int(rd.mapq[r])
        if mapq_null_every and r % mapq_null_every == 3:
            mq = None                                                                             # `*` in column 12
        reads.append(dict(walk=ids, strand=st, read_start=int(rd.pstart[r]), read_end=int(rd.pend[r]), read_len=int(rd.qlen[r]), mapq=mq))
    for r, x in enumerate(reads):
        x["path"] = "".join(("<" if s else ">") + str(v) for v, s in zip(x["walk"], x["strand"]))
        x["read_id"] = "S0R%d/1" % r
        x["read_path_len"] = int(rd.plen[r])
        if null_start_every and r % null_start_every == 5:
            x["read_start"] = None                                                                # `*` in column 8: counted at species level, dropped at strain level
    if dup_ids:   # ids that repeat: pairs far apart in the file, within one species (kept, renamed) and across two (both dropped)
        rng = np.random.default_rng(seed + 7)
        pick = rng.choice(len(reads), size=(dup_ids, 2), replace=False)
        for a_, b_ in pick:
            reads[int(b_)]["read_id"] = reads[int(a_)]["read_id"]
    # ---- rcls + species level
    for x in reads:
        x["species"] = ls.process_single_read_simple(x["path"], species_info)
    rcls_df = [x for x in reads if x["species"] != "U"]                                           # profile.rs:3353-3357
    species_profile = ls.species_profiling(rcls_df, species_len, args["filtered"])
    # ---- strain level: load_species_range's -a cut, then optimize_otu per species (profile.rs:3297-3319)
    all_metrics, per_species, unique_ok = [], {}, True
    clustered = ls.group_reads_by_species(rcls_df)                                                # profile.rs:3293
    for row in species_profile:
        if row["predicted_abundance"] is None or not row["predicted_abundance"] > args["min_species_abundance"]:
            continue
        g = [g for g in sset.species if g.name == row["species_taxid"]][0]
        nodes_len = [int(v) for v in g.node_len]
        paths = {hn: [int(v) for v in g.path_nodes[int(g.path_off[h]):int(g.path_off[h + 1])]] for h, hn in enumerate(g.hap_names)}
        sp_reads = clustered.get(g.name)                                                          # read_clustered_by_species.get(otu) (:3301-3303)
        if sp_reads is None:
            continue
        met, obj1, obj2, extra = ls.optimize_otu(g.name, nodes_len, paths, int(g.range_start) - 1, int(g.range_end) - 1, sp_reads, args)   # start - 1, end - 1 (profile.rs:2886-2887)
        ls.abundace_constraint(species_profile, met)
        per_species[g.name] = dict(metrics=met, obj1=obj1, obj2=obj2, **extra)
        all_metrics += met
    ori, final = ls.abundance_est(all_metrics, args["sd"], args["min_cov"])
    case = dict(
        comment="generated by oracle/gen_golden_literal_strain.py from oracle/ref_literal_strain.py + ref_literal.py (literal Python reading of rcls.rs / profile.rs; LP by SciPy-HiGHS)",
        args=args,
        species=[dict(name=g.name, range_start=int(g.range_start), range_end=int(g.range_end), genome_len=species_len[g.name],
                      node_len=[int(v) for v in g.node_len], hap_names=list(g.hap_names),
                      paths={hn: [int(v) for v in g.path_nodes[int(g.path_off[h]):int(g.path_off[h + 1])]] for h, hn in enumerate(g.hap_names)})
                 for g in sset.species],
        reads=[dict(read_id=x["read_id"], walk=x["walk"], strand=x["strand"], read_path_len=x["read_path_len"], read_start=x["read_start"], read_end=x["read_end"],
                    read_len=x["read_len"], mapq=x["mapq"]) for x in reads],
        expect=dict(read_species=[x["species"] for x in reads], species_profile=species_profile, per_species=per_species,
                    final_rows=[dict(species_taxid=m["otu"], hap_id=m["hap_id"], predicted_coverage=m["second_sol"], predicted_abundance=m["predicted_abundance"]) for m in final]))
    return case


def check_unique(case):
    """every LP of the case has a unique optimum (re-solved with tilted objectives through the same literal code path)"""
    import ref_literal as lit
    ok = True
    for sp in case["species"]:
        ps = case["expect"]["per_species"].get(sp["name"])
        if ps is None or ps["obj1"] is None:
            continue
        met = ps["metrics"]
        haps = sorted(sp["paths"])
        cand = [i for i, m in enumerate(met) if m["first_sol"] is not None]
        V = len(sp["node_len"])
        coeff = np.zeros((V, len(cand)), dtype=np.float32)
        for k, i in enumerate(cand):
            for v in sp["paths"][haps[i]]:
                coeff[v, k] = 1.0
        # node abundances again (cheap): the literal coverage of this species' reads
        df = [dict(x, path="".join(("<" if s else ">") + str(v) for v, s in zip(x["walk"], x["strand"])), species=spn)
              for x, spn in zip(case["reads"], case["expect"]["read_species"]) if spn != "U"]
        reads = ls.group_reads_by_species(df).get(sp["name"], [])
        uniq, ulen, rows = lit.trio_nodes_info(sp["node_len"], sp["paths"])
        ab = lit.get_node_abundances(sp["node_len"], uniq, ulen, sp["range_start"] - 1, reads)[0]
        valid = [v for v, a in enumerate(ab) if a > 0.0]
        ub = 1.05 * max(ab)
        x1 = [met[i]["first_sol"] for i in cand]
        ok = ok and lp_is_unique(coeff, valid, ab, ub, [False] * len(cand), x1, ps["obj1"])
        if ps["obj2"] is not None:
            fixed = [met[i]["second_sol"] is None for i in cand]
            x2 = [met[i]["second_sol"] if met[i]["second_sol"] is not None else 0.0 for i in cand]
            # second_sol was already passed through abundace_constraint (min / scaling): compare on the raw LP only when untouched
            ok = ok and lp_is_unique(coeff, valid, ab, ub, fixed, ls._solve_lad(coeff, valid, ab, ub, fixed)[0], ps["obj2"])
    return ok


def main():
    out = os.path.join(HERE, "..", "tests", "golden")
    base = dict(fr=0.3, fc=0.46, sr=0.85, sd=0.2, min_cov=0, min_depth=0, shift=False, filtered=True, min_species_abundance=1e-4)
    plans = [  # (seed list to try, S, H, genome_len, reads, args, single_every, present_frac, mapq_null_every)
        (range(101, 140), 3, 4, 6000, 2500, dict(base), 0, 0.5, 0),
        (range(201, 240), 4, 5, 5000, 4000, dict(base, shift=True, fr=0.4), 3, 0.6, 0),
        (range(301, 340), 3, 3, 5000, 2500, dict(base, filtered=False, min_depth=2, fc=0.3), 2, 0.7, 17),
        (range(401, 460), 4, 4, 5000, 4000, dict(base), 0, 0.6, 0, 150, 40),      # duplicate read ids + null read_start rows (a5)
        # --solver highs (round 6): highs_opt's slice of the second solution (profile.rs:2865-2879) -- most strains present and a tight --fc, so that
        # candidates fall in the second filter and a survivor sits BEHIND one that fell: under Gurobi's reading it would keep its second_sol
        (range(501, 700), 2, 5, 6000, 5000, dict(base, solver="highs", fc=0.08, sr=0.99), 0, 0.8, 0),
    ]
    only = [int(x) for x in sys.argv[1:]]           # plan indices to (re)write; none = all
    for k, plan in enumerate(plans):
        if only and k not in only:
            continue
        seeds, S, H, gl, nr, args, single_every, pf, mqn = plan[:9]
        dup_ids, null_start = (plan[9], plan[10]) if len(plan) > 9 else (0, 0)
        for seed in seeds:
            case = make_case(seed, S, H, gl, nr, args, single_every, pf, mqn, dup_ids, null_start)
            n_lp = sum(1 for v in case["expect"]["per_species"].values() if v["obj1"] is not None)
            if args.get("solver") == "highs":       # the slice must matter: the same case read Gurobi's way hands out more second solutions
                other = make_case(seed, S, H, gl, nr, dict(args, solver="gurobi"), single_every, pf, mqn, dup_ids, null_start)
                n_h = sum(m["second_sol"] is not None for v in case["expect"]["per_species"].values() for m in v["metrics"])
                n_g = sum(m["second_sol"] is not None for v in other["expect"]["per_species"].values() for m in v["metrics"])
                if not (0 < n_h < n_g):
                    continue
            if n_lp and len(case["expect"]["final_rows"]) >= 2 and check_unique(case):
                break
        else:
            raise SystemExit("no seed with unique LP optima for plan %d" % k)
        case["seed"] = seed
        fn = os.path.join(out, "literal_strain_%d.json" % k)
        with open(fn, "w") as f:
            json.dump(case, f, separators=(",", ":"))
        print("wrote", fn, "seed", seed, "species", S, "reads", len(case["reads"]), "LPs", n_lp, "final rows", len(case["expect"]["final_rows"]),
              "U reads", sum(1 for s in case["expect"]["read_species"] if s == "U"))


if __name__ == "__main__":
    main()
