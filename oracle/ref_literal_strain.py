"""A SECOND, independently written restatement of the reference's species / strain level logic -- TEST INFRASTRUCTURE ONLY.

Companion of oracle/ref_literal.py (which covers trio_nodes_info / get_node_abundances): the read binning, the species
table, the two path filters around the LP, the LP itself (handed to SciPy's HiGHS, the reference's own open backend), the
abundance constraint and the final table filters, written in plain Python while mirroring the reference's statements and
data shapes (lists of row dicts for the polars frames, dicts for the maps, `None` for Option::None).  It shares no code
with oracle/pantax_oracle.c.  Fixtures generated from this file (oracle/gen_golden_literal_strain.py ->
tests/golden/literal_strain_*.json) are compared with the C oracle (CPU test) and with the HIP path (GPU test).

    process_single_read_simple      rcls.rs:237-258
    dataframe_to_records_and_check_unique / process_with_duplicates / group_reads_by_species   profile.rs:361-463
    equal_length_read_cls           profile.rs:208-251
    non_equal_length_read_cls       profile.rs:253-297
    species_profiling               profile.rs:299-349
    zscore_filter                   profile.rs:1028-1051
    first_filter_paths              profile.rs:1080-1227
    second_filter_paths             profile.rs:1229-1285
    gurobi_opt (model and flow)     profile.rs:1297-1511
    optimize_otu (glue)             profile.rs:2884-3026
    abundace_constraint             profile.rs:3028-3070
    abundance_est (filters)         profile.rs:3167-3248
"""
import math
import re

import numpy as np

import ref_literal as lit

_RE = re.compile(r"-?\d+")


def rust_round(x):
    """f64::round: half away from zero"""
    return math.floor(x + 0.5) if x >= 0 else -math.floor(-x + 0.5)


def round2(x):
    return rust_round(x * 100.0) / 100.0


# ---------------------------------------------------------------------------------------------- rcls.rs:237-258
def process_single_read_simple(node_path, species_info):
    nodes = [int(m) for m in _RE.findall(node_path)]
    if len(nodes) == 0:
        mn, mx = -1, -1
    elif len(nodes) == 1:
        mn, mx = nodes[0], nodes[0]
    else:
        mn, mx = min(nodes), max(nodes)
    for s, start, end in species_info:                       # .iter().find(..): first match in file order
        if mn >= start and mx <= end:
            return s
    return "U"


# ---------------------------------------------------------------------------------------------- profile.rs:208-349
def _group_stable(rows, key):
    groups, order = {}, []
    for r in rows:
        k = r[key]
        if k not in groups:
            groups[k] = []
            order.append(k)
        groups[k].append(r)
    return [(k, groups[k]) for k in order]


def equal_length_read_cls(df, read_len, isfilter):
    if not isfilter:
        return [dict(species=k, base_count=len(g) * read_len) for k, g in _group_stable(df, "species")]       # :212-216
    read_count = {k: len(g) for k, g in _group_stable(df, "species")}                                          # :220-222
    filtered = [r for r in df if r["mapq"] is not None and r["mapq"] >= 3 and r["mapq"] <= 60]               # :225-226 (null fails the comparison)
    agg = [dict(species=k, less_multi=len(g), uniq_count=sum(1 for r in g if r["mapq"] == 60)) for k, g in _group_stable(filtered, "species")]
    out = []
    for a in agg:                                                                                              # inner join :236-237
        if a["species"] not in read_count:
            continue
        rc = read_count[a["species"]]
        if a["uniq_count"] > 0 and a["less_multi"] > float(rc) / 10.0:                                         # :240-245
            out.append(dict(species=a["species"], base_count=rc * read_len))                                   # :246
    return out


def non_equal_length_read_cls(df, isfilter):
    if not isfilter:
        return [dict(species=k, base_count=sum(r["read_len"] for r in g)) for k, g in _group_stable(df, "species")]
    rc_bc = {k: (len(g), sum(r["read_len"] for r in g)) for k, g in _group_stable(df, "species")}              # :264-266
    filtered = [r for r in df if r["mapq"] is not None and r["mapq"] >= 3 and r["mapq"] <= 60]
    agg = [dict(species=k, less_multi=len(g), uniq_count=sum(1 for r in g if r["mapq"] == 60)) for k, g in _group_stable(filtered, "species")]
    out = []
    for a in agg:
        if a["species"] not in rc_bc:
            continue
        rc, bc = rc_bc[a["species"]]
        if a["uniq_count"] > 0 and a["less_multi"] > float(rc) / 10.0:
            out.append(dict(species=a["species"], base_count=bc))
    return out


def species_profiling(rcls_df, species_len, filtered):
    """rcls_df: rows without "U" (profile.rs:3353-3357); species_len: dict species -> f64.
    -> rows (species_taxid, predicted_abundance, predicted_coverage) sorted by abundance, descending (stable)"""
    head = [r["read_len"] for r in rcls_df[:1000]]                                    # .limit(1000) (:315)
    uniq = []
    for v in head:                                                                    # .unique(First) (:316)
        if v not in uniq:
            uniq.append(v)
    if len(uniq) == 1:                                                                # :320-323
        grouped = equal_length_read_cls(rcls_df, uniq[0], filtered)
    else:
        grouped = non_equal_length_read_cls(rcls_df, filtered)
    rows = []
    for g in grouped:                                                                 # left join with the lengths (:332-337)
        ln = species_len.get(g["species"])
        rows.append(dict(species=g["species"], absolute_abund=None if ln is None else g["base_count"] / ln))
    total = sum(r["absolute_abund"] for r in rows if r["absolute_abund"] is not None)   # .sum() skips nulls (:341)
    prof = [dict(species_taxid=r["species"], predicted_abundance=None if r["absolute_abund"] is None else r["absolute_abund"] / total,
                 predicted_coverage=r["absolute_abund"]) for r in rows]
    # sort descending, nulls last (polars default for descending sorts puts nulls last)
    prof.sort(key=lambda r: (r["predicted_abundance"] is None, -(r["predicted_abundance"] or 0.0)))
    return prof


# ---------------------------------------------------------------------------------------------- profile.rs:361-463
def dataframe_to_records_and_check_unique(df):
    """df: rows of the frame without "U" reads (read_id, path, read_path_len, read_start, read_end, species; None = null)"""
    seen, unique, records = set(), True, []
    for r in df:                                                      # :373
        if r["read_id"] in seen:                                      # !seen.insert(read_id) (:375-377): before the null check
            unique = False
        seen.add(r["read_id"])
        if all(r[k] is not None for k in ("path", "read_path_len", "read_start", "read_end", "species")):   # :379-385
            records.append(dict(read_id=r["read_id"], path=r["path"], read_path_len=r["read_path_len"], read_start=r["read_start"],
                                read_end=r["read_end"], species=r["species"]))
    return records, unique


def process_with_duplicates(records):
    grouped = {}                                                      # FxHashMap<String, Vec<Record>> (:407-410)
    for r in records:
        grouped.setdefault(r["read_id"], []).append(r)
    species_map = {}                                                  # BTreeMap (:412)
    for _read_id, group in grouped.items():
        species_set = set(r["species"] for r in group)                # :415
        if len(species_set) == 1:                                     # :416
            species = group[0]["species"]
            entry = species_map.setdefault(species, [])
            for i, r in enumerate(group):
                rid = r["read_id"] if i == 0 else "%s_%d" % (r["read_id"], i + 1)   # :420-423
                entry.append(dict(r, read_id=rid))
    return species_map


def group_reads_by_species(df):
    records, unique = dataframe_to_records_and_check_unique(df)
    if unique:                                                        # :441
        m = {}
        for r in records:
            m.setdefault(r["species"], []).append(r)
        return m
    return process_with_duplicates(records)                           # :461


# ---------------------------------------------------------------------------------------------- profile.rs:1028-1051
def zscore_filter(data, threshold):
    if len(data) == 0:
        return []
    mean = sum(data) / len(data)
    std = math.sqrt(sum((x - mean) ** 2 for x in data) / len(data))
    if std == 0.0:
        return []
    return [x for x in data if abs((x - mean) / std) < threshold]


def new_metrics():
    return dict(otu=None, hap_id=None, unique_trio_nodes_fraction=None, frequencies_mean=None, path_cov_ratio=None, first_sol=None,
                divergence=None, second_sol=None, is_rescue=None, total_cov_diff=None)


# ---------------------------------------------------------------------------------------------- profile.rs:1080-1227
def first_filter_paths(var, paths, hap2trio_rows, trio_node_abundances, node_abundance_vec, args):
    """var: dict(otu, hap_metrics, possible_paths_idx, second_possible_paths_idx, orign_n_haps, hap2trio_nodes_m_size,
    same_path_flag, second_opt); hap2trio_rows: one presence row (list over haps) per unique trio."""
    haps = sorted(paths.keys())
    for i, hap_id in enumerate(haps):
        var["hap_metrics"][i]["otu"] = var["otu"]
        var["hap_metrics"][i]["hap_id"] = hap_id
    orign_n_haps = len(paths)
    m_size = len(hap2trio_rows) * orign_n_haps                         # DMatrix::len() = rows x cols (:1096)
    var["orign_n_haps"], var["hap2trio_nodes_m_size"] = orign_n_haps, m_size
    if orign_n_haps != 1 and m_size != 0:
        for hap_idx, hap_id in enumerate(haps):
            trio_idxs = [i for i in range(len(hap2trio_rows)) if hap2trio_rows[i][hap_idx] > 0]          # :1114-1116
            if len(trio_idxs) == 0:
                continue
            trio_idx_set = set(trio_idxs)
            abundances = [v if i in trio_idx_set else 0.0 for i, v in enumerate(trio_node_abundances)]   # :1123-1127
            non_zero = [x for x in abundances if x > 0.0]
            fraction = len(non_zero) / len(trio_idxs)                                                    # :1135
            var["hap_metrics"][hap_idx]["unique_trio_nodes_fraction"] = round2(fraction)                 # :1136-1138
            if args["shift"]:
                filt = zscore_filter(non_zero, 3.0)
                fmean = 0.0 if len(filt) == 0 else sum(filt) / len(filt)
                if fmean >= 1.0:                                                                          # :1148-1157
                    sh = args["fr"] + (0.8 - args["fr"]) * fmean / 100.0
                    if sh > 0.8:
                        sh = 0.8
                else:
                    sh = args["fr"] * fmean
                if fraction < sh:
                    continue
                var["hap_metrics"][hap_idx]["frequencies_mean"] = fmean
            else:
                if fraction < args["fr"]:                                                                 # :1168
                    continue
                filt = zscore_filter(non_zero, 3.0)
                fmean = 0.0 if len(filt) == 0 else sum(filt) / len(filt)
                var["hap_metrics"][hap_idx]["frequencies_mean"] = fmean
            var["possible_paths_idx"].append(hap_idx)
    elif orign_n_haps != 1 and m_size == 0:
        vals = [paths[h] for h in haps]
        all_same = all(v == vals[0] for v in vals[1:])
        if all_same:
            var["same_path_flag"] = True
            non_zero = [x for x in node_abundance_vec if x > 0.0]
            fmean = 0.0 if len(non_zero) == 0 else sum(non_zero) / len(non_zero)
            var["hap_metrics"][0]["frequencies_mean"] = round2(fmean)
            var["possible_paths_idx"].append(0)
        else:
            var["possible_paths_idx"] = list(range(orign_n_haps))
    elif orign_n_haps == 1:
        non_zero = [x for x in node_abundance_vec if x > 0.0]
        fmean = 0.0 if len(non_zero) == 0 else sum(non_zero) / len(non_zero)
        var["hap_metrics"][0]["frequencies_mean"] = round2(fmean)
        var["possible_paths_idx"].append(0)


# ---------------------------------------------------------------------------------------------- profile.rs:1229-1285
def second_filter_paths(var, args):
    keep = []
    if var["orign_n_haps"] != 1 and var["hap2trio_nodes_m_size"] > 0:
        var["second_opt"] = True
        for idx in var["possible_paths_idx"]:
            m = var["hap_metrics"][idx]
            fmean = m["frequencies_mean"] if m["frequencies_mean"] is not None else 0.0
            if fmean == 0.0:
                continue
            sol = m["first_sol"]
            f = abs(sol - fmean) / (sol + fmean)
            f_rounded = round2(f)
            m["divergence"] = f_rounded
            if f_rounded > args["fc"]:
                if f_rounded <= 0.6:
                    this_ratio = m["unique_trio_nodes_fraction"] * m["path_cov_ratio"]
                    if this_ratio < args["sr"] or sol == 0.0:
                        continue
                    m["is_rescue"] = True
                    keep.append(idx)
                else:
                    continue
            elif f_rounded <= args["fc"] and sol != 0.0:
                keep.append(idx)
        var["second_possible_paths_idx"] = keep
    elif (var["orign_n_haps"] != 1 and var["hap2trio_nodes_m_size"] == 0 and var["same_path_flag"]) or var["orign_n_haps"] == 1:
        m = var["hap_metrics"][0]
        fmean = m["frequencies_mean"]
        if fmean > 0.0:
            sol = m["first_sol"]
            m["divergence"] = round2(abs(sol - fmean) / (sol + fmean))
            m["second_sol"] = sol
    elif var["orign_n_haps"] != 1 and var["hap2trio_nodes_m_size"] == 0 and not var["same_path_flag"]:
        for idx in var["possible_paths_idx"]:
            var["hap_metrics"][idx]["second_sol"] = var["hap_metrics"][idx]["first_sol"]


# ---------------------------------------------------------------------------------------------- profile.rs:1297-1511
def _solve_lad(coeff, rows, abund, ub, fixed_zero):
    """min (1/n) sum_v y_v, y_v >= +-(sum_j coeff[v][j] x_j - a_v), 0 <= x_j <= ub (x_j == 0 where fixed): SciPy's HiGHS.
    The binary strain indicators of the reference model (:1363-1377) do not bind (sum z <= npaths; z_j >= something <= 0.525),
    so the continuous relaxation is the model's optimum (tests/golden/lp_milp_cases.npz pins exactly that)."""
    from scipy import sparse
    from scipy.optimize import linprog
    n, p = len(rows), coeff.shape[1]
    A = sparse.csr_matrix(coeff[rows].astype(np.float64))
    I = sparse.identity(n, format="csr")
    Aub = sparse.vstack([sparse.hstack([A, -I]), sparse.hstack([-A, -I])]).tocsr()
    a = np.array([abund[v] for v in rows], dtype=np.float64)
    bub = np.concatenate([a, -a])
    c = np.concatenate([np.zeros(p), np.ones(n) / n])
    bounds = [((0.0, 0.0) if fixed_zero[j] else (0.0, ub)) for j in range(p)] + [(0, None)] * n
    r = linprog(c, A_ub=Aub, b_ub=bub, bounds=bounds, method="highs")
    assert r.status == 0, r.message
    return [float(x) for x in r.x[:p]], float(r.fun)


def gurobi_opt(var, nvert, paths, node_abundance_vec, node_base_cov, node_len, args):
    haps = sorted(paths.keys())
    npaths = len(var["possible_paths_idx"])
    max_val = max(node_abundance_vec) if len(node_abundance_vec) else float("-inf")          # :1316-1319
    coeff = np.zeros((nvert, npaths), dtype=np.float32)                                       # :1333
    for i, hap in enumerate(haps):
        if i in var["possible_paths_idx"]:
            pos = var["possible_paths_idx"].index(i)
            for v in paths[hap]:
                coeff[v, pos] = 1.0
    node_cov = np.array([float(x) for x in node_base_cov], dtype=np.float32)                 # :1344-1347 (usize -> f32)
    path_cov = node_cov @ coeff                                                               # f32 product (:1349)
    node_len32 = np.array([float(x) for x in node_len], dtype=np.float32)
    path_len = node_len32 @ coeff
    ratio = path_cov / path_len                                                               # component_div (:1356)
    for i in range(npaths):
        var["hap_metrics"][var["possible_paths_idx"][i]]["path_cov_ratio"] = float(ratio[i])  # :1359-1361
    valid_nodes = [v for v, ab in enumerate(node_abundance_vec) if ab > 0.0]                  # :1380-1385 (sampling off: --sample 0)
    sols, obj1 = _solve_lad(coeff, valid_nodes, node_abundance_vec, 1.05 * max_val, [False] * npaths)
    for i, sol in enumerate(sols):
        var["hap_metrics"][var["possible_paths_idx"][i]]["first_sol"] = sol                   # :1467-1469
    second_filter_paths(var, args)                                                            # :1474
    if not var["second_opt"]:
        return obj1, None
    fixed = [idx not in var["second_possible_paths_idx"] for idx in var["possible_paths_idx"]]   # :1484-1488
    sols2, obj2 = _solve_lad(coeff, valid_nodes, node_abundance_vec, 1.05 * max_val, fixed)
    if args.get("solver", "gurobi") == "highs":
        # highs_opt (profile.rs:2849-2879): `let sols2 = &all_sols[..all_sols.len().min(opt_var.second_possible_paths_idx.len())];` -- the solution is cut
        # to as many columns as candidates SURVIVED (:2865), and THAT slice is zipped with all the candidates (:2871): zip stops at the shorter one
        sols2 = sols2[:min(len(sols2), len(var["second_possible_paths_idx"]))]
    for path_idx, sol in zip(var["possible_paths_idx"], sols2):                               # :1500-1508 / :2871-2879
        if path_idx in var["second_possible_paths_idx"]:
            var["hap_metrics"][path_idx]["second_sol"] = sol
    return obj1, obj2


# ---------------------------------------------------------------------------------------------- profile.rs:2884-3026
def optimize_otu(otu, nodes_len, paths, start, end, reads, args):
    """-> (hap_metrics in BTreeMap (sorted hap) order, obj1, obj2, intermediate integers for the fixtures)"""
    unique_trio_nodes, unique_lengths, rows = lit.trio_nodes_info(nodes_len, paths)                       # :2936
    node_abundance_vec, trio_abund, node_base_cov, bases, tbases, n_abort = lit.get_node_abundances(nodes_len, unique_trio_nodes, unique_lengths, start, reads)
    extra = dict(n_unique_trios=len(unique_trio_nodes), n_abort=n_abort, bases_total=sum(bases), trio_bases_total=sum(tbases))
    nvert = end - start + 1
    opt_vec = [x if x > args["min_depth"] else 0.0 for x in node_abundance_vec]                           # :2941-2944
    var = dict(otu=otu, hap_metrics=[new_metrics() for _ in paths], possible_paths_idx=[], second_possible_paths_idx=[], orign_n_haps=0,
               hap2trio_nodes_m_size=0, same_path_flag=False, second_opt=False)
    first_filter_paths(var, paths, rows, trio_abund, opt_vec, args)                                       # :2967
    obj1 = obj2 = None
    if len(var["possible_paths_idx"]) != 0:
        obj1, obj2 = gurobi_opt(var, nvert, paths, node_abundance_vec, node_base_cov, nodes_len, args)
    return var["hap_metrics"], obj1, obj2, dict(extra, n_candidates=len(var["possible_paths_idx"]))


# ---------------------------------------------------------------------------------------------- profile.rs:3028-3070
def abundace_constraint(species_profile, metrics):
    strain_abs = []
    for m in metrics:
        if m["is_rescue"] is True and m["first_sol"] is not None and m["second_sol"] is not None:
            m["second_sol"] = min(m["first_sol"], m["second_sol"])
        strain_abs.append(m["second_sol"] if m["second_sol"] is not None else 0.0)
    species_abs = [r["predicted_coverage"] for r in species_profile if r["species_taxid"] == metrics[0]["otu"]][0]
    ssum = sum(strain_abs)
    diff = abs(ssum - species_abs) / ((ssum + species_abs) / 2.0)
    for m in metrics:
        m["total_cov_diff"] = diff
    if len(strain_abs) and max(strain_abs) > 1.05 * species_abs:
        factor = species_abs / ssum
        for m in metrics:
            if not (m["is_rescue"] or False) and m["second_sol"] is not None:
                m["second_sol"] = m["second_sol"] * factor


# ---------------------------------------------------------------------------------------------- profile.rs:3167-3248
def abundance_est(hap_metrics_vec, single_cov_diff, min_cov):
    """hap_metrics_vec: every species' metrics, concatenated.  -> (ori rows, final rows) with predicted_abundance; the join
    with genomes_info.txt only adds names and is left to the table writer."""
    merged = [dict(m) for m in hap_metrics_vec]
    tot = sum(m["second_sol"] for m in merged if m["second_sol"] is not None)                              # :3198 (sum skips nulls)
    for m in merged:
        m["predicted_abundance"] = None if m["second_sol"] is None else m["second_sol"] / tot
    group_size = {}
    for m in merged:                                                                                        # count of hap_id per species (:3219-3223)
        group_size[m["otu"]] = group_size.get(m["otu"], 0) + (1 if m["hap_id"] is not None else 0)
    kept = []
    for m in merged:
        c1 = group_size[m["otu"]] > 1 or (m["total_cov_diff"] is not None and m["total_cov_diff"] <= single_cov_diff)   # :3232-3235
        c2 = m["second_sol"] is not None and m["second_sol"] >= min_cov and m["second_sol"] != 0.0                      # :3237-3241
        if c1 and c2:
            kept.append(dict(m))
    tot2 = sum(m["second_sol"] for m in kept)
    for m in kept:
        m["predicted_abundance"] = m["second_sol"] / tot2                                                   # :3243
    kept.sort(key=lambda m: -m["predicted_abundance"])                                                      # :3247-3248 (stable)
    return merged, kept
