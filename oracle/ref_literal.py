"""A SECOND, independently written restatement of the reference's integer hot path -- TEST INFRASTRUCTURE ONLY.

oracle/pantax_oracle.c restates trio_nodes_info / get_node_abundances in flat-array C (sorted key tables, bit sets).
This file restates the same two reference functions in plain Python while MIRRORING THE REFERENCE'S DATA STRUCTURES
statement by statement -- a dict where it has a DashMap / FxHashMap, a set for FxHashSet, one 0/1 list per node for the
per-base Vec<u8>, a per-read dict for read_nodes_len -- so that the two readings share as little as possible.  Fixtures
generated from THIS file (oracle/gen_golden_literal.py -> tests/golden/literal_cov_*.json) are compared with the C oracle
(CPU test) and with the HIP path (GPU test): two independent readings of profile.rs agreeing is the strongest pin
available while the reference ships no vectors and cannot be compiled here (Rust; no cargo).  Pure-Python loops: small
cases only.

    trio_nodes_info        profile.rs:658-740
    get_node_abundances    profile.rs:743-1026
"""
import re


class Abort(Exception):
    """the reference would panic here (assert profile.rs:854, index out of bounds :849/:850)"""


def trio_nodes_info(nodes_len, paths):
    """paths: dict hap -> list of node ids (BTreeMap order = sorted(hap)).  -> (unique_trio_nodes: dict trio -> index,
    unique_lengths: list, rows: list of presence rows (one 0/1 list over haps per unique trio)).  profile.rs:658-740"""
    trio_nodes_set = set()                                   # FxHashSet (:659)
    hap_trio_paths = {}                                      # FxHashMap (:660)
    haps = sorted(paths.keys())                              # graph.paths.keys() of a BTreeMap (:661)
    for hap in haps:                                         # :666
        path = paths[hap]
        trio_path = []
        for i in range(len(path) - 2):                       # path.windows(3) (:670)
            w = path[i:i + 3]
            trio_path.append((w[2], w[1], w[0]) if w[0] > w[2] else (w[0], w[1], w[2]))   # :672-678
        trio_nodes_set.update(trio_path)                     # :680
        hap_trio_paths[hap] = trio_path                      # :681
    trio_nodes = list(trio_nodes_set)                        # :684 (iteration order of the set: arbitrary)
    trio_index_map = {t: i for i, t in enumerate(trio_nodes)}   # :685
    presence = [[0] * len(haps) for _ in trio_nodes]         # DMatrix::zeros (:686)
    count_per_trio = [0] * len(trio_nodes)                   # :688
    for hap_idx, hap in enumerate(haps):                     # :689
        for t in hap_trio_paths.get(hap, []):                # :690-691
            idx = trio_index_map.get(t)
            if idx is not None:
                presence[idx][hap_idx] = 1                   # :693
                count_per_trio[idx] += 1                     # :694
    unique_trio_nodes, unique_lengths, rows = {}, [], []     # :705-707
    for i, count in enumerate(count_per_trio):               # :708
        if count == 1:                                       # :709
            trio = trio_nodes[i]
            unique_trio_nodes[trio] = len(unique_trio_nodes)     # :711
            unique_lengths.append(nodes_len[trio[0]] + nodes_len[trio[1]] + nodes_len[trio[2]])   # :712-713
            rows.append(list(presence[i]))                   # :714
    return unique_trio_nodes, unique_lengths, rows


_RE = re.compile(r"-?\d+")                                   # :769


def get_node_abundances(nodes_len, trio_nodes, trio_nodes_len, start, reads):
    """reads: list of dict(path=str, read_start=int, read_end=int) (Record, profile.rs:351-359); start = range_start - 1
    (optimize_otu, :2886).  -> (node_abundance_vec, trio_node_abundance_vec, node_base_cov, bases_per_node, trio_bases,
    n_abort).  A read on which the reference would panic is skipped WHOLE and counted (the oracle's convention, stated in
    pantax_oracle.h); state changes it made before the panic are rolled back so that "whole" is exact."""
    bases_per_node = {i: 0 for i in range(len(nodes_len))}                       # DashMap (:774, :779)
    trio_nodes_bases_count = {i: 0 for i in range(len(trio_nodes))}              # :775, :783-785
    node_base_cov_info = {i: [0, 0, [0] * ln] for i, ln in enumerate(nodes_len)}  # (u8, usize, Vec<u8>) (:776, :780)
    n_abort = 0
    for read in reads:                                                           # par_iter (:787)
        read_nodes = [int(m) - 1 - start for m in _RE.findall(read["path"])]     # :788-792
        if not read_nodes:                                                       # :794
            continue
        try:
            _one_read(read, read_nodes, nodes_len, trio_nodes, bases_per_node, trio_nodes_bases_count, node_base_cov_info)
        except Abort:
            n_abort += 1
    node_abundance_vec = [bases_per_node[i] / nodes_len[i] for i in range(len(nodes_len))]        # :980-990
    trio_node_abundance_vec = [trio_nodes_bases_count[i] / ln for i, ln in enumerate(trio_nodes_len)]   # :1005-1015
    node_base_cov = [node_base_cov_info[i][1] for i in range(len(nodes_len))]                    # :1017-1022
    return (node_abundance_vec, trio_node_abundance_vec, node_base_cov,
            [bases_per_node[i] for i in range(len(nodes_len))], [trio_nodes_bases_count[i] for i in range(len(trio_nodes))], n_abort)


def _one_read(read, read_nodes, nodes_len, trio_nodes, bases_per_node, trio_nodes_bases_count, node_base_cov_info):
    # the reference mutates shared maps as it goes and panics mid-read; to skip such a read whole, first walk it dry
    for dry in (True, False):
        start_node, end_node = read_nodes[0], read_nodes[-1]                     # :798-799
        target_len = read["read_end"] - read["read_start"]                       # :800
        seen = 0                                                                 # :801
        read_nodes_len = {}                                                      # FxHashMap (:802)
        undup_read_nodes = set()                                                 # FxHashSet (:803)
        for node in read_nodes:                                                  # :806-808
            read_nodes_len[node] = 0
        if start_node == end_node and len(read_nodes) == 1:                      # :811
            if target_len < 0:                                                   # :821
                return                                                           # :826
            read_nodes_len[start_node] = read_nodes_len.get(start_node, 0) + target_len   # :828
            if not dry and start_node in bases_per_node:                         # entry().and_modify(): nothing if the key is absent (:829)
                bases_per_node[start_node] += target_len
            if not dry and start_node in node_base_cov_info:                     # if let Some(..) = get_mut (:831)
                entry = node_base_cov_info[start_node]
                if read["read_start"] < read["read_end"] <= len(entry[2]):       # :832
                    for j in range(read["read_start"], read["read_end"]):        # :833-835
                        entry[2][j] = 1
                entry[1] = sum(entry[2])                                         # :844
                entry[0] = 1 if entry[1] == nodes_len[start_node] else 0         # :845
        else:
            for i, node in enumerate(read_nodes):                                # :848
                if not (0 <= node < len(nodes_len)):                             # nodes_len[node] panics (:849)
                    raise Abort()
                node_len = nodes_len[node]
                if i == 0:                                                       # :853
                    if not read["read_start"] <= node_len:                       # assert (:854)
                        raise Abort()
                    node_aln_len, start_idx = node_len - read["read_start"], read["read_start"]   # :856
                elif i == len(read_nodes) - 1:                                   # :857
                    if target_len < seen:
                        target_len = seen                                        # :858
                    node_aln_len, start_idx = target_len - seen, 0               # :859
                else:
                    node_aln_len, start_idx = node_len, 0                        # :861
                if not dry:
                    entry = node_base_cov_info[node]                             # :870
                    for j in range(start_idx, min(start_idx + node_aln_len, len(entry[2]))):   # :871
                        entry[2][j] = 1                                          # :872
                    entry[1] = sum(entry[2])                                     # :874
                    entry[0] = 1 if entry[1] == nodes_len[node] else 0           # :875
                seen += node_aln_len                                             # :878
                if node not in undup_read_nodes:                                 # insert() returned true (:879)
                    undup_read_nodes.add(node)
                    read_nodes_len[node] = read_nodes_len.get(node, 0) + node_aln_len   # :880
                    if not dry:
                        bases_per_node[node] += node_aln_len                     # :881
        if dry:
            continue
        if len(read_nodes) < 3:                                                  # :886
            return
        for k in range(len(read_nodes) - 2):                                     # windows(3) (:890-893)
            a, b, c = read_nodes[k:k + 3]
            len_sum = sum(read_nodes_len.get(n, 0) for n in (a, b, c))           # :897-900
            i = trio_nodes.get((a, b, c))                                        # :902
            if i is None:
                i = trio_nodes.get((c, b, a))                                    # :903-904
            if i is not None:
                trio_nodes_bases_count[i] += len_sum                             # :906
