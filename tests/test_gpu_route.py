"""SURVEY 8e, reads over N GPUs: every rank bins a slice of the reads and routes the packed records to the rank that owns
their species (pantax_hip_reads_route_pack / pantax_hip_reads_from_routed).  The ranks are simulated one after the other on
the one GPU of the box: the messages are checked word for word against a numpy statement of the layout, and the owners'
node coverage against the one-process run bit for bit.  The multi-process exchange itself is covered by
tests/test_distributed.py (gloo) and the file-seam tests."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from pantax_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def _slice_reads(rd, a, b):
    so = rd.step_off.astype(np.int64)
    return (so[a:b + 1] - so[a]).astype(np.uint64), rd.node_id[so[a]:so[b]], rd.pstart[a:b], rd.pend[a:b], rd.qlen[a:b], rd.mapq[a:b]


def _expected_message(cols, sp, flags, owner, d):
    """numpy statement of the message layout: reads of the slice whose species is owned by rank d, slice order kept"""
    step_off, node_id, pstart, pend, qlen, mapq = cols
    so = step_off.astype(np.int64)
    k = so[1:] - so[:-1]
    own = np.where(sp >= 0, owner[np.maximum(sp, 0)], -1)
    if flags is not None:
        own = np.where(flags != 0, -1, own)
    sel = np.nonzero(own == d)[0]
    ns = k[sel]
    idx = np.repeat(so[:-1][sel], ns) + (np.arange(int(ns.sum())) - np.repeat(np.cumsum(ns) - ns, ns))
    mq = np.where((mapq < 0) | (mapq > 254), 255, mapq)
    return np.concatenate([ns, pstart[sel], pend[sel], qlen[sel], mq[sel], node_id[idx]]).astype(np.uint32), len(sel), int(ns.sum())


@pytest.mark.parametrize("seed,S,H,R,L,W,long_reads", [(11, 5, 4, 30000, 20000, 2, False), (12, 7, 3, 50001, 15000, 3, False),
                                                      (13, 3, 4, 900, 60000, 4, True), (14, 6, 3, 20000, 15000, 8, False)])
def test_route_messages_and_owner_coverage_equal_one_process(eng, seed, S, H, R, L, W, long_reads):
    import synthdata as synth
    from pantax_amd.pipeline import partition_species
    sset = synth.make_set(seed, S, H, R, L, long_reads=long_reads)
    rd = sset.reads
    rng = np.random.default_rng(seed)
    flags_all = (rng.random(rd.n_reads) < 0.02).astype(np.uint8) * np.uint8(1 + (seed & 1))
    # ---- one process: everything on one GPU
    eng.upload_db(sset.species)
    eng.upload_packed(rd, flags_all)
    sp_all, rc, *_ = eng.rcls_profile()
    eng.trio_nodes_info(fetch=False)
    bases1, cov1, tb1, nab1 = eng.get_node_abundances()
    hto = None
    owner = np.array(partition_species([8.0 * rc[i] + g.n_nodes for i, g in enumerate(sset.species)], W), dtype=np.int32)
    owner[S - 1] = -1 if S > 3 else owner[S - 1]          # one species nobody owns: its reads are left behind
    # ---- W ranks, one after the other: slice -> bin against ALL ranges -> pack
    cuts = np.linspace(0, rd.n_reads, W + 1).astype(np.int64)
    msgs, n_r, n_t = [], np.zeros((W, W), dtype=np.uint64), np.zeros((W, W), dtype=np.uint64)
    for r in range(W):
        a, b = int(cuts[r]), int(cuts[r + 1])
        cols = _slice_reads(rd, a, b)
        mq = np.where((cols[5] < 0) | (cols[5] > 254), 255, cols[5])
        eng.upload_reads(cols[0], cols[1], cols[2], cols[3], cols[4], mq, flags_all[a:b])
        sp, *_ = eng.rcls_profile()
        assert np.array_equal(sp, sp_all[a:b])
        rt, nr, nt = eng.route_pack(owner, W)
        got = eng.route_messages(rt, W)
        eng.route_free(rt)
        for d in range(W):
            exp, er, et = _expected_message(cols, sp, flags_all[a:b], owner, d)
            assert (int(nr[d]), int(nt[d])) == (er, et)
            assert np.array_equal(got[d], exp), (r, d)
        msgs.append(got)
        n_r[r], n_t[r] = nr, nt
    # ---- every owner: messages in source order -> resident reads -> coverage of its species == the one-process run
    nb = np.concatenate([[0], np.cumsum([g.n_nodes for g in sset.species])])
    seen_abort = 0
    for d in range(W):
        mine = [i for i in range(S) if owner[i] == d]
        if not mine:
            continue
        eng.upload_db([sset.species[i] for i in mine])
        recv = np.concatenate([msgs[r][d] for r in range(W)]) if W else np.zeros(0, dtype=np.uint32)
        eng.reads_from_routed(recv, n_r[:, d], n_t[:, d])
        sp_d, *_ = eng.rcls_profile()
        assert (sp_d >= 0).all() and len(sp_d) == int(n_r[:, d].sum())       # only reads of owned species arrived
        eng.trio_nodes_info(fetch=False)
        bases, cov, tb, nab = eng.get_node_abundances()
        seen_abort += nab
        off = 0
        for i in mine:
            n = sset.species[i].n_nodes
            assert np.array_equal(bases[off:off + n], bases1[nb[i]:nb[i + 1]]), (d, i)
            assert np.array_equal(cov[off:off + n], cov1[nb[i]:nb[i + 1]]), (d, i)
            off += n
    if (owner >= 0).all():
        assert seen_abort == nab1


def test_route_edge_cases(eng):
    """no reads at all, a slice whose reads all stay behind, an owner that receives nothing, and the argument checks"""
    import synthdata as synth
    from pantax_amd._ffi import PantaxHipError
    sset = synth.make_set(21, 2, 3, 500, 8000)
    eng.upload_db(sset.species)
    eng.upload_reads(np.zeros(1), np.zeros(0), np.zeros(0), np.zeros(0), np.zeros(0), np.zeros(0))
    eng.rcls_profile()
    rt, nr, nt = eng.route_pack([0, 1], 2)
    assert nr.sum() == 0 and nt.sum() == 0 and all(len(m) == 0 for m in eng.route_messages(rt, 2))
    eng.route_free(rt)
    eng.upload_packed(sset.reads)
    eng.rcls_profile()
    rt, nr, nt = eng.route_pack([-1, -1], 3)
    assert nr.sum() == 0
    eng.route_free(rt)
    rt, nr, nt = eng.route_pack([2, 2], 3)                       # ranks 0 and 1 receive nothing
    assert nr[0] == 0 and nr[1] == 0 and nr[2] > 0
    m = eng.route_messages(rt, 3)
    eng.route_free(rt)
    eng.reads_from_routed(np.concatenate([m[2], np.zeros(0, dtype=np.uint32)]), [nr[2], 0], [nt[2], 0])
    sp, *_ = eng.rcls_profile()
    assert len(sp) == int(nr[2]) and (sp >= 0).all()
    with pytest.raises(PantaxHipError):
        eng.upload_packed(sset.reads)
        eng.rcls_profile()
        eng.route_pack([0, 5], 2)                                # owner outside the world
    with pytest.raises(PantaxHipError):
        eng.reads_from_routed(m[2], [nr[2]], [nt[2] + 1])        # announced steps disagree with the message
    eng.upload_packed(sset.reads)                                 # leave the engine usable


def test_route_reads_through_rccl_on_device_buffers(tmp_path):
    """pipeline.route_reads with backend nccl (= RCCL): the all-to-all(v) runs on the library's DEVICE buffers (no staging)
    and the result equals the host-staged exchange.  One rank here (one GPU per box); the N-rank exchange logic is covered
    under gloo by tests/test_distributed.py."""
    import subprocess
    import sys
    from tests.conftest import ROOT
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, %r)
import torch, torch.distributed as dist
import synthdata as synth
from pantax_amd.engine import Engine
from pantax_amd.pipeline import TorchComm, route_reads
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
comm = TorchComm(device=torch.device("cuda", 0))
sset = synth.make_set(41, 4, 3, 20000, 15000)
res = []
for on_device in (True, False):
    eng = Engine(0)
    eng.upload_ranges([g.range_start for g in sset.species], [g.range_end for g in sset.species])
    eng.upload_packed(sset.reads)
    eng.rcls_profile()
    st = route_reads(eng, [0, 0, -1, 0], comm, on_device=on_device)
    eng.upload_db(sset.species)
    sp, rc, *_ = eng.rcls_profile()
    eng.trio_nodes_info(fetch=False)
    bases, cov, tb, nab = eng.get_node_abundances()
    res.append((st, sp.copy(), bases.copy(), cov.copy()))
    eng.close()
assert res[0][0] == res[1][0] and res[0][0]["recv_reads"] == res[0][0]["sent_reads"] > 0
assert (res[0][1] != 2).all()                     # species 2 has no owner: its reads stayed behind
for a, b in zip(res[0][1:], res[1][1:]):
    assert np.array_equal(a, b)
assert comm.allreduce_sum(np.array([1.5, 2.0])).tolist() == [1.5, 2.0]
assert comm.alltoall_counts(np.array([[7], [9]])).tolist() == [[7], [9]]
dist.destroy_process_group()
print("rccl route ok")
''' % ROOT
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "rccl route ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.gpu
def test_bench_over_two_gpus_gives_the_one_gpu_tables():
    """`python bench.py --gpus 2 --workload cfg3` over RCCL on TWO devices (the driver's SCALE run; skipped on a one-GPU box): the species are
    sharded over the ranks, the reads routed in one all-to-all(v) on device buffers, one all-reduce per step -- and the tables are those of
    N = 1 (same rows, same top strains), the line says rccl_ranks = 2 and carries the abundance-L1 figure, the route time and the ranks' step
    times (profile.rs:3297-3319: species are independent; :341, :3198, :3243 the normalisers that cross ranks)."""
    import json
    import subprocess
    import sys
    import torch
    from tests.conftest import ROOT
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    common = ["--workload", "cfg3", "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-gaf", "--no-hard"]
    lines = []
    for n in (1, 2):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + common, capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert p.returncode == 0, p.stderr[-2000:]
        lines.append(json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]))
    one, two = lines
    assert two["n_gpus"] == 2 and two["config"]["rccl_ranks"] == 2
    assert two["result"]["n_species_rows"] == one["result"]["n_species_rows"] and two["result"]["n_strain_rows"] == one["result"]["n_strain_rows"]
    for a, b in zip(one["result"]["top_strains"], two["result"]["top_strains"]):
        assert a[:2] == b[:2] and a[2] == pytest.approx(b[2], rel=1e-9) and a[3] == pytest.approx(b[3], rel=1e-9)
    assert two["config"]["abundance_l1_vs_oracle"] is not None and two["config"]["abundance_l1_vs_oracle"] <= 1e-4
    assert two["config"]["ingest_route_ms"] is not None and len(two["config"]["ms_per_step_ranks_min_max"]) == 2
