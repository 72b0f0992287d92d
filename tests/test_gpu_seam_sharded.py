"""SURVEY 8e through the file seam: with an `alltoallv` callback pantax_hip_profile shards the INPUT as well -- rank r
tokenises and bins only its line-aligned byte range of the GAF, the species counters are summed, the duplicate-id rule is
decided on exchanged (id hash, species) records and the packed reads travel to the rank that owns their species.  The ranks
are threads with their own ctx on the one GPU of the box; the collectives are barrier-based host implementations of the two
callbacks.  Everything is compared with the one-process run of the same files and with the oracle's tables."""
import os
import threading

import numpy as np
import pytest

from tests.test_gpu_pipeline import _check_outputs, _oracle_tables, world  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


class ThreadComm:
    """sum all-reduce and byte all-to-all(v) between `n` threads of this process"""

    def __init__(self, n, timeout=180):
        self.n, self.timeout = n, timeout
        self.bar = threading.Barrier(n)
        self.slots = [None] * n
        self.total = [None]
        self.calls = {"allreduce": 0, "alltoallv": 0, "bytes": 0}

    def allreduce(self, rank):
        def fn(buf):
            self.slots[rank] = buf.copy()
            if self.bar.wait(self.timeout) == 0:
                self.total[0] = np.sum(self.slots, axis=0)
                self.calls["allreduce"] += 1
            self.bar.wait(self.timeout)
            buf[:] = self.total[0]
            self.bar.wait(self.timeout)
        return fn

    def alltoallv(self, rank):
        def fn(send, send_off, recv, recv_off):
            self.slots[rank] = (send.copy(), send_off.copy())
            self.bar.wait(self.timeout)
            for i in range(self.n):
                s, so = self.slots[i]
                part = s[so[rank]:so[rank + 1]]
                assert len(part) == recv_off[i + 1] - recv_off[i], "recv offsets disagree with what rank %d sends" % i
                recv[recv_off[i]:recv_off[i + 1]] = part
            if rank == 0:
                self.calls["alltoallv"] += 1
                self.calls["bytes"] += sum(int(self.slots[i][1][-1]) for i in range(self.n))
            self.bar.wait(self.timeout)
        return fn


def _run_ranks(n, wd, call):
    """call(eng, rank, comm) on n threads, each with its own ctx; -> (errors by rank, comm)"""
    from pantax_amd.engine import Engine, PantaxHipError
    comm = ThreadComm(n)
    errs = {}

    def run(rank):
        eng = Engine(0)
        try:
            call(eng, rank, comm)
        except PantaxHipError as e:
            errs[rank] = str(e)
        except Exception as e:   # noqa: BLE001
            errs[rank] = "unexpected: %r" % (e,)
            comm.bar.abort()
        finally:
            eng.close()
    cwd = os.getcwd()
    os.chdir(str(wd))
    try:
        ths = [threading.Thread(target=run, args=(r,)) for r in range(n)]
        for t in ths:
            t.start()
        for t in ths:
            t.join(timeout=300)
            assert not t.is_alive(), "a rank is still waiting in a collective"
    finally:
        os.chdir(cwd)
    return errs, comm


@pytest.mark.parametrize("world_size", [2, 3, 5])
def test_sharded_ingest_equals_one_process(world, world_size):
    sset, root, db, gaf, eng0 = world
    exp_species, exp_strain, _ = _oracle_tables(sset)
    one = root / ("wd_one_for_%d" % world_size)
    one.mkdir()
    cwd = os.getcwd()
    os.chdir(str(one))
    try:
        eng0.profile(str(db), str(one), str(gaf), out_binning_file=str(one / "reads_classification.tsv"))
    finally:
        os.chdir(cwd)
    wd = root / ("wd_sharded_ingest%d" % world_size)
    wd.mkdir()

    def call(eng, rank, comm):
        eng.profile(str(db), str(wd), str(gaf), rank=rank, world_size=world_size, allreduce=comm.allreduce(rank),
                    alltoallv=comm.alltoallv(rank), out_binning_file=str(wd / "reads_classification.tsv"))
    errs, comm = _run_ranks(world_size, wd, call)
    assert not errs, errs
    _check_outputs(str(wd), sset, exp_species, exp_strain)
    # integer counters summed over the ranks: the species table and the binning report are the one-process files byte for byte
    assert open(wd / "species_abundance.txt").read() == open(one / "species_abundance.txt").read()
    assert open(wd / "reads_classification.tsv", "rb").read() == open(one / "reads_classification.tsv", "rb").read()
    a = [l.split("\t") for l in open(one / "strain_abundance.txt")]
    b = [l.split("\t") for l in open(wd / "strain_abundance.txt")]
    assert [r[:3] for r in a] == [r[:3] for r in b]                         # same rows in the same order
    for ra, rb in zip(a[1:], b[1:]):   # the same kernels on the same reads in the same order; reductions whose shape follows the batch differ in the last digits
        for x, y in zip(ra[3:], rb[3:]):
            assert (x.strip() == "" and y.strip() == "") or float(x) == pytest.approx(float(y), rel=1e-9, abs=1e-12)
    assert not [f for f in os.listdir(wd) if ".part" in f]
    assert comm.calls["alltoallv"] == 2                                     # id records + packed reads (no id spans species here)
    # the payload is the packed reads of selected species, not the text: well under the size of the GAF
    assert comm.calls["bytes"] < os.path.getsize(gaf)
    # resume (profile.rs:3365-3417): strain level redone from the saved report, every rank taking the rows of its own slice
    os.remove(wd / "strain_abundance.txt")

    def call2(eng, rank, comm):
        eng.profile(str(db), str(wd), str(gaf), rank=rank, world_size=world_size, allreduce=comm.allreduce(rank), alltoallv=comm.alltoallv(rank))
    errs, _ = _run_ranks(world_size, wd, call2)
    assert not errs, errs
    _check_outputs(str(wd), sset, exp_species, exp_strain)


def test_sharded_ingest_duplicate_ids_across_slices(world):
    """An id whose alignments sit in DIFFERENT ranks' byte ranges: same species -> all kept, two species -> all dropped at the
    strain level (profile.rs:361-463), exactly as the one-process run decides."""
    import copy
    import synthdata as synth
    sset, root, db, gaf, eng0 = world
    rd = copy.copy(sset.reads)
    R = rd.n_reads
    _, _, sp = _oracle_tables(sset)
    ids = ["S0R%d/1" % r for r in range(R)]
    rng = np.random.default_rng(7)
    mapped = np.nonzero(sp >= 0)[0]
    lo, hi = mapped[mapped < R // 3], mapped[mapped > 2 * R // 3]           # partners far apart in the file
    drop = np.zeros(R, bool)
    n_same = n_mixed = 0
    for a, b in zip(rng.choice(lo, 1500, replace=False), rng.choice(hi, 1500, replace=False)):
        ids[b] = ids[a]
        if sp[a] == sp[b]:
            n_same += 1
        else:
            drop[a] = drop[b] = True
            n_mixed += 1
    assert n_same > 50 and n_mixed > 50
    rd.read_id = ids
    gaf2 = root / "dup_far.gaf"
    synth.write_gaf(rd, str(gaf2))
    exp_species, exp_strain, _ = _oracle_tables(sset, strain_drop=drop)
    wd = root / "wd_sharded_dup"
    wd.mkdir()

    def call(eng, rank, comm):
        eng.profile(str(db), str(wd), str(gaf2), rank=rank, world_size=3, allreduce=comm.allreduce(rank), alltoallv=comm.alltoallv(rank))
    errs, comm = _run_ranks(3, wd, call)
    assert not errs, errs
    _check_outputs(str(wd), sset, exp_species, exp_strain)
    assert comm.calls["alltoallv"] == 3                                     # + the ids to drop
    _, plain, _ = _oracle_tables(sset)
    assert any(abs(x[2] - y[2]) > 1e-9 for x, y in zip(plain, exp_strain)) or len(plain) != len(exp_strain)


@pytest.mark.parametrize("sharded", [False, True])
def test_rank_local_failure_before_the_strain_step_reaches_every_rank(world, sharded):
    """Rank 0 cannot write the species table (its output directory is below a regular file): the failure travels in the
    next collective's flag, every rank returns an error and nobody waits (the round-1 seam returned on rank 0 only)."""
    sset, root, db, gaf, eng0 = world
    wd = root / ("wd_fail_early_%d" % int(sharded))
    wd.mkdir()
    blocker = wd / "not_a_dir"
    blocker.write_text("x")

    def call(eng, rank, comm):
        eng.profile(str(db), str(wd), str(gaf), rank=rank, world_size=2, allreduce=comm.allreduce(rank),
                    alltoallv=comm.alltoallv(rank) if sharded else None, output_dir=str(blocker / "out"))
    errs, _ = _run_ranks(2, wd, call)
    assert set(errs) == {0, 1}, errs
    assert "cannot write" in errs[0] and "another rank failed" in errs[1]
    assert not os.path.exists(wd / "strain_abundance.txt")


def test_sharded_ingest_with_slices_cut_into_pieces_and_long_walks(tmp_path_factory, monkeypatch):
    """Every rank's byte range is itself tokenised in several pieces (PANTAX_GAF_PIECE_BYTES: the 4-GiB logic at small scale,
    read from the file at the slice's offset) and the reads are long (walks of hundreds of steps, routed by the workgroup-wide
    copy): tables == the one-process run of the same files, report byte-identical."""
    import synthdata as synth
    from pantax_amd.engine import Engine
    sset = synth.make_set(55, 3, 4, 600, 60000, long_reads=True, present_frac=0.6, with_ids=False)
    root = tmp_path_factory.mktemp("pantax_long")
    db = root / "db"
    db.mkdir()
    synth.write_db(sset, str(db))
    gaf = root / "long.gaf"
    synth.write_gaf(sset.reads, str(gaf))
    one = root / "wd_one"
    one.mkdir()
    eng0 = Engine(0)
    cwd = os.getcwd()
    os.chdir(str(one))
    try:
        eng0.profile(str(db), str(one), str(gaf), fr=0.5, out_binning_file=str(one / "reads_classification.tsv"))
    finally:
        os.chdir(cwd)
        eng0.close()
    monkeypatch.setenv("PANTAX_GAF_PIECE_BYTES", str(max(200000, os.path.getsize(gaf) // 11)))
    wd = root / "wd_sharded"
    wd.mkdir()

    def call(eng, rank, comm):
        eng.profile(str(db), str(wd), str(gaf), fr=0.5, rank=rank, world_size=3, allreduce=comm.allreduce(rank), alltoallv=comm.alltoallv(rank),
                    out_binning_file=str(wd / "reads_classification.tsv"))
    errs, comm = _run_ranks(3, wd, call)
    assert not errs, errs
    assert open(wd / "species_abundance.txt").read() == open(one / "species_abundance.txt").read()
    assert open(wd / "reads_classification.tsv", "rb").read() == open(one / "reads_classification.tsv", "rb").read()
    a = [l.rstrip("\n").split("\t") for l in open(one / "strain_abundance.txt")]
    b = [l.rstrip("\n").split("\t") for l in open(wd / "strain_abundance.txt")]
    assert len(a) > 1 and [r[:3] for r in a] == [r[:3] for r in b]
    for ra, rb in zip(a[1:], b[1:]):
        for x, y in zip(ra[3:], rb[3:]):
            assert (x == "" and y == "") or float(x) == pytest.approx(float(y), rel=1e-9, abs=1e-12)


def test_one_rank_world_goes_through_the_callbacks(world):
    """world_size 1 WITH the callbacks (an MPI-style program started on one rank): the whole multi-rank protocol -- byte range,
    counter all-reduce, id-record exchange, read routing (everything to itself), part files -- runs, and gives the files of
    the plain one-process call."""
    sset, root, db, gaf, eng0 = world
    exp_species, exp_strain, _ = _oracle_tables(sset)
    one = root / "wd_one_plain"
    one.mkdir()
    cwd = os.getcwd()
    os.chdir(str(one))
    try:
        eng0.profile(str(db), str(one), str(gaf), out_binning_file=str(one / "reads_classification.tsv"))
    finally:
        os.chdir(cwd)
    wd = root / "wd_one_rank_comm"
    wd.mkdir()

    def call(eng, rank, comm):
        eng.profile(str(db), str(wd), str(gaf), rank=0, world_size=1, allreduce=comm.allreduce(0), alltoallv=comm.alltoallv(0),
                    out_binning_file=str(wd / "reads_classification.tsv"))
    errs, comm = _run_ranks(1, wd, call)
    assert not errs, errs
    assert comm.calls["alltoallv"] == 2 and comm.calls["allreduce"] >= 4
    _check_outputs(str(wd), sset, exp_species, exp_strain)
    for f in ("species_abundance.txt", "strain_abundance.txt", "reads_classification.tsv"):
        assert open(wd / f, "rb").read() == open(one / f, "rb").read(), f      # one rank: also the sums are the same bits


def test_cli_over_rccl_with_one_rank(world):
    """pantax-hip --ranks 1 --rank 0: the CLI's RCCL communicator (unique id through a file in the work directory,
    ncclAllReduce for the sums, grouped ncclSend / ncclRecv on device buffers for the all-to-all(v)) carries the sharded
    protocol; one rank is what a one-GPU box can run -- the N-rank protocol itself is covered by the thread-rank tests above."""
    import subprocess
    from tests.conftest import ROOT
    sset, root, db, gaf, eng0 = world
    exe = os.path.join(ROOT, "pantax_amd", "lib", "pantax-hip")
    ref = root / "wd_cli_plain"
    ref.mkdir()
    wd = root / "wd_cli_rccl"
    wd.mkdir()
    base = [exe, "-db", str(db), "--gaf", str(gaf), "--species", "--strain", "--short-read", "--sample", "0"]
    r0 = subprocess.run(base + ["-T", str(ref)], cwd=str(ref), capture_output=True, text=True, timeout=300)
    assert r0.returncode == 0, r0.stderr
    env = dict(os.environ, PANTAX_HIP_TRACE="1")
    r1 = subprocess.run(base + ["-T", str(wd), "--ranks", "1", "--rank", "0"], cwd=str(wd), capture_output=True, text=True, timeout=300, env=env)
    assert r1.returncode == 0, r1.stderr[-3000:]
    assert "route reads to owners" in r1.stderr            # the sharded path ran (phase trace)
    for f in ("species_abundance.txt", "strain_abundance.txt"):
        assert open(wd / f).read() == open(ref / f).read(), f
    assert not os.path.exists(wd / ".pantax_hip_rccl_id") and not [f for f in os.listdir(wd) if ".part" in f]
    r2 = subprocess.run(base + ["-T", str(wd), "--ranks", "2", "--rank", "5"], capture_output=True, text=True, timeout=60)
    assert r2.returncode != 0
