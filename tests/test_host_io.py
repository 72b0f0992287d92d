"""CPU tests of the host readers behind the pipeline seam (a1 GAF tokenizer, a6 graph loaders).
They run through the C ABI of libpantax_hip.so but touch no GPU."""
import os

import numpy as np
import pytest

from pantax_amd import io as pio
import synthdata as synth


@pytest.fixture(scope="module")
def small(tmp_path_factory):
    sset = synth.make_set(5, 2, 4, 500, 8000, adversarial_frac=0.02)
    root = tmp_path_factory.mktemp("io")
    synth.write_db(sset, str(root))
    synth.write_gaf(sset.reads, str(root / "x.gaf"))
    return sset, root


def test_gaf_tokenizer_roundtrip(small):
    """rcls.rs:119-146: the packed arrays reproduce the generator's arrays exactly, for any thread count."""
    sset, root = small
    rd = sset.reads
    for nt in (1, 3, 8):
        g = pio.load_gaf(root / "x.gaf", n_threads=nt)
        assert np.array_equal(g["step_off"], rd.step_off.astype(np.uint32))
        assert np.array_equal(g["node_id"], rd.node_id)
        assert np.array_equal(g["pstart"], rd.pstart.astype(np.uint32)) and np.array_equal(g["pend"], rd.pend.astype(np.uint32))
        assert np.array_equal(g["qlen"], rd.qlen.astype(np.uint32)) and np.array_equal(g["mapq"], rd.mapq.astype(np.uint8))
        assert not g["flags"].any()


def test_gaf_nulls_comments_and_ragged(tmp_path):
    p = tmp_path / "y.gaf"
    p.write_text(
        "@HD\tVN:1.0\n"
        "r1\t150\t0\t150\t+\t>12<7>300\t400\t3\t153\t150\t150\t60\tNM:i:0\n"
        "r2\t150\t0\t150\t+\t*\t*\t*\t*\t*\t*\t255\n"            # unmapped: null path/len/start/end
        "r3\t100\t0\t100\t+\t<5\t30\t20\t10\t100\t100\t*\n"       # end<start single node, null mapq
        "\n"
        "r4\t90\t0\t90\t+\t>8>9\t200\t0\t90\n")                   # ragged (no mapq column), no trailing newline
    g = pio.load_gaf(p)
    assert g["step_off"].tolist() == [0, 3, 3, 4, 6]
    assert g["node_id"].tolist() == [12, 7, 300, 5, 8, 9]
    assert g["pstart"].tolist() == [3, 0, 20, 0] and g["pend"].tolist() == [153, 0, 10, 90]
    assert g["qlen"].tolist() == [150, 150, 100, 90]
    assert g["mapq"].tolist() == [60, 255, 255, 255]
    assert g["flags"].tolist() == [0, 1, 0, 0]
    with pytest.raises(Exception):
        pio.load_gaf(tmp_path / "missing.gaf")


def test_host_tokenizer_equals_the_gaf_reader_oracle(tmp_path, small):
    """a1 against a reading of the format that is not this repo's tokenizer: oracle/gaf_reader.py restates load_gaf_file_lazy
    (rcls.rs:119-137) and the rules of the CSV reader it configures; the host tokenizer reproduces its packed arrays on every quirk
    of the contract and on a generated GAF."""
    from oracle import gaf_reader
    from tests.helpers import gaf_quirks_text
    sset, root = small
    for name, text in (("quirks", gaf_quirks_text()), ("generated", (root / "x.gaf").read_bytes())):
        p = tmp_path / (name + ".gaf")
        p.write_bytes(text)
        want = gaf_reader.packed(text)
        for nt in (1, 5):
            got = pio.load_gaf(p, n_threads=nt)
            for k in want:
                assert np.array_equal(want[k], got[k]), (name, nt, k)


def test_graph_loaders_agree(small):
    """read_gfa (profile.rs:466-545) and the bincode .bin reader (zip.rs:236-247) give the generator's graph."""
    sset, root = small
    for g in sset.species:
        for fmt, path in (("gfa", root / "species_gfa" / (g.name + ".gfa")), ("bin", root / "species_graph_info" / (g.name + ".bin"))):
            node_len, names, path_off, path_nodes = pio.load_graph(path, fmt)
            assert np.array_equal(node_len, g.node_len) and names == g.hap_names
            assert np.array_equal(path_off, g.path_off) and np.array_equal(path_nodes, g.path_nodes)


def test_gfa_quirks(tmp_path):
    """P lines (hap = field 1 up to '#', ids by \\d+), several contigs of one hap concatenated (profile.rs:540),
    BTreeMap byte order of hap names, walks are NOT reversed by the GFA loader (profile.rs:502-534)."""
    p = tmp_path / "g.gfa"
    p.write_text("H\tVN:Z:1.1\nS\t1\tACGT\nS\t2\tA\nS\t3\tGG\n"
                 "P\tzeta#1#c1\t1+,2+\t*\n"
                 "W\tAlpha\t1\tc1\t0\t5\t<3<2\n"
                 "P\tzeta#1#c2\t3+\t*\n"
                 "W\tB10\t1\tc1\t0\t4\t>1\n")
    node_len, names, path_off, path_nodes = pio.load_graph(p, "gfa")
    assert node_len.tolist() == [4, 1, 2]
    assert names == ["Alpha", "B10", "zeta"]
    assert path_off.tolist() == [0, 2, 3, 6] and path_nodes.tolist() == [2, 1, 0, 0, 1, 2]
    bad = tmp_path / "bad.gfa"
    bad.write_text("S\t2\tAC\n")
    with pytest.raises(Exception):
        pio.load_graph(bad, "gfa")


def test_graph_codecs_lz4_and_zstd(small, tmp_path):
    """`--lz` / `--zstd` graph files (zip.rs:191-223): the bincode image behind an LZ4 frame or a zstd stream
    (written here with pyarrow's codecs, which produce those container formats) load to the same graph as the .bin."""
    import pyarrow as pa
    sset, db = small
    g0 = sset.species[0]
    bin_path = os.path.join(db, "species_graph_info", g0.name + ".bin")
    raw = open(bin_path, "rb").read()
    ref = pio.load_graph(bin_path, "bin")
    for fmt, codec in (("lz4", "lz4"), ("zst", "zstd")):
        p = tmp_path / (g0.name + ".bin." + fmt)
        p.write_bytes(pa.Codec(codec).compress(raw, asbytes=True))
        got = pio.load_graph(p, fmt)
        assert np.array_equal(got[0], ref[0]) and got[1] == ref[1] and np.array_equal(got[2], ref[2]) and np.array_equal(got[3], ref[3])
        bad = tmp_path / ("bad.bin." + fmt)
        bad.write_bytes(p.read_bytes()[: len(p.read_bytes()) // 2])
        with pytest.raises(Exception):
            pio.load_graph(bad, fmt)


# ---------------------------------------------------------------------------------------------
# a11: the generator behind sample_sorted (profile.rs:1287-1295).  rand 0.9.2 is not vendored under the reference, so
# what can be pinned here is the ChaCha core (published keystreams) and that the library and the oracle, written
# separately, restate the same sampler.
# ---------------------------------------------------------------------------------------------
CHACHA_ZERO_KEY = {   # first 32 keystream bytes, 256-bit zero key, zero nonce, block 0 (RFC 7539 A.1 #1; Strombergson TC1)
    20: "76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7",
    12: "9bf49a6a0755f953811fce125f2683d50429c3bb49e074147e0089a52eae155f",
    8: "3e00ef2f895f40d67f5bb8e81f09a5a12c840ec3ce9a7f3b181be188ef711a1e",
}


def test_chacha_core_known_answers():
    from oracle import oracle as orc
    from pantax_amd.engine import Engine
    z = np.zeros(8, dtype=np.uint32)
    for rounds, hexs in CHACHA_ZERO_KEY.items():
        for fn in (orc.chacha_block, Engine.chacha_block):
            assert fn(z, 0, rounds)[:8].astype("<u4").tobytes().hex() == hexs
    # RFC 7539 2.3.2 uses a 32-bit counter + 96-bit nonce; with a zero nonce the layouts coincide: block 1 of the zero key
    blk1 = Engine.chacha_block(z, 1, 20).astype("<u4").tobytes().hex()
    assert blk1.startswith("9f07e7be5551387a98ba977c732d080d")
    assert (orc.chacha_block(z, 1, 20) == Engine.chacha_block(z, 1, 20)).all()
    # RFC 7539 A.1 test vector #3 (key = 00..01, block counter 1): the vector rand_chacha's own test_chacha_true_values_b uses
    k3 = np.zeros(8, dtype=np.uint32)
    k3[7] = 0x01000000
    tv3 = ("3aeb5224ecf849929b9d828db1ced4dd832025e8018b8160b82284f3c949aa5a8eca00bbb4a73bdad192b5c42f73f2fd"
           "4e273644c8b36125a64addeb006c13a0")
    for fn in (orc.chacha_block, Engine.chacha_block):
        assert fn(k3, 1, 20).astype("<u4").tobytes().hex() == tv3
    key = np.arange(8, dtype=np.uint32) * 0x01010101
    assert (orc.chacha_block(key, 5, 12) == Engine.chacha_block(key, 5, 12)).all()


def test_row_sampler_library_equals_oracle_and_is_a_sample():
    from oracle import oracle as orc
    from pantax_amd.engine import Engine
    # every branch of rand::seq::index::sample: in-place (both cost-model halves), Floyd, rejection
    for n, k in [(600000, 500000), (1000, 500), (200000, 500), (700000, 500), (50, 10), (1000, 20), (5000, 100),
                 (501, 500), (20_000_000, 500000), (1, 1), (13, 12), (3000, 162), (3000, 163)]:
        pos = orc.sample_sorted_positions(n, k)
        bits = Engine.sample_ranks(n, k)
        assert len(pos) == k and (np.diff(pos.astype(np.int64)) > 0).all() and pos[-1] < n
        assert int(bits.sum()) == k and (np.nonzero(bits)[0] == pos).all(), (n, k)
    # deterministic in (n, amount, seed); a different seed gives a different set
    assert (orc.sample_sorted_positions(10000, 300) == orc.sample_sorted_positions(10000, 300)).all()
    assert (orc.sample_sorted_positions(10000, 300, seed=43) != orc.sample_sorted_positions(10000, 300)).any()
    # no gross bias: the mean chosen position sits in the middle
    pos = orc.sample_sorted_positions(1_000_000, 500000).astype(np.float64)
    assert abs(pos.mean() / 1e6 - 0.5) < 0.002


def test_table_float_text_matches_the_reference_exemplar_rows():
    """The only known-answer rows the reference ships (README.md:343 species table, README.md:354 strain table): every float
    cell re-emitted by the table writer's formatter is the same text (shortest round-trip digits, `16.0` / `1.0` for integral
    values), and the header lines are the reference's (README.md:342, :353)."""
    import ctypes as C
    from pantax_amd import _ffi
    lib = _ffi.load()
    lib.pantax_hip_format_f64.argtypes = [C.c_double, C.c_char_p, C.c_size_t]

    def fmt(v):
        buf = C.create_string_buffer(64)
        n = lib.pantax_hip_format_f64(v, buf, 64)
        assert n > 0
        return buf.value.decode()
    species_row = "34\t0.5005489240249426\t6.723225501680235"
    strain_row = "34\t34.4\tGCF_006401215.1_ASM640121v1\t16.0\t0.39983790355261384\t0.9967217217217217\t1.0\t15.54\t16.0\t0.01\t0.0010005002501250622"
    for cell in species_row.split("\t")[1:] + strain_row.split("\t")[3:]:
        assert fmt(float(cell)) == cell
    # beyond the exemplar: values polars prints in scientific notation take ryu's exponent form (unverified against polars
    # itself, which is not in this image: DESIGN.md); integral values keep their ".0"
    assert fmt(0.0) == "0.0" and fmt(100.0) == "100.0" and fmt(1e-7) == "1e-7" and fmt(2.5e-10) == "2.5e-10"
    buf = C.create_string_buffer(4)
    assert lib.pantax_hip_format_f64(0.39983790355261384, buf, 4) < 0          # too small a buffer is an error, not a truncation
