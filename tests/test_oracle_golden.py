"""CPU: pins the oracle (oracle/kmer_oracle.c and oracle/pyoracle.py) against the golden vectors that
the REFERENCE produced (tests/golden/manifest.json, written by oracle/gen_golden.py)."""
import hashlib
import os

import numpy as np
import pytest

import inputs
import oracle
from oracle import pyoracle

SMALL = ["G1_kat_k3", "G1_kat_k5", "G1_kat_k7", "G2_c1_k7", "G3_edge_k3", "G3_edge_k7", "G3_edge_k9",
         "G4_edge_flush1000_k7", "G5_c2_2M_k7"]


def _stats_fields(table):
    hist, vals = oracle.table_stats(table)
    return {"hist": hist.tolist(), "hist_sum": int(hist.sum()), "hist_count": int(np.count_nonzero(hist)),
            "hist_min": int(hist.min()), "hist_max": int(hist.max()), "vals_sum": int(vals[0]),
            "vals_count": int(vals[1]), "vals_min": int(vals[2]), "vals_max": int(vals[3])}


def _check_case(case, data, got):
    e = case["expect"]
    assert got["num_kmers"] == e["num_kmers"]
    assert [list(c) for c in oracle.chromosomes(data, got["records"])] == e["chromosomes"]
    for k, v in _stats_fields(got["table"]).items():
        assert v == e[k], k
    assert hashlib.sha256(got["table"].tobytes()).hexdigest() == e["output_file_cheksum"]
    assert got["table"].size == e["output_file_size"] == e["data_size"]


@pytest.mark.parametrize("name", SMALL)
def test_c_oracle_matches_reference_small(manifest, small_tables, name):
    case = manifest["indexer"][name]
    data = inputs.make_input(case["input"])
    assert inputs.sha256(data) == case["input_sha256"], "seeded input drifted from the one the reference saw"
    got = oracle.count_fasta(data, case["k"])
    assert np.array_equal(got["table"], small_tables[name])
    _check_case(case, data, got)


def test_c_oracle_gz_input(manifest, small_tables):
    """G3 .gz: the reference reads it through gzip.open('rt') (indexer.py:112-115); the table is that of the text."""
    import gzip
    case = manifest["indexer"]["G3_edge_gz_k7"]
    data = gzip.decompress(inputs.make_input(case["input"]))
    got = oracle.count_fasta(data, 7)
    assert np.array_equal(got["table"], small_tables["G3_edge_gz_k7"])
    assert got["num_kmers"] == case["expect"]["num_kmers"]


@pytest.mark.parametrize("name", ["G3_edge_k15", "G5_c2_20M_k15", "G5_c1_4M_k13"])
def test_c_oracle_matches_reference_k13_k15(manifest, name):
    case = manifest["indexer"][name]
    data = inputs.make_input(case["input"])
    assert inputs.sha256(data) == case["input_sha256"]
    _check_case(case, data, oracle.count_fasta(data, case["k"]))


@pytest.mark.skipif(os.environ.get("PK_TEST_K17") != "1", reason="needs a 16 GiB host table; set PK_TEST_K17=1")
def test_c_oracle_matches_reference_k17(manifest):
    case = manifest["indexer"]["G5_c2_8M_k17"]
    data = inputs.make_input(case["input"])
    _check_case(case, data, oracle.count_fasta(data, 17))


@pytest.mark.parametrize("name", ["G1_kat_k3", "G1_kat_k5", "G3_edge_k3", "G3_edge_k7", "G3_edge_k9"])
def test_python_restatement_matches_reference(manifest, small_tables, name):
    case = manifest["indexer"][name]
    data = inputs.make_input(case["input"])
    table, num_kmers, chromosomes, _ = pyoracle.count_fasta(data, case["k"])
    assert np.array_equal(table, small_tables[name])
    assert num_kmers == case["expect"]["num_kmers"]
    assert [list(c) for c in chromosomes] == case["expect"]["chromosomes"]
    st = pyoracle.table_stats(table)
    for k, v in st.items():
        assert v == case["expect"][k], k


def test_python_restatement_is_batch_independent(small_tables):
    """G4: the reference's table does not depend on flush_every (indexer.py:262 saturating add)."""
    data = inputs.edge_fasta()
    t1, *_ = pyoracle.count_fasta(data, 7, flush_every=1000)
    t2, *_ = pyoracle.count_fasta(data, 7, flush_every=7)
    assert np.array_equal(t1, small_tables["G3_edge_k7"]) and np.array_equal(t2, t1)
    assert np.array_equal(small_tables["G4_edge_flush1000_k7"], small_tables["G3_edge_k7"])


def test_kat_analytic_answer(small_tables):
    """all 4^k k-mers once, k odd => every canonical address holds 2, every other 0."""
    for k in (3, 5, 7):
        t = small_tables[f"G1_kat_k{k}"]
        a = np.arange(4 ** k, dtype=np.uint64)
        rc = np.zeros_like(a)
        x = a.copy()
        for _ in range(k):
            rc = (rc << np.uint64(2)) | (np.uint64(3) - (x & np.uint64(3)))
            x >>= np.uint64(2)
        assert np.array_equal(t, np.where(a <= rc, 2, 0).astype(np.uint8))


def test_c_and_python_oracles_agree_on_fuzz():
    rng = np.random.default_rng(21)
    alphabet = np.frombuffer(b"ACGTacgtNn>> \t\r\n\n\n\x0b\x0cXR", dtype=np.uint8)
    for trial in range(25):
        n = int(rng.integers(1, 4000))
        w = rng.random(alphabet.size) ** 3
        data = alphabet[rng.choice(alphabet.size, size=n, p=w / w.sum())].tobytes()
        for k in (1, 3, 5):
            c = oracle.count_fasta(data, k)
            table, nk, chrom, everything = pyoracle.count_fasta(data, k)
            assert c["num_kmers"] == nk and np.array_equal(c["table"], table)
            assert oracle.chromosomes(data, c["records"]) == chrom
            assert [int(x) for x in c["records"]["seq_len"]] == [e[1] for e in everything]


@pytest.mark.parametrize("tag", ["default", "min2", "max3", "min2max5"])
def test_merge_oracle_matches_reference(manifest, tag):
    """G7: matrix written by the reference's merger.py for 13 reference-indexed tables (off-diagonal)."""
    case = manifest["merger"][f"G7_k7_n13_{tag}"]
    tables = [oracle.count_fasta(inputs.make_input(spec), case["k"])["table"] for spec in case["inputs"]]
    args = case["args"]
    mn = int(args[args.index("--min-count") + 1]) if "--min-count" in args else 1
    mx = int(args[args.index("--max-count") + 1]) if "--max-count" in args else 255
    want = np.array(case["matrix"], dtype=np.uint64)
    assert np.array_equal(oracle.gram(tables, mn, mx), want)
    assert np.array_equal(pyoracle.gram(tables, mn, mx), want)
    assert case["order"] == sorted(case["order"])           # matrix order = sorted paths (merger.py:228)


def test_oracle_rejects_even_k():
    with pytest.raises(ValueError):
        oracle.count_fasta(b">a\nACGT\n", 4)
    with pytest.raises(AssertionError):
        pyoracle.count_fasta(b">a\nACGT\n", 4)


@pytest.mark.parametrize("threads", [1, 3, 8, 61])
def test_all_cores_port_equals_scalar_port(threads):
    """oracle/kmer_oracle_mt.c (bench.py's all-cores CPU leg) against pko_count_fasta: any number of threads,
    splits landing next to headers, N runs, CRLF, lone \\r, text before the first header."""
    import synth
    cases = {
        "c2": synth.c2(2_000_000)[0], "c1": synth.c1(1_000_000)[0],
        "crlf": synth.generate(5, 300_000, 3, crlf=True, pm_ngap=20)[0],
        "many_records": synth.generate(6, 200_000, 400)[0],
        "preamble": np.frombuffer(b"ACGTACGTAC\nACGT\n>r1\nACGTTGCA\nAC\n>r2 x y\nNNACGTACGTA\n", dtype=np.uint8),
        "lone_cr": np.frombuffer(b">h\nAC\rGT\rACGTAC\n\n\nACGTTT", dtype=np.uint8),
        "empty": np.zeros(0, dtype=np.uint8),
    }
    for name, data in cases.items():
        for k in (3, 7, 15):
            if k == 15 and not (name == "c2" and threads == 8):          # 1 GiB tables: once is enough
                continue
            want = oracle.count_fasta(data, k)
            got = oracle.count_fasta_mt(data, k, threads)
            assert got is not None, name
            assert got["num_kmers"] == want["num_kmers"] and got["total_bp"] == want["total_bp"], (name, k)
            assert np.array_equal(got["table"], want["table"]), (name, k)


def test_all_cores_port_declines_blanks_in_sequence_lines():
    assert oracle.count_fasta_mt(inputs.edge_fasta(), 7, 4) is None          # interior / leading blanks: scalar port only


def test_gram_mt_equals_scalar_gram():
    """bench.py's all-cores merge baseline (pair loop on a thread pool, merger.py:137-153) gives the scalar port's matrix."""
    rng = np.random.default_rng(31)
    tables = []
    for _ in range(6):
        t = rng.integers(0, 8, size=40_000, dtype=np.uint8)
        t[rng.random(t.size) < 0.5] = 0
        tables.append(t)
    for mn, mx in ((1, 255), (2, 5)):
        want = oracle.gram(tables, mn, mx)
        for threads in (1, 3):
            assert np.array_equal(oracle.gram_mt(tables, mn, mx, threads), want)
