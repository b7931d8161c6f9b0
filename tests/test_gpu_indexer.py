"""GPU parity tests for the indexer path: HIP kernels (through the C-ABI) vs the oracle and vs the
golden vectors the reference itself produced (tests/golden/, oracle/gen_golden.py).  Bit-exact."""
import hashlib

import numpy as np
import pytest

import inputs
import oracle

pytestmark = pytest.mark.gpu


def _check_against_oracle(gpu, data, k):
    got = gpu.count_fasta(data, k)
    want = oracle.count_fasta(data, k)
    assert got["num_kmers"] == want["num_kmers"]
    assert got["total_bp"] == want["total_bp"]
    assert len(got["records"]) == len(want["records"])
    for f in ("name_off", "name_len", "seq_len", "n_valid_kmers"):
        assert np.array_equal(got["records"][f], want["records"][f]), f
    assert np.array_equal(got["table"], want["table"])
    hist, vals = oracle.table_stats(want["table"])
    assert np.array_equal(got["hist256"][1:], hist)
    assert int(got["hist256"].sum()) == 4 ** k
    return got


def _expect_fields(got, data, case):
    e = case["expect"]
    assert got["num_kmers"] == e["num_kmers"]
    assert [list(c) for c in oracle.chromosomes(data, got["records"])] == e["chromosomes"]
    h = got["hist256"]
    assert h[1:].tolist() == e["hist"]
    v = np.arange(256, dtype=np.uint64)
    assert int((h * v).sum()) == e["vals_sum"]
    assert int(h[1:].sum()) == e["vals_count"]
    nz = np.nonzero(h)[0]
    assert int(nz.min()) == e["vals_min"] and int(nz.max()) == e["vals_max"]
    assert hashlib.sha256(got["table"].tobytes()).hexdigest() == e["output_file_cheksum"]


@pytest.mark.parametrize("k", [1, 3, 5, 7, 9, 11, 13])
def test_edge_fasta_vs_oracle(gpu, k):
    _check_against_oracle(gpu, inputs.edge_fasta(), k)


@pytest.mark.parametrize("name", ["G1_kat_k3", "G1_kat_k5", "G1_kat_k7", "G2_c1_k7", "G3_edge_k3", "G3_edge_k7",
                                  "G3_edge_k9", "G5_c2_2M_k7"])
def test_golden_small(gpu, manifest, small_tables, name):
    case = manifest["indexer"][name]
    data = inputs.make_input(case["input"])
    assert inputs.sha256(data) == case["input_sha256"], "seeded input drifted"
    got = gpu.count_fasta(data, case["k"])
    assert np.array_equal(got["table"], small_tables[name])
    _expect_fields(got, data, case)


def test_kat_analytic(gpu):
    """every k-mer once, k odd -> each canonical address holds exactly 2."""
    k = 7
    got = gpu.count_fasta(inputs.kat_fasta(k), k)
    assert got["num_kmers"] == 4 ** k and got["hist256"][2] == 4 ** k // 2 and got["hist256"][0] == 4 ** k // 2
    assert len(got["records"]) == 4 ** k and (got["records"]["n_valid_kmers"] == 1).all()


@pytest.mark.parametrize("name", ["G3_edge_k15", "G5_c2_20M_k15", "G5_c1_4M_k13"])
def test_golden_k13_k15(gpu, manifest, name):
    case = manifest["indexer"][name]
    data = inputs.make_input(case["input"])
    assert inputs.sha256(data) == case["input_sha256"]
    _expect_fields(gpu.count_fasta(data, case["k"]), data, case)


def test_golden_k17(gpu, manifest):
    """config 4 scaled: 4^17 = 16 GiB table resident in HBM (64 GiB of u32 counters while counting)."""
    case = manifest["indexer"]["G5_c2_8M_k17"]
    data = inputs.make_input(case["input"])
    assert inputs.sha256(data) == case["input_sha256"]
    _expect_fields(gpu.count_fasta(data, 17), data, case)


def test_full_config4_k17(gpu):
    """config 4 at full size: the ~800 Mbp C2 genome at k = 17 (16 GiB table in HBM), the whole table byte for byte against
    the oracle's, with the byte-counter wrap -> recount path (k_bucket_count_bytes) known to have run.  The oracle here
    is the all-cores twin (pinned to the scalar one in tests/test_oracle_golden.py), the scalar one if it declines."""
    import os
    data = inputs.make_input({"gen": "c2"})
    k = 17
    want = oracle.count_fasta_mt(data, k, len(os.sched_getaffinity(0))) or oracle.count_fasta(data, k)
    with gpu.Indexer(k) as ix:
        ix.feed(data)
        fin = ix.finish()
        t = ix.timings()
        assert t["buckets_recounted"] > 0, "no bucket took the wrap -> recount path: the test no longer covers it"
        assert fin["num_kmers"] == want["num_kmers"] and fin["total_bp"] == want["total_bp"] == 800_000_000
        wt = want["table"]
        hist = np.zeros(256, dtype=np.uint64)
        step = 1 << 30
        got = np.empty(step, dtype=np.uint8)
        for off in range(0, 4 ** k, step):                                # 16 slices of 1 GiB: compare, tally, move on
            ix.table_slice_to_host(got, off)
            assert np.array_equal(got, wt[off:off + step]), f"table differs in [{off}, {off + step})"
            hist += np.bincount(got, minlength=256).astype(np.uint64)
        assert np.array_equal(fin["hist256"], hist)                       # Header.update_stats input (tools.py:246-263)
        assert int(hist[255]) > 0 and int(wt[0]) == 255                   # saturated addresses exist (indexer.py:239,262)
        print(f"k=17 full size: {t['buckets_recounted']} buckets recounted, {t['relayouts']} relayouts, "
              f"{int(hist[1:].sum())} distinct k-mers")


def test_device_feed_larger_than_one_piece(gpu):
    """pk_indexer_feed_device cuts a buffer into pieces (1 GiB for 32-bit k-mers: record positions stay below 2^31 there) at
    arbitrary bytes -- mid-line, mid-k-mer.  1.6 GB in HBM (the C2 genome twice, as two records sets in one stream) in ONE call
    must give what the same bytes give fed host-side in 256 MiB pieces, which in turn is checked against the reference's
    golden by test_full_config2_k15 (a table with every count doubled, saturating)."""
    import torch
    data = np.frombuffer(inputs.make_input({"gen": "c2"}), dtype=np.uint8)
    both = np.concatenate([data, data])
    assert both.size > (1 << 30) + (1 << 29)
    d = torch.empty(both.size + 64, dtype=torch.uint8, device="cuda")
    d[: both.size].copy_(torch.from_numpy(both))
    torch.cuda.synchronize()
    with gpu.Indexer(15) as ix:
        ix.feed_device(d.data_ptr(), int(both.size))
        one = ix.finish()
        assert ix.timings()["feeds"] == 2                                # cut once, at 1 GiB
        t_one = ix.table_to_host()
    del d
    torch.cuda.empty_cache()
    with gpu.Indexer(15) as ix:
        ix.feed(both)
        two = ix.finish()
        t_two = ix.table_to_host()
    assert one["num_kmers"] == two["num_kmers"] and one["total_bp"] == two["total_bp"] == 1_600_000_000
    assert np.array_equal(one["hist256"], two["hist256"])
    assert np.array_equal(t_one, t_two)
    single = gpu.count_fasta(data, 15)
    assert one["num_kmers"] == 2 * single["num_kmers"]
    assert np.array_equal(t_one, np.minimum(2 * single["table"].astype(np.uint16), 255).astype(np.uint8))


def test_full_config2_k15(gpu, manifest):
    """config 2: the ~800 Mbp synthetic genome at k=15, bit-exact against the reference's own run when
    that golden exists (G6, ~1 h of reference time), and through size-independent properties always."""
    data = inputs.make_input({"gen": "c2"})
    got = gpu.count_fasta(data, 15)
    h = got["hist256"]
    v = np.arange(256, dtype=np.uint64)
    assert int(h.sum()) == 4 ** 15
    assert got["num_kmers"] == int(got["records"]["n_valid_kmers"].sum())
    assert got["total_bp"] == int(got["records"]["seq_len"].sum()) == 800_000_000
    assert int((h * v).sum()) <= got["num_kmers"]                       # saturation only ever loses counts
    assert int((h[:255] * v[:255]).sum()) + 255 * int(h[255]) == int((h * v).sum())
    tab = got["table"]
    assert int(np.count_nonzero(tab[: 1 << 24])) == int(np.count_nonzero(tab[: 1 << 24] > 0))
    assert int(tab[0]) == 255                                           # poly-A / poly-T runs saturate address 0
    # a second, independent path to the same table: stream the same bytes in uneven pieces
    with gpu.Indexer(15) as ix:
        cuts = [0, 1, 17, 4097, 1 << 20, (1 << 28) + 5, len(data)]
        for a, b in zip(cuts[:-1], cuts[1:]):
            ix.feed(data[a:b])
        fin = ix.finish()
        assert fin["num_kmers"] == got["num_kmers"] and np.array_equal(fin["hist256"], h)
        assert np.array_equal(ix.table_to_host(), tab)
    case = manifest["indexer"].get("G6_c2_800M_k15")
    if case is not None:
        assert inputs.sha256(data) == case["input_sha256"]
        _expect_fields(got, data, case)


def test_streaming_feed_equals_one_shot(gpu):
    """Chunks may split lines, headers, records and k-mers anywhere (pk_indexer_feed)."""
    data = inputs.edge_fasta()
    for k in (7, 15):
        want = oracle.count_fasta(data, k)
        rng = np.random.default_rng(5)
        for trial in range(3):
            with gpu.Indexer(k) as ix:
                if trial == 0:
                    cuts = list(range(0, 600)) + [len(data)]            # one byte at a time through the preamble
                else:
                    cuts = sorted(set([0, len(data)] + rng.integers(0, len(data), size=40).tolist()))
                for a, b in zip(cuts[:-1], cuts[1:]):
                    ix.feed(data[a:b])
                fin = ix.finish()
                recs = ix.records(fin["n_records"])
                assert fin["num_kmers"] == want["num_kmers"] and fin["total_bp"] == want["total_bp"]
                for f in ("name_off", "name_len", "seq_len", "n_valid_kmers"):
                    assert np.array_equal(recs[f], want["records"][f]), (k, trial, f)
                assert np.array_equal(ix.table_to_host(), want["table"])
                assert np.array_equal(fin["hist256"][1:], oracle.table_stats(want["table"])[0]), (k, trial)


@pytest.mark.parametrize("k", [9, 11])
def test_histogram_kept_across_dense_and_sparse_feeds(gpu, k):
    """The value histogram is maintained feed by feed (k_bucket_count): buckets that receive many records
    and buckets that receive few take different tally paths, on a fresh table and on top of earlier
    feeds; saturated counters must stop moving between bins."""
    import synth
    dense, _ = synth.generate(11, 3_000_000, 3, pm_dup=200, pm_tandem=100)
    sparse, _ = synth.generate(12, 1_500, 1)
    again, _ = synth.generate(13, 1_000_000, 2, pm_tandem=300)
    hot = np.frombuffer(b">hot\n" + b"ACGTTGCAAC" * 4000 + b"\n", dtype=np.uint8)       # pushes a few addresses past 255
    pieces = [sparse, dense, hot, sparse, again, hot]
    whole = np.concatenate(pieces)
    want = oracle.count_fasta(whole, k)
    for upto in range(1, len(pieces) + 1):                              # every prefix: the last feed lands on a different history
        part = oracle.count_fasta(np.concatenate(pieces[:upto]), k)
        with gpu.Indexer(k) as ix:
            for piece in pieces[:upto]:
                ix.feed(piece)
            fin = ix.finish()
            assert np.array_equal(fin["hist256"][1:], oracle.table_stats(part["table"])[0]), upto
            assert int(fin["hist256"].sum()) == 4 ** k
            if upto == len(pieces):
                assert np.array_equal(ix.table_to_host(), want["table"])
                assert fin["num_kmers"] == want["num_kmers"]


@pytest.mark.parametrize("k,n_bp", [(13, 10_000_000)])
def test_dense_feeds_use_half_size_buckets(gpu, k, n_bp):
    """A feed with >= 1 byte per 8 table addresses takes 2^15-address final buckets (two bucket-count workgroups per
    CU, make_part_plan); smaller feeds on the same table take the 2^16 ones.  Fresh and on top of earlier feeds."""
    import synth
    dense, _ = synth.generate(21, n_bp, 3, pm_dup=150, pm_tandem=150)
    small, _ = synth.generate(22, 40_000, 2, pm_tandem=200)
    for order in ([dense], [small, dense, small], [dense, dense]):
        whole = np.concatenate(order)
        want = oracle.count_fasta(whole, k)
        with gpu.Indexer(k) as ix:
            for piece in order:
                ix.feed(piece)
            fin = ix.finish()
            assert fin["num_kmers"] == want["num_kmers"]
            assert np.array_equal(ix.table_to_host(), want["table"])
            assert np.array_equal(fin["hist256"][1:], oracle.table_stats(want["table"])[0])
    # the same through an address slice: 13 bits of buckets over a 2^24-address range
    with gpu.Indexer(k, slice_index=2, n_slices=4) as ix:
        ix.feed(dense)
        ix.finish()
        quarter = 4 ** k // 4
        assert np.array_equal(ix.table_to_host(), oracle.count_fasta(dense, k)["table"][2 * quarter: 3 * quarter])


@pytest.mark.parametrize("k", [5, 15])
def test_many_runs_of_invalid_characters_per_piece(gpu, k):
    """The structure pass pushes a piece's valid bases together by deleting RUNS of other characters (classify_piece):
    pieces with one run, with a dozen, with more non-bases than the byte-by-byte path takes (> 6: four at a time),
    at every density, wrapped at odd widths and unwrapped, lower case mixed in."""
    rng = np.random.default_rng(77)
    parts = []
    for i, p_bad in enumerate((0.01, 0.05, 0.2, 0.5, 0.9)):
        n = 20_000
        seq = rng.choice(np.frombuffer(b"ACGTacgt", dtype=np.uint8), size=n)
        bad = rng.random(n) < p_bad
        seq[bad] = rng.choice(np.frombuffer(b"NnRYKMSWBDHV-*.", dtype=np.uint8), size=int(bad.sum()))
        for a in range(0, n, 997):                                              # a few longer gaps
            seq[a: a + int(rng.integers(1, 90))] = ord("N")
        width = (61, 7, 64, 0, 130)[i]
        body = seq.tobytes() if width == 0 else b"\n".join(seq[j: j + width].tobytes() for j in range(0, n, width))
        parts.append(b">r%d\n" % i + body + b"\n")
    _check_against_oracle(gpu, np.frombuffer(b"".join(parts), dtype=np.uint8), k)


def test_control_bytes_inside_sequence_lines(gpu):
    """Bytes below 0x21 other than \\n / \\r (NUL, \\x01, tab, VT, FS..US, space) and DEL inside sequence
    text: blanks are stripped at line ends and map to None inside, the rest are plain non-bases.  Such
    pieces must leave the clean-piece fast path, also in the last, partly filled chunk."""
    rng = np.random.default_rng(77)
    body = rng.choice(np.frombuffer(b"ACGTacgt", dtype=np.uint8), size=70_000)
    odd = np.frombuffer(b"\x00\x01\x08\x09\x0b\x0c\x0e\x1b\x1c\x1f \x7f", dtype=np.uint8)
    at = rng.choice(body.size, size=600, replace=False)
    body[at] = rng.choice(odd, size=at.size)
    body[-40:] = np.frombuffer(b"ACGTACGTAC\x01GTACGTACGT\x00ACGTACGTACGTACG\x1fTA", dtype=np.uint8)   # tail piece
    lines = [b">ctl one"]
    for i in range(0, body.size, 61):
        lines.append(body[i:i + 61].tobytes())
    data = np.frombuffer(b"\n".join(lines), dtype=np.uint8)           # no trailing newline
    for k in (5, 9, 15):
        _check_against_oracle(gpu, data, k)


def test_indexer_reuse_after_reset(gpu):
    """One indexer, several genomes in a row (bench.py and a multi-sample run do this): nothing of an earlier
    genome -- table bytes, histogram, record tallies, partition cursors -- may leak into the next one, whether
    the next input is larger, much smaller, or empty."""
    import synth
    big, _ = synth.c2(6_000_000, seed=21)
    small, _ = synth.c1(200_000, seed=22)
    tiny = np.frombuffer(b">t\nACGTACGTTTGACCA\n", dtype=np.uint8)
    for k in (9, 15):
        with gpu.Indexer(k) as ix:
            for data in (big, small, tiny, np.zeros(0, dtype=np.uint8), big, tiny):
                ix.reset()
                if data.size:
                    ix.feed(data)
                fin = ix.finish()
                want = oracle.count_fasta(data, k)
                assert fin["num_kmers"] == want["num_kmers"] and fin["total_bp"] == want["total_bp"], (k, data.size)
                assert np.array_equal(ix.table_to_host(), want["table"]), (k, data.size)
                assert np.array_equal(fin["hist256"][1:], oracle.table_stats(want["table"])[0]), (k, data.size)
                assert int(fin["hist256"].sum()) == 4 ** k
                recs = ix.records(fin["n_records"])
                assert np.array_equal(recs["n_valid_kmers"], want["records"]["n_valid_kmers"])


def test_random_structure_fuzz(gpu):
    """Random byte soup over the FASTA-relevant alphabet: every parser state transition, every seam."""
    rng = np.random.default_rng(11)
    alphabet = np.frombuffer(b"ACGTacgtNn>> \t\r\n\n\n\x0b\x0cXR", dtype=np.uint8)
    for trial in range(6):
        n = int(rng.integers(1, 70000))
        w = rng.random(alphabet.size) ** 3
        data = alphabet[rng.choice(alphabet.size, size=n, p=w / w.sum())].tobytes()
        for k in (3, 9):
            _check_against_oracle(gpu, data, k)


def test_read_set_headers_fuzz(gpu):
    """Short records with every kind of header line: the squeeze pass takes pieces with header text by masks (several
    headers in one 64-byte piece, names with blanks / control bytes / trailing whitespace, CR LF, empty records, a '>'
    inside sequence text, text in front of the first header) and the rest byte by byte; both must agree with the oracle."""
    rng = np.random.default_rng(2024)
    names = [b"r", b"read_%d len=100", b"x y\tz ", b"n\x01\x02 ", b"t\x1c\x1d", b"", b" lead", b"tail   \t", b"q\x0b\x0c", b">>dbl", b"a" * 70, b"b" * 130]
    ends = [b"\n", b"\r\n", b"\n\n", b"\r"]
    for trial in range(5):
        parts = [b"ACGTTTGA\nAC\n"] if trial % 2 else []                   # text before the first header is dropped
        for i in range(int(rng.integers(200, 1500))):
            name = names[int(rng.integers(len(names)))]
            if b"%d" in name:
                name = name % i
            parts.append(b">" + name + ends[int(rng.integers(len(ends)))])
            shape = int(rng.integers(6))
            n = int(rng.integers(0, 4)) if shape == 0 else int(rng.integers(1, 400))
            body = bytearray(b"ACGT"[j] for j in rng.integers(0, 4, size=n))
            if shape == 1 and n > 4:
                body[int(rng.integers(n))] = ord("N")
            if shape == 2 and n > 4:
                body[int(rng.integers(1, n))] = ord(">")                     # not at a line start: maps to None
            width = int(rng.choice([0, 60, 61, 64, 7]))
            if width and n:
                body = b"\n".join(bytes(body[j:j + width]) for j in range(0, n, width))
            parts.append(bytes(body) + ends[int(rng.integers(len(ends)))])
        data = b"".join(parts)
        for k in ((3, 15) if trial < 2 else (9,)):
            _check_against_oracle(gpu, data, k)
        if trial in (1, 2):
            # the same bytes through the streaming interface, cut at random places (inside header lines, between CR and
            # LF, one-byte feeds): the carried parser state meets header pieces at every offset
            k = 7
            want = oracle.count_fasta(data, k)
            cuts = np.unique(np.concatenate([[0, len(data)], rng.integers(0, len(data), size=40), rng.integers(0, len(data), size=5) + 1]))
            cuts = cuts[cuts <= len(data)]
            with gpu.Indexer(k) as ix:
                for a, b in zip(cuts[:-1], cuts[1:]):
                    ix.feed(data[int(a):int(b)])
                fin = ix.finish()
                recs = ix.records(fin["n_records"])
                assert fin["num_kmers"] == want["num_kmers"] and fin["total_bp"] == want["total_bp"]
                assert np.array_equal(ix.table_to_host(), want["table"])
                for f in ("name_off", "name_len", "seq_len", "n_valid_kmers"):
                    assert np.array_equal(recs[f], want["records"][f]), f


def test_unwrapped_long_lines_and_empty(gpu):
    rng = np.random.default_rng(3)
    seq = "".join("ACGT"[i] for i in rng.integers(0, 4, size=300_000))
    data = (">one_line_record\n" + seq + "\n>second\n" + seq[:100_000]).encode()      # 300 kbp on one line
    _check_against_oracle(gpu, data, 11)
    for blob in (b"", b"\n\n", b">only_header", b"ACGTACGTACGT\n", b">x\n" + b"A" * 20):
        got = gpu.count_fasta(blob, 5)
        want = oracle.count_fasta(blob, 5)
        assert got["num_kmers"] == want["num_kmers"] and np.array_equal(got["table"], want["table"])
        assert len(got["records"]) == len(want["records"])


def test_table_stats(gpu):
    rng = np.random.default_rng(1)
    for n in (4, 64, 1000, 1 << 20):
        t = rng.integers(0, 256, size=n, dtype=np.uint8)
        t[rng.random(n) < 0.6] = 0
        h = gpu.table_stats(t)
        assert np.array_equal(h, np.bincount(t, minlength=256).astype(np.uint64))


def test_rejects_bad_k(gpu):
    for k in (0, -3, 4, 16):                                  # tools.py:165-167
        with pytest.raises(ValueError):
            gpu.count_fasta(b">a\nACGT\n", k)
    with pytest.raises(ValueError):
        gpu.count_fasta(b">a\nACGT\n", 19)                    # beyond the device path (256 GiB table)


def test_structure_across_chunk_boundaries(gpu):
    """Headers, CR/LF pairs, blank runs and tandem repeats placed right on the 64-byte piece and 16 KiB
    chunk seams of the kernels, a header longer than a chunk, and many tiny records."""
    rng = np.random.default_rng(77)

    def seq(n):
        return bytes(b"ACGT"[i] for i in rng.integers(0, 4, size=n))

    parts = [b">first\n"]
    pos = len(parts[0])
    for target in (64, 128, 16384, 16384 + 64, 2 * 16384, 3 * 16384 - 1, 3 * 16384 + 1, 5 * 16384):
        for delta in (-2, -1, 0, 1, 2):
            want = target * (1 + len(parts) // 7) + delta           # keep moving forward
            if want <= pos + 8:
                continue
            body = seq(want - pos - 1) + b"\n"
            parts.append(body); pos += len(body)
            hdr = b">rec_at_%d extra words\r\n" % pos                # the '>' lands on / next to the seam
            parts.append(hdr); pos += len(hdr)
    long_header = b">" + b"H" * 40000 + b" tail \n"                  # header text spanning three chunks
    parts.append(seq(100) + b"\n" + long_header + seq(5000) + b"\n")
    parts.append(b">blanks\n" + seq(30) + b"   \t  " * 3000 + seq(30) + b"\n" + seq(20) + b" " * 20000 + b"\n" + seq(40) + b"\n")
    parts.append(b">tandem\n" + b"A" * 70000 + b"\n" + b"AT" * 20000 + b"\n" + b"AAG" * 9000 + b"\n" + b"ACGT" * 5000 + b"\n")
    parts.append(b"".join(b">t%d\nACGTTGCA%s\n" % (i, b"ACGT"[i % 4:i % 4 + 1] * (i % 9)) for i in range(5000)))
    parts.append(b">crlf\r\n" + b"\r\n".join(seq(70) for _ in range(600)) + b"\r\n>last_no_newline\n" + seq(333))
    data = b"".join(parts)
    assert len(data) > 10 * 16384
    for k in (5, 15):
        _check_against_oracle(gpu, data, k)
    # the same bytes through the streaming interface with cuts on and around the seams
    want = oracle.count_fasta(data, 11)
    cuts = sorted(set([0, len(data)] + [c for b in (64, 16384, 40000, 65536) for c in (b - 1, b, b + 1, 3 * b, 3 * b + 1) if c < len(data)]))
    with gpu.Indexer(11) as ix:
        for a, b in zip(cuts[:-1], cuts[1:]):
            ix.feed(data[a:b])
        fin = ix.finish()
        recs = ix.records(fin["n_records"])
        assert fin["num_kmers"] == want["num_kmers"] and np.array_equal(ix.table_to_host(), want["table"])
        for f in ("name_off", "name_len", "seq_len", "n_valid_kmers"):
            assert np.array_equal(recs[f], want["records"][f]), f


def test_many_distinct_tandem_repeats(gpu):
    """Thousands of short tandem runs with different motifs: overflows the per-workgroup hot-key table
    (direct side-list appends, mid-kernel flushes) and the re-aggregation table of k_apply_side."""
    rng = np.random.default_rng(2024)
    parts = [b">short_tandems\n"]
    line = []
    for _ in range(60000):
        period = int(rng.integers(1, 4))
        motif = bytes(b"ACGT"[i] for i in rng.integers(0, 4, size=period))
        run = motif * int(rng.integers(12, 30))
        spacer = bytes(b"ACGT"[i] for i in rng.integers(0, 4, size=int(rng.integers(0, 6))))
        line.append(run + spacer)
        if len(line) == 3:
            parts.append(b"".join(line) + b"\n")
            line = []
    data = b"".join(parts)
    assert len(data) > 2_000_000
    for k in (9, 15):
        _check_against_oracle(gpu, data, k)


@pytest.mark.parametrize("k,unit_len", [(13, 61), (15, 61), (15, 997), (17, 61)])
def test_bucket_overflow_takes_the_exact_relayout(gpu, k, unit_len):
    """Bucket rooms come from a sample (one wave's stretch out of every 16 of a feed of >= 1024 chunks; k = 17: + a sample
    of the level-1 records).  A text whose sampled chunks look nothing like the rest must overflow them: the overflow flag makes every
    later kernel of the feed return untouched and the host repeats the passes with exact sizes (`relayouts`; for k = 17
    that is the counting pass k_count2 / k_rows2_scan).  Second feed on the same indexer: the same against a table that
    is no longer fresh."""
    stretch = 1024 if k == 17 else 2048
    data = inputs.skewed_fasta(20_000_040, unit_len, seed=7 + unit_len, stretch=stretch)
    assert len(data) >= 1024 * 16384
    more = inputs.skewed_fasta(18_000_000, unit_len + 2, seed=11, stretch=stretch)
    kmers = np.concatenate([oracle.kmer_list(data, k), oracle.kmer_list(more, k)])
    u, c = np.unique(kmers, return_counts=True)
    sat = np.minimum(c, 255).astype(np.uint8)
    with gpu.Indexer(k) as ix:
        ix.feed(data)
        first = ix.timings()["relayouts"]
        assert first >= 1, "the skewed text was meant to overflow the sampled layout"
        ix.feed(more)
        assert ix.timings()["relayouts"] > first
        fin = ix.finish()
        assert fin["num_kmers"] == kmers.size
        h = fin["hist256"]
        assert int(h.sum()) == 4 ** k
        assert np.array_equal(h[1:], np.bincount(sat, minlength=256)[1:].astype(np.uint64))
        table = ix.table_to_host()
        assert np.array_equal(table[u.astype(np.int64)], sat)
        step = 1 << 30
        assert sum(int(np.count_nonzero(table[o:o + step])) for o in range(0, table.size, step)) == u.size


def _interspersed_repeat(n_copies, unit_len, spacer, seed):
    """`n_copies` of one unit, each followed by `spacer` random bases: every k-mer of the unit occurs n_copies times, but
    never with a period of 1-3 bases (the hot-key path does not see it) and never as a long run of records in one bucket."""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    unit = acgt[rng.integers(0, 4, size=unit_len)]
    parts = []
    for _ in range(n_copies):
        parts.append(unit)
        parts.append(acgt[rng.integers(0, 4, size=spacer)])
    seq = np.concatenate(parts)
    seq = seq[: seq.size // 60 * 60]
    lines = np.empty((seq.size // 60, 61), dtype=np.uint8)
    lines[:, :60] = seq.reshape(-1, 60)
    lines[:, 60] = 10
    return b">interspersed\n" + lines.tobytes()


@pytest.mark.parametrize("k", [9, 17])
def test_byte_counters_wrap_and_are_recounted(gpu, k):
    """Sparse tables count in byte counters (k_bucket_count_bytes).  A k-mer that occurs more than 254 times in a bucket
    visit wraps its byte; the add that sees 255 raises the flag and the bucket is counted again with 16-bit counters
    (`buckets_recounted`).  Second feed of the same text: the bytes come back from HBM already at 255."""
    copies = 300 if k == 9 else 400
    data = _interspersed_repeat(copies, 70, 20 if k == 9 else 5000, seed=k)
    assert k != 9 or len(data) < 4 * 8192                                  # k = 9: four final buckets, still "sparse"
    kmers = oracle.kmer_list(data, k)
    u, c = np.unique(kmers, return_counts=True)
    assert (c >= copies).sum() >= 70 - k + 1
    with gpu.Indexer(k) as ix:
        for feed in (1, 2):
            ix.feed(data)
            assert ix.timings()["buckets_recounted"] >= feed
        fin = ix.finish()
        sat = np.minimum(2 * c, 255).astype(np.uint8)
        assert fin["num_kmers"] == 2 * kmers.size
        assert np.array_equal(fin["hist256"][1:], np.bincount(sat, minlength=256)[1:].astype(np.uint64))
        table = ix.table_to_host()
        assert np.array_equal(table[u.astype(np.int64)], sat)
        step = 1 << 30
        assert sum(int(np.count_nonzero(table[o:o + step])) for o in range(0, table.size, step)) == u.size
