"""CPU: the host-side mirror of the reference interface (pykmer_amd.header / indexer / merger) and the
C-ABI surface.  No compute call touches a GPU here; scans are stood in for by the oracle where a
number is needed to exercise the file-format code."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

import inputs
import oracle
from pykmer_amd import _lib, merger
from pykmer_amd.header import Header, HeaderVars, Timer, gen_checksum, stats_from_hist256
from pykmer_amd.tools import Header as ToolsHeader

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------ C-ABI surface ---------------
def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "pykmer_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pk_[a-z0-9_]+)\s*\(", text)))


def _exported_symbols():
    """The dynamic symbol table of the built library (nm -D): every defined pk_* function."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    return sorted({line.split()[-1] for line in out.splitlines() if line.split()[-1].startswith("pk_") and line.split()[-2] in "TW"})


def test_library_exports_exactly_the_declared_symbols():
    lib = _lib.load()                                   # raises if the .so was not built
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/pykmer_hip.h but not exported"
    assert sorted(_lib.EXPORTS) == declared, "ctypes signatures out of sync with the header"
    assert _exported_symbols() == declared, "the library exports a pk_* symbol that include/pykmer_hip.h does not declare (or the reverse)"
    assert lib.pk_version() == 3


def test_feed_pieces_keep_record_positions_in_32_bits():
    """Record positions inside one feed piece are 32-bit (bucket starts, cursors, limits): the largest piece the library
    cuts a feed into must leave both bucket areas + the dump tile below 2^32 for every k -- k = 17 with its 2^18 final
    buckets of fixed slack is the tight one -- and a plan that does not fit is refused, not wrapped."""
    for k in (3, 9, 13, 15, 17, 19, 21):
        pl = _lib.diag_plan(k)                          # n_bytes = 0: the largest piece
        assert pl["feed_max"] % 16 == 0
        assert pl["fits_u32"] == 1, (k, pl)
        # 32-bit k-mers store through 32-bit BYTE offsets (2-byte records): their positions stay below 2^31
        assert max(pl["capacity1"], pl["capacity2"]) + 16384 + 64 < (2 ** 31 if k <= 15 else 2 ** 32), (k, pl)
    assert _lib.diag_plan(15, 2 << 30)["fits_u32"] == 0
    assert _lib.diag_plan(17, 3 << 30)["fits_u32"] == 0                            # the old 3 GiB piece did not fit at k = 17


def test_numpy_free_loader_serves_the_same_library():
    """pykmer_amd/_rt.py (ctypes only) is what the CLIs warm the device with before numpy is imported: it must open the
    very library the binding uses, and importing it must not pull numpy in."""
    import subprocess
    import sys
    from pykmer_amd import _rt
    assert _rt.LIB_PATH == _lib.LIB_PATH
    assert _rt.open_library() is _rt.open_library()
    assert hasattr(_rt.open_library(), "pk_warm")
    code = "import sys; sys.path.insert(0, %r); from pykmer_amd import _rt; assert 'numpy' not in sys.modules; print(_rt.LIB_PATH)" % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip() == _lib.LIB_PATH


def test_argument_errors_need_no_gpu():
    lib = _lib.load()
    buf = ctypes.create_string_buffer(256)
    for k in (0, -1, 2, 16, 19, 33):                    # tools.py:165-167 + device limit
        rc = lib.pk_count_fasta(None, 0, k, None, None, None, None, None, 0, None, 0)
        assert rc == _lib.PK_ERR_ARG
        lib.pk_last_error(buf, 256)
        assert b"kmer_len" in buf.value
    t = np.zeros(64, np.uint8)
    ptrs = (ctypes.c_void_p * 2)(t.ctypes.data, t.ctypes.data)
    m = np.zeros((2, 2, 3), np.uint64)
    for mn, mx in ((0, 255), (1, 256)):                 # merger.py:90-91
        assert lib.pk_gram(ptrs, 2, 64, mn, mx, m.ctypes.data, None, 0) == _lib.PK_ERR_ARG
    with pytest.raises(ValueError):
        _lib.count_fasta(b">a\nACGT\n", 4)


def test_gram_expand_layout():
    """merger.py:175-176: [k][l] = (total_k, total_l, shared); diagonal zero (never assigned, merger.py:136)."""
    rng = np.random.default_rng(0)
    tables = [rng.integers(0, 3, 256, dtype=np.uint8) for _ in range(5)]
    want = oracle.gram(tables)
    pair = np.zeros((5, 5), np.uint64)
    for i in range(5):
        pair[i, i] = want[i, (i + 1) % 5, 0]
        for j in range(i + 1, 5):
            pair[i, j] = want[i, j, 2]
    assert np.array_equal(_lib.gram_expand(pair), want)


# ------------------------------------------------------------------ Header ----------------------
def test_sizes_names_and_frag_size(tmp_path):
    # values the reference's own Header computes, k=3 exceeds its 64-byte table on purpose
    expect = {3: 1000, 7: 9000, 11: 2_098_000, 13: 33_555_000, 15: 357_914_000, 17: 954_438_000, 19: 999_557_000}
    for k, fs in expect.items():
        h = Header("proj", input_file=str(tmp_path / "g.fa"), kmer_len=k)
        assert h.frag_size == fs
        assert h.kmer_size == h.data_size == h.max_size == 4 ** k
        assert h.index_file_root == f"{tmp_path}/g.fa.{k:02d}.kin"
        assert h.index_tmp_file == h.index_file_root + ".tmp" and h.metadata_file == h.index_file_root + ".json"
    assert Header("p", input_file="x.fa", kmer_len=15, frag_size=123).frag_size == 123
    assert h.file_ver == "KMER001" and h.max_val == 255
    assert ToolsHeader is Header and HeaderVars.DEFAULT_FLUSH_EVERY == 100_000_000


def test_rejects_even_or_missing_k():
    for k in (0, 2, 14, None):                          # tools.py:165-167 asserts
        with pytest.raises(AssertionError):
            Header("p", input_file="x.fa", kmer_len=k)


def test_bgz_is_preferred_when_present(tmp_path):
    h = Header("p", input_file=str(tmp_path / "s.fa"), kmer_len=7)
    assert h.index_file == h.index_file_root
    open(h.index_file_root + ".bgz", "wb").close()
    assert h.index_file == h.index_file_root + ".bgz" and h.index_file_basename == "s.fa.07.kin.bgz"


def test_stats_from_hist256_is_np_histogram():
    """tools.py:250: np.histogram(arr, bins=255, range=(1,255)) == bincount[1:] for every byte value."""
    rng = np.random.default_rng(4)
    t = rng.integers(0, 256, 100_000, dtype=np.uint8)
    t[:500] = 255
    t[500:900] = 254
    ref_hist, _ = np.histogram(t, bins=255, range=(1, 255))
    st = stats_from_hist256(np.bincount(t, minlength=256))
    assert st["hist"] == ref_hist.tolist()
    assert st["hist_sum"] == int(ref_hist.sum()) and st["hist_count"] == int(np.count_nonzero(ref_hist))
    assert st["hist_min"] == int(ref_hist.min()) and st["hist_max"] == int(ref_hist.max())
    assert st["vals_sum"] == int(t.sum(dtype=np.uint64)) and st["vals_count"] == int(np.count_nonzero(t))
    assert st["vals_min"] == int(t.min()) and st["vals_max"] == int(t.max())
    z = stats_from_hist256(np.bincount(np.array([2, 2, 9], np.uint8), minlength=256))
    assert z["vals_min"] == 2 and z["vals_max"] == 9 and z["hist_min"] == 0


def _write_index(tmp_path, name, data, k):
    """Writes <name>.<kk>.kin + .json the way pykmer_amd.indexer does, with the oracle standing in for the GPU."""
    fa = tmp_path / name
    fa.write_bytes(data)
    got = oracle.count_fasta(data, k)
    h = Header(str(fa), sample_name="s", input_file=str(fa), kmer_len=k)
    h._init_clean(overwrite=True)
    h.timer.update(got["total_bp"])
    h.num_kmers = got["num_kmers"]
    h.chromosomes = oracle.chromosomes(data, got["records"])
    got["table"].tofile(h.index_tmp_file)
    h.write_metadata_index_tmp_file(hist256=np.bincount(got["table"], minlength=256))
    os.rename(h.index_tmp_file, h.index_file_root)
    return h, got


def test_kin_json_schema_and_values_match_reference(tmp_path, manifest):
    """Every key of the reference's .kin.json, deterministic fields equal to what the reference wrote (G3)."""
    case = manifest["indexer"]["G3_edge_k7"]
    h, got = _write_index(tmp_path, case["input_file"], inputs.make_input(case["input"]), 7)
    with open(h.metadata_file) as fh:
        text = fh.read()
    meta = json.loads(text)
    assert sorted(meta.keys()) == case["reference_keys"] and len(meta) == 33
    for f, v in case["expect"].items():
        assert meta[f] == v, f
    assert text == json.dumps(meta, indent=1, sort_keys=True)            # tools.py:378 formatting
    for f, typ in (("creation_speed", int), ("hostname", str), ("checksum_script", str), ("input_file_ctime", float),
                   ("creation_duration", str), ("creation_time_start", str), ("input_file_path", str), ("project_name", str)):
        assert isinstance(meta[f], typ), f
    assert os.path.getsize(h.index_file_root) == 4 ** 7 and not os.path.exists(h.index_tmp_file)
    assert meta["input_file_cheksum"] == gen_checksum(str(tmp_path / case["input_file"]))

    # consumer contract: Header(index_file=...) parses the name, reads the JSON, checks the fixed keys
    r = Header(h.index_file_root, index_file=h.index_file_root)
    assert r.kmer_len == 7 and r.num_kmers == case["expect"]["num_kmers"] and r.input_file_name == case["input_file"]
    lean = r.to_dict(lean=True)
    assert "chromosomes" not in lean and len(lean) == 32
    bad = dict(meta, data_size=5)
    with open(h.metadata_file, "w") as fh:
        json.dump(bad, fh)
    with pytest.raises(AssertionError):
        Header(h.index_file_root, index_file=h.index_file_root)
    del bad["hist"]
    with open(h.metadata_file, "w") as fh:
        json.dump(bad, fh)
    with pytest.raises(KeyError):
        Header(h.index_file_root, index_file=h.index_file_root)


def test_empty_input_refused_like_reference(tmp_path):
    fa = tmp_path / "n.fa"
    fa.write_bytes(b">only_n\nNNNNNNNNNNNN\n")
    h = Header(str(fa), input_file=str(fa), kmer_len=7)
    h.num_kmers, h.chromosomes = 0, []
    with pytest.raises(AssertionError):                  # tools.py:367-368
        h.write_metadata_file(str(fa), hist256=np.zeros(256, np.uint64))


def test_timer():
    t = Timer()
    t.update(1000)
    assert t.val_last == 1000 and t.val_delta == 1000 and t.speed_ela > 0 and "val" in str(t)


# ------------------------------------------------------------------ merger host logic -----------
def _oracle_partial(headers, lo, hi, windows, device, threads):
    tabs = [h.read_table_slice(lo, hi) for h in headers]      # a rank reads its slice of every file, nothing else
    N = len(tabs)
    parts = []
    for min_count, max_count in windows:
        m = oracle.gram(tabs, min_count, max_count)
        pair = np.zeros((N, N), np.uint64)
        for i in range(N):
            pair[i, i] = np.count_nonzero((tabs[i] >= min_count) & (tabs[i] <= max_count))
            for j in range(i + 1, N):
                pair[i, j] = m[i, j, 2]
        parts.append(pair)
    return parts


def _family_indexes(tmp_path, manifest, n=13):
    case = manifest["merger"]["G7_k7_n13_default"]
    paths = []
    for i, spec in enumerate(case["inputs"][:n]):
        h, _ = _write_index(tmp_path, f"s{i:02d}.fa", inputs.make_input(spec), case["k"])
        paths.append(h.index_file_root)
    return paths


def test_merge_writes_kma_like_reference(tmp_path, manifest):
    import gzip
    paths = _family_indexes(tmp_path, manifest)
    with open(paths[4], "rb") as fh, gzip.open(paths[4] + ".bgz", "wb") as out:      # one input as .kin.bgz (tools.py:300-302)
        out.write(fh.read())
    case = manifest["merger"]["G7_k7_n13_min2max5"]
    proj = str(tmp_path / "proj")
    data, matrix = merger.merge(proj, sorted(paths), min_count=2, max_count=5, partial_fn=_oracle_partial, devices=(0, 1, 2))
    want = np.array(case["matrix"], dtype=np.uint64)
    assert np.array_equal(matrix, want)
    kma = np.load(proj + ".002-005.kma")                                             # merger.py:96,207
    assert list(kma.keys()) == ["matrix"] and kma["matrix"].dtype == np.uint64 and np.array_equal(kma["matrix"], want)
    with open(proj + ".002-005.kma.json") as fh:
        meta = json.load(fh)
    assert sorted(meta.keys()) == case["kma_json_keys"]
    assert sorted(meta["data"][0].keys()) == case["kma_json_data0_keys"]
    assert sorted(meta["data"][0]["header"].keys()) == case["kma_json_header_keys"]
    assert [os.path.basename(d["index_file"]) for d in meta["data"]] == case["order"]
    assert meta["min_count"] == 2 and meta["max_count"] == 5 and isinstance(meta["data"][0]["index_file"], str)
    assert not os.path.exists(proj + ".002-005.kma.tmp")
    with pytest.raises(AssertionError):                                              # refuses to overwrite (merger.py:98-99)
        merger.merge(proj, sorted(paths), min_count=2, max_count=5, partial_fn=_oracle_partial)


def test_merge_sweep_writes_one_kma_per_window(tmp_path, manifest):
    """--min/--max sweeps in one run (tables staged once), each window equal to its own golden."""
    paths = sorted(_family_indexes(tmp_path, manifest))
    proj = str(tmp_path / "sweep")
    wins = merger.parse_sweep("1-255, 2-255,1-3,2-5")
    assert wins == [(1, 255), (2, 255), (1, 3), (2, 5)]
    calls = []

    def counting_partial(*a):
        calls.append(a[3])
        return _oracle_partial(*a)
    data, first = merger.merge(proj, paths, partial_fn=counting_partial, windows=wins)
    assert len(calls) == 1 and list(calls[0]) == wins                       # one staging pass for all windows
    for (mn, mx), tag in zip(wins, ("default", "min2", "max3", "min2max5")):
        want = np.array(manifest["merger"][f"G7_k7_n13_{tag}"]["matrix"], dtype=np.uint64)
        got = np.load(f"{proj}.{mn:03d}-{mx:03d}.kma")["matrix"]
        assert np.array_equal(got, want), tag
        meta = json.load(open(f"{proj}.{mn:03d}-{mx:03d}.kma.json"))
        assert meta["min_count"] == mn and meta["max_count"] == mx
    assert np.array_equal(first, np.array(manifest["merger"]["G7_k7_n13_default"]["matrix"], dtype=np.uint64))
    with pytest.raises(AssertionError):
        merger.merge(str(tmp_path / "bad"), paths, partial_fn=_oracle_partial, windows=[(0, 5)])


def test_merge_validation(tmp_path, manifest):
    paths = _family_indexes(tmp_path, manifest, n=2)
    for kw in ({"min_count": 0}, {"max_count": 256}, {"buffer_size": 0}, {"block_size": 0}):   # merger.py:90-93
        with pytest.raises(AssertionError):
            merger.merge(str(tmp_path / "p"), paths, partial_fn=_oracle_partial, **kw)
    h9, _ = _write_index(tmp_path, "other.fa", inputs.edge_fasta(), 9)
    with pytest.raises(AssertionError, match="kmer_length differs"):                 # merger.py:119-122
        merger.merge(str(tmp_path / "p"), paths + [h9.index_file_root], partial_fn=_oracle_partial)
    os.remove(paths[1] + ".json")
    with pytest.raises(AssertionError):                                              # merger.py:112-115
        merger.merge(str(tmp_path / "p"), paths, partial_fn=_oracle_partial)
    with pytest.raises(SystemExit):
        merger.main(["proj", paths[0]])                                              # argparse: Kmer_N needs one more (merger.py:53-54)
    args = merger.build_parser().parse_args(["p", "a.kin", "b.kin", "--min-count", "3", "--threads", "2"])
    assert args.min_count == 3 and args.max_count == 255 and args.block_size == 100_000_000 and args.threads == 2


def _cli_rank(rank, world, port, workdir, argv):
    """One rank of `merger.py` under a launcher's environment, scans stood in for by the oracle (no GPU in this suite)."""
    import sys
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      PK_DIST_BACKEND="gloo")
    from pykmer_amd import merger as m
    import test_host_layer as here
    m.gpu_partial = here._oracle_partial                  # the default partial_fn is looked up when pair_matrix runs
    import contextlib
    import io
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        m.main(argv)
    with open(os.path.join(workdir, f"stdout_rank{rank}.txt"), "w") as fh:
        fh.write(out.getvalue())


def test_cli_joins_the_process_group_of_its_launcher(tmp_path, manifest):
    """merger.main under WORLD_SIZE / RANK (what torchrun or `--gpus N` set): every rank validates and scans its address
    slice, one all-reduce (gloo here, RCCL on GPUs) sums the partials, rank 0 alone prints and writes both windows."""
    import socket
    import torch.multiprocessing as mp
    paths = sorted(_family_indexes(tmp_path, manifest))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    proj = str(tmp_path / "cli")
    argv = [proj] + list(reversed(paths)) + ["--sweep", "2-255,1-3", "--threads", "2"]
    mp.spawn(_cli_rank, args=(2, port, str(tmp_path), argv), nprocs=2, join=True)
    for (mn, mx), tag in (((2, 255), "min2"), ((1, 3), "max3")):
        want = np.array(manifest["merger"][f"G7_k7_n13_{tag}"]["matrix"], dtype=np.uint64)
        assert np.array_equal(np.load(f"{proj}.{mn:03d}-{mx:03d}.kma")["matrix"], want), tag
    out0, out1 = ((tmp_path / f"stdout_rank{r}.txt").read_text() for r in (0, 1))
    assert out0.count("saving") == 4 and "verifying" in out0 and out1 == ""


def test_cli_spawn_ranks_relays_failures(tmp_path, monkeypatch):
    """`--gpus N` outside a launcher: N child processes with RANK / WORLD_SIZE / MASTER_* set, the parent's exit code is
    the worst of theirs (exercised with a stand-in script: no GPU here)."""
    import sys
    script = tmp_path / "fake_rank.py"
    script.write_text("import os, sys\n"
                      "open(os.path.join(os.path.dirname(__file__), 'seen_' + os.environ['RANK']), 'w').write(' '.join(["
                      "os.environ['WORLD_SIZE'], os.environ['LOCAL_RANK'], os.environ['MASTER_ADDR']] + sys.argv[1:]))\n"
                      "sys.exit(3 if os.environ['RANK'] == '1' and 'fail' in sys.argv else 0)\n")
    assert merger.spawn_ranks(3, ["a", "b"], script=str(script)) == 0
    assert sorted(p.name for p in tmp_path.glob("seen_*")) == ["seen_0", "seen_1", "seen_2"]
    assert (tmp_path / "seen_2").read_text() == "3 2 127.0.0.1 a b"
    assert merger.spawn_ranks(2, ["fail"], script=str(script)) == 3


def test_address_slices_cover_range():
    for n in (4, 64, 4 ** 7, 4 ** 15, 1000):
        for world in (1, 2, 3, 8):
            cuts = [merger.address_slice(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(cuts[:-1], cuts[1:]))
            assert all(lo % 32 == 0 for lo, hi in cuts if hi > lo)
