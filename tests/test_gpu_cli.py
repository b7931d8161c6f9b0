"""GPU end to end: the two drop-in CLIs, files on disk, against what the reference wrote (goldens)."""
import gzip
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import inputs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, cwd):
    r = subprocess.run([sys.executable] + list(argv), cwd=cwd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


@pytest.mark.parametrize("name", ["G2_c1_k7", "G3_edge_k9", "G3_edge_gz_k7", "G3_edge_k15"])
def test_indexer_cli_writes_reference_files(gpu, tmp_path, manifest, small_tables, name):
    case = manifest["indexer"][name]
    fa = tmp_path / case["input_file"]
    data = inputs.make_input(case["input"])
    fa.write_bytes(data)
    _run(os.path.join(ROOT, "indexer.py"), str(fa), "sample", str(case["k"]), cwd=str(tmp_path))
    kin = f"{fa}.{case['k']:02d}.kin"
    assert os.path.getsize(kin) == 4 ** case["k"] and not os.path.exists(kin + ".tmp")
    with open(kin + ".json") as fh:
        meta = json.load(fh)
    assert sorted(meta.keys()) == case["reference_keys"]
    skip = {"input_file_cheksum", "input_file_size"} if name == "G3_edge_gz_k7" else set()   # gzip bytes depend on the zlib build
    for f, v in case["expect"].items():
        if f not in skip:
            assert meta[f] == v, f
    if name in small_tables:
        assert np.array_equal(np.fromfile(kin, dtype=np.uint8), small_tables[name])
    assert meta["project_name"] == str(fa) and meta["input_file_path"] == str(fa)


def test_indexer_cli_readme_form_and_refusal(gpu, tmp_path):
    fa = tmp_path / "r.fa"
    fa.write_bytes(inputs.edge_fasta())
    _run(os.path.join(ROOT, "indexer.py"), str(fa), "7", cwd=str(tmp_path))          # README.md:22: indexer.py <file> <K>
    assert os.path.exists(f"{fa}.07.kin.json")
    empty = tmp_path / "n.fa"
    empty.write_bytes(b">all_n\nNNNNNNNNNNNNNNNN\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "indexer.py"), str(empty), "s", "7"], capture_output=True, text=True)
    assert r.returncode != 0 and "AssertionError" in r.stderr                        # tools.py:367: no k-mers -> assert


def test_merger_cli_matches_reference_matrix(gpu, tmp_path, manifest):
    case = manifest["merger"]["G7_k7_n13_max3"]
    kins = []
    for i, spec in enumerate(case["inputs"]):
        fa = tmp_path / f"s{i:02d}.fa"
        fa.write_bytes(inputs.make_input(spec))
        _run(os.path.join(ROOT, "indexer.py"), str(fa), f"s{i}", "7", cwd=str(tmp_path))
        kins.append(f"{fa}.07.kin")
    with open(kins[4], "rb") as fh, gzip.open(kins[4] + ".bgz", "wb") as out:        # Header.index_file then prefers the .bgz
        out.write(fh.read())
    proj = str(tmp_path / "proj")
    _run(os.path.join(ROOT, "merger.py"), proj, *reversed(kins), "--max-count", "3", "--threads", "3", cwd=str(tmp_path))
    m = np.load(proj + ".001-003.kma")["matrix"]
    assert m.dtype == np.uint64 and np.array_equal(m, np.array(case["matrix"], dtype=np.uint64))
    with open(proj + ".001-003.kma.json") as fh:
        meta = json.load(fh)
    assert sorted(meta.keys()) == case["kma_json_keys"]
    assert sorted(meta["data"][0]["header"].keys()) == case["kma_json_header_keys"]
    assert [os.path.basename(d["index_file"]) for d in meta["data"]] == case["order"]  # sorted, whatever the argv order
    # a sweep stages the tables once and writes one .kma per window
    sw = str(tmp_path / "sweep")
    _run(os.path.join(ROOT, "merger.py"), sw, *kins, "--sweep", "1-3,2-255", cwd=str(tmp_path))
    assert np.array_equal(np.load(sw + ".001-003.kma")["matrix"], m)
    assert np.array_equal(np.load(sw + ".002-255.kma")["matrix"], np.array(manifest["merger"]["G7_k7_n13_min2"]["matrix"], dtype=np.uint64))
    # the pair API the reference's pool workers call (merger.py:62-78)
    from pykmer_amd import merger
    assert merger.calculate_distance(kins[0], kins[1], max_count=3) == tuple(int(x) for x in m[0, 1])
    # read-back validator (indexer.py:416-444, working here)
    from pykmer_amd import indexer
    indexer.read_fasta_index("p", index_file=kins[2])


def test_indexer_cli_counts_in_address_slices(gpu, tmp_path, manifest):
    """The k = 19 route of the CLI (one address slice of the table at a time, slices spread over the listed devices,
    the file filled and hashed in address order) exercised at a size with a reference golden: PK_SLICES forces four
    slices at k = 9, PK_DEVICES names the one GPU twice.  The files must be what the reference wrote."""
    case = manifest["indexer"]["G3_edge_k9"]
    fa = tmp_path / case["input_file"]
    fa.write_bytes(inputs.make_input(case["input"]))
    env = dict(os.environ, PK_SLICES="4", PK_DEVICES="0,0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "indexer.py"), str(fa), "sample", "9"], cwd=str(tmp_path), capture_output=True,
                       text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    kin = f"{fa}.09.kin"
    with open(kin + ".json") as fh:
        meta = json.load(fh)
    for f, v in case["expect"].items():
        assert meta[f] == v, f


def _family_kins(tmp_path, manifest, tag="G7_k7_n13_default"):
    case = manifest["merger"][tag]
    kins = []
    for i, spec in enumerate(case["inputs"]):
        fa = tmp_path / f"s{i:02d}.fa"
        fa.write_bytes(inputs.make_input(spec))
        _run(os.path.join(ROOT, "indexer.py"), str(fa), f"s{i}", "7", cwd=str(tmp_path))
        kins.append(f"{fa}.07.kin")
    return kins


def test_merger_cli_one_process_per_gpu(gpu, tmp_path, manifest):
    """merger.py:213-239 with the reference's pool (merger.py:137-178) replaced by ranks: `--gpus 2` starts two ranks (here both
    on the one GPU, so the rehearsal backend gloo: RCCL refuses two ranks on one device), each scans its half of the k-mer
    address range, one all-reduce sums the partials, rank 0 writes.  Then the RCCL form of the same entry: a launcher's
    environment (WORLD_SIZE=1, as torchrun sets it) makes the CLI join an nccl group."""
    kins = _family_kins(tmp_path, manifest)
    want = np.array(manifest["merger"]["G7_k7_n13_min2"]["matrix"], dtype=np.uint64)
    env = dict(os.environ, PK_DIST_BACKEND="gloo")
    for v in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(v, None)
    proj = str(tmp_path / "two")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "merger.py"), proj, *kins, "--min-count", "2", "--gpus", "2"], cwd=str(tmp_path),
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert np.array_equal(np.load(proj + ".002-255.kma")["matrix"], want)
    assert r.stdout.count("saving") == 2                                             # rank 0 alone prints and writes
    with open(proj + ".002-255.kma.json") as fh:
        assert [os.path.basename(d["index_file"]) for d in json.load(fh)["data"]] == manifest["merger"]["G7_k7_n13_min2"]["order"]
    # RCCL: one rank, joined through the launcher's environment
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29617")
    env.pop("PK_DIST_BACKEND", None)
    proj = str(tmp_path / "one")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "merger.py"), proj, *kins, "--sweep", "2-255,1-3"], cwd=str(tmp_path),
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert np.array_equal(np.load(proj + ".002-255.kma")["matrix"], want)
    assert np.array_equal(np.load(proj + ".001-003.kma")["matrix"], np.array(manifest["merger"]["G7_k7_n13_max3"]["matrix"], dtype=np.uint64))


@pytest.mark.parametrize("piece", [0, 3000])
def test_indexer_cli_on_bgzipped_fasta(gpu, tmp_path, manifest, small_tables, piece, monkeypatch):
    """indexer.py:112-115 opens .gz / .bgz inputs through gzip.open; here a BGZF FASTA (what `bgzip` writes) is inflated
    block-parallel in pieces that are fed while the next one inflates.  The reference's golden for the gzipped edge FASTA
    (G3_edge_gz_k7) must come out, also with pieces so small that headers, records and k-mers straddle them."""
    from pykmer_amd import bgzf, indexer
    case = manifest["indexer"]["G3_edge_gz_k7"]
    data = inputs.make_input(case["input"])
    plain = tmp_path / "edge.fa"
    plain.write_bytes(gzip.decompress(data) if data[:2] == b"\x1f\x8b" else data)
    fa = tmp_path / case["input_file"]
    if piece:
        monkeypatch.setattr(bgzf, "BLOCK_INPUT", 1000)                               # small blocks, so that pieces of 3 blocks exist
    bgzf.compress_file(str(plain), str(fa), index=False)
    assert bgzf.is_bgzf(str(fa)) and str(fa).endswith(".gz")
    if piece:
        assert len(list(bgzf.iter_pieces(str(fa), piece))) >= 5
        monkeypatch.setattr(indexer, "GZ_PIECE", piece)
        indexer.main([str(fa), "sample", "7"])
    else:
        _run(os.path.join(ROOT, "indexer.py"), str(fa), "sample", "7", cwd=str(tmp_path))
    kin = f"{fa}.07.kin"
    with open(kin + ".json") as fh:
        meta = json.load(fh)
    for f, v in case["expect"].items():
        if f not in ("input_file_cheksum", "input_file_size"):                       # the compressed bytes differ from python-gzip's
            assert meta[f] == v, f
    assert np.array_equal(np.fromfile(kin, dtype=np.uint8), small_tables["G3_edge_gz_k7"])
