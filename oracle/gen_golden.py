#!/usr/bin/env python3
"""gen_golden.py -- run the REFERENCE (/root/reference) on seeded inputs and commit its outputs
as fixtures under tests/golden/.  Runs only in the build container (the reference never travels).

The reference is imported unmodified, with the two harness-side shims SURVEY.md 8c documents:
  1. `import bgzip` (tools.py:17) names a module that is absent and unused -> empty stand-in module;
  2. indexer.create_fasta_index passes sample_name= to Header (indexer.py:311-320), which
     Header.__init__ (tools.py:111-121) does not accept -> a subclass that swallows it.
cwd is /root/reference because tools.py:285 hashes `tools.py` relative to cwd.

usage: python oracle/gen_golden.py [small|k15|k17|full|merge|all]
"""
import contextlib
import gzip
import hashlib
import io
import json
import os
import shutil
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import inputs  # noqa: E402

REF = "/root/reference"
SCRATCH = os.path.join(ROOT, ".scratch", "golden")
GOLDEN = inputs.GOLDEN

DETERMINISTIC = ["chromosomes", "data_size", "file_ver", "flush_every", "frag_size", "hist", "hist_count",
                 "hist_max", "hist_min", "hist_sum", "input_file_cheksum", "input_file_name", "input_file_size",
                 "kmer_len", "kmer_size", "max_size", "num_kmers", "output_file_cheksum", "output_file_size",
                 "vals_count", "vals_max", "vals_min", "vals_sum"]


def _import_reference():
    sys.dont_write_bytecode = True
    sys.modules.setdefault("bgzip", types.ModuleType("bgzip"))          # shim 1
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import tools, indexer, merger                                         # noqa: E401

    class H(tools.Header):                                                # shim 2
        def __init__(self, project_name, sample_name=None, **kw):
            self.sample_name = sample_name
            super().__init__(project_name, **kw)

    indexer.Header = H
    return tools, indexer, merger


def ref_index(fasta_path: str, k: int, flush_every=None):
    """Runs the reference indexer; returns (json dict, path of .kin)."""
    tools, indexer, _ = _import_reference()
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            if flush_every is None:
                sys.argv = ["indexer.py", fasta_path, "sample", str(k)]
                indexer.main()
            else:
                indexer.create_fasta_index(fasta_path, "sample", fasta_path, k, overwrite=True,
                                           flush_every=flush_every, buffer_size=2 ** 16)
    finally:
        os.chdir(cwd)
    kin = f"{os.path.abspath(fasta_path)}.{k:02d}.kin"
    with open(kin + ".json") as fh:
        meta = json.load(fh)
    return meta, kin


def ref_merge(project: str, kins, extra_args=()):
    """Runs the reference merger CLI; returns (matrix, kma.json dict)."""
    _, _, merger = _import_reference()
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            sys.argv = ["merger.py", project] + list(kins) + list(extra_args)
            merger.main()
    finally:
        os.chdir(cwd)
    args = list(extra_args)
    mn = int(args[args.index("--min-count") + 1]) if "--min-count" in args else 1
    mx = int(args[args.index("--max-count") + 1]) if "--max-count" in args else 255
    out = f"{project}.{mn:03d}-{mx:03d}.kma"
    matrix = np.load(out)["matrix"]
    with open(out + ".json") as fh:
        meta = json.load(fh)
    return matrix, meta


def sha256_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for chunk in iter(lambda: fh.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def index_case(name, spec, k, fname="input.fa", keep_table=False, flush_every=None, tables=None):
    d = os.path.join(SCRATCH, name)
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    data = inputs.make_input(spec)
    path = os.path.join(d, fname)
    with open(path, "wb") as fh:
        fh.write(data)
    t0 = time.time()
    meta, kin = ref_index(path, k, flush_every=flush_every)
    dt = time.time() - t0
    case = {"name": name, "input": spec, "input_file": fname, "k": k, "input_sha256": inputs.sha256(data),
            "input_size": len(data), "reference_seconds": round(dt, 2),
            "reference_keys": sorted(meta.keys()),
            "expect": {f: meta[f] for f in DETERMINISTIC}}
    if flush_every is not None:
        case["flush_every_arg"] = flush_every
    assert sha256_file(kin) == meta["output_file_cheksum"]
    if keep_table:
        tables[name] = np.fromfile(kin, dtype=np.uint8)
    print(f"[golden] {name}: k={k} num_kmers={meta['num_kmers']} vals_max={meta['vals_max']} "
          f"reference took {dt:.1f}s", flush=True)
    return case, kin


def load_manifest():
    p = os.path.join(GOLDEN, "manifest.json")
    if os.path.exists(p):
        with open(p) as fh:
            return json.load(fh)
    return {"indexer": {}, "merger": {}}


def save_manifest(m):
    os.makedirs(GOLDEN, exist_ok=True)
    cur = load_manifest()                      # merge: several groups may be generated concurrently
    for section in ("indexer", "merger", "bgzf"):
        cur.setdefault(section, {}).update(m.get(section, {}))
    m = cur
    with open(os.path.join(GOLDEN, "manifest.json"), "w") as fh:
        json.dump(m, fh, indent=1, sort_keys=True)


def group_small(m):
    tables = {}
    npz = os.path.join(GOLDEN, "tables_small.npz")
    if os.path.exists(npz):
        tables.update(dict(np.load(npz)))
    for k in (3, 5, 7):                                                   # G1: analytic KAT
        c, _ = index_case(f"G1_kat_k{k}", {"gen": "kat", "k": k}, k, keep_table=True, tables=tables)
        m["indexer"][c["name"]] = c
    c, _ = index_case("G2_c1_k7", {"gen": "c1"}, 7, keep_table=True, tables=tables)      # G2: config 1
    m["indexer"][c["name"]] = c
    for k in (3, 7, 9):                                                   # G3: edge cases, small k
        c, _ = index_case(f"G3_edge_k{k}", {"gen": "edge"}, k, keep_table=True, tables=tables)
        m["indexer"][c["name"]] = c
    c, _ = index_case("G3_edge_gz_k7", {"gen": "edge_gz"}, 7, fname="input.fa.gz", keep_table=True, tables=tables)
    m["indexer"][c["name"]] = c
    c, _ = index_case("G4_edge_flush1000_k7", {"gen": "edge"}, 7, keep_table=True, flush_every=1000, tables=tables)
    m["indexer"][c["name"]] = c
    assert np.array_equal(tables["G4_edge_flush1000_k7"], tables["G3_edge_k7"]), "batching changed the table"
    c, _ = index_case("G5_c2_2M_k7", {"gen": "c2", "args": {"total_bp": 2_000_000}}, 7, keep_table=True, tables=tables)
    m["indexer"][c["name"]] = c
    np.savez_compressed(npz, **tables)


def group_k15(m):
    c, _ = index_case("G3_edge_k15", {"gen": "edge"}, 15)
    m["indexer"][c["name"]] = c
    c, _ = index_case("G5_c2_20M_k15", {"gen": "c2", "args": {"total_bp": 20_000_000}}, 15)
    m["indexer"][c["name"]] = c
    c, _ = index_case("G5_c1_4M_k13", {"gen": "c1", "args": {"total_bp": 4_000_000, "seed": 7}}, 13)
    m["indexer"][c["name"]] = c


def group_k17(m):
    c, kin = index_case("G5_c2_8M_k17", {"gen": "c2", "args": {"total_bp": 8_000_000}}, 17)
    m["indexer"][c["name"]] = c
    os.remove(kin)


def group_full(m):
    c, kin = index_case("G6_c2_800M_k15", {"gen": "c2"}, 15)
    m["indexer"][c["name"]] = c
    os.remove(kin)


def group_merge(m):
    d = os.path.join(SCRATCH, "merge")
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    kins = []
    N = 13
    for i in range(N):                                                    # G7: 13 tables, k=7
        spec = {"gen": "family", "args": {"index": i, "total_bp": 3000 + 600 * i, "n_records": 2}}
        path = os.path.join(d, f"s{i:02d}.fa")
        with open(path, "wb") as fh:
            fh.write(inputs.make_input(spec))
        meta, kin = ref_index(path, 7)
        kins.append(kin)
    # one input handed over as .kin.bgz (python gzip, tools.py:300-302); Header.index_file prefers it
    with open(kins[4], "rb") as fh, gzip.open(kins[4] + ".bgz", "wb") as out:
        out.write(fh.read())
    inputs_spec = [{"gen": "family", "args": {"index": i, "total_bp": 3000 + 600 * i, "n_records": 2}} for i in range(N)]
    for tag, extra in (("default", []), ("min2", ["--min-count", "2"]), ("max3", ["--max-count", "3"]),
                       ("min2max5", ["--min-count", "2", "--max-count", "5"])):
        proj = os.path.join(d, f"proj_{tag}")
        t0 = time.time()
        matrix, meta = ref_merge(proj, kins, extra)
        for i in range(N):
            matrix[i, i, :] = 0                                           # diagonal is unassigned garbage in the reference
        m["merger"][f"G7_k7_n13_{tag}"] = {
            "k": 7, "inputs": inputs_spec, "args": extra, "matrix": matrix.tolist(),
            "kma_json_keys": sorted(meta.keys()),
            "kma_json_data0_keys": sorted(meta["data"][0].keys()),
            "kma_json_header_keys": sorted(meta["data"][0]["header"].keys()),
            "order": [os.path.basename(x["index_file"]) for x in meta["data"]],
            "reference_seconds": round(time.time() - t0, 1),
        }
        print(f"[golden] merge {tag}: shared[0,1]={matrix[0, 1, 2]} took {time.time() - t0:.0f}s", flush=True)


def group_bgzf(m):
    """f1/f2: a .kin.bgz + .gzi written by THIS build's BGZF writer (the reference shells out to htslib's bgzip,
    README.md:26, which is not in the image), read back by the reference's own tools: gzireader.print_index on
    the index, and tools.Header.open_file -> gzip.open on the table (tools.py:294-305).  What the reference printed
    and what it read are committed; tests/test_bgzf.py regenerates the file and compares."""
    from pykmer_amd import bgzf
    _import_reference()
    import gzireader                                                      # the reference's reader of the index layout
    os.makedirs(SCRATCH, exist_ok=True)
    spec = {"gen": "bgzf_table", "n": 300_000, "seed": 77}
    table = inputs.make_input(spec)
    raw = os.path.join(SCRATCH, "bgzf_case.07.kin")
    with open(raw, "wb") as fh:
        fh.write(table)
    dst, gzi = bgzf.compress_file(raw, level=9, threads=2)
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        gzireader.print_index(gzi)
    with gzip.open(dst, "rb") as fh:                                      # what tools.py:300-302 does with a .bgz
        back = fh.read()
    assert back == bytes(table)
    m.setdefault("bgzf", {})["gzi_300k_level9"] = {
        "input": spec, "input_sha256": hashlib.sha256(bytes(table)).hexdigest(), "level": 9,
        "bgz_sha256": sha256_file(dst), "gzi_sha256": sha256_file(gzi),
        "gzireader_stdout": out.getvalue(), "gzip_open_sha256": hashlib.sha256(back).hexdigest(),
        "zlib_version": __import__("zlib").ZLIB_VERSION}
    print(f"[golden] bgzf: reference gzireader printed {len(out.getvalue().splitlines())} lines", flush=True)


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "small"
    m = {"indexer": {}, "merger": {}}
    groups = {"small": group_small, "k15": group_k15, "k17": group_k17, "full": group_full, "merge": group_merge, "bgzf": group_bgzf}
    for name in (groups if what == "all" else [what]):
        groups[name](m)
        save_manifest(m)


if __name__ == "__main__":
    main()
