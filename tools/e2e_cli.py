"""End-to-end CLI timings on the GPU box (disk -> .kin/.kin.json, 13 x .kin -> .kma): the t_e2e figures of SURVEY 8d."""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import synth

def run(*argv):
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable] + list(argv), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return time.perf_counter() - t0

out = {}
with tempfile.TemporaryDirectory(dir=os.environ.get("PK_TMP", "/tmp")) as d:
    fa, bp = synth.c2(800_000_000)
    big = os.path.join(d, "genome.fa"); fa.tofile(big)
    out["indexer_cli_k15_800Mbp_s"] = run(os.path.join(ROOT, "indexer.py"), big, "genome", "15")
    out["indexer_cli_k15_800Mbp_bp_per_s"] = bp / out["indexer_cli_k15_800Mbp_s"]
    meta = json.load(open(big + ".15.kin.json"))
    out["indexer_json_creation_speed"] = meta["creation_speed"]
    # the same genome bgzipped (what `bgzip genome.fa` writes; indexer.py:112-115 reads .gz / .bgz through gzip.open) and as a
    # plain gzip stream: inflated in pieces while the GPU counts the piece before
    from pykmer_amd import bgzf
    gz = os.path.join(d, "genome_bgzf.fa.gz")
    t0 = time.perf_counter(); bgzf.compress_file(big, gz, level=1, threads=16, index=False); out["bgzf_level1_fasta_16thr_s"] = time.perf_counter() - t0
    out["indexer_cli_k15_800Mbp_bgzf_s"] = run(os.path.join(ROOT, "indexer.py"), gz, "genome", "15")
    out["indexer_cli_bgzf_over_plain"] = out["indexer_cli_k15_800Mbp_bgzf_s"] / out["indexer_cli_k15_800Mbp_s"]
    assert json.load(open(gz + ".15.kin.json"))["output_file_cheksum"] == meta["output_file_cheksum"]
    os.remove(gz + ".15.kin")
    import gzip, shutil
    pgz = os.path.join(d, "genome_plain.fa.gz")
    with open(big, "rb") as fi, gzip.open(pgz, "wb", compresslevel=1) as fo:
        shutil.copyfileobj(fi, fo, 1 << 24)
    out["indexer_cli_k15_800Mbp_gzip_s"] = run(os.path.join(ROOT, "indexer.py"), pgz, "genome", "15")
    assert json.load(open(pgz + ".15.kin.json"))["output_file_cheksum"] == meta["output_file_cheksum"]
    os.remove(pgz + ".15.kin"); os.remove(gz); os.remove(pgz)
    kins = []
    t_idx = 0.0
    for i in range(13):
        g, _ = synth.family(i, 20_000_000)
        p = os.path.join(d, f"s{i:02d}.fa"); g.tofile(p)
        t_idx += run(os.path.join(ROOT, "indexer.py"), p, f"s{i}", "15")
        kins.append(p + ".15.kin")
    out["indexer_cli_13x20Mbp_total_s"] = t_idx
    out["merger_cli_n13_k15_raw_kin_s"] = run(os.path.join(ROOT, "merger.py"), os.path.join(d, "proj"), *kins, "--threads", "8")
    t0 = time.perf_counter()
    bgzf.compress_file(kins[0], level=1, threads=16)
    out["bgzf_level1_1GiB_16thr_s"] = time.perf_counter() - t0
    t0 = time.perf_counter(); bgzf.decompress_file(kins[0] + ".bgz", threads=16); out["bgzf_inflate_1GiB_16thr_s"] = time.perf_counter() - t0
    t0 = time.perf_counter(); bgzf.decompress_file(kins[0] + ".bgz"); out["bgzf_inflate_1GiB_default_threads_s"] = time.perf_counter() - t0
    out["bgzf_inflate_default_threads"] = bgzf.INFLATE_THREADS
    # the reference's own input form: merger.py over *.kin.bgz (README.md:57-61) -- every table inflated block-parallel on
    # native threads, only the bytes of each device slice
    for kin in kins[1:]:
        bgzf.compress_file(kin, level=1, threads=16)
    bgz = [kin + ".bgz" for kin in kins]
    out["merger_cli_n13_k15_kin_bgz_s"] = run(os.path.join(ROOT, "merger.py"), os.path.join(d, "proj_bgz"), *bgz, "--threads", "13")
    import numpy as np
    assert np.array_equal(np.load(os.path.join(d, "proj_bgz.001-255.kma"))["matrix"], np.load(os.path.join(d, "proj.001-255.kma"))["matrix"])
print(json.dumps(out))
