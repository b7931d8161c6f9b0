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
    from pykmer_amd import bgzf
    bgzf.compress_file(kins[0], level=1, threads=16)
    out["bgzf_level1_1GiB_16thr_s"] = time.perf_counter() - t0
    t0 = time.perf_counter(); bgzf.decompress_file(kins[0] + ".bgz", threads=16); out["bgzf_inflate_1GiB_16thr_s"] = time.perf_counter() - t0
print(json.dumps(out))
