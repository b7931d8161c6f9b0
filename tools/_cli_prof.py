import os, sys, time
t00 = time.perf_counter()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
t_np = time.perf_counter()
from pykmer_amd import _lib, indexer
t_imp = time.perf_counter()
_lib.load()
t_load = time.perf_counter()
import synth, hashlib, io, contextlib, tempfile
print(f"numpy {t_np-t00:.3f} pkg {t_imp-t_np:.3f} lib load {t_load-t_imp:.3f}", flush=True)
with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
    fa, bp = synth.c2(800_000_000)
    big = os.path.join(d, "genome.fa"); fa.tofile(big)
    for rep in range(2):
        t0 = time.perf_counter()
        ix = _lib.Indexer(15); t1 = time.perf_counter()
        src = indexer._Input(big)
        with contextlib.redirect_stdout(io.StringIO()):
            for piece in src.pieces(): ix.feed(piece)
        t2 = time.perf_counter(); fin = ix.finish(); recs = ix.records(fin["n_records"]); t3 = time.perf_counter()
        n = 4 ** 15
        out = os.path.join(d, "t.kin")
        with open(out, "wb") as fh: fh.truncate(n)
        table = np.memmap(out, dtype=np.uint8, mode="r+", shape=(n,)); t4 = time.perf_counter()
        for off in range(0, n, 64 << 20): ix.table_slice_to_host(table[off:off + (64 << 20)], off)
        t5 = time.perf_counter()
        h = hashlib.sha256(); h.update(table); t6 = time.perf_counter()
        table.flush(); t7 = time.perf_counter()
        h2 = indexer.gen_checksum(big, 1 << 22); t8 = time.perf_counter()
        ix.close(); t9 = time.perf_counter()
        print(f"create {t1-t0:.3f} feed {t2-t1:.3f} finish {t3-t2:.3f} memmap {t4-t3:.3f} d2h->memmap {t5-t4:.3f} sha256 table {t6-t5:.3f} flush {t7-t6:.3f} sha256 input {t8-t7:.3f} close {t9-t8:.3f}", flush=True)
        os.remove(out)
