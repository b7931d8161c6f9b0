import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, synth
from pykmer_amd import _lib
fa, bp = synth.c2(800_000_000, seed=2)
tab = np.empty(4 ** 15, dtype=np.uint8)
for rep in range(4):
    t0 = time.perf_counter(); r = _lib.count_fasta(fa, 15, table_out=tab); dt = time.perf_counter() - t0
    print(f"count_fasta host buffers: {dt*1e3:.1f} ms -> {bp/dt/1e9:.2f} Gbp/s", flush=True)
with _lib.Indexer(15) as ix:
    for rep in range(3):
        ix.reset(); t0 = time.perf_counter(); ix.feed(fa); t1 = time.perf_counter(); fin = ix.finish(); t2 = time.perf_counter(); ix.table_to_host(tab); t3 = time.perf_counter()
        print(f"feed {1e3*(t1-t0):.1f} ms finish {1e3*(t2-t1):.1f} table_to_host {1e3*(t3-t2):.1f} ms", flush=True)
