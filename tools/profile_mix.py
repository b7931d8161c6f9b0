"""Counting throughput on inputs unlike the headline genome: no repeats, mostly tandem repeats, many short
records, lowercase + N heavy.  Robustness check of the hot-key / partition machinery, not a benchmark."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch, synth
from pykmer_amd import _lib

cases = {
    "uniform": dict(seed=31, total_bp=200_000_000, n_records=8),
    "tandem_50pct": dict(seed=32, total_bp=200_000_000, n_records=8, pm_tandem=500, pm_dup=100),
    "dup_60pct": dict(seed=33, total_bp=200_000_000, n_records=8, pm_dup=600),
    "reads_100k_records": dict(seed=34, total_bp=100_000_000, n_records=100_000),
    "reads_400k_records": dict(seed=34, total_bp=400_000_000, n_records=400_000),
    "n_and_lowercase": dict(seed=35, total_bp=200_000_000, n_records=8, pm_ngap=300, pm_lower=400),
}
k = 15
for name, kw in cases.items():
    fa, bp = synth.generate(kw.pop("seed"), kw.pop("total_bp"), kw.pop("n_records"), **kw)
    d = torch.empty(fa.size + 64, dtype=torch.uint8, device="cuda"); d[:fa.size].copy_(torch.from_numpy(fa)); torch.cuda.synchronize()
    with _lib.Indexer(k) as ix:
        best = 1e9
        for _ in range(4):
            ix.reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
            ix.feed_device(d.data_ptr(), int(fa.size)); fin = ix.finish(); best = min(best, time.perf_counter() - t0)
        t = ix.timings()
    print(f"{name:22s} {bp/1e6:6.0f} Mbp  {best*1e3:7.2f} ms  {bp/best/1e9:6.1f} Gbp/s  kmers {fin['num_kmers']:>11d}  records {fin['n_records']}"
          f"   [all feeds, ms: structure {t['scan_s']*1e3:.2f} squeeze {t['squeeze_s']*1e3:.2f} sort {t['walk_sort_s']*1e3:.2f} "
          f"layout+level2 {(t['partition_s'] - t['walk_sort_s'])*1e3:.2f} count {t['bucket_s']*1e3:.2f} relayouts {t['relayouts']}]", flush=True)
    del d
