import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, synth
from pykmer_amd import _lib
for name, kw in {"reads_400k": dict(seed=34, total_bp=400_000_000, n_records=400_000), "uniform": dict(seed=31, total_bp=400_000_000, n_records=8)}.items():
    fa, bp = synth.generate(kw.pop("seed"), kw.pop("total_bp"), kw.pop("n_records"), **kw)
    d = torch.empty(fa.size + 64, dtype=torch.uint8, device="cuda"); d[:fa.size].copy_(torch.from_numpy(fa)); torch.cuda.synchronize()
    with _lib.Indexer(15) as ix:
        for _ in range(3):
            ix.reset(); t0 = time.perf_counter(); ix.feed_device(d.data_ptr(), int(fa.size)); fin = ix.finish(); dt = time.perf_counter() - t0
            t = ix.timings()
        print(name, "%.2f ms" % (dt * 1e3), {k: round(v * 1e3, 3) for k, v in t.items() if k.endswith("_s")}, flush=True)
