import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, synth
from pykmer_amd import _lib
fasta, bp = synth.c2(800_000_000, seed=2)
n = int(fasta.size)
d = torch.empty(n + 64, dtype=torch.uint8, device='cuda'); d[:n].copy_(torch.from_numpy(fasta)); torch.cuda.synchronize()
for dbg in (0, 1, 2, 3, 4, 7, 8, 15):
    os.environ['PK_DEBUG_WALK'] = str(dbg)
    ix = _lib.Indexer(15)
    ts = []
    for i in range(3):
        ix.reset(); ix.feed_device(d.data_ptr(), n); t = ix.timings(); ts.append(t['count_s'] * 1e3)
    print(f"dbg={dbg:2d} walk_ms={min(ts):.3f} scans_ms={t['scan_s']*1e3:.3f}", flush=True)
    ix.close()
