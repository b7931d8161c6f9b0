"""A read-set-shaped input (400 000 records of 1 kbp) counted a few times: for rocprofv3 counter / stats passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, synth
from pykmer_amd import _lib
fa, bp = synth.generate(34, 400_000_000, 400_000)
d = torch.empty(fa.size + 64, dtype=torch.uint8, device="cuda"); d[:fa.size].copy_(torch.from_numpy(fa)); torch.cuda.synchronize()
with _lib.Indexer(15) as ix:
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
        ix.reset(); ix.feed_device(d.data_ptr(), int(fa.size)); fin = ix.finish()
    t = ix.timings()
print(fin["num_kmers"], fin["n_records"], {k: round(v * 1e3, 3) if isinstance(v, float) else v for k, v in t.items()})
