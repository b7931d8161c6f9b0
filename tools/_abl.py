import os, sys, json, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, synth
from pykmer_amd import _lib
fasta, bp = synth.c2(800_000_000, seed=2)
d = torch.empty(fasta.size + 64, dtype=torch.uint8, device="cuda"); d[:fasta.size].copy_(torch.from_numpy(fasta)); torch.cuda.synchronize()
ix = _lib.Indexer(15)
for dbg in (0, 1, 4):
    os.environ["PK_DBG"] = str(dbg)
    ts = []
    for _ in range(4):
        ix.reset(); ix.feed_device(d.data_ptr(), fasta.size); t = ix.timings(); ts.append(t["walk_sort_s"] * 1e3)
    print(dbg, ["%.3f" % x for x in ts], flush=True)
