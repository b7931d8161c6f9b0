"""One N-table sweep of W windows on random tables in HBM (for rocprofv3 counter passes): python tools/sweep_probe.py N W"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pykmer_amd import _lib
N, W = int(sys.argv[1]), int(sys.argv[2])
n = 4 ** 15
sweep = [(1, 255), (2, 255), (3, 255), (4, 255), (5, 255), (8, 255), (1, 50), (2, 20)][:W]
g = torch.Generator(device="cuda").manual_seed(N)
tabs = [(torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda", generator=g) * (torch.rand(n, device="cuda", generator=g) < 0.4)) for _ in range(N)]
acc = torch.zeros((W, N, N), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
ptrs = [t.data_ptr() for t in tabs]
for _ in range(3):
    t = _lib.gram_device_accumulate_windows(ptrs, n, acc.data_ptr(), sweep)
print(f"N={N} W={W}: {t * 1e3:.3f} ms")
