"""Times the merge scan on random tables resident in HBM for several N (kernel seconds from HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pykmer_amd import _lib
n = 4 ** 15
for N in [int(x) for x in (sys.argv[1:] or ["13", "16", "24", "32", "48"])]:
    g = torch.Generator(device="cuda").manual_seed(N)
    tabs = [(torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda", generator=g) * (torch.rand(n, device="cuda", generator=g) < 0.4)) for _ in range(N)]
    torch.cuda.synchronize()
    ptrs = [t.data_ptr() for t in tabs]
    best = min(_lib.gram_device_partial(ptrs, n)[1] for _ in range(5))
    print(f"N={N:3d} kernel {best*1e3:8.3f} ms  {N*n/best/1e12:6.2f} TB/s", flush=True)
    del tabs
    torch.cuda.empty_cache()

# threshold sweeps: eight windows from one call (k_gram_mw) against one single-window scan
import numpy as np
sweep = [(1, 255), (2, 255), (3, 255), (4, 255), (5, 255), (8, 255), (1, 50), (2, 20)]
for N in (8, 13, 16, 24, 32):
    g = torch.Generator(device="cuda").manual_seed(N)
    tabs = [(torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda", generator=g) * (torch.rand(n, device="cuda", generator=g) < 0.4)) for _ in range(N)]
    acc = torch.zeros((8, N, N), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    ptrs = [t.data_ptr() for t in tabs]
    one = min(_lib.gram_device_partial(ptrs, n)[1] for _ in range(3))
    for W in (2, 4, 8):
        t = min(_lib.gram_device_accumulate_windows(ptrs, n, acc.data_ptr(), sweep[:W]) for _ in range(3))
        print(f"N={N:3d} sweep of {W} windows {t*1e3:8.3f} ms = {t/one:5.2f} x one scan ({one*1e3:.3f} ms)", flush=True)
    del tabs
    torch.cuda.empty_cache()
