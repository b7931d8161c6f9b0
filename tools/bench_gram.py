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
