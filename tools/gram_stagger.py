"""Does the placement of the N tables in HBM move the merge scan?  One allocation, table i at i * (4^15 + stagger) bytes,
for several staggers; and N separate allocations as the allocator hands them out."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pykmer_amd import _lib
n = 4 ** 15
N = int(sys.argv[1]) if len(sys.argv) > 1 else 13
g = torch.Generator(device="cuda").manual_seed(5)
src = (torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda", generator=g) * (torch.rand(n, device="cuda", generator=g) < 0.4)).to(torch.uint8)
for stagger in (0, 256, 2048, 4096, 4096 + 256, 65536, 65536 + 4096 + 256, (1 << 21), (1 << 21) + 4096 + 256, 3 * 4096 * 7 + 256):
    big = torch.empty(N * (n + stagger) + 4096, dtype=torch.uint8, device="cuda")
    base = (big.data_ptr() + 4095) // 4096 * 4096
    ptrs = []
    for i in range(N):
        p = base + i * (n + stagger)
        off = p - big.data_ptr()
        big[off:off + n].copy_(torch.roll(src, i * 1000003))
        ptrs.append(p)
    torch.cuda.synchronize()
    ts = sorted(_lib.gram_device_partial(ptrs, n)[1] for _ in range(7))
    print(f"N={N} one allocation, stagger {stagger:8d}: best {ts[0]*1e3:6.3f} ms  median {ts[3]*1e3:6.3f} ms  {N*n/ts[0]/1e12:5.2f} TB/s", flush=True)
    del big
    torch.cuda.empty_cache()
tabs = [torch.roll(src, i * 1000003).clone() for i in range(N)]
torch.cuda.synchronize()
print("separate allocations, base addresses mod 2^30:", [hex(t.data_ptr() % (1 << 30)) for t in tabs][:6], "...")
ts = sorted(_lib.gram_device_partial([t.data_ptr() for t in tabs], n)[1] for _ in range(7))
print(f"N={N} separate allocations: best {ts[0]*1e3:6.3f} ms  median {ts[3]*1e3:6.3f} ms")
