"""Reads a rocprofv3 --kernel-trace CSV and prints one indexer step (from one k_chunk_l1 launch to the next): every
dispatch with its duration and the idle gap in front of it, then the totals -- where a step's time goes between kernels.
    python tools/trace_gaps.py <dir-or-csv> [step index, default: the last complete one]"""
import csv
import glob
import os
import re
import sys

path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
start = next(c for c in rows[0] if "start" in c.lower())
end = next(c for c in rows[0] if "end" in c.lower())
name = next(c for c in rows[0] if c.lower() in ("kernel_name", "name"))
ev = sorted(((int(r[start]), int(r[end]), r[name]) for r in rows), key=lambda e: e[0])
heads = [i for i, e in enumerate(ev) if "k_chunk_l1" in e[2]]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(heads) - 2
a, b = heads[which], heads[which + 1]
step = ev[a:b]
short = lambda n: re.sub(r"\(.*", "", re.sub(r"^void ", "", n)).replace("pk::", "")[:70]
busy = gaps = 0
prev_end = step[0][0]
for s, e, n in step:
    gap = max(0, s - prev_end)
    gaps += gap
    busy += e - s
    print(f"{(e - s) / 1e3:9.1f} us  gap {gap / 1e3:7.1f}  {short(n)}")
    prev_end = max(prev_end, e)
print(f"step: {len(step)} dispatches, kernels {busy / 1e6:.3f} ms, gaps {gaps / 1e6:.3f} ms, first start -> next step's first start {(ev[b][0] - step[0][0]) / 1e6:.3f} ms")
