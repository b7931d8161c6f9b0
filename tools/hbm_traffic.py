#!/usr/bin/env python3
"""Derive profiles/hbm_traffic.json from the two PMC passes.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d D1 -o fetch -- python3 bench.py --steps 2 --warmup 0 --no-cpu
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d D2 -o write -- python3 bench.py --steps 2 --warmup 0 --no-cpu
    python tools/hbm_traffic.py fetch_counter_collection.csv write_counter_collection.csv > profiles/hbm_traffic.json

Per kernel the LARGEST dispatch is taken (= the 800 Mbp k=15 step; bench.py also runs 40 Mbp genomes for
the merge section).  bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB and FETCH_SIZE
reports half of a wide coalesced read stream on gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section).
`_pipeline_bytes_per_step` sums the kernels one indexer feed + finish launches.
"""
import csv
import json
import re
import sys

PIPELINE = ("k_chunk_l1", "k_chunk_l2", "k_scan_l1_reduce", "k_scan_l1_tiles", "k_scan_l1_apply", "k_scan_l2_reduce",
            "k_scan_l2_tiles", "k_scan_l2_apply", "k_squeeze", "k_walk_sort_count", "k_tally_sum", "k_provision", "k_walk_sort",
            "k_level1_finish", "k_sample2", "k_rooms2", "k_bases2", "k_starts2", "k_count2", "k_rows2_scan", "k_scatter2", "k_bucket_count", "k_hist_reduce", "k_apply_side")


def short(name: str) -> str:
    m = re.search(r"pk::(k_\w+)(<[^>]*>)?", name)
    if not m:
        return ""
    base = m.group(1)
    if base.startswith("k_gram") and m.group(2):
        return base + "<" + m.group(2)[1:-1].split(",")[0] + ">"
    if base == "k_walk_sort" and re.search(r"k_walk_sort<[^,]+, true", name):
        return "k_walk_sort_count"                                   # the sampling launch (tallies only)
    if base.startswith("k_bucket_count"):
        return "k_bucket_count"                                      # whole / half / half_lean: one of them per feed
    return base


def largest(path: str, counter: str) -> dict:
    best = {}
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] != counter:
                continue
            k = short(row["Kernel_Name"])
            if k:
                best[k] = max(best.get(k, 0.0), float(row["Counter_Value"]))
    return best


def main() -> None:
    fetch_csv, write_csv = sys.argv[1], sys.argv[2]
    fetch, write = largest(fetch_csv, "FETCH_SIZE"), largest(write_csv, "WRITE_SIZE")
    out = {
        "_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, `bench.py --steps 2 --warmup 0 --no-cpu`), "
                 "largest dispatch per kernel (= the 800 Mbp step); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per "
                 "MI355X_MICROARCH.md HBM section (FETCH_SIZE reads half of a wide coalesced stream on gfx950). "
                 "Calibration: k_gram_blk (N=32) vs 34.36 GB algorithmic; k_chunk_l2 vs 0.813 GB FASTA + 0.102 GB lane states + 0.407 GB piece packs. "
                 "The factor 2 is calibrated for 16-byte-per-lane streams only; k_walk_sort reads 4 bytes per lane (0.3 GB of packed "
                 "bases) and k_squeeze reads 8 + 32 bytes per lane (states and piece packs), so their read side is approximate -- their traffic is mostly writes. "
                 "Derived by tools/hbm_traffic.py.",
        "_source": [fetch_csv, write_csv],
    }
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        out[k] = {"fetch_size_kib": f, "write_size_kib": w, "bytes_per_launch": int((2 * f + w) * 1024)}
    out["_pipeline_bytes_per_step"] = sum(out[k]["bytes_per_launch"] for k in PIPELINE if k in out)
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
