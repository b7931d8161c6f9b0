"""Seeded test inputs shared by oracle/gen_golden.py (which ran the reference on them) and the tests.

Every golden fixture names its input by a small spec dict; `make_input(spec)` rebuilds the exact
bytes (checked against the sha256 stored in the manifest), so only expected outputs are committed.
"""
import gzip
import hashlib
import io
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import synth  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def kat_fasta(k: int) -> bytes:
    """Every one of the 4^k k-mers as its own record -- the layout /root/reference/test.py:8-27 writes
    (header `>examples/example--KK-NNNNNNNNNN`, one k-mer per record, lexicographic ACGT order)."""
    out = io.BytesIO()
    for num, tup in enumerate(itertools.product("ACGT", repeat=k)):
        out.write(f">examples/example--{k:02d}-{num + 1:010d}\n{''.join(tup)}\n".encode())
    return out.getvalue()


def edge_fasta() -> bytes:
    """Hand-built FASTA exercising every parser corner SURVEY.md 8c lists (G3)."""
    state = [12345]

    def nxt():                                   # splitmix64: no dependence on numpy's generators
        state[0] = (state[0] + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = state[0]
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def rand(n, alphabet="ACGT"):
        return "".join(alphabet[(nxt() >> 33) % len(alphabet)] for _ in range(n))

    def wrap(s, w=60, eol="\n"):
        return eol.join(s[i:i + w] for i in range(0, len(s), w)) + eol

    p = []
    p.append("ACGTACGTACGTACGTACGTACGT\nthis text precedes the first header and is dropped\n")
    p.append(">rec01 plain uppercase with a description  \n" + wrap(rand(3000)))
    p.append(">rec02_lowercase\n" + wrap(rand(1500, "acgt")))
    p.append(">rec03_mixed_case_and_N_runs\n" + wrap(rand(400) + "N" * 37 + rand(300, "acgt") + "n" * 5 + rand(700) + "N" + rand(90)))
    p.append(">rec04_iupac\n" + wrap(rand(200) + "RYKMSWBDHVN" + rand(200) + "ryk" + rand(100) + "U" + rand(50) + "-*." + rand(64)))
    p.append(">rec05_crlf\r\n" + wrap(rand(900), 70, "\r\n"))
    p.append(">rec06_blank_lines\n\n\n" + rand(80) + "\n\n   \n\t\n" + rand(75) + "\n\n")
    p.append(">rec07_lead_trail_ws\n   " + rand(60) + "\n" + rand(60) + "   \t\n\t " + rand(45) + " \n")
    p.append(">rec08_interior_space\n" + rand(40) + " " + rand(40) + "\n" + rand(30) + "\t\t" + rand(30) + "\n")
    p.append(">rec09_empty\n")
    p.append(">rec10_shorter_than_k\nACGTA\n")
    p.append(">rec11_exactly_7\nACGTTGC\n")
    p.append(">rec12_exactly_15\nACGTTGCAAGCTTAG\n")
    p.append(">rec13_all_N\n" + wrap("N" * 333))
    p.append(">rec14 gt mid-line\n" + rand(50) + ">" + rand(50) + "\n")
    p.append(">rec15_polyA\n" + wrap("A" * 700 + rand(20) + "T" * 400))
    p.append(">rec16_microsat\n" + wrap("AT" * 300 + rand(33) + "AAG" * 250 + rand(10) + "ACGT" * 100))
    p.append(">rec17_lone_cr\r" + rand(100) + "\r" + rand(77) + "\r")
    p.append("  \t>rec18_header_with_leading_ws\n" + wrap(rand(500)))
    p.append(">rec19_vt_ff_fs\n" + rand(50) + "\x0b\n\x0c" + rand(50) + "\x1c\n" + rand(20) + "\x0b" + rand(20) + "\n")
    p.append(">\n" + wrap(rand(120)))                       # empty name
    p.append(">rec21_kmers_only_across_lines\nACG\nTAC\nGTA\nCGT\nACG\nTTG\n")
    p.append(">rec22_saturation\n" + wrap(("ACGTTGCAAGCTTAGGCTAACGTAT" + "C") * 300))
    p.append(">rec23_no_trailing_newline\n" + rand(500))     # last line has no \n
    return "".join(p).encode("ascii")


def make_input(spec: dict) -> bytes:
    kind = spec["gen"]
    if kind == "kat":
        data = kat_fasta(spec["k"])
    elif kind == "edge":
        data = edge_fasta()
    elif kind == "edge_gz":
        data = gzip.compress(edge_fasta(), mtime=0)
    elif kind == "c1":
        data = synth.c1(**spec.get("args", {}))[0].tobytes()
    elif kind == "c2":
        data = synth.c2(**spec.get("args", {}))[0].tobytes()
    elif kind == "family":
        data = synth.family(**spec["args"])[0].tobytes()
    elif kind == "bgzf_table":                        # a count-table-like byte string: mostly zeros, small counts (seeded, numpy-free stream)
        import struct
        x, out = spec["seed"] * 0x9E3779B97F4A7C15 & 0xFFFFFFFFFFFFFFFF, bytearray()
        for _ in range(spec["n"]):
            x = (x * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
            r = x >> 40
            out.append(0 if r % 10 < 7 else 1 + (r >> 8) % 5)
        data = bytes(out)
    else:
        raise KeyError(kind)
    return data


def skewed_fasta(n_bp: int, unit_len: int, seed: int = 7, stretch: int = 2048, chunk: int = 16384, stride: int = 16) -> bytes:
    """One record that defeats the bucket-size sample on purpose.  The indexer sizes its buckets from one wave's stretch
    of bases (`stretch`: 64 threads x 32 bases for 32-bit k-mers, x 16 for 64-bit ones) out of every `stride` stretches:
    stretch number q of the text's 16 KiB chunks is sampled iff q % stride == (q // stride) % stride (kmer_fuse.hip,
    locate).  Here exactly those stretches hold uniform random sequence and everything else repeats one `unit_len`-base
    unit (a period the hot-key path does not look for), so the estimate is wrong by an order of magnitude, the buckets
    overflow and the exact re-layout has to run."""
    import numpy as np
    assert n_bp % 60 == 0 and chunk % stretch == 0
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    head = b">skewed_against_the_sample\n"
    i = np.arange(n_bp, dtype=np.int64)
    off = len(head) + i + i // 60                              # byte offset of base i: 60 bases and a newline per line
    slot = off // chunk
    first = np.searchsorted(slot, np.arange(slot[-1] + 1))     # index of the first base of every chunk
    q = slot * (chunk // stretch) + (i - first[slot]) // stretch
    sampled = q % stride == (q // stride) % stride
    unit = acgt[rng.integers(0, 4, size=unit_len)]
    seq = acgt[rng.integers(0, 4, size=n_bp)]
    rep = ~sampled
    seq[rep] = unit[i[rep] % unit_len]
    lines = np.empty((n_bp // 60, 61), dtype=np.uint8)
    lines[:, :60] = seq.reshape(-1, 60)
    lines[:, 60] = 10
    return head + lines.tobytes()


def sha256(data) -> str:
    return hashlib.sha256(data).hexdigest()
