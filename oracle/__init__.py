"""oracle -- CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
`kmer_oracle.c` (fast, streaming) and `kmer_oracle_mt.c` (the same count on all cores, for the
bench baseline) are built with gcc into oracle/_build/; `pyoracle.py` is the
literal pure-Python twin for small cases.  The reference is pure Python, so there is no
oracle/_ref build: the pin is tests/golden/, written by oracle/gen_golden.py running the reference.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libpkoracle.so")
_lib = None

RECORD_DTYPE = np.dtype([("name_off", "<u8"), ("name_len", "<u8"), ("seq_len", "<u8"),
                         ("n_valid_kmers", "<u8")])


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, "kmer_oracle.c"), os.path.join(_HERE, "kmer_oracle_mt.c")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(s) for s in srcs):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        tmp = f"{_SO}.{os.getpid()}.tmp"               # several ranks may build at once: write aside, rename atomically
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", *srcs, "-o", tmp])
        os.replace(tmp, _SO)
    return _SO


def _load():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(build())
        u64p = ctypes.POINTER(ctypes.c_uint64)
        lib.pko_count_fasta.restype = ctypes.c_int
        lib.pko_count_fasta.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p,
                                        u64p, u64p, ctypes.c_void_p, ctypes.c_uint64, u64p]
        lib.pko_count_fasta_ex.restype = ctypes.c_int
        lib.pko_count_fasta_ex.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p,
                                           u64p, u64p, ctypes.c_void_p, ctypes.c_uint64, u64p, ctypes.c_void_p, ctypes.c_uint64]
        lib.pko_count_fasta_mt.restype = ctypes.c_int
        lib.pko_count_fasta_mt.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p, u64p, u64p,
                                           ctypes.c_int]
        lib.pko_table_stats.restype = ctypes.c_int
        lib.pko_table_stats.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
        lib.pko_gram.restype = ctypes.c_int
        lib.pko_gram.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_int, ctypes.c_int,
                                 ctypes.c_void_p]
        lib.pko_gram_mt.restype = ctypes.c_int
        lib.pko_gram_mt.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_void_p, ctypes.c_int]
        _lib = lib
    return _lib


def count_fasta(fasta, k: int, table: np.ndarray = None):
    """Returns dict(table, num_kmers, total_bp, records[RECORD_DTYPE]).  `fasta`: bytes or u8 array."""
    lib = _load()
    buf = np.frombuffer(fasta, dtype=np.uint8) if isinstance(fasta, (bytes, bytearray)) else np.ascontiguousarray(fasta)
    if table is None:
        table = np.zeros(4 ** k, dtype=np.uint8)
    nk, bp, nr = ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_uint64(0)
    cap = 1024
    while True:
        recs = np.zeros(cap, dtype=RECORD_DTYPE)
        scratch = table if cap == 1024 else np.zeros_like(table)
        rc = lib.pko_count_fasta(buf.ctypes.data, buf.size, k, scratch.ctypes.data, ctypes.byref(nk),
                                 ctypes.byref(bp), recs.ctypes.data, cap, ctypes.byref(nr))
        if rc != 0:
            raise ValueError(f"oracle rejected k={k}")
        if nr.value <= cap:
            break
        cap = int(nr.value)                       # rare: re-run only to collect every record
    return {"table": table, "num_kmers": int(nk.value), "total_bp": int(bp.value),
            "records": recs[: nr.value].copy()}


def kmer_list(fasta, k: int) -> np.ndarray:
    """Canonical value of every valid window, in text order (u64) -- for k where the 4^k table is out of reach."""
    lib = _load()
    buf = np.frombuffer(fasta, dtype=np.uint8) if isinstance(fasta, (bytes, bytearray)) else np.ascontiguousarray(fasta)
    out = np.zeros(max(1, buf.size), dtype=np.uint64)
    nk, bp, nr = ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_uint64(0)
    rc = lib.pko_count_fasta_ex(buf.ctypes.data, buf.size, k, None, ctypes.byref(nk), ctypes.byref(bp), None, 0, ctypes.byref(nr),
                                out.ctypes.data, out.size)
    if rc != 0:
        raise ValueError(f"oracle rejected k={k}")
    return out[: nk.value]


def count_fasta_mt(fasta, k: int, threads: int, table: np.ndarray = None):
    """All-cores variant (kmer_oracle_mt.c): dict(table, num_kmers, total_bp), or None when the input has
    blanks other than line terminators (not handled there -- use count_fasta)."""
    lib = _load()
    buf = np.frombuffer(fasta, dtype=np.uint8) if isinstance(fasta, (bytes, bytearray)) else np.ascontiguousarray(fasta)
    if table is None:
        table = np.zeros(4 ** k, dtype=np.uint8)
    nk, bp = ctypes.c_uint64(0), ctypes.c_uint64(0)
    rc = lib.pko_count_fasta_mt(buf.ctypes.data, buf.size, k, table.ctypes.data, ctypes.byref(nk), ctypes.byref(bp),
                                int(threads))
    if rc < 0:
        raise ValueError(f"oracle rejected k={k}")
    if rc == 1:
        return None
    return {"table": table, "num_kmers": int(nk.value), "total_bp": int(bp.value)}


def chromosomes(fasta, records) -> list:
    """[(name, seq_len)] for records with >= 1 valid k-mer (indexer.py:349-351)."""
    raw = bytes(fasta) if not isinstance(fasta, (bytes, bytearray)) else fasta
    out = []
    for r in records:
        if r["n_valid_kmers"]:
            off, ln = int(r["name_off"]), int(r["name_len"])
            out.append((raw[off:off + ln].decode("utf-8"), int(r["seq_len"])))
    return out


def table_stats(table: np.ndarray):
    lib = _load()
    t = np.ascontiguousarray(table, dtype=np.uint8)
    hist = np.zeros(255, dtype=np.uint64)
    vals = np.zeros(4, dtype=np.uint64)
    lib.pko_table_stats(t.ctypes.data, t.size, hist.ctypes.data, vals.ctypes.data)
    return hist, vals


def gram(tables, min_count: int = 1, max_count: int = 255) -> np.ndarray:
    lib = _load()
    ts = [np.ascontiguousarray(t, dtype=np.uint8) for t in tables]
    n = len(ts)
    ptrs = (ctypes.c_void_p * n)(*[t.ctypes.data for t in ts])
    m = np.zeros((n, n, 3), dtype=np.uint64)
    rc = lib.pko_gram(ptrs, n, ts[0].size, min_count, max_count, m.ctypes.data)
    if rc != 0:
        raise ValueError("oracle rejected min/max count")
    return m


def gram_mt(tables, min_count: int = 1, max_count: int = 255, threads: int = 1) -> np.ndarray:
    """The pair loop spread over `threads` host threads (merger.py:137-153's pool); same matrix as gram()."""
    lib = _load()
    ts = [np.ascontiguousarray(t, dtype=np.uint8) for t in tables]
    n = len(ts)
    ptrs = (ctypes.c_void_p * n)(*[t.ctypes.data for t in ts])
    m = np.zeros((n, n, 3), dtype=np.uint64)
    if lib.pko_gram_mt(ptrs, n, ts[0].size, min_count, max_count, m.ctypes.data, int(threads)) != 0:
        raise ValueError("oracle rejected min/max count")
    return m
