"""Deterministic synthetic FASTA inputs -- test/bench infrastructure.

The generator itself is `synth.c` (integer-only xoshiro256**, identical bytes on every host);
this module builds it with gcc on first use and exposes the named workloads:

    C1        10 Mbp, 5 records, uniform + 1 % N runs + 5 % lowercase           (k=7 plumbing case)
    C2        ~800 Mbp repeat-rich "genome", 12 records                          (k=15 / k=17 headline)
    family()  genomes derived from one ancestor by substitutions + indels        (merge cases C3 / C5)

Nothing here is on the product path and nothing here is the oracle.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libpksynth.so")
_lib = None


class Params(ctypes.Structure):
    _fields_ = [
        ("seed", ctypes.c_uint64),
        ("total_bp", ctypes.c_uint64),
        ("n_records", ctypes.c_uint32),
        ("line_width", ctypes.c_uint32),
        ("pm_dup", ctypes.c_uint32),
        ("pm_tandem", ctypes.c_uint32),
        ("pm_ngap", ctypes.c_uint32),
        ("pm_lower", ctypes.c_uint32),
        ("big_gap", ctypes.c_uint32),
        ("crlf", ctypes.c_uint32),
        ("mut_seed", ctypes.c_uint64),
        ("sub_ppm", ctypes.c_uint32),
        ("indel_ppm", ctypes.c_uint32),
    ]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "synth.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        tmp = f"{_SO}.{os.getpid()}.tmp"               # several ranks may build at once: write aside, rename atomically
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", src, "-o", tmp])
        os.replace(tmp, _SO)
    return _SO


def _load():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(build())
        lib.pk_synth_bound.restype = ctypes.c_uint64
        lib.pk_synth_bound.argtypes = [ctypes.POINTER(Params)]
        lib.pk_synth_fasta.restype = ctypes.c_uint64
        lib.pk_synth_fasta.argtypes = [ctypes.POINTER(Params), ctypes.c_void_p, ctypes.c_uint64,
                                       ctypes.POINTER(ctypes.c_uint64)]
        _lib = lib
    return _lib


def generate(seed, total_bp, n_records, *, line_width=60, pm_dup=0, pm_tandem=0, pm_ngap=0,
             pm_lower=0, big_gap=0, crlf=False, mut_seed=0, sub_ppm=0, indel_ppm=0):
    """Returns (fasta_bytes: np.ndarray[uint8], total_bp_written)."""
    lib = _load()
    p = Params(seed, total_bp, n_records, line_width, pm_dup, pm_tandem, pm_ngap, pm_lower, big_gap,
               1 if crlf else 0, mut_seed, sub_ppm, indel_ppm)
    cap = lib.pk_synth_bound(ctypes.byref(p))
    buf = np.empty(cap, dtype=np.uint8)
    bp = ctypes.c_uint64(0)
    n = lib.pk_synth_fasta(ctypes.byref(p), buf.ctypes.data, cap, ctypes.byref(bp))
    if n == 0:
        raise RuntimeError("synthetic FASTA generation failed")
    return buf[:n], int(bp.value)


def c1(total_bp=10_000_000, seed=1):
    """5 records, uniform ACGT + 1 % N runs + 5 % lowercase."""
    return generate(seed, total_bp, 5, pm_ngap=10, pm_lower=50)


def c2(total_bp=800_000_000, seed=2, n_records=12):
    """60 % uniform / 25 % segmental dup / 10 % tandem / 3 % N / 2 % lowercase."""
    return generate(seed, total_bp, n_records, pm_dup=250, pm_tandem=100, pm_ngap=30, pm_lower=20,
                    big_gap=100_000 if total_bp >= 10_000_000 else 0)


def family(index, total_bp, seed=3, n_records=4):
    """member `index` of a family derived from one ancestor.

    index 0 is the ancestor; higher indices get 0.1 % .. 20 % substitutions (geometric ladder) plus
    indels, so pairwise Jaccard spans a wide range.
    """
    if index == 0:
        return generate(seed, total_bp, n_records, pm_dup=100, pm_tandem=50, pm_ngap=10, pm_lower=20)
    ladder = [1000, 2000, 5000, 10000, 20000, 30000, 50000, 70000, 100000, 130000, 160000, 200000]
    sub = ladder[(index - 1) % len(ladder)]
    return generate(seed, total_bp, n_records, pm_dup=100, pm_tandem=50, pm_ngap=10, pm_lower=20,
                    mut_seed=1000 + index, sub_ppm=sub, indel_ppm=sub // 20)
