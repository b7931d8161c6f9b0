"""pyoracle.py -- literal pure-Python / numpy restatement of the pykmer hot path.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, never by the product package.  It keeps the reference's *shape* (record tuples, O(k) window
evaluation, unique+saturating-add batches) so that it can be read side by side with
/root/reference; `kmer_oracle.c` is the fast streaming twin used for large inputs.  Both are pinned
against tests/golden/ (outputs of the reference itself, see oracle/gen_golden.py).

Citations are into /root/reference.
"""
from typing import Iterator, List, Optional, Tuple

import numpy as np

# indexer.py:36-41 -- ALFA / CONV.  The reference's list has 255 slots; ord >= 255 raises there.
_CODES = {}
for _v, _c in enumerate("ACGT"):
    _CODES[_c] = _v
    _CODES[_c.lower()] = _v


def records(text: str) -> Iterator[Tuple[str, Tuple[Optional[int], ...], int]]:
    """indexer.py:45-99 (parse_fasta) on an already-decoded text stream.

    Text mode with universal newlines (indexer.py:110,115) ends lines at \\n, \\r and \\r\\n.
    """
    name = None
    parts: List[str] = []
    for raw in text.replace("\r\n", "\n").replace("\r", "\n").split("\n"):
        line = raw.strip()                       # :56
        if not line:                             # :58-59
            continue
        if line[0] == ">":                       # :66
            if name is not None:
                seq = tuple(_CODES.get(ch) for part in parts for ch in part)   # :75-76
                yield name, seq, len(seq)
            name = line[1:]                      # :80
            parts = []                           # :82 (also drops lines seen before the first header)
        else:
            parts.append(line)                   # :84
    if name is not None:                         # :86-95
        seq = tuple(_CODES.get(ch) for part in parts for ch in part)
        yield name, seq, len(seq)


def windows(seq, k: int) -> Iterator[Tuple[int, int, int]]:
    """indexer.py:130-160 (gen_kmers body): (position, fwd, rev) for every None-free window."""
    weight = [4 ** (k - p - 1) for p in range(k)]           # :131
    for i in range(0, len(seq) - k + 1):                    # :141
        w = seq[i:i + k]
        if None in w:                                       # :144
            continue
        fwd = rev = 0
        for p, b in enumerate(w):                           # :148-150
            fwd += weight[p] * b
            rev += weight[k - p - 1] * (3 - b)
        yield i, fwd, rev


def apply_batch(table: np.ndarray, batch: np.ndarray) -> None:
    """indexer.py:162-297 (process_kmers) without the fragment loop: unique, clip, saturating add."""
    uniq, cnt = np.unique(batch, return_counts=True)        # :169
    cnt = np.minimum(cnt, 255).astype(np.uint8)             # :239
    cur = table[uniq]
    table[uniq] = cur + np.minimum(255 - cur, cnt)          # :262


def count_fasta(data: bytes, k: int, flush_every: int = 100_000_000):
    """indexer.py:299-414 (create_fasta_index) up to the point where the table is complete.

    Returns (table u8[4^k], num_kmers, chromosomes [(name, seq_len)], all_records [(name, seq_len, n_valid)]).
    `chromosomes` lists only records that produced a k-mer (indexer.py:349-351).
    """
    assert k > 0 and k % 2 == 1                             # tools.py:165-167
    table = np.zeros(4 ** k, dtype=np.uint8)
    buf: List[int] = []
    num_kmers = 0
    chromosomes, everything = [], []
    for name, seq, seq_len in records(data.decode("utf-8")):
        n_valid = 0
        for _, fwd, rev in windows(seq, k):
            buf.append(fwd if fwd < rev else rev)           # :341
            n_valid += 1
            if len(buf) >= flush_every:                     # :345,358-372
                apply_batch(table, np.asarray(buf, dtype=np.uint64))
                buf = []
        num_kmers += n_valid
        if n_valid:
            chromosomes.append((name, seq_len))
        everything.append((name, seq_len, n_valid))
    if buf:                                                 # :380-384
        apply_batch(table, np.asarray(buf, dtype=np.uint64))
    return table, num_kmers, chromosomes, everything


def table_stats(table: np.ndarray):
    """tools.py:246-263 (Header.update_stats), same numpy calls."""
    hist, _ = np.histogram(table, bins=255, range=(1, 255))
    return {
        "hist": hist.tolist(),
        "hist_sum": int(hist.sum()),
        "hist_count": int(np.count_nonzero(hist)),
        "hist_min": int(hist.min()),
        "hist_max": int(hist.max()),
        "vals_sum": int(table.sum(dtype=np.uint64)),
        "vals_count": int(np.count_nonzero(table)),
        "vals_min": int(table.min()),
        "vals_max": int(table.max()),
    }


def pair_distance(a: np.ndarray, b: np.ndarray, min_count: int = 1, max_count: int = 255,
                  block_size: int = 100_000_000):
    """tools.py:439-493 (Header.calculate_distance) on in-memory tables."""
    assert a.shape == b.shape                               # :444
    s = o = c = 0
    for start in range(0, a.shape[0], block_size):          # :449
        sb, ob = a[start:start + block_size], b[start:start + block_size]
        sv = (sb >= min_count) & (sb <= max_count)          # :473-475
        ov = (ob >= min_count) & (ob <= max_count)
        s += int(sv.sum()); o += int(ov.sum()); c += int((sv & ov).sum())
    return s, o, c


def gram(tables, min_count: int = 1, max_count: int = 255) -> np.ndarray:
    """merger.py:136-176: (N,N,3) u64; off-diagonal per merger.py:175-176, diagonal zero.

    The reference never assigns the diagonal of its uninitialised np.ndarray (merger.py:136); zero
    is what it holds when the allocator hands back fresh pages, and what this build writes.
    """
    assert min_count >= 1 and max_count <= 255              # merger.py:90-91
    n = len(tables)
    m = np.zeros((n, n, 3), dtype=np.uint64)
    for i in range(n - 1):
        for j in range(i + 1, n):
            ti, tj, sh = pair_distance(tables[i], tables[j], min_count, max_count)
            m[i, j, :] = (ti, tj, sh)
            m[j, i, :] = (tj, ti, sh)
    return m
