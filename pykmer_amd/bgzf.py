"""BGZF (blocked gzip) reader / writer for `.kin.bgz` tables and `.fa.gz` inputs -- SURVEY.md 8f rows f1/f2.

The reference compresses finished tables with the htslib CLI (`bgzip -i -I $F.bgz.gzi -l 9 -c $F > $F.bgz`,
README.md:26, data/README.md:24) and reads them back through one single-threaded `gzip.open`
(tools.py:294-305).  A BGZF file is a series of independent gzip members of <= 64 KiB each carrying
their compressed size in a `BC` extra field (SAM spec 4.1), so both directions parallelise over
blocks.  Reading goes through the library (pk_bgzf_scan / pk_bgzf_inflate, csrc/bgzf_host.cpp: zlib on native threads
straight into the destination array -- from Python threads the per-block interpreter work capped a 1 GiB table at ~1 GB/s);
the writer goes the same way (pk_bgzf_deflate).  The `.gzi` index is the layout
gzireader.py:12-34 prints: u64 count, then (compressed_offset, uncompressed_offset) u64 pairs for
every block but the first.

Files that are gzip but not BGZF (what python's gzip module writes, and what the reference's own
tests use as `.bgz`) are detected and handed to the sequential gzip reader.
"""
import gzip
import os
import struct
import zlib
from typing import List, Tuple

import numpy as np

BLOCK_INPUT = 0xFF00                    # uncompressed bytes per block (htslib BGZF_BLOCK_SIZE)
_HEADER = b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00"
EOF_BLOCK = _HEADER + struct.pack("<H", 27) + b"\x03\x00" + struct.pack("<II", 0, 0)
INFLATE_THREADS = max(1, min(64, len(os.sched_getaffinity(0))))  # native threads of the library calls (inflate and deflate)
DEFAULT_THREADS = INFLATE_THREADS


def compress_file(src: str, dst: str = None, level: int = 9, threads: int = None, index: bool = True,
                  batch_blocks: int = 4096) -> Tuple[str, str]:
    """`bgzip -i -I dst.gzi -l 9 -c src > dst`: writes dst (default src + '.bgz') via .tmp + rename and,
    if `index`, dst + '.gzi'.  Returns (dst, gzi path or None).  The blocks are deflated by the library on native
    threads (pk_bgzf_deflate), `batch_blocks` at a time."""
    dst = dst or src + ".bgz"
    tmp = dst + ".tmp"
    entries: List[Tuple[int, int]] = []
    c_off = u_off = 0
    size = os.path.getsize(src)
    raw = np.memmap(src, dtype=np.uint8, mode="r") if size else np.zeros(0, np.uint8)
    step = BLOCK_INPUT * batch_blocks
    with open(tmp, "wb") as fout:
        for at in range(0, size, step):
            chunk = np.ascontiguousarray(raw[at:at + step])
            packed, sizes = _native().bgzf_deflate(chunk, level, BLOCK_INPUT, threads or INFLATE_THREADS)
            for i, z in enumerate(sizes):
                if c_off:                                    # htslib lists every block except the first
                    entries.append((c_off, u_off))
                c_off += int(z)
                u_off += min(BLOCK_INPUT, chunk.size - i * BLOCK_INPUT)
            fout.write(packed.data)
        fout.write(EOF_BLOCK)
    os.replace(tmp, dst)
    gzi = None
    if index:
        gzi = dst + ".gzi"
        with open(gzi, "wb") as fh:
            fh.write(struct.pack("<Q", len(entries)))
            for c, u in entries:
                fh.write(struct.pack("<QQ", c, u))
    return dst, gzi


def read_gzi(path: str) -> List[Tuple[int, int]]:
    """gzireader.py:12-34."""
    with open(path, "rb") as fh:
        (n,) = struct.unpack("<Q", fh.read(8))
        return [struct.unpack("<QQ", fh.read(16)) for _ in range(n)]


def scan_blocks(buf) -> List[Tuple[int, int]]:
    """[(offset, block_size)] of every BGZF block in a bytes-like, or [] if the data is not BGZF."""
    blocks, pos, n = [], 0, len(buf)
    while pos < n:
        if n - pos < 18 or buf[pos:pos + 4] != b"\x1f\x8b\x08\x04":
            return []
        xlen = struct.unpack_from("<H", buf, pos + 10)[0]
        x, end, bsize = pos + 12, pos + 12 + xlen, None
        while x + 4 <= end:
            si1, si2, slen = buf[x], buf[x + 1], struct.unpack_from("<H", buf, x + 2)[0]
            if si1 == 66 and si2 == 67 and slen == 2:
                bsize = struct.unpack_from("<H", buf, x + 4)[0] + 1
            x += 4 + slen
        if bsize is None or pos + bsize > n:
            return []
        blocks.append((pos, bsize))
        pos += bsize
    return blocks


def _native():
    from . import _lib
    return _lib


def decompress_file(path: str, expected_size: int = None, threads: int = None) -> np.ndarray:
    """Whole file -> uint8 array.  BGZF blocks are inflated in parallel straight into the result; any
    other gzip stream goes through gzip.open like tools.py:300-302."""
    if os.path.getsize(path) and is_bgzf(path):
        c_offs, c_sizes, u_offs = block_index(path)
        raw = np.memmap(path, dtype=np.uint8, mode="r")
        data = np.empty(int(u_offs[-1]), dtype=np.uint8)
        _native().bgzf_inflate(raw, c_offs, c_sizes, u_offs, data, threads or INFLATE_THREADS)
    else:
        with gzip.open(path, "rb") as fh:
            data = np.frombuffer(fh.read(), dtype=np.uint8)
    if expected_size is not None and data.size != expected_size:
        raise AssertionError(f"{path}: {data.size} bytes after inflating, expected {expected_size}")
    return data


_INDEX_CACHE = {}


def block_index(path: str):
    """(compressed offsets, compressed sizes, uncompressed offsets [n_blocks + 1]) of every data block of a BGZF file, as
    int64 arrays.  From the `.gzi` beside the file when there is one (gzireader.py:12-34 layout: every block but the
    first), else from one walk over the block headers (no inflation).  Cached per (path, size, mtime): a sharded merge
    asks for many ranges of the same file."""
    st = os.stat(path)
    key = (os.path.abspath(path), st.st_size, st.st_mtime_ns)
    hit = _INDEX_CACHE.get(key)
    if hit is not None:
        return hit
    raw = np.memmap(path, dtype=np.uint8, mode="r")
    buf = memoryview(raw)
    gzi = path + ".gzi"
    if os.path.exists(gzi) and os.path.getmtime(gzi) >= st.st_mtime:
        entries = [(0, 0)] + read_gzi(gzi)
        c_offs = np.array([c for c, _ in entries], dtype=np.int64)
        u_offs = np.array([u for _, u in entries], dtype=np.int64)
        last = int(c_offs[-1])                               # the last indexed block: its size is in its own header
        last_size = struct.unpack_from("<H", buf, last + 16)[0] + 1
        c_sizes = np.append(np.diff(c_offs), last_size)
        last_isize = struct.unpack_from("<I", buf, last + last_size - 4)[0]
        u_offs = np.append(u_offs, u_offs[-1] + last_isize)
    else:
        try:
            c_offs, c_sizes, isizes = _native().bgzf_scan(raw)
        except ValueError as exc:
            raise OSError(f"{path}: {exc}") from None
        keep = isizes > 0                                    # the empty end-of-file block (and any other empty block) is no data
        if keep.any():
            c_offs, c_sizes, isizes = c_offs[keep], c_sizes[keep], isizes[keep]
        else:
            c_offs, c_sizes, isizes = c_offs[:1], c_sizes[:1], isizes[:1]
        u_offs = np.concatenate(([0], np.cumsum(isizes)))
    if len(_INDEX_CACHE) > 64:
        _INDEX_CACHE.clear()
    _INDEX_CACHE[key] = (c_offs, c_sizes, u_offs)
    return _INDEX_CACHE[key]


def decompress_range(path: str, lo: int, hi: int, threads: int = None) -> Tuple[np.ndarray, int]:
    """Uncompressed bytes [lo, hi) of a gzip / BGZF file -> (array, bytes inflated to get them).

    BGZF: the blocks overlapping the range are found in the (cached) block index -- the `.gzi` beside the file, or one
    walk over the block headers -- and only they are inflated, in parallel.  Any other gzip stream is read
    sequentially up to `hi`."""
    if hi <= lo:
        return np.zeros(0, dtype=np.uint8), 0
    if not is_bgzf(path):
        with gzip.open(path, "rb") as fh:
            left = lo
            while left:                                      # no seeking in a plain deflate stream
                got = fh.read(min(left, 1 << 24))
                if not got:
                    break
                left -= len(got)
            data = np.frombuffer(fh.read(hi - lo), dtype=np.uint8)
        return data, hi
    c_offs, c_sizes, u_offs = block_index(path)
    if hi > int(u_offs[-1]):
        raise OSError(f"{path}: ends before byte {hi}")
    first = int(np.searchsorted(u_offs, lo, side="right")) - 1
    end = int(np.searchsorted(u_offs, hi, side="left"))       # blocks [first, end) overlap [lo, hi)
    raw = np.memmap(path, dtype=np.uint8, mode="r")
    base = int(u_offs[first])
    span = np.empty(int(u_offs[end]) - base, dtype=np.uint8)
    _native().bgzf_inflate(raw, c_offs[first:end], c_sizes[first:end], u_offs[first:end + 1], span, threads or INFLATE_THREADS)
    return span[lo - base: hi - base], int(span.size)


def iter_pieces(path: str, piece_bytes: int, threads: int = None):
    """The inflated stream of a BGZF file in pieces of about `piece_bytes` (whole blocks), in order.  Piece i + 1 is
    inflated (block-parallel, native threads) while the caller still works on piece i -- the indexer feeds piece i to the
    GPU meanwhile -- so a bgzipped FASTA never sits inflated in host memory as a whole."""
    import threading
    c_offs, c_sizes, u_offs = block_index(path)
    n_blocks = len(c_offs)
    raw = np.memmap(path, dtype=np.uint8, mode="r")
    cuts, i = [], 0
    while i < n_blocks:
        j = int(np.searchsorted(u_offs, u_offs[i] + piece_bytes, side="right")) - 1
        j = min(n_blocks, max(j, i + 1))
        cuts.append((i, j))
        i = j

    def start(cut):
        i, j = cut
        out = np.empty(int(u_offs[j] - u_offs[i]), dtype=np.uint8)
        box = {}

        def run():
            try:
                _native().bgzf_inflate(raw, c_offs[i:j], c_sizes[i:j], u_offs[i:j + 1], out, threads or INFLATE_THREADS)
            except BaseException as exc:                    # re-raised by the consumer
                box["exc"] = exc
        t = threading.Thread(target=run, daemon=True)
        t.start()
        return out, t, box

    ahead = start(cuts[0]) if cuts else None
    for n in range(len(cuts)):
        out, t, box = ahead
        t.join()
        if "exc" in box:
            raise box["exc"]
        ahead = start(cuts[n + 1]) if n + 1 < len(cuts) else None       # runs while the caller holds `out`
        if out.size:
            yield out


def is_bgzf(path: str) -> bool:
    with open(path, "rb") as fh:
        head = fh.read(18)
    return len(head) == 18 and head[:4] == b"\x1f\x8b\x08\x04" and head[12:14] == b"BC"


def main(argv=None) -> None:
    """`python -m pykmer_amd.bgzf file.kin` -> file.kin.bgz + file.kin.bgz.gzi (the README's bgzip step)."""
    import argparse
    ap = argparse.ArgumentParser(description="BGZF-compress a file the way `bgzip -i -l 9` does")
    ap.add_argument("file")
    ap.add_argument("-l", "--level", type=int, default=9)
    ap.add_argument("-@", "--threads", type=int, default=DEFAULT_THREADS)
    ap.add_argument("-d", "--decompress", action="store_true")
    ap.add_argument("--keep", action="store_true", help="keep the input (the README recipe removes it)")
    a = ap.parse_args(argv)
    if a.decompress:
        out = a.file[:-4] if a.file.endswith(".bgz") else a.file + ".out"
        decompress_file(a.file, threads=a.threads).tofile(out)
    else:
        compress_file(a.file, level=a.level, threads=a.threads)
        if not a.keep:
            os.remove(a.file)


if __name__ == "__main__":
    main()
