"""Indexer host side: FASTA file in, `<fa>.<kk>.kin` + `<fa>.<kk>.kin.json` out.

Same entry points and outputs as the reference's indexer.py (create_fasta_index indexer.py:299-414,
read_fasta_index :416-444, main :475-495); the work the reference does in parse_fasta / gen_kmers /
process_kmers (indexer.py:45-297) happens in HBM behind pk_indexer_* (include/pykmer_hip.h).
"""
import gzip
import os
import sys
from typing import List, Union

import numpy as np

from . import _lib
from .header import Header, gen_checksum

import time

FEED_BYTES = 1 << 30            # host -> device staging granularity
_T0 = time.perf_counter()


def _mark(what: str) -> None:
    """PK_TIMING=1: where the wall time of a CLI run goes (seconds since this module was imported), on stderr."""
    if os.environ.get("PK_TIMING"):
        print(f"[pk timing] {time.perf_counter() - _T0:7.3f} s  {what}", file=sys.stderr)

RESIDENT_LIMIT = 16 << 30       # inputs up to this size (decompressed) are kept in host memory for the header names


def _open_input(input_file: str):
    """indexer.py:101-128: .gz / .bgz through gzip, anything else as is (bytes here, text there)."""
    if input_file.endswith((".gz", ".bgz")):
        print(f"READING FASTA FROM PYGZ {input_file}")
        return gzip.open(input_file, "rb")
    print(f"READING FASTA FROM {input_file}")
    return open(input_file, "rb")


def _names_from_stream(input_file: str, records: np.ndarray) -> List[bytes]:
    """Second pass for inputs too large to keep: pull the header byte ranges out of the stream."""
    want = [(int(r["name_off"]), int(r["name_len"])) for r in records]
    names, pos, nxt, tail = [], 0, 0, b""
    with _open_input(input_file) as fh:
        while nxt < len(want):
            piece = fh.read(1 << 24)
            if not piece:
                break
            buf, start = tail + piece, pos - len(tail)
            while nxt < len(want) and want[nxt][0] + want[nxt][1] <= start + len(buf):
                off, ln = want[nxt]
                names.append(buf[off - start: off - start + ln])
                nxt += 1
            keep = min(len(buf), 1 << 16)
            if nxt < len(want):
                keep = max(keep, start + len(buf) - want[nxt][0])
            tail, pos = buf[len(buf) - keep:], pos + len(piece)
    assert len(names) == len(want), "input changed between passes"
    return names


GZ_PIECE = 256 << 20            # compressed inputs: inflated bytes per piece (piece i + 1 inflates while piece i is fed)


class _Input:
    """The decompressed byte stream of a FASTA file, in pieces (anything with the buffer protocol), and
    afterwards the header texts at given byte ranges of that stream.

    A plain file is mapped, not read: the pieces are views of the page cache that the library copies to HBM from
    several threads, and the names are sliced from the mapping.  A BGZF-compressed FASTA (bgzip output) is inflated
    block-parallel, GZ_PIECE bytes at a time, the next piece while the GPU takes the current one (bgzf.iter_pieces).
    Other gzip streams are read through gzip.open like the reference does (indexer.py:112-115), by a producer thread
    that stays one piece ahead of the feed.  Inflated pieces are kept for the names while they fit RESIDENT_LIMIT
    (`keep`; the later address slices of a sliced count never ask for names), otherwise the stream is inflated a
    second time for them."""

    def __init__(self, input_file: str, keep: bool = True, quiet: bool = False):
        self.path = input_file
        self.gz = input_file.endswith((".gz", ".bgz"))
        self.quiet = quiet
        self.kept, self.kept_bytes = ([] if keep else None), 0   # compressed inputs: pieces in order, or None once they no longer fit

    def _say(self, text: str) -> None:
        if not self.quiet:
            print(text)

    def _keep(self, piece) -> None:
        if self.kept is not None:
            self.kept.append(piece)
            self.kept_bytes += len(piece)
            if self.kept_bytes > RESIDENT_LIMIT:
                self.kept = None

    def _gzip_pieces(self):
        """gzip.open read by a producer thread, one piece ahead of the consumer."""
        import queue
        import threading
        q: "queue.Queue" = queue.Queue(maxsize=1)

        def produce():
            try:
                with gzip.open(self.path, "rb") as fh:
                    while True:
                        piece = fh.read(GZ_PIECE)
                        q.put(piece)
                        if not piece:
                            return
            except BaseException as exc:                    # handed to the consumer, which raises it
                q.put(exc)
        t = threading.Thread(target=produce, daemon=True)
        t.start()
        while True:
            piece = q.get()
            if isinstance(piece, BaseException):
                raise piece
            if not piece:
                break
            yield piece
        t.join()

    def pieces(self):
        if not self.gz:
            self._say(f"READING FASTA FROM {self.path}")
            if os.path.getsize(self.path) == 0:
                return
            import mmap
            with open(self.path, "rb") as fh:
                mm = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
            arr = np.frombuffer(mm, dtype=np.uint8)
            for off in range(0, arr.size, FEED_BYTES):
                yield arr[off:off + FEED_BYTES]
            del arr
            try:
                mm.close()
            except BufferError:                             # a caller still holds a piece: the mapping goes with it
                pass
            return
        from . import bgzf
        if bgzf.is_bgzf(self.path):
            self._say(f"READING FASTA FROM BGZF {self.path}")
            source = bgzf.iter_pieces(self.path, GZ_PIECE)
        else:
            self._say(f"READING FASTA FROM PYGZ {self.path}")
            source = self._gzip_pieces()
        for piece in source:
            self._keep(piece)
            yield piece

    def names(self, records: np.ndarray) -> List[bytes]:
        spans = [(int(r["name_off"]), int(r["name_len"])) for r in records]
        if not spans:
            return []
        if not self.gz:
            import mmap
            with open(self.path, "rb") as fh, mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ) as mm:
                return [mm[off:off + ln] for off, ln in spans]
        if self.kept is not None:
            # header texts out of the kept pieces: a name may straddle two of them
            starts = np.cumsum([0] + [len(p) for p in self.kept])
            out = []
            for off, ln in spans:
                i = int(np.searchsorted(starts, off, side="right")) - 1
                got = bytes(memoryview(self.kept[i])[off - starts[i]: off - starts[i] + ln])
                while len(got) < ln and i + 1 < len(self.kept):
                    i += 1
                    got += bytes(memoryview(self.kept[i])[: ln - len(got)])
                out.append(got)
            return out
        return _names_from_stream(self.path, records)


TABLE_SLICE = 64 << 20          # the table leaves HBM in slices: slice i is hashed while slice i+1 crosses PCIe


def n_address_slices(kmer_len: int) -> int:
    """How many address slices the 4^k table is counted in: one indexer holds 2^34 addresses (16 GiB), so k <= 17 is one
    slice, k = 19 sixteen (README.md:51-52: the reference never ran it).  PK_SLICES forces a (power-of-two) number."""
    forced = int(os.environ.get("PK_SLICES", "0"))
    return forced or max(1, 4 ** kmer_len >> 34)


def count_file(input_file: str, kmer_len: int, device: int = 0, table_file: str = None, devices=None):
    """Streams one FASTA file through the GPU indexer.

    Returns (table, summary dict, all_records [(name, seq_len, n_valid)]).  With `table_file` the 4^k-byte
    table is written straight into that file (tools.py:333-341: exactly 4^k bytes, no header) piece by piece as
    it arrives, hashed on the way (summary["table_sha256"]), and `table` is the mapping of the file; otherwise
    `table` is a host array.

    A table of more than 2^34 addresses (or PK_SLICES) is counted one address slice at a time: every slice streams
    the whole input and keeps the k-mers of its own range (option A).  `devices` spreads the slices over
    several GPUs, one host thread each; the file is filled in address order, so the hash is still one stream."""
    import hashlib
    import queue
    import threading
    from concurrent.futures import ThreadPoolExecutor
    devices = tuple(devices) if devices else (device,)
    n = 4 ** kmer_len
    n_slices = n_address_slices(kmer_len)
    size = n // n_slices
    source = _Input(input_file)
    if table_file is None:
        table = np.empty(n, dtype=np.uint8)
    else:
        with open(table_file, "wb") as fh:
            fh.truncate(n)
        table = np.memmap(table_file, dtype=np.uint8, mode="r+", shape=(n,))
    digest = hashlib.sha256()
    todo: "queue.Queue" = queue.Queue(maxsize=8)

    def hasher():
        while True:
            part = todo.get()
            if part is None:
                return
            digest.update(part)                                # hashlib releases the GIL

    def count_slice(s: int):
        """Counts slice s and copies it into table[s * size : (s + 1) * size]; the first slice's pieces feed the hasher
        as they land, later slices are hashed whole once it is their turn (the file is hashed in address order)."""
        src = source if s == 0 else _Input(input_file, keep=False, quiet=True)   # only slice 0's names and messages are used
        _mark("creating the indexer (library load, HIP start-up, table and workspace allocation)")
        with _lib.Indexer(kmer_len, device=devices[s % len(devices)], slice_index=s, n_slices=n_slices) as ix:
            _mark("indexer ready")
            for piece in src.pieces():
                ix.feed(piece)
            fin = ix.finish()
            _mark("text counted")
            fin["records"] = ix.records(fin["n_records"]) if s == 0 else None
            fin["timings"] = ix.timings()
            for off in range(0, size, TABLE_SLICE):
                part = table[s * size + off: s * size + min(size, off + TABLE_SLICE)]
                ix.table_slice_to_host(part, off)
                if s == 0:
                    todo.put(part)
            _mark("table slice on the host")
        return fin

    worker = threading.Thread(target=hasher)
    worker.start()
    fin, hist = None, np.zeros(256, dtype=np.uint64)
    try:
        with ThreadPoolExecutor(max_workers=len(devices)) as pool:
            for s, part_fin in enumerate(pool.map(count_slice, range(n_slices))):        # results in slice order
                hist += part_fin["hist256"]
                if s == 0:
                    fin = part_fin
                else:
                    assert part_fin["num_kmers"] == fin["num_kmers"] and part_fin["total_bp"] == fin["total_bp"]
                    for off in range(0, size, TABLE_SLICE):
                        todo.put(table[s * size + off: s * size + min(size, off + TABLE_SLICE)])
    finally:
        todo.put(None)
        worker.join()
    _mark("table hashed")
    if table_file is not None:
        table.flush()
    _mark("table flushed")
    fin["hist256"] = hist
    fin["table_sha256"] = digest.hexdigest()
    fin["n_slices"] = n_slices
    recs = fin.pop("records")
    raw = source.names(recs)                                   # header text: byte ranges of the decompressed stream
    everything = [(nm.decode("utf-8", "replace"), int(r["seq_len"]), int(r["n_valid_kmers"])) for nm, r in zip(raw, recs)]
    return table, fin, everything


def create_fasta_index(
        project_name: str,
        sample_name: str,
        input_file: str,
        kmer_len: int,
        overwrite: bool,
        flush_every: int = Header.DEFAULT_FLUSH_EVERY,
        min_frag_size: int = Header.DEFAULT_MIN_FRAG_SIZE,
        max_frag_size: int = Header.DEFAULT_MAX_FRAG_SIZE,
        buffer_size: int = Header.DEFAULT_BUFFER_SIZE,
        debug: bool = False,
        device: int = 0) -> Header:
    """indexer.py:299-414.  flush_every / frag sizes only travel into the .kin.json (the table is
    independent of batching, indexer.py:262); the 4^k table is resident in HBM instead."""
    header = Header(project_name, sample_name=sample_name, input_file=input_file, kmer_len=kmer_len,
                    flush_every=flush_every, min_frag_size=min_frag_size, max_frag_size=max_frag_size,
                    buffer_size=buffer_size, device=device)
    print(f"project_name {header.project_name} sample_name {header.sample_name} kmer_len {header.kmer_len:15,d} "
          f"kmer_size {header.kmer_size:15,d} max_size {header.max_size:15,d} bytes {header.max_size // 1024:15,d} Kb "
          f"{header.max_size // 1024 // 1024:15,d} Mb {header.max_size // 1024 // 1024 // 1024:15,d} Gb")
    header._init_clean(overwrite=overwrite)                    # indexer.py:327 (the tmp file is written whole below)

    # the two sha256 sums of the .kin.json (tools.py:280,283) run beside the work instead of after it: the input
    # file is hashed in a thread while the GPU counts, the table slice by slice as it comes back from HBM
    import concurrent.futures
    pool = concurrent.futures.ThreadPoolExecutor(max_workers=1)
    input_sum = pool.submit(gen_checksum, header.input_file_path, 1 << 22)
    devices = tuple(int(d) for d in os.environ.get("PK_DEVICES", str(device)).split(",") if d != "")
    table, fin, everything = count_file(input_file, kmer_len, device=device, table_file=header.index_tmp_file, devices=devices)
    del table
    for num, (name, seq_len, n_valid) in enumerate(everything):
        print(f"{num + 1:03d} {name} {seq_len:15,d}")           # indexer.py:136
    header.timer.update(fin["total_bp"])
    header.num_kmers = fin["num_kmers"]
    # only records that produced a k-mer are listed (indexer.py:349-351); tuples serialise as JSON lists
    header.chromosomes = [(name, seq_len) for name, seq_len, n_valid in everything if n_valid]
    print(f"project_name {header.project_name} kmer_len {header.kmer_len:15,d} num_kmers {header.num_kmers:15,d} "
          f"kmer_size {header.kmer_size:15,d} max_size {header.max_size:15,d}")
    print("  indexing finished. creating header")
    checksums = {"input": input_sum.result(), "output": fin["table_sha256"]}
    _mark("input hashed")
    pool.shutdown()
    header.write_metadata_index_tmp_file(hist256=fin["hist256"], checksums=checksums)   # asserts num_kmers and chromosomes (tools.py:367-368)
    print("renaming")
    os.rename(header.index_tmp_file, header.index_file_root)  # indexer.py:412
    _mark("files renamed")
    print("done")
    return header


def read_fasta_index(project_name: str, input_file: Union[str, None] = None, kmer_len: Union[int, None] = None,
                     index_file: Union[str, None] = None, debug: bool = False, device: int = 0) -> Header:
    """indexer.py:416-444: load the metadata and verify the table against it (stats recomputed on the GPU)."""
    header = Header(project_name, input_file=input_file, kmer_len=kmer_len, index_file=index_file, device=device)
    header.read_metadata()
    print(header)
    header.check_data_index()
    print("OK")
    return header


def main(argv: List[str] = None) -> None:
    """indexer.py:475-495: `indexer.py <fasta[.gz|.bgz]> <sample_name> <k>`; also the README's `indexer.py <fasta> <k>`."""
    argv = list(sys.argv[1:] if argv is None else argv)
    if len(argv) == 2 and argv[1].isdigit():                   # README.md:22
        argv = [argv[0], os.path.basename(argv[0]), argv[1]]
    if len(argv) != 3:
        print("usage: indexer.py <input.fa[.gz|.bgz]> <sample_name> <kmer_len>")
        sys.exit(1)
    input_file, sample_name, kmer_len = argv[0], argv[1], int(argv[2])
    print(f"project_name {input_file:s} input_file {input_file:s} sample_name {sample_name:s} kmer_len {kmer_len:15,d}")
    create_fasta_index(input_file, sample_name, input_file, kmer_len, buffer_size=2 ** 16, overwrite=True, debug=False,
                       device=int(os.environ.get("PK_DEVICE", "0")))
    print()
