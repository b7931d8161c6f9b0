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

FEED_BYTES = 1 << 30            # host -> device staging granularity
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


def _pieces(input_file: str):
    """The decompressed byte stream in FEED_BYTES pieces.  A BGZF-compressed FASTA (bgzip output) is
    inflated block-parallel in one go when it is small enough to keep; everything else streams."""
    if input_file.endswith((".gz", ".bgz")):
        from . import bgzf
        if bgzf.is_bgzf(input_file) and os.path.getsize(input_file) * 6 <= RESIDENT_LIMIT:
            print(f"READING FASTA FROM BGZF {input_file}")
            data = bgzf.decompress_file(input_file)
            for off in range(0, data.size, FEED_BYTES):
                yield data[off:off + FEED_BYTES].tobytes()
            return
    with _open_input(input_file) as fh:
        while True:
            piece = fh.read(FEED_BYTES)
            if not piece:
                return
            yield piece


def count_file(input_file: str, kmer_len: int, device: int = 0):
    """Streams one FASTA file through the GPU indexer.

    Returns (table u8[4^k] on the host, summary dict, all_records [(name, seq_len, n_valid)])."""
    kept, total = [], 0
    with _lib.Indexer(kmer_len, device=device) as ix:
        for piece in _pieces(input_file):
            ix.feed(piece)
            total += len(piece)
            if kept is not None:
                kept.append(piece)
                if total > RESIDENT_LIMIT:
                    kept = None
        fin = ix.finish()
        recs = ix.records(fin["n_records"])
        table = ix.table_to_host()
        fin["timings"] = ix.timings()
    if kept is not None:
        blob = kept[0] if len(kept) == 1 else b"".join(kept)
        raw = [blob[int(r["name_off"]): int(r["name_off"]) + int(r["name_len"])] for r in recs]
    else:
        raw = _names_from_stream(input_file, recs)
    everything = [(n.decode("utf-8", "replace"), int(r["seq_len"]), int(r["n_valid_kmers"])) for n, r in zip(raw, recs)]
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
    header._init_clean(overwrite=overwrite)                    # indexer.py:327 (the sparse tmp file is written whole below)

    # the two sha256 sums of the .kin.json (tools.py:280,283) run beside the work instead of after it: the input
    # file is hashed while the GPU counts, the table while it is written (hashlib releases the GIL)
    import concurrent.futures
    import hashlib
    pool = concurrent.futures.ThreadPoolExecutor(max_workers=2)
    input_sum = pool.submit(gen_checksum, header.input_file_path, 1 << 22)
    table, fin, everything = count_file(input_file, kmer_len, device=device)
    for num, (name, seq_len, n_valid) in enumerate(everything):
        print(f"{num + 1:03d} {name} {seq_len:15,d}")           # indexer.py:136
    header.timer.update(fin["total_bp"])
    header.num_kmers = fin["num_kmers"]
    # only records that produced a k-mer are listed (indexer.py:349-351); tuples serialise as JSON lists
    header.chromosomes = [(name, seq_len) for name, seq_len, n_valid in everything if n_valid]
    print(f"project_name {header.project_name} kmer_len {header.kmer_len:15,d} num_kmers {header.num_kmers:15,d} "
          f"kmer_size {header.kmer_size:15,d} max_size {header.max_size:15,d}")

    table_sum = pool.submit(lambda: hashlib.sha256(memoryview(table)).hexdigest())
    with open(header.index_tmp_file, "wb") as fh:              # tools.py:333-341: exactly 4^k bytes, no header
        table.tofile(fh)
    print("  indexing finished. creating header")
    checksums = {"input": input_sum.result(), "output": table_sum.result()}
    pool.shutdown()
    header.write_metadata_index_tmp_file(hist256=fin["hist256"], checksums=checksums)   # asserts num_kmers and chromosomes (tools.py:367-368)
    print("renaming")
    os.rename(header.index_tmp_file, header.index_file_root)  # indexer.py:412
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
