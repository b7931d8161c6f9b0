"""Host-side format layer: file naming, 4^k sizing, .kin.json metadata, stats, pair distance.

Mirrors the reference's tools.py interface (HeaderVars / Header / Timer / gen_checksum; citations are
file:line into the reference) so that its callers and tests read the same, but every scan over a
table -- statistics (tools.py:246-263) and pair distance (tools.py:439-493) -- goes through the
C-ABI to the HIP kernels.  There is no CPU fallback: without the library and a GPU those calls raise.
"""
import datetime
import gzip
import hashlib
import io
import json
import math
import os
import socket
from collections import OrderedDict
from typing import Any, BinaryIO, Dict, Iterator, List, Tuple, Union

import numpy as np

from . import _lib


class Timer:
    """tools.py:24-64."""

    def __init__(self):
        self.time_begin = datetime.datetime.now()
        self.time_last = self.time_begin
        self.val_last = 0
        self.val_delta = 0
        self.time_ela = datetime.timedelta(seconds=0)
        self.time_delta = datetime.timedelta(seconds=0)
        self.time_ela_s = "none"
        self.time_delta_s = "none"
        self.speed_ela = 0
        self.speed_delta = 0

    @property
    def time_delta_seconds(self):
        return (datetime.datetime.now() - self.time_last).total_seconds()

    def update(self, val):
        now = datetime.datetime.now()
        self.time_ela = now - self.time_begin
        self.time_delta = now - self.time_last
        self.time_ela_s = str(self.time_ela).split(".", 2)[0]
        self.time_delta_s = str(self.time_delta).split(".", 2)[0]
        self.val_delta = val - self.val_last
        self.speed_ela = int(val // max(self.time_ela.total_seconds(), 1e-9))
        self.speed_delta = int(self.val_delta // max(self.time_delta.total_seconds(), 1e-9))
        self.time_last = now
        self.val_last = val

    def __str__(self):
        return (f"ela   time {self.time_ela_s} val {self.val_last:15,d} speed {self.speed_ela:15,d}\n"
                f"delta time {self.time_delta_s} val {self.val_delta:15,d} speed {self.speed_delta:15,d}")


class HeaderVars:
    """tools.py:67-106: constants and the key lists that define the .kin.json schema."""
    HEADER_VER: str = "KMER001"
    HEADER_FIXED: List[str] = ["file_ver", "kmer_size", "data_size", "max_size"]
    HEADER_DATA: List[str] = [
        "project_name", "kmer_len", "flush_every", "frag_size",
        "input_file_name", "input_file_path",
        "input_file_size", "input_file_ctime", "input_file_cheksum",
        "output_file_size", "output_file_ctime", "output_file_cheksum",
        "num_kmers", "chromosomes",
        "creation_time_start", "creation_time_end", "creation_duration", "creation_speed",
        "hostname", "checksum_script",
        "hist", "hist_sum", "hist_count", "hist_min", "hist_max",
        "vals_sum", "vals_count", "vals_min", "vals_max",
    ]
    NOT_LEAN: List[str] = ["chromosomes"]

    IND_EXT: str = "kin"
    DESC_EXT: str = "json"
    TMP: str = "tmp"
    COMP_EXT: str = "bgz"

    DEFAULT_FLUSH_EVERY: int = 100_000_000
    DEFAULT_MIN_FRAG_SIZE: int = 500_000_000
    DEFAULT_MAX_FRAG_SIZE: int = 1_000_000_000
    DEFAULT_BUFFER_SIZE: int = io.DEFAULT_BUFFER_SIZE

    DEFAULT_MIN_COUNT = 1
    DEFAULT_MAX_COUNT = 255
    DEFAULT_BLOCK_SIZE = 100_000_000


def stats_from_hist256(hist256) -> Dict[str, Any]:
    """The nine statistics of Header.update_stats (tools.py:250-263) from a 256-bin value histogram.

    np.histogram(arr, bins=255, range=(1,255)) counts value v in bin v-1 (255 lands in the last,
    closed bin), i.e. hist == hist256[1:]."""
    h = np.asarray(hist256, dtype=np.uint64)
    hist = h[1:]
    v = np.arange(256, dtype=np.uint64)
    present = np.nonzero(h)[0]
    return {
        "hist": [int(x) for x in hist],
        "hist_sum": int(hist.sum()),
        "hist_count": int(np.count_nonzero(hist)),
        "hist_min": int(hist.min()),
        "hist_max": int(hist.max()),
        "vals_sum": int((h * v).sum()),
        "vals_count": int(hist.sum()),
        "vals_min": int(present.min()) if present.size else 0,
        "vals_max": int(present.max()) if present.size else 0,
    }


class Header(HeaderVars):
    """tools.py:110-545.  `device` selects the GPU used for statistics / distance scans."""

    def __init__(self,
                 project_name: str,
                 input_file: Union[str, None] = None,
                 kmer_len: Union[int, None] = None,
                 index_file: Union[str, None] = None,
                 frag_size: int = None,
                 flush_every: int = HeaderVars.DEFAULT_FLUSH_EVERY,
                 min_frag_size: int = HeaderVars.DEFAULT_MIN_FRAG_SIZE,
                 max_frag_size: int = HeaderVars.DEFAULT_MAX_FRAG_SIZE,
                 buffer_size: int = HeaderVars.DEFAULT_BUFFER_SIZE,
                 sample_name: Union[str, None] = None,
                 device: int = 0):
        self.project_name = project_name
        self.sample_name = sample_name          # indexer.py:313 passes it; the reference's Header rejects it
        self.input_file_name = os.path.basename(input_file) if input_file else input_file
        self.input_file_path = os.path.abspath(input_file) if input_file else input_file
        self.kmer_len = kmer_len
        self.flush_every = flush_every
        self._buffer_size = buffer_size
        self._device = device

        self.input_file_size = None
        self.input_file_ctime = None
        self.input_file_cheksum = None
        self.output_file_size = None
        self.output_file_ctime = None
        self.output_file_cheksum = None
        self.num_kmers = None
        self.chromosomes = None
        self.timer = Timer()
        self.creation_time_start = None
        self.creation_time_end = None
        self.creation_duration = None
        self.creation_speed = None
        self.hostname = None
        self.checksum_script = None
        self.hist = None
        self.hist_sum = None
        self.hist_count = None
        self.hist_min = None
        self.hist_max = None
        self.vals_sum = None
        self.vals_count = None
        self.vals_min = None
        self.vals_max = None

        if index_file is not None:
            self._parse_index_file_name(index_file)
            self.read_metadata()

        assert self.kmer_len                                  # tools.py:165-167
        assert self.kmer_len > 0
        assert self.kmer_len % 2 == 1

        if frag_size is not None:
            self.frag_size = frag_size
        else:                                                 # tools.py:173-182, verbatim arithmetic
            fs = self.data_size // 10
            if max_frag_size is not None and fs > max_frag_size:
                fs = max_frag_size
            if min_frag_size is not None and fs < min_frag_size:
                fs = min_frag_size
            if fs > self.data_size:
                fs = self.data_size
            if (self.data_size % fs) < (self.data_size // 2):
                pieces = self.data_size // fs
                fs = self.data_size // (pieces + 1)
                fs = fs + (pieces + 1) + 1
                fs = int(math.ceil(fs / 1_000) * 1_000)
            self.frag_size = fs

    # ---- names and sizes (tools.py:185-217)
    @property
    def index_file(self) -> str:
        return f"{self.index_file_root}.bgz" if os.path.exists(f"{self.index_file_root}.bgz") else self.index_file_root

    @property
    def index_file_basename(self) -> str:
        return os.path.basename(self.index_file)

    @property
    def index_file_root(self) -> str:
        return f"{self.input_file_path}.{self.kmer_len:02d}.{self.IND_EXT}"

    @property
    def index_tmp_file(self) -> str:
        return f"{self.index_file_root}.{self.TMP}"

    @property
    def metadata_file(self) -> str:
        return f"{self.index_file_root}.{self.DESC_EXT}"

    @property
    def kmer_size(self) -> int:
        return 4 ** self.kmer_len

    @property
    def data_size(self) -> int:
        return self.kmer_size

    @property
    def max_size(self) -> int:
        return self.data_size

    @property
    def file_ver(self) -> str:
        return self.HEADER_VER

    @property
    def max_val(self) -> int:
        return 255

    def _parse_index_file_name(self, index_file: str) -> None:
        """tools.py:220-238: `<input>.<KK>.kin[.bgz]` -> input path and k."""
        if index_file.endswith("." + self.COMP_EXT):
            index_file = index_file[:-(len(self.COMP_EXT) + 1)]
        ext_len = 2 + 1 + len(self.IND_EXT) + 1
        ext = index_file[-(ext_len - 1):]
        if self.input_file_name is None:
            stem = index_file[:-ext_len]
            self.input_file_name = os.path.basename(stem)
            self.input_file_path = os.path.abspath(stem)
        if self.kmer_len is None:
            self.kmer_len = int(ext[:2])

    # ---- statistics (tools.py:246-271): the table scan runs on the GPU
    def update_stats_from_hist256(self, hist256) -> None:
        for k, v in stats_from_hist256(hist256).items():
            setattr(self, k, v)

    def update_stats(self, fhd: BinaryIO) -> None:
        print("updating stats")
        table = np.frombuffer(fhd.read(self.data_size), dtype=np.uint8)
        assert table.size == self.data_size
        self.update_stats_from_hist256(_lib.table_stats(table, device=self._device))

    def update_stats_index_file(self) -> None:
        for fhd in self.open_index_file():
            self.update_stats(fhd)

    def update_stats_index_tmp_file(self) -> None:
        for fhd in self.open_index_tmp_file():
            self.update_stats(fhd)

    def update_metadata(self, index_file: str, checksums=None) -> None:
        """tools.py:273-291.  `checksums` = {"input": hex, "output": hex} when the caller already hashed the
        very bytes of the two files (the indexer does, in threads, while the GPU counts and the table is
        written); otherwise both files are read back and hashed here, as the reference does."""
        print("updating metadata")
        checksums = checksums or {}
        self.input_file_size = os.path.getsize(self.input_file_path)
        self.input_file_ctime = os.path.getctime(self.input_file_path)
        self.input_file_cheksum = checksums.get("input") or gen_checksum(self.input_file_path)
        self.output_file_size = os.path.getsize(index_file)
        self.output_file_ctime = os.path.getctime(index_file)
        self.output_file_cheksum = checksums.get("output") or gen_checksum(index_file)
        self.hostname = socket.gethostname()
        # the reference hashes `tools.py` found in the cwd (tools.py:285); this build hashes its own format module
        self.checksum_script = gen_checksum(os.path.abspath(__file__))
        time_end = datetime.datetime.now()
        self.creation_time_start = str(self.timer.time_begin)
        self.creation_time_end = str(time_end)
        self.creation_duration = str(time_end - self.timer.time_begin)
        self.creation_speed = self.timer.speed_ela

    # ---- files (tools.py:294-363)
    def open_file(self, index_file: str, mode: str = "r+b") -> Iterator[BinaryIO]:
        if index_file.endswith(".bgz"):
            with open(index_file, "rb", buffering=self._buffer_size) as fhd:
                with gzip.open(fhd, "rb") as fhz:
                    yield fhz
        else:
            with open(index_file, mode, buffering=self._buffer_size) as fhd:
                yield fhd

    def open_index_file(self, mode: str = "r+b") -> Iterator[BinaryIO]:
        return self.open_file(self.index_file, mode=mode)

    def open_index_tmp_file(self, mode: str = "r+b") -> Iterator[BinaryIO]:
        return self.open_file(self.index_tmp_file, mode=mode)

    def _init_clean(self, overwrite: bool = False) -> None:
        for path in (self.index_file, self.index_file_root):
            if os.path.exists(path):
                if not overwrite:
                    raise ValueError(f"file {path} already exists and overwritting disabled")
                os.remove(path)
        for path in (self.metadata_file, self.index_tmp_file):
            if os.path.exists(path):
                os.remove(path)

    def init_file(self, index_file: str, mode: str = "r+b") -> None:
        if not os.path.exists(index_file):
            with open(index_file, "w"):
                pass
        for fhd in self.open_file(index_file, mode=mode):
            fhd.seek(self.max_size - 1)
            fhd.write(b"\0")

    def init_index_file(self, overwrite: bool = False, mode: str = "r+b") -> None:
        self._init_clean(overwrite=overwrite)
        return self.init_file(self.index_file, mode=mode)

    def init_index_tmp_file(self, overwrite: bool = False, mode: str = "r+b") -> None:
        self._init_clean(overwrite=overwrite)
        return self.init_file(self.index_tmp_file, mode=mode)

    def read_table(self, index_file: str = None) -> np.ndarray:
        """The whole table as a host array (raw .kin, or python-gzip .kin.bgz as tools.py:300-302 reads it)."""
        path = index_file or self.index_file
        if path.endswith(".bgz"):
            from . import bgzf                                # BGZF blocks inflate in parallel; plain gzip falls back to gzip.open
            table = bgzf.decompress_file(path)
        else:
            table = np.fromfile(path, dtype=np.uint8)
        assert table.size == self.data_size, f"{path}: {table.size} bytes, expected {self.data_size}"
        return table

    def read_table_slice(self, lo: int, hi: int, index_file: str = None, threads: int = None) -> np.ndarray:
        """Table bytes [lo, hi) only: a rank of the address-range-sharded merge never touches the rest of the
        file.  Raw .kin: one positioned read.  BGZF .kin.bgz: only the blocks that cover the slice are
        inflated.  Plain gzip (what tools.py:300-302 reads through gzip.open) has no random access: the
        stream is inflated up to `hi` and the front discarded.  `bytes_delivered` counts what came out of
        the reader, for the tests that check a rank reads its share only."""
        assert 0 <= lo <= hi <= self.data_size
        path = index_file or self.index_file
        if path.endswith(".bgz"):
            from . import bgzf
            part, delivered = bgzf.decompress_range(path, lo, hi, threads=threads)   # native inflate threads for this table
        else:
            assert os.path.getsize(path) == self.data_size, f"{path}: not {self.data_size} bytes"
            # a view of the page cache, not a copy: the library moves it to HBM from several threads
            part = np.memmap(path, dtype=np.uint8, mode="r", offset=lo, shape=(hi - lo,)) if hi > lo else np.zeros(0, np.uint8)
            delivered = part.size
        assert part.size == hi - lo, f"{path}: short read"
        self.bytes_delivered = getattr(self, "bytes_delivered", 0) + delivered
        return part

    def get_array_from_fhd(self, fhd: BinaryIO, mode: str = "r+") -> Iterator[np.memmap]:
        yield np.memmap(fhd, dtype=np.uint8, mode=mode, offset=0, shape=(self.data_size,))

    def get_array_from_index_file(self, fhd_mode: str = "r+b", mm_mode: str = "r+") -> Iterator[np.memmap]:
        for fhd in self.open_index_file(mode=fhd_mode):
            return self.get_array_from_fhd(fhd, mode=mm_mode)

    def get_array_from_index_tmp_file(self, fhd_mode: str = "r+b", mm_mode: str = "r+") -> Iterator[np.memmap]:
        for fhd in self.open_index_tmp_file(mode=fhd_mode):
            return self.get_array_from_fhd(fhd, mode=mm_mode)

    # ---- metadata (tools.py:366-401)
    def write_metadata_file(self, index_file: str, hist256=None, checksums=None) -> None:
        assert self.num_kmers
        assert self.chromosomes
        self.update_metadata(index_file, checksums=checksums)
        if hist256 is not None:                               # histogram already built in HBM by the indexer
            self.update_stats_from_hist256(hist256)
        else:
            for fhd in self.open_file(index_file):
                self.update_stats(fhd)
        with open(self.metadata_file, "wt") as fhm:
            json.dump(self.to_dict(), fhm, indent=1, sort_keys=1)

    def write_metadata_index_file(self, hist256=None) -> None:
        self.write_metadata_file(self.index_file, hist256=hist256)

    def write_metadata_index_tmp_file(self, hist256=None, checksums=None) -> None:
        self.write_metadata_file(self.index_tmp_file, hist256=hist256, checksums=checksums)

    def read_metadata(self) -> None:
        with open(self.metadata_file, "rt") as fhd:
            header_data = json.load(fhd)
        for k in self.HEADER_DATA:
            setattr(self, k, header_data[k])                  # KeyError on a missing key, like the reference
        for k in self.HEADER_FIXED:
            v, h = getattr(self, k), header_data[k]
            assert v == h, f"self.{k} != header_data[{k}]: {v} != {h}"

    # ---- validation (tools.py:404-436; the reference's check_data_file is broken at HEAD, this one works)
    def check_data(self, fhd: BinaryIO) -> None:
        self.read_metadata()
        other = self.__class__(self.project_name, input_file=self.input_file_path, kmer_len=self.kmer_len, device=self._device)
        other.read_metadata()
        for f in ("project_name", "input_file_name", "input_file_path", "kmer_len", "num_kmers"):
            assert getattr(self, f) == getattr(other, f), f
        other.update_stats(fhd)
        for f in ("hist", "hist_sum", "hist_count", "hist_min", "hist_max", "vals_sum", "vals_count", "vals_min", "vals_max"):
            assert getattr(self, f) == getattr(other, f), f

    def check_data_file(self, filename: str) -> None:
        for fhd in self.open_file(filename, mode="rb"):
            self.check_data(fhd)

    def check_data_index(self) -> None:
        self.check_data_file(self.index_file)

    def check_data_index_tmp(self) -> None:
        self.check_data_file(self.index_tmp_file)

    # ---- pair distance (tools.py:439-493): one GPU scan instead of blocked numpy passes
    def calculate_distance(self, other: "Header", min_count: int = HeaderVars.DEFAULT_MIN_COUNT,
                           max_count: int = HeaderVars.DEFAULT_MAX_COUNT,
                           block_size: int = HeaderVars.DEFAULT_BLOCK_SIZE, threading=False) -> Tuple[int, int, int]:
        assert self.data_size == other.data_size              # tools.py:444
        m = _lib.gram([self.read_table(), other.read_table()], min_count, max_count, devices=(self._device,))
        return int(m[0, 1, 0]), int(m[0, 1, 1]), int(m[0, 1, 2])

    def calculate_distance2(self, other: "Header", min_count: int = HeaderVars.DEFAULT_MIN_COUNT,
                            max_count: int = HeaderVars.DEFAULT_MAX_COUNT) -> Tuple[int, int, int]:
        return self.calculate_distance(other, min_count=min_count, max_count=max_count)   # tools.py:495-512: same numbers

    # ---- serialisation (tools.py:515-545)
    def to_dict(self, lean=False) -> Dict[str, Any]:
        data = OrderedDict()
        for k in self.HEADER_FIXED + self.HEADER_DATA:
            if lean and k in self.NOT_LEAN:
                continue
            data[k] = getattr(self, k)
        return data

    def to_json(self, indent: int = 1, sort_keys: bool = True) -> str:
        return json.dumps(self.to_dict(), indent=indent, sort_keys=sort_keys)

    def __iter__(self) -> Iterator[int]:
        for fhd in self.open_index_file():
            cs = fhd.read(self._buffer_size)
            while cs:
                yield from cs
                cs = fhd.read(self._buffer_size)

    def __str__(self) -> str:
        res = []
        for k, v in self.to_dict().items():
            res.append(f"{k:20s}: {v:15,d}" if isinstance(v, int) else f"{k:20s}: {str(v)[:50]}")
        return "\n".join(res) + "\n"

    __repr__ = __str__


def gen_checksum(filename: str, chunk_size: int = 2 ** 16) -> str:
    """tools.py:548-556: sha256 of a file, streamed."""
    file_hash = hashlib.sha256()
    with open(filename, "rb") as f:
        for chunk in iter(lambda: f.read(chunk_size), b""):
            file_hash.update(chunk)
    return file_hash.hexdigest()
