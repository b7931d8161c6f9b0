"""ctypes binding of libpykmer_hip.so (include/pykmer_hip.h).

The engine has no CPU fallback: if the shared library is missing or a call fails, this raises.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
from ._rt import LIB_PATH, open_library

PK_OK, PK_ERR_ARG, PK_ERR_HIP, PK_ERR_RECS_CAP, PK_ERR_STATE = 0, -1, -2, -3, -4

RECORD_DTYPE = np.dtype([("name_off", "<u8"), ("name_len", "<u8"), ("seq_len", "<u8"), ("n_valid_kmers", "<u8")])

_u64p = ctypes.POINTER(ctypes.c_uint64)
_SIGNATURES = {
    "pk_version": (ctypes.c_int, []),
    "pk_last_error": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_size_t]),
    "pk_device_count": (ctypes.c_int, []),
    "pk_warm": (ctypes.c_int, [ctypes.c_int]),
    "pk_dev_alloc": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint64, ctypes.c_int]),
    "pk_dev_free": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "pk_dev_upload": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int]),
    "pk_dev_download": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int]),
    "pk_dev_mem_info": (ctypes.c_int, [_u64p, _u64p, ctypes.c_int]),
    "pk_count_fasta": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p, _u64p, _u64p,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, _u64p, ctypes.c_int]),
    "pk_count_release": (ctypes.c_int, []),
    "pk_indexer_create": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int]),
    "pk_indexer_create_slice": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "pk_indexer_reset": (ctypes.c_int, [ctypes.c_void_p]),
    "pk_indexer_feed": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]),
    "pk_indexer_feed_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]),
    "pk_indexer_finish": (ctypes.c_int, [ctypes.c_void_p, _u64p, _u64p, ctypes.c_void_p, _u64p]),
    "pk_indexer_records": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]),
    "pk_indexer_table_to_host": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "pk_indexer_table_slice_to_host": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64]),
    "pk_indexer_table_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]),
    "pk_indexer_table_slice_to_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64]),
    "pk_indexer_timings": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "pk_indexer_destroy": (None, [ctypes.c_void_p]),
    "pk_table_stats": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int]),
    "pk_gram": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_int, ctypes.c_int,
                                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]),
    "pk_gram_device_partial": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                               ctypes.POINTER(ctypes.c_double)]),
    "pk_gram_device_accumulate": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_int, ctypes.c_int,
                                                  ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]),
    "pk_gram_device_accumulate_windows": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p,
                                                          ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]),
    "pk_gram_expand": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "pk_bgzf_scan": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _u64p]),
    "pk_bgzf_inflate": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p,
                                        ctypes.c_int]),
    "pk_bgzf_deflate": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64,
                                        ctypes.c_void_p, _u64p, ctypes.c_int]),
    "pk_diag_occupancy": (ctypes.c_int, [ctypes.c_int]),
    "pk_diag_plan": (ctypes.c_int, [ctypes.c_int, ctypes.c_uint64, ctypes.c_void_p]),
}
EXPORTS = tuple(_SIGNATURES)

_lib = None


class PkError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libpykmer_hip error {code}: {message}")
        self.code = code


def load():
    """Loads the library (building nothing: run pykmer_amd.build or __graft_entry__.build first)."""
    global _lib
    if _lib is None:
        lib = open_library()
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def _check(rc):
    if rc != PK_OK:
        buf = ctypes.create_string_buffer(512)
        load().pk_last_error(buf, 512)
        msg = buf.value.decode("utf-8", "replace")
        if rc == PK_ERR_ARG:
            raise ValueError(msg)
        raise PkError(rc, msg)


def _as_u8(data) -> np.ndarray:
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data, dtype=np.uint8)
    return np.frombuffer(data, dtype=np.uint8)


def device_count() -> int:
    return load().pk_device_count()


def warm(device: int = 0) -> None:
    """pk_warm: HIP start-up and kernel load on `device` (the CLIs call this from a thread while they set up)."""
    _check(load().pk_warm(device))


def mem_info(device: int = 0):
    """(free, total) bytes of HBM on `device`."""
    f, t = ctypes.c_uint64(0), ctypes.c_uint64(0)
    _check(load().pk_dev_mem_info(ctypes.byref(f), ctypes.byref(t), device))
    return int(f.value), int(t.value)


def _hist_out():
    return np.zeros(256, dtype=np.uint64)


class DeviceBuffer:
    """n bytes of HBM on one device (pk_dev_*)."""

    def __init__(self, n_bytes: int, device: int = 0):
        self.n, self.device = int(n_bytes), device
        p = ctypes.c_void_p()
        _check(load().pk_dev_alloc(ctypes.byref(p), self.n, device))
        self.ptr = p.value

    def zero(self):
        self.upload(np.zeros(self.n, dtype=np.uint8))

    def upload(self, data, offset: int = 0):
        buf = _as_u8(data)
        assert offset + buf.size <= self.n
        _check(load().pk_dev_upload(ctypes.c_void_p(self.ptr + offset), buf.ctypes.data, buf.size, self.device))

    def download(self, n_bytes: int = None, offset: int = 0) -> np.ndarray:
        n = self.n - offset if n_bytes is None else n_bytes
        out = np.empty(n, dtype=np.uint8)
        _check(load().pk_dev_download(out.ctypes.data, ctypes.c_void_p(self.ptr + offset), n, self.device))
        return out

    def free(self):
        if getattr(self, "ptr", None):
            load().pk_dev_free(ctypes.c_void_p(self.ptr), self.device)
            self.ptr = None

    __del__ = free


class Indexer:
    """One 4^k count table resident in HBM on one device (pk_indexer_*)."""

    def __init__(self, k: int, device: int = 0, slice_index: int = 0, n_slices: int = 1):
        self._h = ctypes.c_void_p()
        self.k, self.device, self.slice_index, self.n_slices = k, device, slice_index, n_slices
        self.table_bytes = 4 ** k // n_slices
        _check(load().pk_indexer_create_slice(ctypes.byref(self._h), k, device, slice_index, n_slices))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            load().pk_indexer_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def reset(self):
        _check(load().pk_indexer_reset(self._h))

    def feed(self, data):
        buf = _as_u8(data)                                   # bytes, bytearray, mmap, numpy: no copy for any of them
        _check(load().pk_indexer_feed(self._h, buf.ctypes.data, buf.size))

    def feed_device(self, dev_ptr: int, n_bytes: int):
        _check(load().pk_indexer_feed_device(self._h, ctypes.c_void_p(dev_ptr), n_bytes))

    def finish(self):
        nk, bp, nr = ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_uint64(0)
        hist = _hist_out()
        _check(load().pk_indexer_finish(self._h, ctypes.byref(nk), ctypes.byref(bp), hist.ctypes.data, ctypes.byref(nr)))
        return {"num_kmers": int(nk.value), "total_bp": int(bp.value), "hist256": hist, "n_records": int(nr.value)}

    def records(self, n_records: int) -> np.ndarray:
        recs = np.zeros(max(n_records, 1), dtype=RECORD_DTYPE)
        _check(load().pk_indexer_records(self._h, recs.ctypes.data, recs.size))
        return recs[:n_records]

    def table_to_host(self, out: np.ndarray = None) -> np.ndarray:
        if out is None:
            out = np.empty(self.table_bytes, dtype=np.uint8)
        assert out.dtype == np.uint8 and out.size == self.table_bytes and out.flags.c_contiguous
        _check(load().pk_indexer_table_to_host(self._h, out.ctypes.data))
        return out

    def table_slice_to_host(self, out: np.ndarray, offset: int):
        """Table bytes [offset, offset + out.size) into `out` (a contiguous uint8 array, e.g. a slice of a memmap)."""
        assert out.dtype == np.uint8 and out.flags.c_contiguous
        _check(load().pk_indexer_table_slice_to_host(self._h, out.ctypes.data, offset, out.size))

    def table_device_ptr(self) -> int:
        p = ctypes.c_void_p()
        _check(load().pk_indexer_table_device(self._h, ctypes.byref(p)))
        return p.value

    def table_slice_to_device(self, dev_dst: int, offset: int, n_bytes: int):
        _check(load().pk_indexer_table_slice_to_device(self._h, ctypes.c_void_p(dev_dst), offset, n_bytes))

    def timings(self) -> dict:
        t = np.zeros(10, dtype=np.float64)
        _check(load().pk_indexer_timings(self._h, t.ctypes.data))
        return {"scan_s": t[0], "squeeze_s": t[1], "finalize_s": t[2], "zero_s": t[3], "feeds": int(t[4]),
                "partition_s": t[5], "bucket_s": t[6], "walk_sort_s": t[7], "relayouts": int(t[8]), "buckets_recounted": int(t[9])}


def count_fasta(data, k: int, device: int = 0, table_out: np.ndarray = None):
    """pk_count_fasta: host FASTA text -> dict(table, num_kmers, total_bp, hist256, records)."""
    buf = _as_u8(data)
    if k <= 0 or k % 2 == 0 or k > 17:                      # let the library word the error (tools.py:165-167)
        _check(load().pk_count_fasta(None, 0, k, None, None, None, None, None, 0, None, device))
    table = table_out if table_out is not None else np.empty(4 ** k, dtype=np.uint8)
    nk, bp, nr = ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_uint64(0)
    hist = _hist_out()
    cap = 4096
    while True:
        recs = np.zeros(cap, dtype=RECORD_DTYPE)
        rc = load().pk_count_fasta(buf.ctypes.data, buf.size, k, table.ctypes.data, ctypes.byref(nk), ctypes.byref(bp),
                                   hist.ctypes.data, recs.ctypes.data, cap, ctypes.byref(nr), device)
        if rc == PK_ERR_RECS_CAP and nr.value > cap:
            cap = int(nr.value)
            continue
        _check(rc)
        break
    return {"table": table, "num_kmers": int(nk.value), "total_bp": int(bp.value), "hist256": hist,
            "records": recs[: nr.value].copy()}


def table_stats(table: np.ndarray, device: int = 0) -> np.ndarray:
    """pk_table_stats: 256-bin histogram of a host u8 table."""
    t = _as_u8(table)
    hist = _hist_out()
    _check(load().pk_table_stats(t.ctypes.data, t.size, hist.ctypes.data, device))
    return hist


def gram(tables, min_count: int = 1, max_count: int = 255, devices=(0,)) -> np.ndarray:
    """pk_gram on host tables -> (N,N,3) uint64 matrix (merger.py:136,175-176; zero diagonal)."""
    ts = [_as_u8(t) for t in tables]
    n = ts[0].size
    if any(t.size != n for t in ts):
        raise AssertionError("tables differ in size (tools.py:444)")
    N = len(ts)
    ptrs = (ctypes.c_void_p * N)(*[t.ctypes.data for t in ts])
    devs = (ctypes.c_int * len(devices))(*devices)
    m = np.zeros((N, N, 3), dtype=np.uint64)
    _check(load().pk_gram(ptrs, N, n, min_count, max_count, m.ctypes.data, devs, len(devices)))
    return m


def gram_device_partial(dev_ptrs, n_slice: int, min_count: int = 1, max_count: int = 255, device: int = 0,
                        dev_pair_out: int = None):
    """pk_gram_device_partial on device-resident slices -> (pair[N,N] uint64, kernel_seconds)."""
    N = len(dev_ptrs)
    ptrs = (ctypes.c_void_p * N)(*dev_ptrs)
    pair = np.zeros((N, N), dtype=np.uint64)
    secs = ctypes.c_double(0)
    _check(load().pk_gram_device_partial(ptrs, N, n_slice, min_count, max_count, pair.ctypes.data,
                                         ctypes.c_void_p(dev_pair_out) if dev_pair_out else None, device,
                                         ctypes.byref(secs)))
    return pair, secs.value


def gram_device_accumulate(dev_ptrs, n_slice: int, dev_pair_accum: int, min_count: int = 1, max_count: int = 255,
                           device: int = 0) -> float:
    """pk_gram_device_accumulate: adds one slice's tallies to an N x N u64 accumulator in HBM; returns kernel seconds."""
    N = len(dev_ptrs)
    ptrs = (ctypes.c_void_p * N)(*dev_ptrs)
    secs = ctypes.c_double(0)
    _check(load().pk_gram_device_accumulate(ptrs, N, n_slice, min_count, max_count, ctypes.c_void_p(dev_pair_accum), device,
                                            ctypes.byref(secs)))
    return secs.value


def gram_device_accumulate_windows(dev_ptrs, n_slice: int, dev_pair_accum: int, windows, device: int = 0) -> float:
    """pk_gram_device_accumulate_windows: adds one slice's tallies for every (min_count, max_count) of `windows` to a
    W x N x N u64 accumulator in HBM -- one pass over the slices per group of windows; returns kernel seconds."""
    N, W = len(dev_ptrs), len(windows)
    ptrs = (ctypes.c_void_p * N)(*dev_ptrs)
    mins = (ctypes.c_int * W)(*[int(w[0]) for w in windows])
    maxs = (ctypes.c_int * W)(*[int(w[1]) for w in windows])
    secs = ctypes.c_double(0)
    _check(load().pk_gram_device_accumulate_windows(ptrs, N, n_slice, mins, maxs, W, ctypes.c_void_p(dev_pair_accum), device,
                                                    ctypes.byref(secs)))
    return secs.value


def bgzf_scan(buf: np.ndarray):
    """pk_bgzf_scan: (offsets, sizes, isizes) of every BGZF block in a u8 array (int64 arrays); ValueError if not BGZF."""
    cap = max(16, buf.size // 4096)
    while True:
        c_off, c_size, isize = (np.zeros(cap, dtype=np.uint64) for _ in range(3))
        n = ctypes.c_uint64(0)
        rc = load().pk_bgzf_scan(buf.ctypes.data, buf.size, cap, c_off.ctypes.data, c_size.ctypes.data, isize.ctypes.data, ctypes.byref(n))
        if rc == PK_ERR_RECS_CAP:
            cap = int(n.value)
            continue
        _check(rc)
        k = int(n.value)
        return c_off[:k].astype(np.int64), c_size[:k].astype(np.int64), isize[:k].astype(np.int64)


def bgzf_inflate(buf: np.ndarray, c_off, c_size, u_off, out: np.ndarray, threads: int) -> None:
    """pk_bgzf_inflate: blocks (c_off[i], c_size[i]) of `buf` -> out[u_off[i] - u_off[0] : u_off[i + 1] - u_off[0]]."""
    n = len(c_off)
    co = np.ascontiguousarray(c_off, dtype=np.uint64)
    cs = np.ascontiguousarray(c_size, dtype=np.uint64)
    uo = np.ascontiguousarray(np.asarray(u_off, dtype=np.int64) - int(u_off[0]), dtype=np.uint64)
    assert uo.size == n + 1 and out.dtype == np.uint8 and out.flags.c_contiguous and out.size >= int(uo[-1])
    _check(load().pk_bgzf_inflate(buf.ctypes.data, buf.size, co.ctypes.data, cs.ctypes.data, uo.ctypes.data, n, out.ctypes.data, int(threads)))


def bgzf_deflate(data: np.ndarray, level: int, block_input: int, threads: int):
    """pk_bgzf_deflate: (compressed blocks back to back as a u8 array, per-block compressed sizes)."""
    n_blocks = (data.size + block_input - 1) // block_input
    dst = np.empty(max(1, n_blocks) * 65536, dtype=np.uint8)
    sizes = np.zeros(max(1, n_blocks), dtype=np.uint64)
    total = ctypes.c_uint64(0)
    _check(load().pk_bgzf_deflate(data.ctypes.data, data.size, level, block_input, dst.ctypes.data, dst.size, sizes.ctypes.data,
                                  ctypes.byref(total), int(threads)))
    return dst[: total.value], sizes[:n_blocks].astype(np.int64)


def diag_plan(k: int, n_bytes: int = 0) -> dict:
    """pk_diag_plan: the partition plan of one feed (n_bytes = 0: the largest piece a feed is cut into)."""
    out = np.zeros(8, dtype=np.uint64)
    _check(load().pk_diag_plan(k, n_bytes, out.ctypes.data))
    names = ("feed_max", "capacity1", "capacity2", "B1", "B2", "fb_bits", "n_chunks", "fits_u32")
    return {n: int(v) for n, v in zip(names, out)}


def gram_expand(pair: np.ndarray) -> np.ndarray:
    p = np.ascontiguousarray(pair, dtype=np.uint64)
    N = p.shape[0]
    m = np.zeros((N, N, 3), dtype=np.uint64)
    _check(load().pk_gram_expand(p.ctypes.data, N, m.ctypes.data))
    return m
