"""Merger host side: N `.kin[.bgz]` tables in, `<project>.<min>-<max>.kma` + `.kma.json` out.

Same CLI, checks and outputs as the reference's merger.py (argparse :51-59, merge :80-210, main
:213-239).  The reference runs Header.calculate_distance once per pair in a process pool
(merger.py:137-153), reading both tables each time; here every table is staged in HBM once and ONE
kernel pass (pk_gram_device_partial) yields all N(N+1)/2 tallies.  With several GPUs the k-mer
address range is split across them; with torch.distributed initialised (one process per GPU) each
rank scans its slice and the N x N partials are summed by one all-reduce (RCCL over xGMI).  The CLI enters that
mode by itself under a launcher that sets WORLD_SIZE / RANK / LOCAL_RANK (torchrun), or with `--gpus N`, which
starts the N ranks before anything touches a GPU.
"""
import argparse
import json
import os
import pathlib
import sys
from concurrent.futures import ThreadPoolExecutor
from json import JSONEncoder
from pathlib import Path
from typing import List, Tuple

import numpy as np

from . import _lib
from .header import Header

EXTS = ("." + Header.IND_EXT, "." + Header.IND_EXT + "." + Header.COMP_EXT, ".kma", ".kma." + Header.COMP_EXT)

DEFAULT_MIN_COUNT = Header.DEFAULT_MIN_COUNT
DEFAULT_MAX_COUNT = Header.DEFAULT_MAX_COUNT
DEFAULT_BUFFER_SIZE = Header.DEFAULT_BUFFER_SIZE
DEFAULT_BLOCK_SIZE = Header.DEFAULT_BLOCK_SIZE
DEFAULT_THREADS = 4


class _Encoder(JSONEncoder):
    """merger.py:23-30 patches JSONEncoder globally so Path objects serialise as strings; same effect, scoped."""

    def default(self, obj):
        if isinstance(obj, pathlib.PurePath):
            return str(obj)
        if hasattr(obj.__class__, "to_dict"):
            return obj.to_dict()
        return super().default(obj)


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(description="Merge kmer databases.")
    parser.add_argument("Project_Name", metavar="P", type=str, help="Project name")
    parser.add_argument("Kmer_1", metavar="K", type=Path, nargs=1, help="List of kin files")
    parser.add_argument("Kmer_N", metavar="K", type=Path, nargs="+", help="List of kin files")
    parser.add_argument("--min-count", type=int, default=DEFAULT_MIN_COUNT, nargs="?", help=f"Minimum Kmer Count [{DEFAULT_MIN_COUNT}]")
    parser.add_argument("--max-count", type=int, default=DEFAULT_MAX_COUNT, nargs="?", help=f"Maximum Kmer Count [{DEFAULT_MAX_COUNT}]")
    parser.add_argument("--buffer-size", type=int, default=DEFAULT_BUFFER_SIZE, nargs="?", help=f"Buffer size [{DEFAULT_BUFFER_SIZE}]")
    parser.add_argument("--block-size", type=int, default=DEFAULT_BLOCK_SIZE, nargs="?", help=f"Block size [{DEFAULT_BLOCK_SIZE}]")
    parser.add_argument("--threads", type=int, default=DEFAULT_THREADS, nargs="?",
                        help=f"Host threads reading / inflating the tables [{DEFAULT_THREADS}]")
    parser.add_argument("--sweep", type=str, default=None,
                        help="several count windows in one run, e.g. 1-255,2-255,1-3 (tables staged once; one .kma each)")
    parser.add_argument("--gpus", type=int, default=0,
                        help="one process per GPU: the k-mer address range is split over N ranks and the N x N partials are "
                             "summed by one RCCL all-reduce (also entered under torchrun, which sets WORLD_SIZE / RANK)")
    return parser


def calculate_distance(k_index_file: str, l_index_file: str, min_count: int = DEFAULT_MIN_COUNT, max_count: int = DEFAULT_MAX_COUNT,
                       buffer_size: int = DEFAULT_BUFFER_SIZE, block_size: int = DEFAULT_BLOCK_SIZE) -> Tuple[int, int, int]:
    """merger.py:62-78: one pair, from paths."""
    k_header = Header(str(k_index_file), index_file=str(k_index_file), buffer_size=buffer_size)
    l_header = Header(str(l_index_file), index_file=str(l_index_file), buffer_size=buffer_size)
    return k_header.calculate_distance(l_header, min_count=min_count, max_count=max_count, block_size=block_size, threading=True)


def address_slice(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice of the k-mer address range owned by `rank` (multiples of 32 addresses)."""
    per = ((n + world - 1) // world + 31) & ~31
    lo = min(n, per * rank)
    return lo, min(n, lo + per)


def _sub_slices(lo: int, hi: int, n_tables: int, device: int):
    """[lo, hi) cut so that n_tables slices fit HBM beside each other (the reference streams pairs and takes
    any N, merger.py:139-153; here a k=17 merge of 32 tables is 512 GiB).  Partials add, so the cuts are free.
    PK_MERGE_HBM_BUDGET (bytes) overrides the 80 % of free HBM used by default."""
    budget = int(os.environ.get("PK_MERGE_HBM_BUDGET", "0")) or int(_lib.mem_info(device)[0] * 0.8)
    per_table = max(2048, (budget // max(1, n_tables) - 64) & ~2047)
    return [(a, min(hi, a + per_table)) for a in range(lo, hi, per_table)]


class ResidentTable:
    """Addresses [first, first + n) of one 4^k-byte table that already lie in HBM on `device` (e.g. the table of an
    indexer that has just finished: pk_indexer_table_device) -- takes a Header's place in pair_matrix / gpu_partial, which
    then scan it where it is instead of staging it from a file."""

    def __init__(self, ptr: int, n: int, data_size: int, device: int = 0, first: int = 0):
        assert ptr % 16 == 0 and first % 32 == 0
        self.ptr, self.n, self.data_size, self.device, self.first = int(ptr), int(n), int(data_size), device, int(first)

    def device_slice(self, lo: int, hi: int) -> int:
        assert self.first <= lo <= hi <= self.first + self.n, "address range outside the resident part of the table"
        return self.ptr + (lo - self.first)


def gpu_partial(headers: List[Header], lo: int, hi: int, windows, device: int, threads: int, acc_ptr: int = None, stats: dict = None):
    """Stage addresses [lo, hi) of every table in HBM on `device` (sub-slice by sub-slice if they do not all
    fit) and tally each staged piece for every (min_count, max_count) window in one kernel pass per group of windows
    (pk_gram_device_accumulate_windows).  Only bytes [lo, hi) of each file are read / inflated; ResidentTable entries are
    scanned where they lie.  The tallies are accumulated in HBM: at `acc_ptr` (W x N x N u64,
    zeroed by the caller -- the buffer an RCCL all-reduce then sums) or in a buffer of this call, which is
    then returned as W host arrays.  `stats["kernel_seconds"]` accumulates the scans' kernel time."""
    N, W = len(headers), len(windows)
    own = None
    if acc_ptr is None:
        own = _lib.DeviceBuffer(W * N * N * 8, device)
        own.zero()
        acc_ptr = own.ptr
    resident = [hasattr(h, "device_slice") for h in headers]
    assert all(resident) or not any(resident), "resident and file-backed tables cannot be mixed in one merge"
    cuts = [(lo, hi)] if all(resident) else _sub_slices(lo, hi, N, device)
    bufs = [] if all(resident) else [_lib.DeviceBuffer(max(b - a for a, b in cuts), device) for _ in range(N)]
    from . import bgzf
    io_threads = max(1, bgzf.INFLATE_THREADS // max(1, min(threads, N)))
    try:
        pool = ThreadPoolExecutor(max_workers=max(1, threads)) if bufs else None
        try:
            for a, b in cuts:
                if bufs:
                    # read / inflate (`threads` tables at a time, each .kin.bgz on its share of the native inflate threads),
                    # upload as each one lands
                    list(pool.map(lambda i: bufs[i].upload(headers[i].read_table_slice(a, b, threads=io_threads)), range(N)))
                    ptrs = [buf.ptr for buf in bufs]
                else:
                    ptrs = [h.device_slice(a, b) for h in headers]
                secs = _lib.gram_device_accumulate_windows(ptrs, b - a, acc_ptr, windows, device=device)
                if stats is not None:
                    stats["kernel_seconds"] = stats.get("kernel_seconds", 0.0) + secs
        finally:
            if pool is not None:
                pool.shutdown()
        if own is None:
            return None
        return list(own.download().view(np.uint64).reshape(W, N, N))
    finally:
        for buf in bufs:
            buf.free()
        if own is not None:
            own.free()


def pair_matrix(headers: List[Header], windows, threads: int = DEFAULT_THREADS, devices=(0,), group=None,
                partial_fn=None, stats: dict = None) -> List[np.ndarray]:
    """One N x N u64 per (min, max) window: [i][i] = valid addresses of table i, [i][j] (i<j) = addresses valid in both.

    Single process: the address range is split over `devices`, one host thread per device.  With `group`
    (a torch.distributed process group, or True for the default group) this rank scans only its own slice
    on devices[0] and the partials of all windows are summed by ONE all-reduce -- over RCCL on the
    accumulator where the kernel left it in HBM.  `partial_fn` computes one slice's tallies (default: gpu_partial,
    looked up when called; the CPU-only tests substitute the oracle)."""
    partial_fn = partial_fn or gpu_partial
    n, N, W = headers[0].data_size, len(headers), len(windows)
    if group is None:
        plan = [(d,) + address_slice(n, i, len(devices)) for i, d in enumerate(devices)]
        plan = [p for p in plan if p[2] > p[1]]
        kw = {"stats": stats} if (stats is not None and partial_fn is gpu_partial) else {}
        with ThreadPoolExecutor(max_workers=max(1, len(plan))) as pool:
            parts = list(pool.map(lambda p: partial_fn(headers, p[1], p[2], windows, p[0], max(1, threads // len(plan)), **kw), plan))
        total = np.zeros((W, N, N), dtype=np.uint64)
        for part in parts:
            for w in range(W):
                total[w] += part[w]
        return [total[w] for w in range(W)]

    import torch
    import torch.distributed as dist
    pg = None if group is True else group
    dev = devices[0]
    lo, hi = address_slice(n, dist.get_rank(pg), dist.get_world_size(pg))
    on_gpu = dist.get_backend(pg) == "nccl"                    # "nccl" is RCCL on ROCm; gloo only for rehearsals and CPU tests
    if partial_fn is gpu_partial and on_gpu:
        acc = torch.zeros((W, N, N), dtype=torch.int64, device=torch.device("cuda", dev))
        torch.cuda.synchronize(dev)                             # zeroed before the scan (its own stream) adds to it
        if hi > lo:
            gpu_partial(headers, lo, hi, windows, dev, threads, acc_ptr=acc.data_ptr(), stats=stats)
        dist.all_reduce(acc, group=pg)                         # RCCL over xGMI: W x N x N u64, a few KB
        total = acc.cpu().numpy().view(np.uint64)
    else:
        total = np.zeros((W, N, N), dtype=np.uint64)
        if hi > lo:
            kw = {"stats": stats} if (stats is not None and partial_fn is gpu_partial) else {}
            for w, part in enumerate(partial_fn(headers, lo, hi, windows, dev, threads, **kw)):
                total[w] += part
        t = torch.from_numpy(total.view(np.int64).copy())
        if on_gpu:                                             # an RCCL group reduces device tensors only
            t = t.to(torch.device("cuda", dev))
        dist.all_reduce(t, group=pg)
        total = t.cpu().numpy().view(np.uint64)
    return [total[w] for w in range(W)]


def merge(project_name: str, indexes: List[Path], min_count: int = DEFAULT_MIN_COUNT, max_count: int = DEFAULT_MAX_COUNT,
          buffer_size: int = DEFAULT_BUFFER_SIZE, block_size: int = DEFAULT_BLOCK_SIZE, threads: int = DEFAULT_THREADS,
          devices=(0,), group=None, partial_fn=None, windows=None):
    """merger.py:80-210.  `windows` (a list of (min_count, max_count)) turns the call into a sweep: the
    tables are staged once and one `.kma` + `.kma.json` is written per window (the reference re-runs
    the whole merge per threshold, README.md:57-61); the first window's matrix is returned."""
    windows = [(min_count, max_count)] if not windows else [tuple(w) for w in windows]
    for mn, mx in windows:
        assert mn >= 1
        assert mx <= 255
    assert buffer_size > 0
    assert block_size > 0
    assert len(indexes) > 0

    outfiles = [Path(f"{project_name}.{mn:03d}-{mx:03d}.kma") for mn, mx in windows]
    assert not Path(project_name).exists(), f"project name ({project_name}) is a file. maybe forgot to pass project name as first argument?"
    for outfile in outfiles:
        assert not outfile.exists(), f"project output file ({outfile}) already exists. not overwriting."

    indexes = [Path(p) for p in indexes]
    assert all(i.exists() for i in indexes)

    data, headers, kmer_len = [], [], None
    for pos, kin in enumerate(indexes):
        print(f"verifying {kin}")
        kins = str(kin)
        assert kins.endswith(EXTS), f"all files must be .{Header.IND_EXT}[.bgz]: {kin}"
        desc = kins[:-(len(Header.COMP_EXT) + 1)] if kins.endswith("." + Header.COMP_EXT) else kin
        desc = Path(f"{desc}.{Header.DESC_EXT}")
        assert desc.exists(), f"all .{Header.IND_EXT}[.{Header.COMP_EXT}] files must have a associated .{Header.IND_EXT}.{Header.DESC_EXT}: {desc}"
        header = Header(kins, index_file=kins, buffer_size=buffer_size, device=devices[0])
        if kmer_len is None:
            kmer_len = header.kmer_len
        assert header.kmer_len == kmer_len, f"kmer_length differs. expected {kmer_len}, got {header.kmer_len}"
        headers.append(header)
        data.append({"pos": pos, "index_file": kin, "description_file": desc, "header": header})
    print()

    pairs = pair_matrix(headers, windows, threads=threads, devices=devices, group=group, partial_fn=partial_fn)
    for v in data:
        v["header"] = v["header"].to_dict(lean=True)          # merger.py:187-188
    is_writer = True
    if group is not None:
        import torch.distributed as dist
        is_writer = dist.get_rank(None if group is True else group) == 0

    matrices = []
    for (mn, mx), outfile, pair in zip(windows, outfiles, pairs):
        # (N,N,3): [k][l] = (total_k, total_l, shared) (merger.py:175-176); the diagonal, which the reference
        # never assigns in its uninitialised array (merger.py:136), is zero here
        matrix = _lib.gram_expand(pair)
        matrices.append(matrix)
        for k in range(len(data) - 1):
            for l in range(k + 1, len(data)):
                print(f"   matrix Total #{k:3d} {int(matrix[k, l, 0]):15,d} Total #{l:3d} {int(matrix[k, l, 1]):15,d} Shared {int(matrix[k, l, 2]):15,d}")
        if not is_writer:
            continue
        output = {"project_name": project_name, "min_count": mn, "max_count": mx, "data": data}
        outfile_json = Path(f"{outfile}.json")
        outfile_json_tmp = Path(f"{outfile_json}.tmp")
        print(f"saving {outfile_json}")
        with outfile_json_tmp.open(mode="wt") as fhd:
            json.dump(output, fhd, sort_keys=True, indent=1, cls=_Encoder)
        outfile_json_tmp.rename(outfile_json)
        print(f"saving {outfile}")
        outfile_tmp = Path(f"{outfile}.tmp")
        with outfile_tmp.open(mode="wb") as fhd:
            np.savez_compressed(fhd, matrix=matrix)            # merger.py:207: key `matrix`
        outfile_tmp.rename(outfile)
    return data, matrices[0]


def parse_sweep(text: str):
    """`"1-255,2-255,1-3"` -> [(1, 255), (2, 255), (1, 3)]."""
    out = []
    for item in text.split(","):
        lo, hi = item.strip().split("-")
        out.append((int(lo), int(hi)))
    return out


def spawn_ranks(world: int, argv: List[str], script: str = None) -> int:
    """`merger.py ... --gpus N` without a launcher: start N ranks of this command line (one per GPU) and wait for them.
    The parent never touches a GPU (the root CLI skips its warm-up thread in this case), so the ranks are plain child
    processes of a process without a HIP context.  Rank 0 inherits stdout; the others print nothing."""
    import socket
    import subprocess
    import time
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    # the root CLI, whoever called main(): a rank must not start whatever sys.argv[0] happens to be (a test runner, say)
    script = script or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "merger.py")
    cmd = [sys.executable, script] + list(argv)
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen(cmd, env=env))
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):       # a rank died: its peers would wait in the all-reduce for ever
            for p in procs:
                if p.poll() is None:
                    p.kill()                                     # exactly the children started above
            break
        time.sleep(0.05)
    return max(abs(p.wait()) for p in procs)


def _join_group():
    """One process per GPU: RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the environment (torchrun, or spawn_ranks).
    Backend "nccl" (= RCCL over xGMI); PK_DIST_BACKEND=gloo rehearses the same path where ranks share a GPU.
    Returns (rank, device ordinal)."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    listed = [int(d) for d in os.environ.get("PK_DEVICES", "").split(",") if d != ""]
    n_dev = torch.cuda.device_count()
    device = listed[local % len(listed)] if listed else (local % n_dev if n_dev else 0)
    backend = os.environ.get("PK_DIST_BACKEND", "nccl")
    if backend == "nccl":
        torch.cuda.set_device(device)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, device


def main(argv: List[str] = None) -> None:
    """merger.py:213-239."""
    argv = list(sys.argv[1:] if argv is None else argv)
    args = build_parser().parse_args(argv)
    indexes: List[Path] = args.Kmer_1 + args.Kmer_N
    if len(indexes) <= 1:
        print("needs at least 2 files")
        sys.exit(1)
    indexes.sort()                                             # matrix order = sorted path order (merger.py:228)
    windows = parse_sweep(args.sweep) if args.sweep else None
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(spawn_ranks(args.gpus, argv))
        devices = tuple(int(d) for d in os.environ.get("PK_DEVICES", "0").split(",") if d != "")
        merge(args.Project_Name, indexes, min_count=args.min_count, max_count=args.max_count, buffer_size=args.buffer_size,
              block_size=args.block_size, threads=args.threads, devices=devices or (0,), windows=windows)
        return
    # one rank of a multi-process merge: every rank validates and scans its address slice, rank 0 prints and writes
    import contextlib
    import torch.distributed as dist
    rank, device = _join_group()
    try:
        with contextlib.redirect_stdout(None) if rank else contextlib.nullcontext():
            merge(args.Project_Name, indexes, min_count=args.min_count, max_count=args.max_count, buffer_size=args.buffer_size,
                  block_size=args.block_size, threads=args.threads, devices=(device,), group=True, windows=windows)
        dist.barrier()                                         # nobody leaves before rank 0 has renamed the outputs
    finally:
        dist.destroy_process_group()
