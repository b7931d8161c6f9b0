"""GPU parity tests for the merge path: one-pass gram scan (C-ABI) vs the oracle's pairwise
restatement of Header.calculate_distance and vs the reference's own merger output (G7)."""
import numpy as np
import pytest

import inputs
import oracle
from oracle import pyoracle

pytestmark = pytest.mark.gpu


def _random_tables(rng, N, n, density=0.4):
    out = []
    for _ in range(N):
        t = rng.integers(1, 256, size=n, dtype=np.uint8)
        t[rng.random(n) > density] = 0
        t[rng.random(n) < 0.05] = rng.integers(1, 6)          # plenty of small counts for min/max windows
        out.append(t)
    return out


@pytest.mark.parametrize("N", [2, 3, 5, 8, 9, 13, 16, 17, 24, 25, 32, 33, 40, 41, 48, 49, 64])
def test_gram_vs_oracle(gpu, N):
    rng = np.random.default_rng(100 + N)
    n = 4 ** 7
    tables = _random_tables(rng, N, n)
    for mn, mx in ((1, 255), (2, 255), (1, 3), (2, 5), (128, 255), (100, 200), (1, 127), (129, 254), (255, 255)):
        got = gpu.gram(tables, mn, mx)
        want = oracle.gram(tables, mn, mx)
        assert got.dtype == np.uint64 and got.shape == (N, N, 3)
        assert np.array_equal(got, want), (N, mn, mx)


def test_gram_matches_numpy_restatement(gpu):
    rng = np.random.default_rng(7)
    tables = _random_tables(rng, 6, 4 ** 5)
    assert np.array_equal(gpu.gram(tables, 2, 9), pyoracle.gram(tables, 2, 9))


@pytest.mark.parametrize("n", [4, 64, 100, 1023, 8192 + 32, 300_001])
def test_gram_ragged_sizes(gpu, n):
    """Table sizes that are not multiples of the 32-address word / 256-word tile."""
    rng = np.random.default_rng(n)
    for N in (3, 20):
        tables = _random_tables(rng, N, n, density=0.7)
        assert np.array_equal(gpu.gram(tables), oracle.gram(tables))


@pytest.mark.parametrize("tag", ["default", "min2", "max3", "min2max5"])
def test_golden_merge_k7_n13(gpu, manifest, tag):
    """G7: 13 tables indexed by OUR indexer from the seeded family, merged by OUR kernel, against the
    matrix the reference's merger.py wrote for the reference-indexed tables."""
    case = manifest["merger"][f"G7_k7_n13_{tag}"]
    tables = [gpu.count_fasta(inputs.make_input(spec), case["k"])["table"] for spec in case["inputs"]]
    args = case["args"]
    mn = int(args[args.index("--min-count") + 1]) if "--min-count" in args else 1
    mx = int(args[args.index("--max-count") + 1]) if "--max-count" in args else 255
    got = gpu.gram(tables, mn, mx)
    want = np.array(case["matrix"], dtype=np.uint64)
    assert np.array_equal(got, want)


def test_rejects_bad_window(gpu):
    t = [np.zeros(64, np.uint8)] * 2
    for mn, mx in ((0, 255), (1, 256), (-1, 3)):               # merger.py:90-91
        with pytest.raises(ValueError):
            gpu.gram(t, mn, mx)
    with pytest.raises(AssertionError):                        # tools.py:444
        gpu.gram([np.zeros(64, np.uint8), np.zeros(128, np.uint8)])


def test_gram_full_size_properties(gpu):
    """config 3 shape: N=13 tables of 4^15 bytes, resident in HBM.  Checked through size-independent
    properties: totals equal per-table valid counts, matrix symmetric, shared <= min(total), and the
    address-range split (the multi-GPU sharding) sums to the unsplit result."""
    import torch
    n, N = 4 ** 15, 13
    g = torch.Generator(device="cuda").manual_seed(9)
    tabs = []
    for i in range(N):
        t = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda", generator=g)
        keep = torch.rand(n, device="cuda", generator=g) < (0.15 + 0.05 * i)
        tabs.append(t * keep)
    ptrs = [t.data_ptr() for t in tabs]
    torch.cuda.synchronize()
    pair, secs = gpu.gram_device_partial(ptrs, n)
    totals = [int((t != 0).sum().item()) for t in tabs]
    assert [int(pair[i, i]) for i in range(N)] == totals
    for i, j in ((0, 1), (3, 11), (5, 12)):
        assert int(pair[i, j]) == int(((tabs[i] != 0) & (tabs[j] != 0)).sum().item())
    m = gpu.gram_expand(pair)
    assert (m[:, :, 2] == m[:, :, 2].T).all() and (np.diagonal(m[:, :, 2]) == 0).all()
    assert (m[:, :, 2] <= np.minimum(m[:, :, 0], m[:, :, 1])).all()
    # the sharded form: 4 address slices -> partial matrices -> sum
    acc = np.zeros_like(pair)
    step = n // 4
    for s in range(4):
        p, _ = gpu.gram_device_partial([q + s * step for q in ptrs], step)
        acc += p
    assert np.array_equal(acc, pair)
    # windowed counts against torch on one pair
    p2, _ = gpu.gram_device_partial(ptrs[:2], n, 3, 200)
    a = (tabs[0] >= 3) & (tabs[0] <= 200)
    b = (tabs[1] >= 3) & (tabs[1] <= 200)
    assert int(p2[0, 0]) == int(a.sum().item()) and int(p2[0, 1]) == int((a & b).sum().item())
    print(f"gram N=13 k=15 kernel {secs * 1e3:.3f} ms -> {N * n / secs / 1e12:.2f} TB/s")


def _torch_tables(n, N, seed):
    import torch
    g = torch.Generator(device="cuda").manual_seed(seed)
    tabs = []
    for i in range(N):
        t = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda", generator=g)
        keep = torch.rand(n, device="cuda", generator=g) < (0.1 + 0.8 * ((i * 7) % N) / N)
        tabs.append(t * keep)
    torch.cuda.synchronize()
    return tabs


@pytest.mark.parametrize("N", [13, 32])
def test_gram_full_size_every_pair(gpu, N):
    """configs 3 and 5 at full size (4^15-byte tables resident in HBM): EVERY total and EVERY shared tally
    against an independent torch computation, default window and a --min/--max window."""
    import torch
    torch.cuda.empty_cache()
    n = 4 ** 15
    tabs = _torch_tables(n, N, seed=40 + N)
    ptrs = [t.data_ptr() for t in tabs]
    for mn, mx in ((1, 255), (3, 200)):
        pair, secs = gpu.gram_device_partial(ptrs, n, mn, mx)
        valid = [(t >= mn) & (t <= mx) for t in tabs]
        for i in range(N):
            assert int(pair[i, i]) == int(valid[i].sum().item()), (N, mn, mx, i)
            for j in range(i + 1, N):
                assert int(pair[i, j]) == int((valid[i] & valid[j]).sum().item()), (N, mn, mx, i, j)
            assert not pair[i, :i].any()
        del valid
    print(f"gram N={N} k=15 kernel {secs * 1e3:.3f} ms -> {N * n / secs / 1e12:.2f} TB/s")
    del tabs
    torch.cuda.empty_cache()


@pytest.mark.parametrize("N,log4n", [(48, 15), (64, 15), (41, 16)])
def test_gram_packed_tallies_worst_case_at_size(gpu, N, log4n):
    """The 16-bit packed tallies (N 41-48, and every launch of N > 48) at the sizes that matter, with DENSE
    tables -- every address valid is the worst case for a tally that must not carry into its neighbour.
    4^15: 128 tiles per workgroup; 4^16 with N = 41: more than 448 tiles per workgroup at the default grid,
    so the launcher's grid-resize branch runs.  Expected values are closed-form: a table whose constant
    value lies inside the window is valid everywhere, so totals and shared tallies are n or 0."""
    import torch
    torch.cuda.empty_cache()
    n = 4 ** log4n
    values = [(i % 5) + 1 for i in range(N)]                         # table i holds the constant values[i]
    tabs = [torch.full((n,), v, dtype=torch.uint8, device="cuda") for v in values]
    torch.cuda.synchronize()
    ptrs = [t.data_ptr() for t in tabs]
    for mn, mx in ((1, 255), (2, 4)):
        pair, secs = gpu.gram_device_partial(ptrs, n, mn, mx)
        ok = [mn <= v <= mx for v in values]
        want = np.zeros((N, N), dtype=np.uint64)
        for i in range(N):
            for j in range(i, N):
                want[i, j] = n if ok[i] and ok[j] else 0
        assert np.array_equal(pair, want), (N, log4n, mn, mx)
    print(f"gram N={N} n=4^{log4n} dense: kernel {secs * 1e3:.2f} ms -> {N * n / secs / 1e12:.2f} TB/s")
    del tabs
    torch.cuda.empty_cache()


def test_merge_host_path_sub_slices_and_threads(gpu, tmp_path, monkeypatch):
    """merger.pair_matrix on real files: a small HBM budget forces several sub-slices per device slice
    (partials accumulate in HBM), two host threads drive two plan entries on the same GPU at once, one
    table arrives as BGZF with a .gzi -- and only the bytes of each slice are read."""
    from pykmer_amd import bgzf, merger
    from pykmer_amd.header import Header
    from test_host_layer import _write_index
    import synth
    k, N = 9, 5
    paths, tables = [], []
    for i in range(N):
        fa, _ = synth.family(i, 60_000)
        h, got = _write_index(tmp_path, f"m{i}.fa", fa.tobytes(), k)
        paths.append(h.index_file_root)
        tables.append(got["table"])
    bgzf.compress_file(paths[2], level=1)
    headers = [Header(p, index_file=p) for p in paths]
    assert headers[2].index_file.endswith(".bgz")
    monkeypatch.setenv("PK_MERGE_HBM_BUDGET", str(N * (40_000 + 64)))
    assert len(merger._sub_slices(0, 4 ** k // 2, N, 0)) >= 3
    windows = [(1, 255), (2, 9)]
    got = merger.pair_matrix(headers, windows, threads=4, devices=(0, 0))
    for (mn, mx), pair in zip(windows, got):
        assert np.array_equal(gpu.gram_expand(pair), oracle.gram(tables, mn, mx)), (mn, mx)
    raw = [h.bytes_delivered for i, h in enumerate(headers) if i != 2]
    assert all(b == 4 ** k for b in raw)                                # every byte of a raw table exactly once
    assert headers[2].bytes_delivered <= 4 ** k + 2 * len(merger._sub_slices(0, 4 ** k // 2, N, 0)) * bgzf.BLOCK_INPUT


# ------------------------------------------------------------------ several windows in one pass ------------
SWEEP9 = [(1, 255), (2, 255), (1, 3), (2, 5), (128, 255), (100, 200), (1, 127), (129, 254), (255, 255)]


def _accumulate_windows(gpu, tables, windows):
    """Host tables -> device buffers -> ONE pk_gram_device_accumulate_windows call -> W x N x N."""
    N, n = len(tables), tables[0].size
    bufs = [gpu.DeviceBuffer(n, 0) for _ in range(N)]
    acc = gpu.DeviceBuffer(len(windows) * N * N * 8, 0)
    try:
        for b, t in zip(bufs, tables):
            b.upload(t)
        acc.zero()
        gpu.gram_device_accumulate_windows([b.ptr for b in bufs], n, acc.ptr, windows)
        return acc.download().view(np.uint64).reshape(len(windows), N, N)
    finally:
        for b in bufs + [acc]:
            b.free()


@pytest.mark.parametrize("N", [2, 5, 8, 9, 13, 16, 17, 24, 25, 32, 33, 49])
def test_window_sweep_in_one_pass_vs_oracle(gpu, N):
    """Every window of a sweep from one call (k_gram_mw: bit planes + ripple comparators; N > 32: one scan per window)
    against the oracle's pair loop (tools.py:473-482), for window lists that do and do not fill a pass, sorted and not."""
    rng = np.random.default_rng(300 + N)
    for n in (4 ** 7, 300_001):
        tables = _random_tables(rng, N, n)
        for windows in (SWEEP9, SWEEP9[:2], [(5, 9)], [(7, 255), (1, 255), (3, 255), (3, 20), (2, 255), (1, 1)]):
            got = _accumulate_windows(gpu, tables, windows)
            for w, (mn, mx) in enumerate(windows):
                assert np.array_equal(gpu.gram_expand(got[w]), oracle.gram(tables, mn, mx)), (N, n, mn, mx)
        if N > 9:
            break                                                       # the ragged size once per kernel shape is enough


def test_window_sweep_accumulates_and_rejects(gpu):
    rng = np.random.default_rng(5)
    tables = _random_tables(rng, 4, 4 ** 6)
    N, n = 4, tables[0].size
    bufs = [gpu.DeviceBuffer(n, 0) for _ in range(N)]
    acc = gpu.DeviceBuffer(2 * N * N * 8, 0)
    for b, t in zip(bufs, tables):
        b.upload(t)
    acc.zero()
    half = (n // 2) & ~31
    wins = [(1, 255), (2, 4)]
    gpu.gram_device_accumulate_windows([b.ptr for b in bufs], half, acc.ptr, wins)                     # two address sub-slices add up
    gpu.gram_device_accumulate_windows([b.ptr + half for b in bufs], n - half, acc.ptr, wins)
    got = acc.download().view(np.uint64).reshape(2, N, N)
    for w, (mn, mx) in enumerate(wins):
        assert np.array_equal(gpu.gram_expand(got[w]), oracle.gram(tables, mn, mx))
    for bad in ([(0, 255)], [(1, 256)], []):                            # merger.py:90-91
        with pytest.raises(ValueError):
            gpu.gram_device_accumulate_windows([b.ptr for b in bufs], n, acc.ptr, bad)


@pytest.mark.parametrize("N", [13, 32])
def test_window_sweep_full_size(gpu, N):
    """Eight thresholds over N tables of 4^15 bytes from one staging: every total and a spread of shared tallies against
    torch; the time of the sweep is printed beside the time of ONE single-window scan."""
    import torch
    torch.cuda.empty_cache()
    n = 4 ** 15
    tabs = _torch_tables(n, N, seed=70 + N)
    ptrs = [t.data_ptr() for t in tabs]
    windows = [(1, 255), (2, 255), (3, 255), (4, 255), (5, 200), (8, 255), (1, 50), (2, 20)]
    acc = torch.zeros((len(windows), N, N), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    secs = gpu.gram_device_accumulate_windows(ptrs, n, acc.data_ptr(), windows)
    acc.zero_()
    torch.cuda.synchronize()
    secs = gpu.gram_device_accumulate_windows(ptrs, n, acc.data_ptr(), windows)
    _, one = gpu.gram_device_partial(ptrs, n, 1, 255)
    got = acc.cpu().numpy()
    for w, (mn, mx) in enumerate(windows):
        valid = [(t >= mn) & (t <= mx) for t in tabs]
        for i in range(N):
            assert int(got[w, i, i]) == int(valid[i].sum().item()), (N, mn, mx, i)
        for i, j in ((0, 1), (2, N - 1), (N // 2, N // 2 + 1), (N - 2, N - 1)):
            assert int(got[w, i, j]) == int((valid[i] & valid[j]).sum().item()), (N, mn, mx, i, j)
        del valid
    print(f"sweep of {len(windows)} windows, N={N} k=15: {secs * 1e3:.2f} ms = {secs / one:.2f} x one scan ({one * 1e3:.2f} ms)")
    del tabs
    torch.cuda.empty_cache()


# ------------------------------------------------------------------ the RCCL branch of the product merge -----------------
def test_merge_through_rccl_group_of_one(gpu, tmp_path, manifest):
    """merger.merge(group=True) with torch.distributed on backend nccl (= RCCL), world_size 1: the branch a multi-GPU merge
    takes on every rank -- accumulator allocated by torch in HBM, pk_gram_device_accumulate_windows adds into it through
    acc_ptr, dist.all_reduce sums it -- on real files, against the reference's matrices and against the single-process path."""
    import socket
    import torch
    import torch.distributed as dist
    from pykmer_amd import merger
    from test_host_layer import _family_indexes
    paths = sorted(_family_indexes(tmp_path, manifest))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        assert dist.get_backend() == "nccl"
        seen = []
        real = merger.gpu_partial

        def spy(*a, **kw):
            seen.append(kw.get("acc_ptr"))
            return real(*a, **kw)
        merger.gpu_partial = spy
        try:
            wins = [(1, 255), (2, 255), (1, 3), (2, 5)]
            _, first = merger.merge(str(tmp_path / "rccl"), paths, group=True, windows=wins)
        finally:
            merger.gpu_partial = real
        assert len(seen) == 1 and seen[0], "the merge did not hand the kernel an accumulator in HBM (acc_ptr)"
    finally:
        dist.destroy_process_group()
    _, plain = merger.merge(str(tmp_path / "plain"), paths, windows=wins)
    assert np.array_equal(first, plain)
    for (mn, mx), tag in zip(wins, ("default", "min2", "max3", "min2max5")):
        want = np.array(manifest["merger"][f"G7_k7_n13_{tag}"]["matrix"], dtype=np.uint64)
        assert np.array_equal(np.load(tmp_path / f"rccl.{mn:03d}-{mx:03d}.kma")["matrix"], want), tag
        assert np.array_equal(np.load(tmp_path / f"plain.{mn:03d}-{mx:03d}.kma")["matrix"], want), tag


def test_merge_of_resident_tables(gpu):
    """merger.ResidentTable: tables that never left HBM (what bench.py merges) go through the same pair_matrix."""
    import synth
    from pykmer_amd import merger
    k, N = 9, 6
    n = 4 ** k
    host, bufs = [], []
    for i in range(N):
        fa, _ = synth.family(i, 50_000)
        host.append(gpu.count_fasta(fa, k)["table"])
        b = gpu.DeviceBuffer(n, 0)
        b.upload(host[-1])
        bufs.append(b)
    tabs = [merger.ResidentTable(b.ptr, n, n, device=0) for b in bufs]
    stats = {}
    got = merger.pair_matrix(tabs, [(1, 255), (2, 7)], devices=(0,), stats=stats)
    assert stats["kernel_seconds"] > 0
    for pair, (mn, mx) in zip(got, ((1, 255), (2, 7))):
        assert np.array_equal(gpu.gram_expand(pair), oracle.gram(host, mn, mx))
    # an address slice held on its own (a rank's share): the resident part starts at `first`
    lo, hi = merger.address_slice(n, 1, 3)
    part = [merger.ResidentTable(b.ptr + lo, hi - lo, n, device=0, first=lo) for b in bufs]
    sl = merger.gpu_partial(part, lo, hi, [(1, 255)], 0, 1)
    assert np.array_equal(gpu.gram_expand(sl[0]), oracle.gram([t[lo:hi] for t in host], 1, 255))
    for b in bufs:
        b.free()
