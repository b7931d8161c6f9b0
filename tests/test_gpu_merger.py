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
