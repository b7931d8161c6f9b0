"""GPU parity of the address-range-sharded indexer (SURVEY 8e option A / 8f f4): every shard streams the whole text and
keeps the canonical k-mers of its own address range; the shards' tables concatenate to the .kin image.  For k <= 17 that
image is compared byte for byte with the unsharded table (and with the oracle); k = 19 -- 4^19 = 256 GiB, which the
reference never ran (README.md:51-52) -- is 16 slices of 16 GiB each, checked against the oracle's k-mer list.
k = 19 has no reference output anywhere ("parity unpinned" beyond the oracle, whose algorithm the goldens pin at k <= 17)."""
import numpy as np
import pytest

import inputs
import oracle

pytestmark = pytest.mark.gpu


def _slices(gpu, data, k, n_slices, want_tables=True):
    out = []
    for s in range(n_slices):
        with gpu.Indexer(k, slice_index=s, n_slices=n_slices) as ix:
            ix.feed(data)
            fin = ix.finish()
            fin["records"] = ix.records(fin["n_records"])
            fin["table"] = ix.table_to_host() if want_tables else None
            out.append(fin)
    return out


@pytest.mark.parametrize("k,n_slices", [(7, 2), (9, 4), (13, 2), (15, 4), (15, 16)])
def test_slices_concatenate_to_the_unsharded_table(gpu, k, n_slices):
    import synth
    body, _ = synth.c2(3_000_000, seed=91)
    data = np.concatenate([np.frombuffer(inputs.edge_fasta(), dtype=np.uint8), body])
    want = oracle.count_fasta(data, k)
    whole = gpu.count_fasta(data, k)
    assert np.array_equal(whole["table"], want["table"])
    parts = _slices(gpu, data, k, n_slices)
    assert np.array_equal(np.concatenate([p["table"] for p in parts]), want["table"])
    for p in parts:
        assert p["num_kmers"] == want["num_kmers"] and p["total_bp"] == want["total_bp"]      # whole-input figures in every shard
        assert np.array_equal(p["records"]["n_valid_kmers"], want["records"]["n_valid_kmers"])
        assert np.array_equal(p["hist256"], np.bincount(p["table"], minlength=256).astype(np.uint64))
    assert sum(int(p["hist256"][1:].sum()) for p in parts) == int(np.count_nonzero(want["table"]))


def test_slices_k17(gpu):
    """4^17 in two slices of 8 GiB (64-bit k-mers, sliced)."""
    import synth
    data, _ = synth.c2(2_000_000, seed=92)
    kmers = oracle.kmer_list(data, 17)
    u, c = np.unique(kmers, return_counts=True)
    half = 4 ** 17 // 2
    for s, p in enumerate(_slices(gpu, data, 17, 2)):
        sel = (u >= s * half) & (u < (s + 1) * half)
        assert int(p["hist256"][1:].sum()) == int(sel.sum())
        assert np.array_equal(p["table"][(u[sel] - s * half).astype(np.int64)], np.minimum(c[sel], 255).astype(np.uint8))
        assert int(np.count_nonzero(p["table"])) == int(sel.sum())
        assert p["num_kmers"] == kmers.size


def test_k19_in_sixteen_slices(gpu):
    """k = 19: 38-bit addresses, windows deeper than one code dword, slices of 2^34 addresses.  Every slice's histogram
    against the oracle's k-mer list; three slices (the first -- poly-A lands at address 0 --, one inside, the last)
    byte for byte at the addresses that must be non-zero, plus the count of non-zero bytes."""
    import synth
    body, _ = synth.generate(93, 400_000, 3, pm_tandem=100, pm_dup=100, pm_ngap=30, pm_lower=50)
    data = np.concatenate([np.frombuffer(inputs.edge_fasta(), dtype=np.uint8), body,
                           np.frombuffer(b">polyA\n" + b"A" * 700 + b"\n" + b"ACGT" * 200 + b"\n", dtype=np.uint8)])
    k, n_slices = 19, 16
    kmers = oracle.kmer_list(data, k)
    u, c = np.unique(kmers, return_counts=True)
    sat = np.minimum(c, 255)
    size = 4 ** k // n_slices
    full = (0, 6, 15)
    for s in range(n_slices):
        sel = (u >> np.uint64(34)) == s
        with gpu.Indexer(k, slice_index=s, n_slices=n_slices) as ix:
            ix.feed(data)
            fin = ix.finish()
            h = fin["hist256"]
            assert fin["num_kmers"] == kmers.size
            assert int(h.sum()) == size and int(h[1:].sum()) == int(sel.sum()), s
            assert np.array_equal(h[1:], np.bincount(sat[sel], minlength=256)[1:].astype(np.uint64)), s
            if s in full:
                table = ix.table_to_host()
                assert np.array_equal(table[(u[sel] & np.uint64(size - 1)).astype(np.int64)], sat[sel].astype(np.uint8)), s
                nz = sum(int(np.count_nonzero(table[o:o + (1 << 30)])) for o in range(0, size, 1 << 30))
                assert nz == int(sel.sum()), s
                del table
    assert int((u >> np.uint64(34)).max()) <= 15 and sat.max() == 255


def test_slice_arguments(gpu):
    with pytest.raises(ValueError):
        gpu.Indexer(19)                                        # 256 GiB: needs slices
    with pytest.raises(ValueError):
        gpu.Indexer(19, slice_index=0, n_slices=8)             # 2^35 addresses per slice
    with pytest.raises(ValueError):
        gpu.Indexer(15, slice_index=3, n_slices=3)             # not a power of two
    with pytest.raises(ValueError):
        gpu.Indexer(15, slice_index=4, n_slices=4)
