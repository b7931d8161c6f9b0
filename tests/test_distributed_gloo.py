"""CPU: the multi-GPU merge path with world_size=2 over gloo.  Each rank tallies only its slice of the
k-mer address range, the N x N partials are summed by one all-reduce, rank 0 writes the .kma.  The
slice tallies come from the oracle here (no GPU in this suite); on GPUs the same code path calls
pk_gram_device_partial and the all-reduce runs over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, workdir, paths, mn, mx):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from pykmer_amd import merger
    from test_host_layer import _oracle_partial
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        seen = []

        def partial(headers, lo, hi, *a):
            out = _oracle_partial(headers, lo, hi, *a)               # (windows, device, threads) pass through
            seen.append((lo, hi, sum(h.bytes_delivered for h in headers)))
            return out
        data, matrix = merger.merge(os.path.join(workdir, "dist"), paths, min_count=mn, max_count=mx, group=True, partial_fn=partial)
        np.save(os.path.join(workdir, f"matrix_rank{rank}.npy"), matrix)
        np.save(os.path.join(workdir, f"slice_rank{rank}.npy"), np.array(seen))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_merge_matches_reference_matrix(tmp_path, manifest, world):
    import torch.multiprocessing as mp
    from test_host_layer import _family_indexes
    case = manifest["merger"]["G7_k7_n13_min2"]
    paths = sorted(_family_indexes(tmp_path, manifest))
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), paths, 2, 255), nprocs=world, join=True)
    want = np.array(case["matrix"], dtype=np.uint64)
    slices = []
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"matrix_rank{r}.npy"), want)       # every rank holds the reduced matrix
        lo, hi, delivered = (int(v) for v in np.load(tmp_path / f"slice_rank{r}.npy")[0])
        slices.append((lo, hi))
        # each rank read only its share of every table: N * 4^k / world bytes (+ rounding to 32 addresses)
        assert delivered == len(paths) * (hi - lo) <= len(paths) * (4 ** 7 // world + 32)
    assert slices[0][0] == 0 and slices[-1][1] == 4 ** 7                             # disjoint cover of the address range
    assert all(a[1] == b[0] for a, b in zip(slices[:-1], slices[1:]))
    kma = np.load(tmp_path / "dist.002-255.kma")["matrix"]                           # written once, by rank 0
    assert np.array_equal(kma, want)
