#!/usr/bin/env python3
"""merger.py Project_Name a.kin[.bgz] b.kin[.bgz] ... [--min-count --max-count --buffer-size --block-size --threads]

Drop-in for the reference's merger.py CLI (merger.py:51-59,213-239): writes
`<project>.<min:03d>-<max:03d>.kma` (np.savez_compressed, key `matrix`, shape (N,N,3) uint64) and
`.kma.json`.  All pairs are tallied in one GPU pass over the tables.  Several GPUs: `--gpus N` (or torchrun) runs one
process per GPU, each scanning its slice of the k-mer address range, and sums the N x N partials with one RCCL
all-reduce; PK_DEVICES=0,1,.. splits the range over several GPUs inside one process instead.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _warm_device():
    """HIP start-up (~0.2 s) runs beside the imports and the argument / file set-up instead of after them."""
    try:
        import time
        t0 = time.perf_counter()
        from pykmer_amd import _rt         # ctypes only: starts before numpy is imported
        t1 = time.perf_counter()
        listed = [d for d in (os.environ.get("PK_DEVICES") or os.environ.get("PK_DEVICE", "0")).split(",") if d != ""] or ["0"]
        _rt.warm(int(listed[int(os.environ.get("LOCAL_RANK", "0")) % len(listed)]))
        if os.environ.get("PK_TIMING"):
            print(f"[pk timing] device warm-up: {t1 - t0:.3f} s to load the library, {time.perf_counter() - t1:.3f} s in pk_warm", file=sys.stderr)
    except Exception:          # whatever is wrong is reported by the call that needs the device
        pass


def _is_rank_launcher() -> bool:
    """`--gpus N` outside a launcher: this process only starts the ranks and must not touch a GPU itself."""
    if "WORLD_SIZE" in os.environ:
        return False
    return any(a == "--gpus" or a.startswith("--gpus=") for a in sys.argv[1:])


if __name__ == "__main__" and not _is_rank_launcher() and "WORLD_SIZE" not in os.environ:
    # (a rank of a multi-process merge picks its device when it joins the group: no warm-up guess here)
    import threading
    threading.Thread(target=_warm_device, daemon=True).start()

from pykmer_amd.merger import main  # noqa: E402

if __name__ == "__main__":
    main()
    # Everything is written, closed and renamed at this point.  PK_FAST_EXIT=1 leaves without tearing down the
    # interpreter and the HIP runtime (~0.15 s of a 0.9 s run): no atexit handlers, no flush of file objects other
    # than the two below -- so it is an opt-in for batch loops, not the default.
    if os.environ.get("PK_FAST_EXIT") == "1":
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)
