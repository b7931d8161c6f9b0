#!/usr/bin/env python3
"""merger.py Project_Name a.kin[.bgz] b.kin[.bgz] ... [--min-count --max-count --buffer-size --block-size --threads]

Drop-in for the reference's merger.py CLI (merger.py:51-59,213-239): writes
`<project>.<min:03d>-<max:03d>.kma` (np.savez_compressed, key `matrix`, shape (N,N,3) uint64) and
`.kma.json`.  All pairs are tallied in one GPU pass over the tables; PK_DEVICES=0,1,.. splits the
k-mer address range over several GPUs.
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
        _rt.warm(int((os.environ.get("PK_DEVICES") or os.environ.get("PK_DEVICE", "0")).split(",")[0] or 0))
        if os.environ.get("PK_TIMING"):
            print(f"[pk timing] device warm-up: {t1 - t0:.3f} s to load the library, {time.perf_counter() - t1:.3f} s in pk_warm", file=sys.stderr)
    except Exception:          # whatever is wrong is reported by the call that needs the device
        pass


if __name__ == "__main__":
    import threading
    threading.Thread(target=_warm_device, daemon=True).start()

from pykmer_amd.merger import main  # noqa: E402

if __name__ == "__main__":
    main()
    # everything is written and renamed: leave without tearing down the interpreter and the HIP runtime (~0.15 s)
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(0)
