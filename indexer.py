#!/usr/bin/env python3
"""indexer.py <input.fa[.gz|.bgz]> <sample_name> <kmer_len>   (also: indexer.py <input.fa> <kmer_len>)

Drop-in for the reference's indexer.py CLI (indexer.py:475-495): writes `<abs input>.<kk>.kin` and
`<abs input>.<kk>.kin.json` next to the input.  The k-mer counting runs on an MI355X through
pykmer_amd (include/pykmer_hip.h); PK_DEVICE selects the GPU.
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
        _rt.warm(int(os.environ.get("PK_DEVICE", "0")))
        if os.environ.get("PK_TIMING"):
            print(f"[pk timing] device warm-up: {t1 - t0:.3f} s to load the library, {time.perf_counter() - t1:.3f} s in pk_warm", file=sys.stderr)
    except Exception:          # whatever is wrong is reported by the call that needs the device
        pass


if __name__ == "__main__":
    import threading
    threading.Thread(target=_warm_device, daemon=True).start()

from pykmer_amd.indexer import main  # noqa: E402

if __name__ == "__main__":
    main()
    # Everything is written, closed and renamed at this point.  PK_FAST_EXIT=1 leaves without tearing down the
    # interpreter and the HIP runtime (~0.15 s of a 0.9 s run): no atexit handlers, no flush of file objects other
    # than the two below -- so it is an opt-in for batch loops, not the default.
    if os.environ.get("PK_FAST_EXIT") == "1":
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)
