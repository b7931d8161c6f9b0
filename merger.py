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

from pykmer_amd.merger import main  # noqa: E402

if __name__ == "__main__":
    main()
