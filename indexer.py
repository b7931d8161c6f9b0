#!/usr/bin/env python3
"""indexer.py <input.fa[.gz|.bgz]> <sample_name> <kmer_len>   (also: indexer.py <input.fa> <kmer_len>)

Drop-in for the reference's indexer.py CLI (indexer.py:475-495): writes `<abs input>.<kk>.kin` and
`<abs input>.<kk>.kin.json` next to the input.  The k-mer counting runs on an MI355X through
pykmer_amd (include/pykmer_hip.h); PK_DEVICE selects the GPU.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from pykmer_amd.indexer import main  # noqa: E402

if __name__ == "__main__":
    main()
