#!/usr/bin/env python3
"""calculate_distance.py <project.MIN-MAX.kma>

Drop-in for the reference's calculate_distance.py CLI (calculate_distance.py:243-251): Jaccard distance
matrix + neighbour-joining tree from a `.kma`, written next to it (`.dist.jaccard.npz`, `.mat.*`,
`.newick`, `.tree`).  No scikit-bio / ete3 needed; the PNG is not rendered.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from pykmer_amd.distance import main  # noqa: E402

if __name__ == "__main__":
    main()
