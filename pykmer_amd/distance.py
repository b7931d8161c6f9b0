"""`.kma` -> Jaccard distance matrix -> neighbour-joining tree  (SURVEY.md 8f row f3).

Host-side closing step of the pipeline, mirroring the reference's calculate_distance.py without its
scikit-bio / ete3 / xvfb dependencies (none is installed here): same inputs (`<proj>.kma`,
`<proj>.kma.json`, optional `<proj>.kma.names.tsv`), same output file names, same distance
formula.  N is a dozen or so samples: plain numpy on the host, nothing for the GPU to do.

Parity notes: the distance arithmetic restates calculate_distance.py:82-97 exactly.  The tree is
built with the Saitou-Nei neighbour joining that skbio.tree.nj implements, in skbio's order of operations
(see neighbor_joining); skbio itself is absent here, so the pin is the worked 5-taxon example the reference
quotes at calculate_distance.py:128-134, whose newick and lsmat texts as the skbio documentation prints them
are committed under tests/golden/ (nj_five_taxa.*).  Anything that example does not exercise (ties among
equal Q entries, float formatting of other values) stays "parity unpinned".  The PNG rendering (ete3 + xvfb)
is not reproduced.
"""
import json
import sys
from pathlib import Path
from typing import Dict, List, Tuple

import numpy as np


def read_names_file(names_file: Path) -> Dict[str, str]:
    """calculate_distance.py:20-26: two tab-separated columns, sample file name -> display name."""
    names = {}
    with Path(names_file).open("rt") as fh:
        for row in fh:
            cols = row.split("\t")
            if len(cols) == 2:
                names[cols[0].strip()] = cols[1].strip()
    return names


def get_matrix(matrix_file: Path) -> np.ndarray:
    """calculate_distance.py:28-41: the (N,N,3) uint64 array stored under key `matrix`."""
    npz = np.load(Path(matrix_file))
    assert "matrix" in npz
    return npz["matrix"]


def jaccard_distance(matrix: np.ndarray, fill_diagonal: bool = True) -> np.ndarray:
    """calculate_distance.py:82-97: 1 - shared / (total_i + total_j - shared), zero diagonal."""
    shared = matrix[:, :, 2].astype(np.float64)
    total = matrix[:, :, 0:2].sum(axis=2).astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):       # the diagonal is 0/0 before it is overwritten
        dist = 1.0 - (shared / (total - shared))
    if fill_diagonal:
        np.fill_diagonal(dist, 0.0)
    return dist


def calc_distance(matrix_file: Path, matrix: np.ndarray, fill_diagonal: bool = True) -> Tuple[Path, np.ndarray]:
    dist = jaccard_distance(matrix, fill_diagonal)
    basefile = Path(f"{matrix_file}.dist.jaccard")
    with Path(f"{basefile}.npz").open("wb") as fh:
        np.savez(fh, distance=dist)                             # calculate_distance.py:105-107
    return basefile, dist


def condensed_form(d: np.ndarray) -> np.ndarray:
    """Upper triangle, row-major (what skbio's DistanceMatrix.condensed_form / scipy squareform return)."""
    iu = np.triu_indices(d.shape[0], k=1)
    return d[iu]


def neighbor_joining(dist: np.ndarray, ids: List[str]) -> str:
    """Saitou & Nei (1987) neighbour joining -> newick string with branch lengths, in the order of operations of
    skbio.tree.nj (calculate_distance.py:189 calls it with result_constructor=str): the pair with the lowest Q is looked
    for in the lower triangle row by row (first strict minimum wins ties), the member with the LARGER index is written
    first, the joined node goes to the front of the id list, negative lengths and collapsed distances are clamped to
    zero, and the last three nodes close as (ids[1], ids[0], ids[2]).  The 5-taxon example the reference quotes
    (calculate_distance.py:128-134) therefore prints exactly what the skbio documentation shows
    (tests/golden/nj_five_taxa.newick)."""
    n = len(ids)
    assert dist.shape == (n, n) and n >= 2
    d = dist.astype(np.float64).copy()
    nodes = [str(i) for i in ids]
    if n == 2:
        return f"({nodes[0]}:{d[0, 1] / 2:f}, {nodes[1]}:{d[0, 1] / 2:f});"

    def member_lengths(d, i, j):
        m = d.shape[0]
        li = 0.5 * d[i, j] + (d[i].sum() - d[j].sum()) / (2 * (m - 2))
        li = max(li, 0.0)                                       # disallow_negative_branch_length=True
        lj = max(d[i, j] - li, 0.0)
        return li, lj

    while len(nodes) > 3:
        m = len(nodes)
        r = d.sum(axis=1)
        q = (m - 2) * d - r[:, None] - r[None, :]
        q[np.triu_indices(m)] = np.inf                          # lower triangle only, row-major: i > j
        i, j = np.unravel_index(np.argmin(q), q.shape)
        li, lj = member_lengths(d, i, j)
        label = f"({nodes[i]}:{li:f}, {nodes[j]}:{lj:f})"
        keep = [x for x in range(m) if x not in (i, j)]
        new = np.maximum(0.5 * (d[i, keep] + d[j, keep] - d[i, j]), 0.0)
        nd = np.zeros((m - 1, m - 1))
        nd[1:, 1:] = d[np.ix_(keep, keep)]
        nd[0, 1:] = nd[1:, 0] = new
        d = nd
        nodes = [label] + [nodes[x] for x in keep]
    # three nodes left: one unrooted trifurcation
    l1, l2 = member_lengths(d, 1, 2)
    l0 = max(0.5 * (d[1, 0] + d[2, 0] - d[1, 2]), 0.0)
    return f"({nodes[1]}:{l1:f}, {nodes[0]}:{l0:f}, {nodes[2]}:{l2:f});"


def _ascii(newick: str) -> str:
    """A plain indented rendering of the tree (stands in for ete3's str(Tree), calculate_distance.py:209-213)."""
    out, depth, tok = [], 0, ""
    for ch in newick:
        if ch in "(),;":
            if tok.strip():
                out.append("   " * depth + "-- " + tok.strip())
            tok = ""
            depth += 1 if ch == "(" else -1 if ch == ")" else 0
        else:
            tok += ch
    return "\n".join(out) + "\n"


def write_lsmat(path: Path, d: np.ndarray, ids: List[str]) -> None:
    with Path(path).open("wt") as fh:
        fh.write("\t" + "\t".join(ids) + "\n")
        for name, row in zip(ids, d):
            fh.write(name + "\t" + "\t".join(str(float(x)) for x in row) + "\n")


def cluster_distance(matrix_file: Path, basefile: Path, distance: np.ndarray, names_file: Path = None,
                     load_header: bool = True) -> np.ndarray:
    """calculate_distance.py:111-231 minus the PNG."""
    if load_header:
        with Path(f"{matrix_file}.json").open("rt") as fh:
            header = json.load(fh)
        ids = [d["header"]["input_file_name"] for d in header["data"]]
        assert len(ids) == distance.shape[0]
    else:
        ids = [str(i + 1) for i in range(distance.shape[0])]
    if names_file:
        names = read_names_file(names_file)
        ids = [names.get(i, i) for i in ids]
    assert np.allclose(distance, distance.T) and not np.isnan(distance).any(), "distance matrix must be symmetric and finite"

    with Path(f"{basefile}.mat.redundant.np").open("wb") as fh:
        np.save(fh, distance, allow_pickle=False)
    write_lsmat(Path(f"{basefile}.mat.redundant.lsmat"), distance, ids)
    dmc = condensed_form(distance)
    with Path(f"{basefile}.mat.condensed.np").open("wb") as fh:
        np.save(fh, dmc, allow_pickle=False)
    with Path(f"{basefile}.mat.condensed.txt").open("wt") as fh:
        np.savetxt(fh, dmc)
    newick = neighbor_joining(distance, ids)
    Path(f"{basefile}.newick").write_text(newick)
    Path(f"{basefile}.tree").write_text(_ascii(newick))
    return distance


def load(matrix_file: Path, names_file: Path = None) -> np.ndarray:
    """calculate_distance.py:233-241."""
    matrix_file = Path(matrix_file)
    if names_file is None and Path(f"{matrix_file}.names.tsv").exists():
        names_file = Path(f"{matrix_file}.names.tsv")
    matrix = get_matrix(matrix_file)
    basefile, distance = calc_distance(matrix_file, matrix, fill_diagonal=True)
    return cluster_distance(matrix_file, basefile, distance, names_file=names_file)


def main(argv=None) -> None:
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 1:
        print("usage: calculate_distance.py <project.MIN-MAX.kma>")
        sys.exit(1)
    load(Path(argv[0]))
