"""CPU: .kma -> Jaccard distance -> neighbour-joining tree."""
import re

import numpy as np

from pykmer_amd import distance, merger


def test_jaccard_formula():
    """calculate_distance.py:82-97: 1 - shared / (total_i + total_j - shared); diagonal forced to 0."""
    m = np.zeros((3, 3, 3), dtype=np.uint64)
    tot = [100, 50, 80]
    sh = {(0, 1): 25, (0, 2): 80, (1, 2): 0}
    for (i, j), s in sh.items():
        m[i, j] = (tot[i], tot[j], s)
        m[j, i] = (tot[j], tot[i], s)
    d = distance.jaccard_distance(m)
    assert d[0, 1] == 1 - 25 / (100 + 50 - 25) and d[1, 0] == d[0, 1]
    assert d[0, 2] == 1 - 80 / (100 + 80 - 80) and d[1, 2] == 1.0
    assert (np.diagonal(d) == 0).all() and not np.isnan(d).any()


def _leaf_lengths(newick):
    return {m.group(1): float(m.group(2)) for m in re.finditer(r"([A-Za-z0-9_.]+):([0-9.]+)", newick)}


def test_neighbor_joining_worked_example():
    """The 5-taxon example quoted in calculate_distance.py:128-134 (from the skbio.tree.nj documentation):
    expected tree (d:2, (c:4, (b:3, a:2):3):2, e:1)."""
    data = np.array([[0, 5, 9, 9, 8], [5, 0, 10, 10, 9], [9, 10, 0, 8, 7], [9, 10, 8, 0, 3], [8, 9, 7, 3, 0]], dtype=float)
    nw = distance.neighbor_joining(data, list("abcde"))
    # the text skbio.tree.nj(dm, result_constructor=str) prints for this matrix in its documentation (skbio is absent here:
    # the committed expected files hold that documented output, newick and lsmat layout)
    import os
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    assert nw == "(d:2.000000, (c:4.000000, (b:3.000000, a:2.000000):3.000000):2.000000, e:1.000000);"
    assert nw == open(os.path.join(golden, "nj_five_taxa.newick")).read()
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        distance.write_lsmat(os.path.join(tmp, "m.lsmat"), data, list("abcde"))
        assert open(os.path.join(tmp, "m.lsmat")).read() == open(os.path.join(golden, "nj_five_taxa.lsmat")).read()
    assert nw.endswith(";") and nw.count("(") == nw.count(")") == 3
    assert _leaf_lengths(nw) == {"a": 2.0, "b": 3.0, "c": 4.0, "d": 2.0, "e": 1.0}
    assert re.search(r"\((a:2\.0+, b:3\.0+|b:3\.0+, a:2\.0+)\):3\.0+", nw)          # cherry (a,b) on a branch of length 3
    # additive check: path lengths in the tree reproduce the input distances (this matrix is tree-like)
    assert distance.neighbor_joining(np.array([[0, 4.0], [4.0, 0]]), ["x", "y"]) == "(x:2.000000, y:2.000000);"


def test_outputs_from_a_kma(tmp_path, manifest):
    from test_host_layer import _family_indexes, _oracle_partial
    paths = sorted(_family_indexes(tmp_path, manifest, n=6))
    proj = str(tmp_path / "proj")
    merger.merge(proj, paths, partial_fn=_oracle_partial)
    kma = proj + ".001-255.kma"
    (tmp_path / "proj.001-255.kma.names.tsv").write_text("s00.fa\tancestor\ns03.fa\tthird\n")
    d = distance.load(kma)
    base = kma + ".dist.jaccard"
    assert np.array_equal(np.load(base + ".npz")["distance"], d)                     # calculate_distance.py:105-107
    assert np.array_equal(np.load(base + ".mat.redundant.np"), d)
    cond = np.load(base + ".mat.condensed.np")
    assert cond.shape == (15,) and cond[0] == d[0, 1] and cond[-1] == d[4, 5]
    assert np.allclose(np.loadtxt(base + ".mat.condensed.txt"), cond)
    rows = open(base + ".mat.redundant.lsmat").read().splitlines()
    assert rows[0].split("\t")[1:] == ["ancestor", "s01.fa", "s02.fa", "third", "s04.fa", "s05.fa"]
    assert rows[1].split("\t")[0] == "ancestor" and float(rows[1].split("\t")[2]) == d[0, 1]
    nw = open(base + ".newick").read()
    assert set(_leaf_lengths(nw)) >= {"ancestor", "third"} and nw.count(",") == 5
    assert open(base + ".tree").read().count("--") >= 6
    # more mutated family members sit further from the ancestor
    assert d[0, 1] < d[0, 3] < d[0, 5]
