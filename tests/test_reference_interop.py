"""Build-container only (skipped wherever /root/reference is absent, e.g. on the GPU box): the
reference's own Header and merger must accept the files this build writes."""
import contextlib
import io
import os
import sys
import types

import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "tools.py")), reason="reference not present")


def _ref_modules():
    sys.dont_write_bytecode = True
    sys.modules.setdefault("bgzip", types.ModuleType("bgzip"))          # tools.py:17 imports a module it never uses
    saved = {k: sys.modules.pop(k) for k in ("tools", "merger") if k in sys.modules}
    sys.path.insert(0, REF)
    try:
        import tools as ref_tools
        import merger as ref_merger
    finally:
        sys.path.remove(REF)
        for k in ("tools", "merger"):
            sys.modules.pop(k, None)
        sys.modules.update(saved)
    return ref_tools, ref_merger


def test_reference_reads_our_kin_and_merges_it(tmp_path, manifest):
    from test_host_layer import _family_indexes
    ref_tools, ref_merger = _ref_modules()
    paths = sorted(_family_indexes(tmp_path, manifest, n=3))
    h = ref_tools.Header(paths[0], index_file=paths[0])                 # parses our name + JSON, asserts the fixed keys
    assert h.kmer_len == 7 and h.data_size == 4 ** 7
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            got = ref_merger.calculate_distance(paths[0], paths[1], min_count=2)     # merger.py:62-78 on our files
    finally:
        os.chdir(cwd)
    case = manifest["merger"]["G7_k7_n13_min2"]
    assert tuple(got) == tuple(case["matrix"][0][1])
