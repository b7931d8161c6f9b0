import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# Switches that select an alternate library or kernel (experiments: tools/build_variant.sh): a test run must
# measure and check the product as shipped, so they have to be unset.
EXPERIMENT_SWITCHES = ("PK_LIB", "PK_K15", "PK_K6_BYTES", "PK_GRAM_MW", "PK_DENSE_SHIFT", "PK_WG1", "PK_GRID2", "PK_XCD", "PK_SPARSE_MAX")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    set_ = [v for v in EXPERIMENT_SWITCHES if v in os.environ]
    assert not set_, f"experiment switches set in the environment: {set_} -- the tests check the library as shipped"
    # The shared library is a build artefact (git-ignored): on a fresh checkout build it once, the way
    # __graft_entry__.build() does (hipcc cross-compiles gfx950 without a GPU).  If hipcc is absent the
    # tests that need the library fail loudly with the ImportError from pykmer_amd._lib.load().
    import shutil
    from pykmer_amd import _lib, build as hip_build
    if not os.path.exists(_lib.LIB_PATH) and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        hip_build.build()


def _gpu_present() -> bool:
    try:
        from pykmer_amd import _lib
        return _lib.device_count() > 0
    except ImportError:
        # a GPU box without the built library must FAIL, not skip: decide from the device nodes
        return os.path.exists("/dev/kfd") and any(n.startswith("renderD") for n in os.listdir("/dev/dri")) \
            if os.path.isdir("/dev/dri") else False


@pytest.fixture(scope="session")
def gpu():
    """The loaded C-ABI library on a box with a GPU; skips only when there is no GPU at all."""
    if not _gpu_present():
        pytest.skip("no GPU visible")
    try:                 # one test uses torch for device tensors: bring it (and its HIP runtime) in before any GPU work
        import torch     # noqa: F401
    except ImportError:
        pass
    from pykmer_amd import _lib
    _lib.load()          # raises ImportError if the extension was not built: loud failure
    return _lib


@pytest.fixture(scope="session")
def manifest():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "manifest.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def small_tables():
    import numpy as np
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "tables_small.npz")))
