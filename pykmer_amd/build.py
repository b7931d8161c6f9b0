"""Builds libpykmer_hip.so (the HIP kernels + C-ABI) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so travels
to the GPU box with the repository snapshot (it is git-ignored, not gpurun-ignored).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpykmer_hip.so")
SOURCES = ["kmer_count.hip", "kmer_pack.hip", "kmer_fuse.hip", "kmer_part.hip", "gram_scan.hip", "pk_api.hip"]


def _deps():
    """Everything the library is compiled from: every .hip / .h under csrc/ plus the public header."""
    import glob
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h"))) + \
        [os.path.join(HERE, "..", "include", "pykmer_hip.h"), os.path.abspath(__file__)]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in _deps())


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-result",
           "-Wno-unused-value"] + SOURCES + ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd, cwd=CSRC)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
