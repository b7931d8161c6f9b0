"""Builds libpykmer_hip.so (the HIP kernels + C-ABI) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so travels
to the GPU box with the repository snapshot (it is git-ignored, not gpurun-ignored).  Every source is
compiled to its own object (in parallel, only when it or a header changed) and the objects are linked.
"""
import glob
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libpykmer_hip.so")
SOURCES = ["kmer_count.hip", "kmer_pack.hip", "kmer_fuse.hip", "kmer_part.hip", "gram_scan.hip", "pk_api.hip", "bgzf_host.cpp"]   # the .cpp is host-only (zlib)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-unused-value"]


def _headers():
    """What every object depends on besides its own source: the headers under csrc/, the public header, this script."""
    return sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "pykmer_hip.h"), os.path.abspath(__file__)]


def _deps():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp"))) + _headers()


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in _deps())


def build(force: bool = False, verbose: bool = False, extra_flags=()) -> str:
    if not force and not extra_flags and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(OBJ, exist_ok=True)
    newest_header = max(os.path.getmtime(f) for f in _headers())
    tag = os.path.join(OBJ, "flags.txt")
    flags = FLAGS + list(extra_flags)
    if not os.path.exists(tag) or open(tag).read() != " ".join(flags):
        force = True

    def compile_one(src: str) -> str:
        obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        path = os.path.join(CSRC, src)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(newest_header, os.path.getmtime(path)):
            cmd = ([hipcc] + flags if src.endswith(".hip") else [hipcc, "-O3", "-std=c++17", "-fPIC"]) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd, cwd=CSRC)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    with open(tag, "w") as fh:
        fh.write(" ".join(flags))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-lz", "-lpthread", "-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd, cwd=CSRC)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, extra_flags=[a for a in sys.argv[1:] if a.startswith("-D")]))
