"""Opening libpykmer_hip.so -- ctypes only (no numpy), so a command-line host can bring the HIP runtime up in a thread
before its heavier imports (numpy alone takes ~0.18 s) have finished: see warm() and the root indexer.py / merger.py."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PK_LIB: another build of the same library (kernel experiments: tools/build_variant.sh); never a CPU stand-in --
# open_library() refuses one that does not answer pk_version() with this ABI or lacks the device entry points
LIB_PATH = os.environ.get("PK_LIB") or os.path.join(_HERE, "libpykmer_hip.so")
ABI_VERSION = 3                      # PK_ABI_VERSION of include/pykmer_hip.h

_handle = None


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch wheels bundle their own libamdhip64 / libhsa-runtime64 with the
    same SONAMEs as /opt/rocm's; whichever is loaded first serves both, and torch fails ("no ROCm-capable
    device") if the system copy got in first.  So when torch is installed, pull ITS copies in before our
    library resolves the SONAMEs (no `import torch` needed); without torch the system runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)


def open_library():
    """The CDLL handle (no signatures attached: pykmer_amd._lib.load does that).  There is no CPU fallback."""
    global _handle
    if _handle is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -m pykmer_amd.build` "
                              "(there is no CPU fallback)")
        _share_torch_hip_runtime()
        lib = ctypes.CDLL(LIB_PATH)
        missing = [n for n in ("pk_version", "pk_warm", "pk_device_count", "pk_indexer_feed_device", "pk_gram_device_partial")
                   if not hasattr(lib, n)]
        if missing:
            raise ImportError(f"{LIB_PATH} is not a build of libpykmer_hip.so: no {', '.join(missing)}")
        lib.pk_version.restype, lib.pk_version.argtypes = ctypes.c_int, []
        if lib.pk_version() != ABI_VERSION:
            raise ImportError(f"{LIB_PATH} speaks ABI {lib.pk_version()}, this package needs {ABI_VERSION}: rebuild it "
                              "(python -m pykmer_amd.build)")
        _handle = lib
    return _handle


def warm(device: int = 0) -> int:
    """pk_warm(device): HIP start-up and kernel load; returns the library's return code."""
    fn = open_library().pk_warm
    fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_int]
    return fn(device)
