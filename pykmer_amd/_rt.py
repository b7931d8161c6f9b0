"""Opening libpykmer_hip.so -- ctypes only (no numpy), so a command-line host can bring the HIP runtime up in a thread
before its heavier imports (numpy alone takes ~0.18 s) have finished: see warm() and the root indexer.py / merger.py."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PK_LIB: another build of the same library (kernel experiments: tools/build_variant.sh); never a CPU stand-in
LIB_PATH = os.environ.get("PK_LIB") or os.path.join(_HERE, "libpykmer_hip.so")

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
        _handle = ctypes.CDLL(LIB_PATH)
    return _handle


def warm(device: int = 0) -> int:
    """pk_warm(device): HIP start-up and kernel load; returns the library's return code."""
    fn = open_library().pk_warm
    fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_int]
    return fn(device)
