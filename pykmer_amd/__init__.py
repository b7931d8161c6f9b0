"""pykmer_amd -- MI355X-native engine for pykmer's k-mer index (.kin) and merge (.kma) hot path.

Host side (Python, mirrors the reference's tools.Header / indexer / merger) over a C-ABI shared
library of hand-written HIP kernels for gfx950.  No CPU fallback: without libpykmer_hip.so and a
GPU the compute calls raise.
"""
__version__ = "0.1.0"
