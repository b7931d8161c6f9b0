"""Prints how many workgroups of the bucket-count / level-2 kernels the runtime places on one CU."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pykmer_amd import _lib
lib = _lib.load()
_lib.device_count()
for i, name in enumerate(["k_bucket_count_half", "k_bucket_count_bytes", "k_bucket_count_half_lean", "k_scatter2<claim>"]):
    print(name, lib.pk_diag_occupancy(i))
