#!/bin/bash
# Builds a variant of libpykmer_hip.so with extra -D flags for a kernel experiment:
#   tools/build_variant.sh <name> [-DPK_PB_L1=8 ...]  ->  pykmer_amd/_build/libpykmer_hip_<name>.so   (select with PK_LIB=...)
set -e
cd "$(dirname "$0")/../pykmer_amd/csrc"
name=$1; shift
mkdir -p ../_build
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-result -Wno-unused-value "$@" \
  kmer_count.hip kmer_pack.hip kmer_fuse.hip kmer_part.hip gram_scan.hip pk_api.hip -o ../_build/libpykmer_hip_$name.so
echo ../_build/libpykmer_hip_$name.so
