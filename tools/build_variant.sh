#!/bin/bash
# Builds a variant of libpykmer_hip.so with extra -D flags for a kernel experiment:
#   tools/build_variant.sh <name> [-DPK_PB_L1=8 ...]  ->  pykmer_amd/_build/libpykmer_hip_<name>.so   (select with PK_LIB=...)
set -e
cd "$(dirname "$0")/../pykmer_amd/csrc"
name=$1; shift
mkdir -p ../_build/$name
for f in kmer_count kmer_pack kmer_fuse kmer_part gram_scan pk_api; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value "$@" -c $f.hip -o ../_build/$name/$f.o &
done
hipcc -O3 -std=c++17 -fPIC -c bgzf_host.cpp -o ../_build/$name/bgzf_host.o &
wait
hipcc --offload-arch=gfx950 -shared -fPIC ../_build/$name/*.o -lz -lpthread -o ../_build/libpykmer_hip_$name.so
echo ../_build/libpykmer_hip_$name.so
