#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp PK_EXPERIMENT=1
O=gpurun_out/exp11; mkdir -p $O
PK_LIB=$PWD/pykmer_amd/_build/libpykmer_hip_prof.so timeout -k 10 200 python bench.py --no-cpu --no-merge --no-e2e --steps 4 --warmup 1 > $O/k15.json 2> $O/k15.err; tail -3 $O/k15.err
PK_LIB=$PWD/pykmer_amd/_build/libpykmer_hip_prof.so timeout -k 10 200 python bench.py --k 17 --no-cpu --no-merge --no-e2e --steps 3 --warmup 1 > $O/k17.json 2> $O/k17.err; tail -2 $O/k17.err
