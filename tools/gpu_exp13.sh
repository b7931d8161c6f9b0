#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/exp13; mkdir -p $O
timeout -k 10 800 python -m pytest tests/test_gpu_indexer.py tests/test_gpu_slices.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
grep -q " passed" $O/pytest.log || exit 1
grep -q "failed" $O/pytest.log && exit 1
bash tools/bench_variants.sh exp13 k17
