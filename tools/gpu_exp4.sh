#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/exp4; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_merger.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
grep -q " passed" $O/pytest.log || exit 1
grep -q "failed" $O/pytest.log && exit 1
timeout -k 10 300 python tools/bench_gram.py 13 16 24 32 48 > $O/gram.txt 2>&1; cat $O/gram.txt
