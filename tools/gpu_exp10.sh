#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/exp10; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_indexer.py -m gpu -x -q -k "device_feed_larger or config2" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
bash tools/bench_variants.sh exp10 k17
