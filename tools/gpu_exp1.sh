#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/exp1; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_merger.py -m gpu -x -q -k "sweep" > $O/pytest_sweep.log 2>&1; tail -3 $O/pytest_sweep.log
timeout -k 10 300 python tools/bench_gram.py 13 32 > $O/gram.txt 2>&1; cat $O/gram.txt
bash tools/bench_variants.sh exp1 k17
