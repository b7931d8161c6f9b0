#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/exp1; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_indexer.py tests/test_gpu_slices.py tests/test_gpu_merger.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
grep -q "passed" $O/pytest.log || exit 1
timeout -k 10 300 python tools/bench_gram.py 13 32 > $O/gram.txt 2>&1; cat $O/gram.txt
bash tools/bench_variants.sh exp1 k17
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-merge --no-e2e > $O/bench_traced.json 2>$O/trace.err
python tools/trace_gaps.py $O/trace > $O/gaps.txt 2>&1; cat $O/gaps.txt
