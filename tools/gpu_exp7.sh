#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/exp7; mkdir -p $O
timeout -k 10 800 python -m pytest tests/test_gpu_indexer.py tests/test_gpu_slices.py tests/test_gpu_cli.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
grep -q " passed" $O/pytest.log || exit 1
grep -q "failed" $O/pytest.log && exit 1
for i in 1 2; do timeout -k 10 200 python bench.py --no-cpu --no-merge --no-e2e --steps 20 --warmup 3 > $O/bench_k15_$i.json 2> $O/bench_k15.err; done
python - <<PY
import json
for i in (1, 2):
    d = json.load(open("$O/bench_k15_%d.json" % i))
    print(round(d["value"] / 1e9, 1), "Gbp/s", round(d["ms_per_step"], 3), "ms", {a: round(b, 3) for a, b in d["stage_ms"].items()})
PY
PK_TMP=/dev/shm timeout -k 10 500 python tools/e2e_cli.py > $O/e2e_cli.json 2> $O/e2e.err; cat $O/e2e_cli.json; tail -3 $O/e2e.err
