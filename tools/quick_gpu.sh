#!/bin/bash
# Quick GPU check of an indexer change (run on the MI355X box: `gpurun -- bash tools/quick_gpu.sh <tag>`):
# the indexer / slice parity tests, then the k=15 and k=17 bench lines without the CPU, merge and end-to-end legs.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
T=${1:-quick}
O=gpurun_out/$T
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_indexer.py tests/test_gpu_slices.py -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
grep -q "pytest rc=0" $O/pytest.log || exit 1
timeout -k 10 200 python bench.py --no-cpu --no-merge --no-e2e --steps 20 --warmup 3 > $O/bench_k15.json 2> $O/bench_k15.err || exit 2
timeout -k 10 200 python bench.py --k 17 --no-cpu --no-merge --no-e2e --steps 10 --warmup 2 > $O/bench_k17.json 2> $O/bench_k17.err || exit 3
python - <<PY
import json
for k in (15, 17):
    d = json.load(open("$O/bench_k%d.json" % k))
    print(k, round(d["value"] / 1e9, 1), "Gbp/s", round(d["ms_per_step"], 3), "ms", {a: round(b, 3) for a, b in d["stage_ms"].items()})
PY
