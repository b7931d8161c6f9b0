#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/exp14; mkdir -p $O
timeout -k 10 800 python -m pytest tests/test_gpu_indexer.py tests/test_gpu_slices.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
grep -q " passed" $O/pytest.log || exit 1
grep -q "failed" $O/pytest.log && exit 1
for i in 1 2; do timeout -k 10 200 python bench.py --no-cpu --no-merge --no-e2e --steps 20 --warmup 3 > $O/bench_k15_$i.json 2> $O/bench_k15.err; done
timeout -k 10 200 python bench.py --k 17 --no-cpu --no-merge --no-e2e --steps 10 --warmup 2 > $O/bench_k17.json 2> $O/bench_k17.err
python - <<PY
import json
for f in ("k15_1", "k15_2", "k17"):
    d = json.load(open("$O/bench_%s.json" % f))
    print(f, round(d["value"] / 1e9, 1), "Gbp/s", round(d["ms_per_step"], 3), "ms", {a: round(b, 3) for a, b in d["stage_ms"].items()})
PY
PK_EXPERIMENT=1 PK_LIB=$PWD/pykmer_amd/_build/libpykmer_hip_prof.so timeout -k 10 200 python bench.py --no-cpu --no-merge --no-e2e --steps 2 --warmup 1 > $O/prof.json 2> $O/prof.err; tail -1 $O/prof.err
