#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/exp2; mkdir -p $O
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
grep -q " passed" $O/pytest.log || exit 1
grep -q "failed" $O/pytest.log && exit 1
timeout -k 10 300 python tools/bench_gram.py 13 > $O/gram.txt 2>&1; cat $O/gram.txt
timeout -k 10 200 python bench.py --no-cpu --no-merge --no-e2e --steps 20 --warmup 3 > $O/bench_k15.json 2> $O/bench_k15.err
timeout -k 10 200 python bench.py --k 17 --no-cpu --no-merge --no-e2e --steps 10 --warmup 2 > $O/bench_k17.json 2> $O/bench_k17.err
python - <<PY
import json
for k in (15, 17):
    d = json.load(open("$O/bench_k%d.json" % k))
    print(k, round(d["value"] / 1e9, 1), "Gbp/s", round(d["ms_per_step"], 3), "ms", {a: round(b, 3) for a, b in d["stage_ms"].items()})
PY
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-merge --no-e2e > $O/bench_traced.json 2>$O/trace.err
python tools/trace_gaps.py $O/trace > $O/gaps.txt 2>&1; cat $O/gaps.txt
timeout -k 10 300 python tools/profile_mix.py > $O/profile_mix.txt 2>&1; cat $O/profile_mix.txt
