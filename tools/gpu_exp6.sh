#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/exp6; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_merger.py -m gpu -x -q -k "sweep or window" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout -k 10 300 python tools/bench_gram.py 8 13 > $O/gram.txt 2>&1; cat $O/gram.txt
PK_DIST_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu --no-e2e > $O/bench_2rank_gloo.json 2> $O/bench_2rank.err; echo "2-rank rc=$?"; tail -3 $O/bench_2rank.err
python - <<PY
import json
d = json.load(open("$O/bench_2rank_gloo.json"))
print(d["n_gpus"], round(d["value"] / 1e9, 1), "Gbp/s", round(d["ms_per_step"], 3), "ms; merge", {k: v for k, v in d["merge"].items() if k in ("seconds", "sharding", "path", "sweep8_seconds")})
PY
bash tools/profile_round.sh > $O/profile_round.log 2>&1; tail -5 $O/profile_round.log
