#!/bin/bash
# Whole GPU suite, then the default bench line (run on the MI355X box: `gpurun --timeout 1200 -- bash tools/gpu_all.sh <tag>`).
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
T=${1:-all}
O=gpurun_out/$T
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu --maxfail=6 -q -s --durations=15 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -30 $O/pytest.log
grep -q "pytest rc=0" $O/pytest.log || exit 1
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 2; }
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print(round(d["value"] / 1e9, 1), "Gbp/s", round(d["ms_per_step"], 3), "ms", {a: round(b, 3) for a, b in d["stage_ms"].items()})
print("roofline", d["roofline"]["frac"], "merge", {k: v for k, v in d.get("merge", {}).items() if "seconds" in k or "sweep" in k})
print("merge32", {k: v for k, v in d.get("merge_n32", {}).items() if "seconds" in k or "sweep" in k or "error" in k})
print("e2e", {k: v.get("t_e2e_s") if isinstance(v, dict) else v for k, v in d.get("e2e", {}).items()})
PY
