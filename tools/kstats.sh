#!/bin/bash
# kernel-time table of a short bench run (run on the MI355X box): tools/kstats.sh <k> -> gpurun_out/kstats_<k>.txt
set -e -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
K=${1:-15}
O=gpurun_out/kstats_$K
rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o ks -- python3 bench.py --k $K --steps 5 --warmup 1 --no-cpu --no-merge --no-e2e > $O/bench.json
python3 - "$O" <<'PY' > gpurun_out/kstats_$K.txt
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:8.2f}")
PY
