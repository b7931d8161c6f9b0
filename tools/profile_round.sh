#!/bin/bash
# Collects what profiles/ holds for a round (run on the MI355X box: `gpurun -- bash tools/profile_round.sh`).
# Separate passes: the bench line, rocprofv3 kernel stats, and one PMC counter per pass (never combined
# with the hip/hsa/sys trace domains).
set -e -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/profile_round
mkdir -p $O
timeout -k 10 300 python bench.py > $O/bench.json
echo "bench done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o ks -- python3 bench.py --steps 5 --warmup 1 --no-cpu --no-merge > $O/bench_under_rocprof.json
echo "stats done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o fetch -- python3 bench.py --steps 2 --warmup 0 --no-cpu > $O/pmc_fetch.log
echo "fetch done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o write -- python3 bench.py --steps 2 --warmup 0 --no-cpu > $O/pmc_write.log
echo "write done"
timeout -k 10 300 python bench.py --k 17 --no-merge --no-cpu --steps 5 --warmup 1 > $O/bench_k17.json
timeout -k 10 300 python tools/e2e_cli.py > $O/e2e_cli.json
find $O -name "*.csv" | head -20
