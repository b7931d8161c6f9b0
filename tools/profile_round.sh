#!/bin/bash
# Collects what profiles/ holds for a round (run on the MI355X box: `gpurun -- bash tools/profile_round.sh`).
# Separate passes: the bench line, rocprofv3 kernel stats, and one PMC counter per pass (never combined
# with the hip/hsa/sys trace domains).
set -e -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/profile_round
mkdir -p $O
timeout -k 10 600 python bench.py > $O/bench.json
echo "bench done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o ks -- python3 bench.py --steps 20 --warmup 5 --no-cpu --no-merge --no-e2e > $O/bench_under_rocprof.json
echo "stats done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o fetch -- python3 bench.py --steps 2 --warmup 0 --no-cpu --no-e2e > $O/pmc_fetch.log
echo "fetch done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o write -- python3 bench.py --steps 2 --warmup 0 --no-cpu --no-e2e > $O/pmc_write.log
echo "write done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_k17 -o fetch -- python3 bench.py --k 17 --steps 2 --warmup 0 --no-cpu --no-merge --no-e2e > $O/pmc_fetch_k17.log
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_k17 -o write -- python3 bench.py --k 17 --steps 2 --warmup 0 --no-cpu --no-merge --no-e2e > $O/pmc_write_k17.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_k17 -o ks -- python3 bench.py --k 17 --steps 10 --warmup 3 --no-cpu --no-merge --no-e2e > $O/bench_k17_under_rocprof.json
echo "k17 passes done"
timeout -k 10 300 python bench.py --k 17 --no-merge --no-cpu --no-e2e --steps 10 --warmup 3 > $O/bench_k17.json
PK_TMP=/dev/shm timeout -k 10 400 python tools/e2e_cli.py > $O/e2e_cli.json
timeout -k 10 300 python tools/profile_mix.py > $O/profile_mix.txt 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq -o sq -- python3 bench.py --steps 2 --warmup 0 --no-cpu --no-merge --no-e2e > $O/pmc_sq.log
find $O -name "*.csv" | head -20
# round 3 additions: threshold sweeps (k_gram_mw) against one scan, SQ counters of the k=17 step, a step's dispatch trace
timeout -k 10 300 python tools/bench_gram.py 13 32 > $O/gram_sweeps.txt 2>&1
bash tools/pmc_sq.sh 17 > /dev/null 2>&1 && cp gpurun_out/pmc_sq_17.txt $O/pmc_sq_k17.txt
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-merge --no-e2e > $O/bench_traced.json 2>/dev/null
python tools/trace_gaps.py $O/trace > $O/step_trace_gaps.txt 2>&1
