#!/bin/bash
# copies what tools/profile_round.sh collected (gpurun_out/profile_round) into profiles/ and derives hbm_traffic.json
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/profile_round; P=profiles; R=${1:-round02}
cp $O/bench.json $P/${R}_bench.json
cp $O/stats/ks_kernel_stats.csv $P/${R}_kernel_stats.csv
cp $O/bench_under_rocprof.json $P/${R}_bench_under_rocprof.json
cp $O/pmc_fetch/fetch_counter_collection.csv $P/${R}_pmc_fetch_counter_collection.csv
cp $O/pmc_write/write_counter_collection.csv $P/${R}_pmc_write_counter_collection.csv
cp $O/bench_k17.json $P/${R}_bench_k17.json
cp $O/e2e_cli.json $P/${R}_e2e_cli.json
cp $O/profile_mix.txt $P/${R}_profile_mix.txt
cp $O/pmc_sq/sq_counter_collection.csv $P/${R}_pmc_sq_counter_collection.csv
python tools/hbm_traffic.py $P/${R}_pmc_fetch_counter_collection.csv $P/${R}_pmc_write_counter_collection.csv > $P/hbm_traffic.json
if [ -f $O/pmc_fetch_k17/fetch_counter_collection.csv ]; then
  cp $O/pmc_fetch_k17/fetch_counter_collection.csv $P/${R}_k17_pmc_fetch_counter_collection.csv
  cp $O/pmc_write_k17/write_counter_collection.csv $P/${R}_k17_pmc_write_counter_collection.csv
  cp $O/stats_k17/ks_kernel_stats.csv $P/${R}_k17_kernel_stats.csv
  python tools/hbm_traffic.py $P/${R}_k17_pmc_fetch_counter_collection.csv $P/${R}_k17_pmc_write_counter_collection.csv > $P/hbm_traffic_k17.json
fi
for f in gram_sweeps.txt pmc_sq_k17.txt step_trace_gaps.txt; do [ -f $O/$f ] && cp $O/$f $P/${R}_$f; done
true
