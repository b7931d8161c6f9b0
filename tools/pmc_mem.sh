#!/bin/bash
# Memory-pipeline counters per kernel for a short bench run (on the MI355X box): tools/pmc_mem.sh <k> -> gpurun_out/pmc_mem_<k>.txt
# Separate passes (one group of counters each); never combined with sys / hip traces.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
K=${1:-15}
O=gpurun_out/pmc_mem_$K; rm -rf $O; mkdir -p $O
i=0
# ONLY the SQ group: on this pool the TA_* / TCP_* / TCC_* groups (tried: TA_TA_BUSY_sum ..., TCP_TCC_WRITE_REQ_sum ..., TCC_REQ_sum ...,
# TCC_EA0_WRREQ_sum ...) end with "rocprofv3 finalizing after signal 6 ... incomplete dispatches" after the time limit -- do not run them.
for G in "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $O/g$i -o c -- python3 bench.py --k $K --steps 2 --warmup 0 --no-cpu --no-merge --no-e2e > $O/bench_$i.json 2> $O/err_$i.txt || echo "group $i failed: $(tail -2 $O/err_$i.txt)"
done
python3 - "$O" <<'PY' > gpurun_out/pmc_mem_$K.txt
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('void ', '').replace('pk::', '')[:34]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
        dur[k].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in agg.items():
    d = sum(dur[k]) / len(dur[k]) / 1e3
    if d > 150: print(k, 'us', round(d, 1), {c.replace('_sum', ''): round(max(x) / 1e6, 2) for c, x in sorted(v.items())})
PY
cat gpurun_out/pmc_mem_$K.txt
