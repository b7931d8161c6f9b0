#!/bin/bash
# SQ counters per kernel for the read-set-shaped input (400 000 records of 1 kbp), on the MI355X box -> gpurun_out/pmc_sq_reads.txt
set -e -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/pmc_sq_reads
rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O -o sq -- python3 tools/reads_probe.py 2 > $O/out.txt
python3 - "$O" <<'PY' > gpurun_out/pmc_sq_reads.txt
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
    dur[r['Kernel_Name'][:70]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in sorted(agg.items(), key=lambda kv: -sum(dur[kv[0]])):
    print(k, 'us', round(sum(dur[k]) / len(dur[k]) / 1e3, 1), {c: round(sum(x) / len(x) / 1e6, 2) for c, x in v.items()})
PY
