#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/exp3; mkdir -p $O
CNT="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"
for W in 2 8; do
  rm -rf $O/pmc_$W
  timeout -k 10 200 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $O/pmc_$W -o sq -- python3 tools/sweep_probe.py 13 $W > $O/probe_$W.txt 2>&1
  python3 - "$O/pmc_$W" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'gram' not in r['Kernel_Name']: continue
    agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    dur[r['Kernel_Name'][:60]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in agg.items():
    print(k, 'us', round(sum(dur[k]) / len(dur[k]) / 1e3, 1), {c: round(sum(x) / len(x) / 1e6, 2) for c, x in v.items()})
PY
done
bash tools/bench_variants.sh exp3 k17
