#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/exp5; mkdir -p $O
CNT="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"
rm -rf $O/pmc
timeout -k 10 200 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $O/pmc -o sq -- python3 tools/reads_probe.py 2 > $O/probe.txt 2>&1
python3 - "$O/pmc" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
    dur[r['Kernel_Name'][:48]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in agg.items():
    d = sum(dur[k]) / len(dur[k]) / 1e3
    if d > 30: print(k, 'us', round(d, 1), {c.replace('SQ_', ''): round(sum(x) / len(x) / 1e6, 2) for c, x in v.items()})
PY
timeout -k 10 300 python tools/bench_gram.py 8 > $O/gram.txt 2>&1; grep "N=  8\|N= 13 sweep" $O/gram.txt
