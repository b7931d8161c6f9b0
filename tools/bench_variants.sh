#!/bin/bash
# On the MI355X box: bench every variant library under pykmer_amd/_build (and the default build), k=15 and k=17.
#   gpurun -- bash tools/bench_variants.sh <tag> [k17]
cd "${GRAFT_REPO_ROOT:-.}"
export PK_EXPERIMENT=1
T=${1:-variants}; O=gpurun_out/$T; mkdir -p $O
for lib in default $(ls pykmer_amd/_build/libpykmer_hip_*.so 2>/dev/null); do
  n=$(basename $lib .so); n=${n#libpykmer_hip_}
  if [ "$lib" = default ]; then unset PK_LIB; else export PK_LIB=$PWD/$lib; fi
  timeout -k 10 120 python bench.py --no-cpu --no-merge --no-e2e --steps 20 --warmup 3 > $O/k15_$n.json 2> $O/k15_$n.err || { echo "$n k15 FAILED"; continue; }
  if [ "$2" = k17 ]; then timeout -k 10 120 python bench.py --k 17 --no-cpu --no-merge --no-e2e --steps 8 --warmup 2 > $O/k17_$n.json 2> $O/k17_$n.err || echo "$n k17 FAILED"; fi
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/k1*_*.json")):
    try: d = json.load(open(f))
    except Exception as e: print(f, "unreadable"); continue
    print(os.path.basename(f)[:-5].ljust(28), round(d["ms_per_step"], 3), "ms", round(d["value"] / 1e9, 1), "Gbp/s", " ".join("%s=%.3f" % (a.replace("bucket_layout_and_level2", "lvl2").replace("walk_sort_kernel", "walk").replace("structure_scans", "struct").replace("bucket_count", "count"), b) for a, b in d["stage_ms"].items() if b > 0.01))
PY
