#!/bin/bash
# Where the wall time of `indexer.py` goes (run on the MI355X box): PK_TIMING=1 marks on stderr, for a small and the 800 Mbp genome.
cd "${GRAFT_REPO_ROOT:-.}"
D=$(mktemp -d -p /dev/shm)
python - "$D" <<PY
import sys, os
sys.path.insert(0, ".")
import synth
synth.c2(800_000_000)[0].tofile(os.path.join(sys.argv[1], "big.fa"))
synth.family(0, 20_000_000)[0].tofile(os.path.join(sys.argv[1], "small.fa"))
PY
for f in small big small big; do
  echo "== $f"; t0=$(date +%s.%N); PK_TIMING=1 python indexer.py $D/$f.fa x 15 2>&1 >/dev/null | grep -E "pk timing|rror"; python -c "import time,sys; print(\"wall\", round(time.time()-float(sys.argv[1]),3), \"s\")" $t0
done
rm -rf $D
