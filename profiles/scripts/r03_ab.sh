#!/bin/bash
# A/B of an environment switch on the setup path, interleaved on one box: r03_ab.sh VAR
VAR=$1
for rep in 1 2; do
  for v in "" 1; do
    if [ -z "$v" ]; then unset $VAR; tag="$VAR unset"; else export $VAR=1; tag="$VAR=1"; fi
    echo "== $tag"
    python3 profiles/scripts/r03_setup.py 15 2>&1 | grep -v destroy | head -1
    python3 profiles/scripts/r03_setup.py 10 4 16 2>&1 | grep -v destroy | head -1
    python3 profiles/scripts/r03_e2e_batch.py 2>&1 | grep "workers 4, group size None\|workers 8, group size None" | cut -c1-140
  done
done
