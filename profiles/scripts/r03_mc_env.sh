#!/bin/bash
# config-5 solver-only throughput under an environment switch: r03_mc_env.sh VAR v1 v2 ...  (3 runs each)
VAR=$1; shift
for v in "$@"; do
  for i in 1 2 3; do
    env_line="$VAR=$v"
    if [ "$v" = "unset" ]; then unset "$VAR"; else export "$VAR=$v"; fi
    python3 bench.py --workload montecarlo --steps 6 --warmup 2 --no-probes 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$env_line', round(d['value']), round(d['ms_per_step'],2))"
  done
done
