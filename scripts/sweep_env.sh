#!/bin/bash
# Sweep one environment knob over bench_configs:  scripts/sweep_env.sh VAR "v1 v2 ..." "C3 C5"
var=$1; vals=$2; cfgs=${3:-"C3"}
for v in $vals; do
  echo "$var=$v"
  env "$var=$v" timeout -k 10 300 python3 scripts/bench_configs.py $cfgs 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(' ', d['config'], round(d['device_ms'], 2), 'ms  rescans', d['exact_rescan'])
" || exit 1
done
