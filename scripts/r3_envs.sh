#!/bin/bash
# Several settings of the environment knobs of scripts/bench_configs.py on one box, twice round:
#   scripts/r3_envs.sh "CFTRT_CLASSES=0 CFTRT_CLASSES=1,4" "C3 C5"     (each word: one run's assignments joined by '+')
sets=$1; cfgs=${2:-"C3"}
for pass in 1 2; do for kv in $sets; do
  echo "== $kv"
  ( for a in ${kv//+/ }; do export "$a"; done
    timeout -k 10 300 python3 scripts/bench_configs.py $cfgs 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(' ', d['config'], round(d['device_ms'], 2), 'ms  rescans', d['exact_rescan'], d['image_sha1'], ' mean', [round(x, 6) for x in d['mean_rgb']])
    else: print(l.rstrip())
" ) || exit 1
done; done
