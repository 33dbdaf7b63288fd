#!/bin/bash
# Several builds of the library on one box, twice round:  scripts/abn.sh "libcftrt_prev.so libcftrt.so libcftrt_vA.so" "C3 C5"
libs=$1; cfgs=${2:-"C3"}
for pass in 1 2; do for L in $libs; do
  echo "== $L"
  CFTRT_LIB=$PWD/ray-tracing_amd/csrc/$L timeout -k 10 300 python3 scripts/bench_configs.py $cfgs 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(' ', d['config'], round(d['device_ms'], 2), 'ms  rescans', d['exact_rescan'], ' mean', [round(x, 6) for x in d['mean_rgb']])
" || exit 1
done; done
