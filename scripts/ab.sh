#!/bin/bash
# A/B of two builds of the library on one box:  scripts/ab.sh "C1 C3 C5"   (libcftrt_prev.so against libcftrt.so, twice each)
cfgs=${1:-"C3"}
for L in libcftrt_prev.so libcftrt.so libcftrt_prev.so libcftrt.so; do
  echo "== $L"
  CFTRT_LIB=$PWD/ray-tracing_amd/csrc/$L timeout -k 10 300 python3 scripts/bench_configs.py $cfgs 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(' ', d['config'], round(d['device_ms'], 2), 'ms  rescans', d['exact_rescan'], ' mean', [round(x, 6) for x in d['mean_rgb']])
" || exit 1
done
