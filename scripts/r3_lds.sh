#!/bin/bash
# LDS-resident grid on / off on one box (S500, scene.rs): time and image hash
cfgs=${1:-"C3"}
for v in 0 1 0 1; do
  echo "== CFTRT_LDS_GRID=$v"
  CFTRT_LDS_GRID=$v timeout -k 10 300 python3 scripts/bench_configs.py $cfgs 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(' ', d['config'], round(d['device_ms'], 2), 'ms  rescans', d['exact_rescan'], d['image_sha1'], ' mean', [round(x, 6) for x in d['mean_rgb']])
    else: print(l.rstrip())
" || exit 1
done
