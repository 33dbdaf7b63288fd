#!/bin/bash
# round-3 experiment sweep (GPU box): microbenchmark, variants, grid cap, finish threshold, X4 build
out=gpurun_out/r3; mkdir -p $out
timeout -k 10 120 scripts/ubench_valu2 > $out/ubench_valu2.log 2>&1 || exit 1
run() { timeout -k 10 300 python3 scripts/bench_configs.py $1 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(' ', d['config'], round(d['device_ms'], 2), 'ms  rescans', d['exact_rescan'], 'mean', round(d['mean_rgb'][0], 6))
"; }
{
echo "== baseline"; run "C3 C5 C2 C1" || exit 1
echo "== baseline again"; run "C3" || exit 1
for v in 2 3; do echo "== VARIANT=$v"; CFTRT_VARIANT=$v run "C3 C5" || exit 1; done
for f in "1,2097152" "1,524288" "1,262144"; do echo "== FINISH=$f"; CFTRT_FINISH=$f run "C3" || exit 1; done
echo "== X4 lib"; CFTRT_LIB=$PWD/ray-tracing_amd/csrc/libcftrt_x4.so run "C3 C5" || exit 1
echo "== base lib"; run "C3 C5" || exit 1
echo "== X4 lib"; CFTRT_LIB=$PWD/ray-tracing_amd/csrc/libcftrt_x4.so run "C3 C5" || exit 1
} > $out/sweep1.log 2>&1
cat $out/ubench_valu2.log $out/sweep1.log
