#!/bin/bash
# Alternating A/B of builds of the library on one box (the clocks drift by a few percent over a minute: two passes of
# "A A B B" cannot resolve 2 %):  scripts/abalt.sh "libcftrt_prev.so libcftrt_vA.so" "C3" 6
libs=$1; cfgs=${2:-"C3"}; rounds=${3:-6}
for i in $(seq 1 $rounds); do for L in $libs; do
  CFTRT_LIB=$PWD/ray-tracing_amd/csrc/$L timeout -k 10 300 python3 scripts/bench_configs.py $cfgs 2>/dev/null | python3 -c "
import sys, json
o = []
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); o.append(d['config'][:2] + ' ' + str(round(d['device_ms'], 2)) + ' ' + d['image_sha1'][:6])
print('$L:', ' | '.join(o))
" || exit 1
done; done
