#!/bin/bash
# Per-launch times of one config of scripts/bench_configs.py under rocprofv3 --kernel-trace:
#   scripts/trace_config.sh TAG CONFIG [env assignments...]     (CONFIG: C1 C2 C2p C3 C4 C5)
tag=$1; cfg=$2; shift; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/trace_$tag
rm -rf $out && mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 scripts/bench_configs.py $cfg > $out/config.jsonl 2> $out/err.txt || exit 1
python3 scripts/level_times.py $out > $out/level_times.txt
python3 -c "import json;d=json.loads(open('$out/config.jsonl').readline());print('$tag', d['config'], round(d['device_ms'],2),'ms', round(d['mrays_per_s']),'Mrays/s', 'rescans', d['exact_rescan'])"
cat $out/level_times.txt
