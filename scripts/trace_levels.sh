#!/bin/bash
# Per-launch times of one frame under rocprofv3 --kernel-trace: scripts/trace_levels.sh TAG [env assignments...]
# (run on the GPU box from the repo root; prints the level times of the last frame)
tag=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/trace_$tag
rm -rf $out && mkdir -p $out
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-plain-scan > $out/bench.json 2> $out/err.txt || exit 1
python3 scripts/level_times.py $out > $out/level_times.txt
python3 -c "import json;d=json.load(open('$out/bench.json'));print('$tag', round(d['ms_per_step'],3),'ms', round(d['value']),'Mrays/s')"
cat $out/level_times.txt
