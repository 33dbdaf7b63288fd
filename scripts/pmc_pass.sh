#!/bin/bash
# One rocprofv3 --pmc pass over a short bench run: scripts/pmc_pass.sh TAG VARIANT COUNTERS...
# (run from the repo root on the GPU box; output gpurun_out/pmc_TAG/, summary gpurun_out/pmc_TAG.json;
#  PMC_WARMUP=1 adds an untimed first frame - scripts/hbm_traffic.py wants two frames)
set -e
tag=$1; shift
export CFTRT_VARIANT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -o p -- python3 bench.py --steps 1 --warmup ${PMC_WARMUP:-0} --no-cpu-baseline --no-plain-scan > gpurun_out/pmc_$tag.bench.json 2> gpurun_out/pmc_$tag.err
python3 scripts/pmc_summary.py gpurun_out/pmc_$tag gpurun_out/pmc_$tag.json > /dev/null
