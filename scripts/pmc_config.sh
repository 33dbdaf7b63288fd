#!/bin/bash
# One rocprofv3 --pmc pass over one config of scripts/bench_configs.py: scripts/pmc_config.sh TAG CONFIG COUNTERS...
# (run from the repo root on the GPU box; output gpurun_out/pmc_TAG/, summary gpurun_out/pmc_TAG.json)
set -e
tag=$1; shift
cfg=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -o p -- python3 scripts/bench_configs.py $cfg > gpurun_out/pmc_$tag.bench.json 2> gpurun_out/pmc_$tag.err
python3 scripts/pmc_summary.py gpurun_out/pmc_$tag gpurun_out/pmc_$tag.json > /dev/null
