#!/bin/bash
# Where a wave's cycles go (SQ_WAIT_ANY + SQ_WAIT_INST_ANY + SQ_ACTIVE_INST_ANY ~ SQ_WAVE_CYCLES), default path and C5
tag=${1:-r03}
mkdir -p gpurun_out
PMC_WARMUP=1 bash scripts/pmc_pass.sh ${tag}_wait 0 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE || { tail -5 gpurun_out/pmc_${tag}_wait.err; exit 1; }
bash scripts/pmc_config.sh ${tag}_c5_wait C5 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE || exit 1
python3 - $tag <<'PY'
import json, sys
for f in (f"gpurun_out/pmc_{sys.argv[1]}_wait.json", f"gpurun_out/pmc_{sys.argv[1]}_c5_wait.json"):
    print(f)
    for k in json.load(open(f))[:4]:
        c = k["largest_dispatch"]; wc = c.get("SQ_WAVE_CYCLES", 0)
        if not wc: continue
        print(f"  {k['name'][:40]:40s} {c['duration_us']:8.0f} us  parked {c['SQ_WAIT_ANY']/wc:.3f}  issue-stall {c['SQ_WAIT_INST_ANY']/wc:.3f}  "
              f"active {c['SQ_ACTIVE_INST_ANY']/wc:.3f} (valu {c['SQ_ACTIVE_INST_VALU']/wc:.3f} scalar {c['SQ_ACTIVE_INST_SCA']/wc:.3f} "
              f"lds {c['SQ_ACTIVE_INST_LDS']/wc:.3f} vmem {c['SQ_ACTIVE_INST_VMEM']/wc:.3f})")
PY
