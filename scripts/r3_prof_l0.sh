#!/bin/bash
# Level-0 traffic accounting: L2 <-> fabric request counters per dispatch (separate --pmc passes, TCC block holds 4).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3
pass() { tag=$1; shift; PMC_WARMUP=1 bash scripts/pmc_pass.sh $tag 0 "$@" || echo "pass $tag failed"; tail -3 gpurun_out/pmc_$tag.err; }
pass r3_tcc1 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum
pass r3_tcc2 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum TCC_READ_sum
pass r3_fetch FETCH_SIZE
pass r3_write WRITE_SIZE
python3 - <<'PY'
import json
for t in ("r3_tcc1","r3_tcc2","r3_fetch","r3_write"):
    try:
        d=json.load(open(f"gpurun_out/pmc_{t}.json"))
    except Exception as e:
        print(t, "missing", e); continue
    for e in d:
        if "k_trace" in e["name"] or "k_resolve" in e["name"]:
            print(t, e["name"][:60], {k:(round(v) if isinstance(v,(int,float)) else v) for k,v in e["largest_dispatch"].items()})
PY
