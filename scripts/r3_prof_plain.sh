#!/bin/bash
# SQ counters of the plain linear scan (grid, wave cull and tile lists off): what roofline.frac is measured on.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export CFTRT_GRID=0 CFTRT_WAVE_CULL=0 CFTRT_TILE_LISTS=0
PMC_WARMUP=1 bash scripts/pmc_pass.sh r3_sq_plain 0 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE
PMC_WARMUP=1 bash scripts/pmc_pass.sh r3_sq_plain2 0 SQ_WAIT_ANY SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
python3 - <<'PY'
import csv, glob, json
from collections import defaultdict
for tag in ("r3_sq_plain", "r3_sq_plain2"):
    f = glob.glob(f"gpurun_out/pmc_{tag}/**/*counter_collection.csv", recursive=True)[0]
    disp = defaultdict(dict); meta = {}
    for row in csv.DictReader(open(f)):
        k = int(row["Dispatch_Id"])
        disp[k][row["Counter_Name"]] = disp[k].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        meta[k] = (row["Kernel_Name"].split("(")[0], int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    ks = sorted(meta)
    res = [k for k in ks if meta[k][0].startswith("k_resolve")]
    ks = [k for k in ks if k > res[0]] if len(res) >= 2 else ks
    for k in ks:
        c = disp[k]; name, ns = meta[k]
        if "k_trace" not in name: continue
        line = f"{tag} {name[:34]:34s} {ns/1e3:8.1f} us"
        if "SQ_ACTIVE_INST_VALU" in c:
            cyc = c["GRBM_GUI_ACTIVE"] / 8
            line += f" valu_insts {c['SQ_INSTS_VALU']/1e6:8.1f}M busy {c['SQ_ACTIVE_INST_VALU']*4/(1024*cyc):.3f} lanes {c['SQ_THREAD_CYCLES_VALU']/(64*c['SQ_ACTIVE_INST_VALU']):.3f} salu {c['SQ_INSTS_SALU']/1e6:.1f}M smem {c['SQ_INSTS_SMEM']/1e6:.1f}M wait_inst/wave_cyc {c['SQ_WAIT_INST_ANY']/c['SQ_WAVE_CYCLES']:.3f} valu/wave_cyc {c['SQ_ACTIVE_INST_VALU']/c['SQ_WAVE_CYCLES']:.3f}"
        else:
            line += " " + " ".join(f"{n}={v/1e6:.1f}M" for n, v in sorted(c.items()))
        print(line)
PY
