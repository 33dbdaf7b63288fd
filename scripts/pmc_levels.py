#!/usr/bin/env python3
"""Per-dispatch table of an SQ counter pass (scripts/pmc_pass.sh) over the last frame of the run: one line per level
launch with duration, VALU instructions, VALU busy (SQ_ACTIVE_INST_VALU x 4 / (SIMDs x cycles)), lanes active per VALU
instruction, and their product.  usage: pmc_levels.py gpurun_out/pmc_TAG"""
import csv, glob, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
disp = defaultdict(dict); meta = {}
for row in csv.DictReader(open(f)):
    k = int(row["Dispatch_Id"])
    disp[k][row["Counter_Name"]] = disp[k].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    meta[k] = (row["Kernel_Name"].split("(")[0].replace("void ", ""), int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
ks = sorted(meta)
res = [k for k in ks if meta[k][0].startswith("k_resolve")]
if len(res) >= 2:
    ks = [k for k in ks if res[-2] < k <= res[-1]]
tot_us = 0.0
for k in ks:
    c = disp[k]; name, ns = meta[k]
    if "k_trace" not in name and "k_resolve" not in name:
        continue
    tot_us += ns / 1e3
    line = f"{name[:36]:36s} {ns / 1e3:9.1f} us"
    if "SQ_ACTIVE_INST_VALU" in c and c.get("GRBM_GUI_ACTIVE"):
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        busy = c["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cyc)
        lanes = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]) if c["SQ_ACTIVE_INST_VALU"] else 0.0
        line += (f"  valu_insts {c.get('SQ_INSTS_VALU', 0) / 1e6:9.1f} M  valu_busy {busy:.3f}  lanes {lanes:.3f}  busy_x_lanes {busy * lanes:.3f}"
                 f"  salu {c.get('SQ_INSTS_SALU', 0) / 1e6:8.1f} M  smem {c.get('SQ_INSTS_SMEM', 0) / 1e6:7.1f} M")
    print(line)
print(f"total {tot_us:.1f} us")
