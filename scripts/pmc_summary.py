#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc run (…_counter_collection.csv): per kernel name, the dispatch with the
largest grid*duration, its counters and a few derived ratios.  Usage: pmc_summary.py DIR [out.json]"""
import csv, glob, json, os, sys
from collections import defaultdict

d = sys.argv[1]
f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
disp = defaultdict(dict)
meta = {}
for row in csv.DictReader(open(f)):
    k = int(row["Dispatch_Id"])
    disp[k][row["Counter_Name"]] = disp[k].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    meta[k] = (row["Kernel_Name"].split("(")[0], int(row["VGPR_Count"]), int(row["Grid_Size"]),
               int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
# with a warm-up frame in the run (two or more k_resolve dispatches) the first frame is left out: its launches pay for
# cold caches and first-touch page faults
res = sorted(k for k, m in meta.items() if m[0].startswith("k_resolve"))
if len(res) >= 2:
    for k in [k for k in meta if k <= res[0]]:
        del meta[k]
best = {}
tot = defaultdict(lambda: defaultdict(float))
for k, (name, vg, grid, ns) in meta.items():
    for c, v in disp[k].items():
        tot[name][c] += v
    tot[name]["duration_us"] += ns / 1e3
    tot[name]["dispatches"] += 1
    if name not in best or ns > meta[best[name]][3]:
        best[name] = k
out = []
for name, k in sorted(best.items(), key=lambda kv: -meta[kv[1]][3]):
    _, vg, grid, ns = meta[k]
    e = {"name": name, "vgpr": vg, "largest_dispatch": dict(disp[k], duration_us=ns / 1e3, grid=grid),
         "all_dispatches": dict(tot[name])}
    c = e["all_dispatches"]
    if "SQ_BUSY_CYCLES" in c and "SQ_ACTIVE_INST_VALU" in c and c["SQ_BUSY_CYCLES"]:
        # SQ_* cycle counters are in quad-cycles summed over SEs/XCDs; ratio of the two is what matters
        e["valu_active_per_busy"] = c["SQ_ACTIVE_INST_VALU"] / c["SQ_BUSY_CYCLES"]
    if "SQ_INSTS_VALU" in c and "SQ_THREAD_CYCLES_VALU" in c and c.get("SQ_ACTIVE_INST_VALU"):
        e["valu_lane_util"] = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
    out.append(e)
s = json.dumps(out, indent=1)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(s)
print(s)
