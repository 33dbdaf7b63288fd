#!/usr/bin/env python3
"""HBM bytes per kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over a bench.py run.
Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md: both counters are in KB, and
FETCH_SIZE reports half the bytes of 16 B/lane reads, so it is doubled.  Only the LAST step's dispatches are
used (the run is `--steps 1 --warmup 1`, so the second half of each kernel's dispatches).
Usage: hbm_traffic.py FETCH_DIR WRITE_DIR RAYS_PER_STEP out.json"""
import csv, glob, json, os, sys
from collections import defaultdict


full, order = {}, defaultdict(list)


def per_kernel(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    rows = defaultdict(list)   # base kernel name -> [(dispatch id, value)]
    acc = defaultdict(float)
    name = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = int(r["Dispatch_Id"])
        acc[k] += float(r["Counter_Value"])
        name[k] = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        full[k] = r["Kernel_Name"].split("(")[0].replace("void ", "")
    for k in sorted(acc):
        rows[name[k]].append(acc[k])
        order[counter].append((full[k], acc[k]))
    return rows


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
rays = int(sys.argv[3])
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 1 --warmup 1`; "
               "last step only; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half for 16 B/lane "
               "reads); KB -> bytes x1024",
       "rays_per_step": rays, "kernels": {}}
for k in fetch:
    if not k.startswith("k_"):
        continue
    fv, wv = fetch[k], write.get(k, [])
    half_f, half_w = fv[len(fv) // 2:], wv[len(wv) // 2:]
    fb = 2.0 * 1024.0 * sum(half_f)
    wb = 1024.0 * sum(half_w)
    out["kernels"][k] = {"launches": len(half_f), "fetch_bytes_corrected": fb, "write_bytes": wb,
                         "bytes_per_ray": (fb + wb) / rays,
                         "largest_launch_fetch_bytes_corrected": 2.0 * 1024.0 * max(half_f),
                         "largest_launch_write_bytes": 1024.0 * max(half_w) if half_w else 0.0}
# the level launches of the last frame one by one (the two passes dispatch the same sequence): the level-0 launch is
# the instance that generates (<..., true>); its own algorithmic bytes are stated beside the counters
lv_f = [x for x in order["FETCH_SIZE"] if x[0].startswith("k_trace_level")]
lv_w = [x for x in order["WRITE_SIZE"] if x[0].startswith("k_trace_level")]
lv_f, lv_w = lv_f[len(lv_f) // 2:], lv_w[len(lv_w) // 2:]
out["level_launches"] = [{"level": i, "kernel": f[0], "fetch_bytes_corrected": 2.0 * 1024.0 * f[1], "write_bytes": 1024.0 * w[1]}
                         for i, (f, w) in enumerate(zip(lv_f, lv_w))]
if len(sys.argv) > 6:   # PATHS_PER_STEP LEVEL1_RAYS: what level 0 has to move by itself
    paths, l1 = int(sys.argv[5]), int(sys.argv[6])
    l0 = out["level_launches"][0]
    own = {"reads: its generated rays back from the queue, 24 B each (16 + 8)": 24.0 * paths,
           "writes: its generated rays into the queue, 24 B each": 24.0 * paths,
           "writes: survivors into the next level's queue, 40 B each": 40.0 * l1,
           "writes: radiance of the paths that end here, 12 B each": 12.0 * (paths - l1)}
    rd = sum(v for k, v in own.items() if k.startswith("reads"))
    wr = sum(v for k, v in own.items() if k.startswith("writes"))
    out["level0_accounting"] = {"paths": paths, "level1_rays": l1, "algorithmic_bytes": own,
                                "fetch_over_own_reads": l0["fetch_bytes_corrected"] / rd,
                                "write_over_own_writes": l0["write_bytes"] / wr}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out, indent=1))
