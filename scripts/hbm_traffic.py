#!/usr/bin/env python3
"""HBM bytes per kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over a bench.py run.
Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md: both counters are in KB, and
FETCH_SIZE reports half the bytes of 16 B/lane reads, so it is doubled.  Only the LAST step's dispatches are
used (the run is `--steps 1 --warmup 1`, so the second half of each kernel's dispatches).
Usage: hbm_traffic.py FETCH_DIR WRITE_DIR RAYS_PER_STEP out.json"""
import csv, glob, json, os, sys
from collections import defaultdict


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
    for k in sorted(acc):
        rows[name[k]].append(acc[k])
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
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out, indent=1))
