"""Per-launch durations of the level kernels from a rocprofv3 --kernel-trace CSV (last frame of the run).
usage: level_times.py DIR   (DIR holds *_kernel_trace.csv)"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# frames are separated by k_generate launches
# frames are separated by k_generate launches (split pipeline) or end with k_resolve (fused: generation is part of level 0)
frames, cur = [], []
frames.append(cur)
for r in rows:
    n = r["Kernel_Name"]
    if "k_generate" in n or "k_resolve" in n:
        if cur:
            cur = []
            frames.append(cur)
    elif ("k_trace" in n or "k_intersect" in n or "k_shade" in n):
        cur.append((n.split("(")[0][:40], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
last = [f for f in frames if f][-1]
tot = sum(d for _, d in last)
print("launches", len(last), "total_us", round(tot, 1))
print(" ".join(f"{d:.0f}" for _, d in last))
