"""Timeline of the last frame in a rocprofv3 --kernel-trace CSV: per launch its start (us after the frame's first
launch), duration and the gap since the previous launch ended.   usage: timeline.py DIR"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
frames, cur = [], []
for r in rows:
    cur.append(r)
    if "k_resolve" in r["Kernel_Name"]:
        frames.append(cur); cur = []
fr = frames[-1]
t0, prev = int(fr[0]["Start_Timestamp"]), None
busy = 0
for r in fr:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0 if prev is None else (s - prev) / 1e3
    busy += e - s
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  gap {gap:7.1f}  {r['Kernel_Name'].split('(')[0][:60]}")
    prev = e
print(f"span {(prev - t0) / 1e3:.1f} us, kernels {busy / 1e3:.1f} us, gaps {(prev - t0 - busy) / 1e3:.1f} us")
