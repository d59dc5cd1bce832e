"""Developer helper: the GPU timeline of the dynamic workload's per-frame update (skin + refit + top-level rebuild), from a rocprofv3
kernel trace:   cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -o dyn -- python3 bench.py --workload dynamic --steps 6 --warmup 2 --no-cpu-baseline
then            python tools/dynamic_timeline.py OUT/.../dyn_kernel_trace.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# a frame's update starts at k_skin and ends before k_gbuffer
frames = []
i = 0
while i < len(rows):
    if "k_skin" in names[i]:
        j = i
        while j < len(rows) and "k_gbuffer" not in names[j]:
            j += 1
        if j < len(rows):
            frames.append((i, j))
        i = j
    i += 1
if not frames:
    sys.exit("no k_skin ... k_gbuffer sequence in the trace")
a, b = frames[-1]
t0 = int(rows[a]["Start_Timestamp"])
print(f"{len(frames)} dynamic frames in the trace; the last one's update: {b - a} kernels")
busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    print(f"  +{(s - t0) / 1e3:8.1f} us  {(e - s) / 1e3:7.1f} us  {r['Kernel_Name'][:70]}")
span = int(rows[b]["Start_Timestamp"]) - t0
print(f"span to the G-buffer kernel {span / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us")
