import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/t/**/*hip_api_trace.csv", recursive=True) or glob.glob(sys.argv[1] + "/t/*hip_api_trace.csv")
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Function"] for r in rows]
g = [i for i, n in enumerate(names) if n == "hipGraphLaunch"]
lo, hi = g[8], g[22]
c = collections.Counter(names[lo:hi])
print("HIP API calls per frame over 14 consecutive dynamic frames (skin -> BLAS refit -> TLAS rebuild -> G-buffer -> path tracer):")
for k, v in sorted(c.items(), key=lambda kv: -kv[1]): print(f"  {k:32s} {v/14:.1f}")
bad = {k: v for k, v in c.items() if k in ("hipMalloc", "hipFree", "hipStreamSynchronize", "hipDeviceSynchronize", "hipHostMalloc", "hipMallocAsync", "hipMemcpy", "hipEventSynchronize")}
print("allocations / synchronisations / blocking copies in the window:", bad or "none")
