"""Summarise rocprofv3 --pmc CSV output per kernel (sum over dispatches)."""
import csv, glob, os, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True) + glob.glob(os.path.join(out, "*", "*counter_collection.csv")):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[k][row["Counter_Name"]] += 1
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", 0)):
    if not k.startswith("pt::"):
        continue
    print(k)
    for c, v in sorted(agg[k].items()):
        print(f"    {c:24s} {v:16.0f}   dispatches {calls[k][c]}")
