#!/bin/bash
# Developer helper: per-launch kernel durations of one frame (rocprofv3 --kernel-trace), single stream.
# usage: tools/launch_trace.sh <tag> <workload>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-t}; WL=${2:-c3}
OUT=gpurun_out/trace_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT/k -o p --output-format csv -- python3 bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --inflight 1 --no-kernel-timing > $OUT/bench.json 2> $OUT/bench.err
python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/k/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t0, out = None, []
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void pt::", "")
    if not any(k in n for k in ("k_round", "k_extend", "k_shade", "k_gbuffer", "k_pt_init")):
        continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.append((n[:30], (e - s) / 1e3, (s - (t0 or s)) / 1e3)); t0 = e
for o in out[-24:]:
    print("%-30s dur %8.1f us  gap %6.1f us" % o)
PY
