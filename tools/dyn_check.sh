#!/bin/bash
# Developer helper (GPU box): builder tests, dynamic-workload bench line, build times and the kernel timeline of one dynamic update.
OUT=gpurun_out/dyn; mkdir -p $OUT
python -m pytest tests/test_gpu_fullscale.py tests/test_skinning.py tests/test_gpu_parity.py -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -20 $OUT/tests.log; exit 1; }
tail -1 $OUT/tests.log
python3 bench.py --workload dynamic --no-cpu-baseline > $OUT/dynamic_bench.json 2> $OUT/dynamic_bench.err || { tail $OUT/dynamic_bench.err; exit 1; }
python3 -c "import json;d=json.load(open('$OUT/dynamic_bench.json'));print('dynamic ms/step',d['ms_per_step'],d.get('dynamic'))"
timeout -k 10 200 python3 tools/build_prof.py c3 c5 > $OUT/build_times.txt 2>&1; cat $OUT/build_times.txt | grep -v amdgpu.ids
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/build_stats -o b -- python3 $GRAFT_REPO_ROOT/tools/build_prof.py c3 > /dev/null 2>&1 )
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/build_stats/**/b_kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]: print("%-60s %5s total %9.1f us"%(r["Name"][:60],r["Calls"],float(r["TotalDurationNs"])/1e3))
PY
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/dyn_trace -o d -- python3 $GRAFT_REPO_ROOT/bench.py --workload dynamic --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1 )
python3 tools/dynamic_timeline.py $(find $OUT/dyn_trace -name "d_kernel_trace.csv" | head -1) > $OUT/dynamic_update_timeline.txt 2>&1
cut -c1-110 $OUT/dynamic_update_timeline.txt
rm -rf $OUT/build_stats $OUT/dyn_trace
