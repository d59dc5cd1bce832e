#!/bin/bash
# Developer helper: HIP API trace of the dynamic workload; counts allocations and synchronisations inside the timed frames.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/hiptrace_dynamic
mkdir -p $OUT
rocprofv3 --hip-trace -d $OUT/t -o p --output-format csv -- python3 bench.py --workload dynamic --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python3 tools/hip_trace_summary.py $OUT | tee $OUT/summary.txt
