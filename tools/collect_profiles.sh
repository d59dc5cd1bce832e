#!/bin/bash
# Round evidence in one gpurun call: bench lines, rocprofv3 kernel statistics and PMC summaries for C2 / C3 / C5 -> gpurun_out/final/
# (copy what is to be judged into profiles/ afterwards: tools/make_traffic_json.py does that and stamps the kernel-source hash).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/final
mkdir -p $OUT
for w in c2 c3 c5; do
  extra="--no-cpu-baseline"; [ $w = c2 ] && extra=""
  timeout -k 10 400 python3 bench.py --workload $w $extra > $OUT/${w}_bench.json 2> $OUT/${w}_bench.err || echo "bench $w failed"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/${w}_stats -o p --output-format csv -- python3 bench.py --workload $w --no-cpu-baseline > $OUT/${w}_bench_under_rocprof.json 2> $OUT/${w}_rocprof.err || echo "rocprof $w failed"
  cp $OUT/${w}_stats/p_kernel_stats.csv $OUT/${w}_kernel_stats.csv 2>/dev/null
  bash tools/pmc.sh final_$w $w > /dev/null 2>&1
  cp gpurun_out/pmc_final_$w/summary.txt $OUT/${w}_pmc_summary.txt
  echo "done $w"
done
timeout -k 10 200 python3 bench.py --workload dynamic --no-cpu-baseline > $OUT/dynamic_bench.json 2> $OUT/dynamic_bench.err || echo "dynamic failed"
timeout -k 10 200 python3 bench.py --rehearse-collective --steps 5 --warmup 2 --no-cpu-baseline > $OUT/c2_rehearse_collective.json 2> $OUT/c2_rehearse_collective.err || echo "rehearse failed"
rm -rf $OUT/*_stats
ls $OUT
