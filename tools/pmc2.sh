#!/bin/bash
# Developer helper: second set of PMC passes (lane utilisation, instruction cache, LDS, texture path) for one bench workload.
# usage: tools/pmc2.sh <tag> <workload>
set -e
TAG=${1:-pmc2}; WL=${2:-c3}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc2_$TAG
mkdir -p $OUT
run() { name=$1; shift; timeout -k 10 280 rocprofv3 --pmc "$@" -d $OUT/$name -o p --output-format csv -- python3 bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err || echo "pass $name failed"; }
run a SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_BRANCH
run b SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run c TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
run d TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
python3 tools/pmc_summary.py $OUT > $OUT/summary.txt
