#!/bin/bash
# Developer helper: PMC passes for one bench workload (separate passes; never combined with traces).
# usage: tools/pmc.sh <tag> <workload>        (--chains 1: every launch of a kernel covers all sub-queues, so "per launch" means one thing)
set -e
TAG=${1:-pmc}; WL=${2:-c2}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
run() { name=$1; shift; timeout -k 10 280 rocprofv3 --pmc "$@" -d $OUT/$name -o p --output-format csv -- python3 bench.py --workload $WL --steps 2 --warmup 1 --chains 1 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err || echo "pass $name failed"; }
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU
run sq2 SQ_INSTS_LDS SQ_INSTS_FLAT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
run lanes SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
python3 tools/pmc_summary.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
