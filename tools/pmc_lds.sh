#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_lds
mkdir -p $OUT
timeout -k 10 280 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $OUT/l -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --inflight 1 > $OUT/l.json 2> $OUT/l.err || echo failed
python3 tools/pmc_summary.py $OUT | grep -A10 "k_extend2<false, true>"
