#!/bin/bash
# Experiment helper (GPU box): a few PMC passes of one frame sequence for two builds of the library, to compare where waves wait.
# usage: tools/experiments/pmc_ab.sh <workload> <libA> <libB>
R=$GRAFT_REPO_ROOT; W=${1:-c2}; shift
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  L=$R/build/ab/libptamd_$n.so
  for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY" "SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_INSTS_FLAT SQ_ACTIVE_INST_MISC"; do
    tag=$(echo $pass | cut -d' ' -f1)
    timeout -k 10 200 rocprofv3 --pmc $pass -d $R/gpurun_out/pmcab/$n/$tag -o p --output-format csv -- python3 $R/tools/ab_bench.py --child $L --workloads $W --frames 6 --inflight 1 > /dev/null 2>&1 || echo "pass $tag of $n failed"
  done
done
cd $R && python3 - "$@" <<'PY'
import csv,glob,sys,collections
for n in sys.argv[1:]:
    tot=collections.Counter()
    for f in glob.glob(f"gpurun_out/pmcab/{n}/**/p_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_round" in r["Kernel_Name"]: tot[r["Counter_Name"]]+=float(r["Counter_Value"])
    print(n, {k:int(v) for k,v in sorted(tot.items())})
PY
