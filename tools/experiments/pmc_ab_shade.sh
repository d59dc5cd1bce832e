#!/bin/bash
# Experiment helper (GPU box): cache and instruction counters of k_shade for several builds of the library on one workload.
# usage: tools/experiments/pmc_ab_shade.sh <workload> <out.json> <lib name | default> ...
R=$GRAFT_REPO_ROOT; W=${1:-c3t}; OUT=$2; shift 2
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  L=$R/build/ab/libptamd_$n.so; [ $n = default ] && L=default
  for pass in "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
    tag=$(echo $pass | cut -d' ' -f1)
    timeout -k 10 200 rocprofv3 --pmc $pass -d $R/gpurun_out/pmcab/$n/$tag -o p --output-format csv -- python3 $R/tools/ab_bench.py --child $L --workloads $W --frames 6 --inflight 1 > /dev/null 2>&1 || echo "pass $tag of $n failed"
  done
done
cd $R && python3 - "$OUT" "$W" "$@" <<'PY'
import csv,glob,sys,collections,json
out,w=sys.argv[1],sys.argv[2]
with open(out,"a") as fh:
    for n in sys.argv[3:]:
        tot=collections.defaultdict(collections.Counter)
        for f in glob.glob(f"gpurun_out/pmcab/{n}/**/p_counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k=r["Kernel_Name"].split("(")[0].replace("void pt::","")
                if k.startswith(("k_shade","k_extend_stream")): tot[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        for k,c in tot.items():
            d={"workload":w,"build":n,"kernel":k,**{a:int(b) for a,b in sorted(c.items())}}
            if c.get("TCC_HIT_sum"): d["tcc_hit_rate"]=c["TCC_HIT_sum"]/(c["TCC_HIT_sum"]+c["TCC_MISS_sum"])
            print(json.dumps(d)); fh.write(json.dumps(d)+"\n")
PY
