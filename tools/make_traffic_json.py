#!/usr/bin/env python3
"""gpurun_out/final/* -> profiles/r04_* and profiles/traffic.json (HBM bytes per launch and VALU instructions per frame from the PMC
summaries, stamped with the hash of the kernel sources they were measured on: bench.py only quotes them for the same sources)."""
import json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SRC = os.path.join(ROOT, "gpurun_out", "final"); DST = os.path.join(ROOT, "profiles"); ROUND = "r04_"
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec)
sys.argv = ["bench.py"]; spec.loader.exec_module(bench)

def parse(path):
    out, cur = {}, None
    for line in open(path):
        if not line.startswith(" "):
            cur = line.strip(); out[cur] = {}
        else:
            p = line.split()
            out[cur][p[0]] = (float(p[1]), int(p[3]))
    return out

KERNEL_KEY = {"k_round": "k_round", "k_extend_stream": "k_extend", "k_extend2": "k_extend", "k_shade": "k_shade"}
traffic = {"_doc": "HBM bytes per launch from rocprofv3 PMC passes: FETCH_SIZE x 2 (gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md) + WRITE_SIZE (KB), "
                   "divided by the dispatch count, for the variant of each kernel the workload runs; valu: SQ_INSTS_VALU of all kernels of one frame. "
                   "lanes_per_instruction: SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU), each divided by its number of passes, per kernel. Sources: profiles/r04_<workload>_pmc_summary.txt. source_hash = bench.source_hash() of the kernel sources measured."}
for f in os.listdir(SRC):                    # everything that is not a per-workload file (extras of tools/collect_profiles.sh)
    if os.path.isfile(os.path.join(SRC, f)) and not f.endswith(".err") and not f.startswith(("c2_b", "c2_k", "c2_p", "c2_i", "c3_", "c3t_", "c5_")):
        shutil.copy(os.path.join(SRC, f), os.path.join(DST, ROUND + f))
for w in [w for w in ("c2", "c3", "c3t", "c5") if os.path.exists(os.path.join(SRC, w + "_pmc_summary.txt"))]:
    for f in os.listdir(SRC):
        if f.startswith(w + "_") and os.path.isfile(os.path.join(SRC, f)) and not f.endswith(".err"):
            shutil.copy(os.path.join(SRC, f), os.path.join(DST, ROUND + f))
    pm = parse(os.path.join(SRC, w + "_pmc_summary.txt"))
    ent = {"source_hash": bench.source_hash()}
    # the kernels of the product path of this workload (the statistics frame of bench.py runs other variants: not counted)
    product = ("k_round<false", "k_gbuffer<false", "k_pt_init", "k_pt_first", "k_set_constants", "k_capture_normals") if w == "c2" else \
              (("k_shade<true" if w == "c3t" else "k_shade<false"), "k_extend_stream<false", "k_gbuffer<false", "k_pt_init", "k_pt_first", "k_set_constants", "k_capture_normals")
    frames = max(c["SQ_INSTS_VALU"][1] for k, c in pm.items() if "k_gbuffer<false" in k and "SQ_INSTS_VALU" in c)   # one G-buffer launch per frame
    valu_total = 0.0
    lanes = {}
    for k, c in pm.items():
        name = k.replace("pt::", "")
        if "SQ_INSTS_VALU" not in c or not name.startswith(product): continue
        valu_total += c["SQ_INSTS_VALU"][0]
        base = name.split("<")[0]
        if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c and c["SQ_ACTIVE_INST_VALU"][0] > 0:
            # a counter that was collected in two passes shows twice the dispatches of one that was collected in one: normalise per dispatch
            tc = c["SQ_THREAD_CYCLES_VALU"][0] / c["SQ_THREAD_CYCLES_VALU"][1]; ai = c["SQ_ACTIVE_INST_VALU"][0] / c["SQ_ACTIVE_INST_VALU"][1]
            lanes[KERNEL_KEY.get(base, base)] = tc / (64.0 * ai)
        if base in KERNEL_KEY and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            ent[KERNEL_KEY[base]] = (2.0 * c["FETCH_SIZE"][0] + c["WRITE_SIZE"][0]) * 1024.0 / c["FETCH_SIZE"][1]
    ent["lanes_per_instruction"] = lanes
    ent["valu"] = {"wave_instructions_per_frame": valu_total / frames, "frames_in_profile": frames,
                   "note": "k_gbuffer dispatches = frames; for c3 / c5 one of them is bench.py's statistics frame, whose k_shade launches are included (one frame in %d)" % frames}
    traffic[w] = ent
json.dump(traffic, open(os.path.join(DST, "traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
