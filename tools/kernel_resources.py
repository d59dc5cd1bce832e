#!/usr/bin/env python3
"""Developer helper: per-kernel register / spill / LDS table from hipcc -Rpass-analysis=kernel-resource-usage.
usage: tools/kernel_resources.py [file.hip ...] [--filter substr]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "directx-physically-based-raytracer_amd", "csrc")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
flt = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--filter=")]
files = args or [os.path.join(CSRC, "pt_kernels.hip"), os.path.join(CSRC, "pt_stream.hip")]
flags = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize".split()
for f in files:
    out = subprocess.run(["/opt/rocm/bin/hipcc", *flags, "-c", f, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                         stderr=subprocess.PIPE, text=True).stderr
    cur = None; rows = {}
    for line in out.splitlines():
        m = re.search(r"remark: \s*(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (.*?) \[-R", line)
        if not m: continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = subprocess.run(["c++filt", v], stdout=subprocess.PIPE, text=True).stdout.strip().split("(")[0].replace("void ", "")
            rows[cur] = {}
        elif cur: rows[cur][k.split(" [")[0]] = v
    print(f"{'kernel':58s} {'VGPR':>5s} {'SGPR':>5s} {'sSpill':>6s} {'vSpill':>6s} {'scratch':>7s} {'occ':>4s} {'LDS':>7s}")
    for k, r in rows.items():
        if flt and not any(x in k for x in flt): continue
        print(f"{k[:58]:58s} {r.get('VGPRs','?'):>5s} {r.get('TotalSGPRs','?'):>5s} {r.get('SGPRs Spill','?'):>6s} {r.get('VGPRs Spill','?'):>6s} {r.get('ScratchSize','?'):>7s} {r.get('Occupancy','?'):>4s} {r.get('LDS Size','?'):>7s}")
