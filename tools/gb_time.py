"""Developer helper: time the G-buffer pass alone (HIP events around N back-to-back pt_gbuffer_render calls) for some workloads and builds
of the library.  usage: tools/gb_time.py [--workloads c3,c5] [name ...]   (names: build/ab/libptamd_<name>.so, 'default' = the product)"""
import argparse, os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def child(lib, w, n):
    sys.path.insert(0, ROOT)
    import torch
    import __graft_entry__ as ge
    ge.load_package()
    import dxpbrt_amd.ptamd as P, dxpbrt_amd.scenes as S
    if lib != "default":
        P.LIB_PATH = lib
    import bench
    kind, W, H, spp, bounces, desc = bench.WORKLOADS[w]
    scene, ext = bench.make_scene(kind, W / H, S)
    ctx = P.DeviceContext(0); g = P.Scene(ctx, scene); r = P.Renderer(ctx, g, W, H)
    tlas = g.GetTopLevelAccelerationStructure()
    for _ in range(3):
        r.gbuffer.Render(tlas, r.constants)
    ctx.sync()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        r.gbuffer.Render(tlas, r.constants)
    b.record(); torch.cuda.synchronize()
    print(json.dumps({"ms": a.elapsed_time(b) / n}))

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", default=None); ap.add_argument("--workloads", default="c3,c5"); ap.add_argument("--n", type=int, default=30)
    ap.add_argument("names", nargs="*")
    a = ap.parse_args()
    if a.child:
        child(a.child, a.workloads, a.n); sys.exit(0)
    for w in a.workloads.split(","):
        for name in a.names or ["default"]:
            lib = "default" if name == "default" else os.path.join(ROOT, "build", "ab", "libptamd_%s.so" % name)
            p = subprocess.run([sys.executable, __file__, "--child", lib, "--workloads", w, "--n", str(a.n)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            print("%-4s %-28s %s" % (w, name, ("%.3f ms per G-buffer pass" % json.loads(line[0])["ms"]) if line else "FAILED " + p.stderr[-300:]), flush=True)
