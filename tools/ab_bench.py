"""Developer helper: time one workload with a given build of libptamd (tools/ab.sh), one process per build.
usage: tools/ab_bench.py [--workloads c3,c5] [--frames 12] [--log gpurun_out/ab/<experiment>.jsonl] [name ...]      (no name: every build/ab/libptamd_*.so)
Every result is also appended as one JSON line to --log (default gpurun_out/ab/ab.jsonl): the A/B evidence copied into profiles/r04_ab/."""
import argparse, glob, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def child(lib, workload, frames, inflight, world=1):
    sys.path.insert(0, ROOT)
    import torch
    import __graft_entry__ as ge
    ge.load_package()
    import dxpbrt_amd.ptamd as P, dxpbrt_amd.scenes as S
    if lib:
        P.LIB_PATH = lib
    import bench
    kind, W, H, spp, bounces, desc = bench.WORKLOADS[workload]
    scene, ext = bench.make_scene(kind, W / H, S)
    lanes = []
    for _ in range(inflight):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            ctx = P.DeviceContext(0, stream=st.cuda_stream)
            if hasattr(ctx, 'set_frames_in_flight'): ctx.set_frames_in_flight(inflight)
            ctx.set_sharding(0, world, 16)                          # rank 0's share of a `world`-rank sharding (bench.py --emulate-world)
            g = P.Scene(ctx, scene)
            lanes.append((st, ctx, g, P.Renderer(ctx, g, W, H)))
    def frame(i):
        st, ctx, g, r = lanes[i % inflight]
        with torch.cuda.stream(st):
            r.render(S.graphics_settings(W, H, spp=spp, bounces=bounces, frame_index=i, ext_flags=ext))
    for i in range(3 * inflight):
        frame(i)
    torch.cuda.synchronize()
    for _, ctx, _, _ in lanes:
        ctx.reset_counters()
    t = time.perf_counter()
    for i in range(frames):
        frame(100 + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    rays = sum(c.PrimaryRays + c.SecondaryRays for c in (ctx.counters() for _, ctx, _, _ in lanes))
    print(json.dumps({"mrays": rays / dt / 1e6, "ms": dt / frames * 1e3}))

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", default=None)
    ap.add_argument("--workloads", default="c3,c5")
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--inflight", type=int, default=3)
    ap.add_argument("--world", type=int, default=1, help="render only rank 0's share of an N-rank band sharding")
    ap.add_argument("--log", default=os.path.join(ROOT, "gpurun_out", "ab", "ab.jsonl"))
    ap.add_argument("--note", default="")
    ap.add_argument("names", nargs="*")
    a = ap.parse_args()
    if a.child is not None:
        child(a.child if a.child != "default" else None, a.workloads, a.frames, a.inflight, a.world)
        sys.exit(0)
    libs = [("default", "default")] + [(os.path.basename(p)[9:-3], p) for p in sorted(glob.glob(os.path.join(ROOT, "build", "ab", "libptamd_*.so")))]
    if a.names:
        libs = [l for l in libs if l[0] in a.names]
    for w in a.workloads.split(","):
        for name, path in libs:
            p = subprocess.run([sys.executable, __file__, "--child", path, "--workloads", w, "--frames", str(a.frames), "--inflight", str(a.inflight), "--world", str(a.world)],
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if line:
                d = json.loads(line[-1]); print(f"{w} {name:28s} {d['mrays']:9.1f} Mrays/s  {d['ms']:7.3f} ms/frame", flush=True)
                os.makedirs(os.path.dirname(a.log), exist_ok=True)
                with open(a.log, "a") as fh:
                    fh.write(json.dumps({"workload": w, "build": name, "mrays_per_s": d["mrays"], "ms_per_frame": d["ms"], "frames": a.frames,
                                         "inflight": a.inflight, "world": a.world, "note": a.note, "time": time.strftime("%Y-%m-%dT%H:%M:%S")}) + "\n")
            else:
                print(w, name, "FAILED", p.stderr[-300:], flush=True)
