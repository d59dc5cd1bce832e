"""Developer helper: build the acceleration structures of one workload (scene upload, bottom levels, top level) and time it; run it
under `rocprofv3 --kernel-trace --stats -- python3 tools/build_prof.py c3` for the per-kernel times of the builder."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.load_package()
import torch
import dxpbrt_amd.ptamd as P, dxpbrt_amd.scenes as S
import bench
for w in sys.argv[1:] or ["c3"]:
    kind, W, H, spp, bounces, desc = bench.WORKLOADS[w]
    scene, ext = bench.make_scene(kind, W / H, S)
    ctx = P.DeviceContext(0)
    g = P.Scene(ctx, scene); ctx.sync()                     # first build: allocations, code load
    t = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g.CreateAccelerationStructures(); ctx.sync()
        t.append((time.perf_counter() - t0) * 1e3)
    st = ctx.accel_stats()
    print(f"{w}: {scene.triangle_count} triangles, {len(scene.objects)} instances: CreateAccelerationStructures {min(t):.2f} ms (best of 3, host clock, "
          f"bottom levels + top level + traversal copy), bottom-level depth {st.MaxBottomLevelDepth}, top-level depth {st.TopLevelDepth}, nodes {st.NodeBytes // 80}")
    ctx.close()
