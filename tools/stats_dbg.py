"""Developer helper: traversal statistics of a bench workload (nodes / triangles per ray, longest walk)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.load_package()
import dxpbrt_amd.ptamd as P, dxpbrt_amd.scenes as S
import bench
for w in sys.argv[1:] or ["c3", "c5"]:
    kind, W, H, spp, bounces, desc = bench.WORKLOADS[w]
    scene, ext = bench.make_scene(kind, W / H, S)
    ctx = P.DeviceContext(0); g = P.Scene(ctx, scene); r = P.Renderer(ctx, g, W, H)
    ctx.set_debug_flags(1); ctx.reset_counters()
    r.render(S.graphics_settings(W, H, spp=spp, bounces=bounces, ext_flags=ext)); ctx.sync()
    c = ctx.counters(); a = ctx.accel_stats(); n = c.PrimaryRays + c.SecondaryRays
    print(w, "rays", n, "nodes/ray %.2f tris/ray %.2f max nodes of a ray %d" % (c.NodesVisited / n, c.TrianglesTested / n, c.MaxNodesPerRay),
          "| node bytes", a.NodeBytes, "tri bytes", a.TriangleBytes, "blas depth", a.MaxBottomLevelDepth, "tlas depth", a.TopLevelDepth, flush=True)
    ctx.close()
