"""Developer helper: traversal statistics of one frame (node visits / triangle tests per ray, longest walk) for a given build of the library.
usage: tools/trav_stats.py [--lib build/ab/libptamd_x.so] c3 c5"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.load_package()
import dxpbrt_amd.ptamd as P, dxpbrt_amd.scenes as S
args = sys.argv[1:]
if args and args[0] == "--lib":
    P.LIB_PATH = os.path.join(ROOT, args[1]); args = args[2:]
import bench
for w in args or ["c3", "c5"]:
    kind, W, H, spp, bounces, desc = bench.WORKLOADS[w]
    scene, ext = bench.make_scene(kind, W / H, S)
    ctx = P.DeviceContext(0); g = P.Scene(ctx, scene); r = P.Renderer(ctx, g, W, H)
    ctx.set_debug_flags(1); ctx.reset_counters()
    r.render(S.graphics_settings(W, H, spp=spp, bounces=bounces, ext_flags=ext)); ctx.sync()
    c = ctx.counters(); st = ctx.accel_stats()
    rays = c.PrimaryRays + c.SecondaryRays
    print(f"{w} {os.path.basename(P.LIB_PATH)}: nodes/ray {c.NodesVisited / rays:.2f}  tris/ray {c.TrianglesTested / rays:.2f}  longest walk {c.MaxNodesPerRay} nodes  "
          f"wide nodes {st.NodeBytes // 80}  depth {st.MaxBottomLevelDepth}+{st.TopLevelDepth}")
    ctx.close()
