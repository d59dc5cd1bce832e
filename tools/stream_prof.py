"""Developer helper: where the lanes of the streaming traversal go. Needs a library built with -DPT_STREAM_PROF
(tools/ab.sh prof "-DPT_STREAM_PROF"); tallies come back through the mismatch record."""
import sys, os, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.load_package()
import dxpbrt_amd.ptamd as P, dxpbrt_amd.scenes as S
P.LIB_PATH = os.path.join(ROOT, "build", "ab", "libptamd_%s.so" % os.environ.get("PROF_LIB", "prof"))
import bench
names = ["steps", "nodeExec", "nodeLanes", "triExec", "triLanes", "enterExec", "enterLanes", "liveLanes", "exhSteps", "exhLive", "refills", "refillLanes", "outer"]
INFLIGHT = int(os.environ.get("INFLIGHT", "1"))
for w in sys.argv[1:] or ["c3", "c5"]:
    kind, W, H, spp, bounces, desc = bench.WORKLOADS[w]
    scene, ext = bench.make_scene(kind, W / H, S)
    ctx = P.DeviceContext(0); g = P.Scene(ctx, scene); r = P.Renderer(ctx, g, W, H)
    ctx.set_frames_in_flight(INFLIGHT); ctx.reset_counters()
    r.render(S.graphics_settings(W, H, spp=spp, bounces=bounces, ext_flags=ext)); ctx.sync()
    buf = np.zeros(16, np.float32); ctx.check(ctx.lib.pt_debug_read_mismatch(ctx.handle, buf.ctypes.data))
    m = buf.view(np.uint32)[:13].astype(np.float64)
    d = dict(zip(names, m)); c = ctx.counters()
    print(w, "secondary rays", c.SecondaryRays, {k: int(v) for k, v in d.items()})
    st = d["steps"]
    print("   live lanes per step %.1f | node section: run in %.0f%% of steps with %.1f lanes | tri: %.0f%% with %.1f | enter: %.0f%% with %.1f"
          % (d["liveLanes"] / st, 100 * d["nodeExec"] / st, d["nodeLanes"] / max(d["nodeExec"], 1), 100 * d["triExec"] / st, d["triLanes"] / max(d["triExec"], 1),
             100 * d["enterExec"] / st, d["enterLanes"] / max(d["enterExec"], 1)))
    print("   steps after the sub-queue ran dry: %.0f%% of all steps, %.1f live lanes in them | refills %d with %.1f rays | steps per ray %.1f"
          % (100 * d["exhSteps"] / st, d["exhLive"] / max(d["exhSteps"], 1), d["refills"], d["refillLanes"] / max(d["refills"], 1), (d["nodeLanes"] + d["triLanes"] + d["enterLanes"]) / max(c.SecondaryRays, 1)))
    ctx.close()
