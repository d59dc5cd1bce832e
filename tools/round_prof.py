"""Developer helper: where a wave of the fused round kernel (k_round, flat schedule) spends its clock. Needs a library built with
-DPT_ROUND_PROF (tools/ab.sh rprof "-DPT_ROUND_PROF"); the per-wave section clocks come back through the mismatch record."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.load_package()
import dxpbrt_amd.ptamd as P, dxpbrt_amd.scenes as S
P.LIB_PATH = os.path.join(ROOT, "build", "ab", "libptamd_%s.so" % os.environ.get("PROF_LIB", "rprof"))
import bench
names = ["load", "scan", "items", "trace", "merge", "reconstruct", "material+scatter", "emit", "fresh", "nItems", "nBatchRounds", "tiles"]
for w in sys.argv[1:] or ["c2"]:
    kind, W, H, spp, bounces, desc = bench.WORKLOADS[w]
    scene, ext = bench.make_scene(kind, W / H, S)
    ctx = P.DeviceContext(0); g = P.Scene(ctx, scene); r = P.Renderer(ctx, g, W, H)
    r.render(S.graphics_settings(W, H, spp=spp, bounces=bounces, ext_flags=ext)); ctx.sync()
    ctx.reset_counters()
    r.render(S.graphics_settings(W, H, spp=spp, bounces=bounces, frame_index=1, ext_flags=ext)); ctx.sync()
    buf = np.zeros(16, np.float32); ctx.check(ctx.lib.pt_debug_read_mismatch(ctx.handle, buf.ctypes.data))
    m = buf.view(np.uint32)[:12].astype(np.float64)
    f = buf.view(np.uint32)[12:16].astype(np.float64)
    c = ctx.counters()
    tot = m[:9].sum()
    print(w, "secondary rays", c.SecondaryRays, "wave tiles", int(m[11]), "items per tile %.1f" % (m[9] / max(m[11], 1)), "item rounds per tile %.2f" % (m[10] / max(m[11], 1)))
    for k in range(9):
        print("   %-18s %5.1f %%   %8.0f wave-cycles per tile" % (names[k], 100 * m[k] / tot, 64 * m[k] / max(m[11], 1)))
    ft = f.sum()
    if ft > 0:
        print("   inside the fresh tiles (the profiling build waits for the loads before each stamp): state load %.0f %%, record load %.0f %%, ray + decode + scatter %.0f %%, reservation + stores %.0f %%"
              % (100 * f[0] / ft, 100 * f[1] / ft, 100 * f[2] / ft, 100 * f[3] / ft))
    ctx.close()
