"""Developer helper: find pixels where GPU and oracle differ for one case."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.load_package()
import dxpbrt_amd.layouts as L, dxpbrt_amd.ptamd as P, dxpbrt_amd.scenes as S
orc = ge.load_oracle()
W, H = 96, 96
scene = S.cornell_box(aspect=W / H, variant="ggx", glass_sphere=False, has_normals=True, jitter=(0.31, -0.17))
for spp in (16,):
    gs = S.graphics_settings(W, H, spp=spp, bounces=16, frame_index=5)
    ctx = P.DeviceContext(0)
    g = P.Scene(ctx, scene); r = P.Renderer(ctx, g, W, H, with_f32=True)
    for dbg in (0, 2):
        ctx.set_debug_flags(dbg)
        ctx.reset_counters(); r.render(gs); ctx.sync()
        out = P.textures_to_numpy(r.textures); c = ctx.counters()
        gb, rays, f32 = orc.render(scene, gs, accel_mode=0, want_f32=True, layouts=L)
        d = (out["RadianceF32"].view(np.uint32) != f32.view(np.uint32)).any(-1)
        if dbg == 2:
            import ctypes as C
            m = np.zeros(16, np.float32)
            ctx.check(ctx.lib.pt_debug_read_mismatch(ctx.handle, C.c_void_p(m.ctypes.data)))
            print("mismatches", c.BvhMismatches, "ray o", m[:4].tolist(), "d", m[4:8].tolist(), "bvh", m[8:10].view(np.uint32), m[10], "brute", m[12:14].view(np.uint32), m[14])
            print("hex", [hex(v) for v in m[:8].view(np.uint32)])
        print("debug", dbg, "spp", spp, "gpu rays", c.PrimaryRays + c.SecondaryRays, "oracle", rays, "diff px", np.argwhere(d).tolist()[:10])
        for y, x in np.argwhere(d)[:4]:
            print("  ", (x, y), out["RadianceF32"][y, x], f32[y, x])
    ctx.close()
