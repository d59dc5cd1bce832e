"""Developer helper: ray counts / image hashes per traversal schedule vs the oracle, + structure + device brute force."""
import sys, os, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
ge.load_package()
import dxpbrt_amd.layouts as L, dxpbrt_amd.ptamd as P, dxpbrt_amd.scenes as S
import bvh_check
orc = ge.load_oracle()

def run(scene, W, H, spp, bounces):
    gs = S.graphics_settings(W, H, spp=spp, bounces=bounces)
    gb, rays, f32 = orc.render(scene, gs, accel_mode=1, want_f32=True, layouts=L)
    print(scene.name, "oracle rays", rays)
    ctx = P.DeviceContext(0)
    g = P.Scene(ctx, scene)
    lay, buf = ctx.download_blob()
    try:
        print("  structure:", bvh_check.check_blob(lay, buf, lambda n: 2 if n <= 32 else 1))
    except AssertionError as e:
        print("  STRUCTURE PROBLEMS:\n   ", str(e).replace("\n", "\n    "))
    for flags in (0, 0x20, 4, 8, 2):
        r = P.Renderer(ctx, g, W, H, with_f32=True)
        ctx.set_debug_flags(flags); ctx.reset_counters()
        r.render(gs); ctx.sync()
        c = ctx.counters()
        out = P.textures_to_numpy(r.textures)
        bad = (out["RadianceF32"].view(np.uint32) != f32.view(np.uint32)).any(-1)
        print(f"  flags {flags:#x}: rays {c.PrimaryRays + c.SecondaryRays} (diff {c.PrimaryRays + c.SecondaryRays - rays}) pixels differing from oracle {int(bad.sum())} mismatches {c.BvhMismatches} overflow {c.StackOverflows}")
        if flags == 2 and c.BvhMismatches:
            m = np.zeros(16, np.float32); ctx.lib.pt_debug_read_mismatch(ctx.handle, m.ctypes.data); u = m.view(np.uint32)
            print("   ray", m[:8].tolist(), "bvh", u[8], u[9], m[10], "brute", u[12], u[13], m[14])
        ctx.set_debug_flags(0)
    ctx.close()

if __name__ == "__main__":
    sc = S.instanced_grid(n=24, aspect=192 / 108); sc.scene_data = S.make_scene_data((0.2, 0.3, 0.4, 1.0))
    run(sc, 192, 108, 2, 6)
