"""Developer helper (not part of the product): GPU vs oracle comparison with details."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.load_package()
import dxpbrt_amd.layouts as L, dxpbrt_amd.ptamd as P, dxpbrt_amd.scenes as S
orc = ge.load_oracle()

def run(scene, W, H, spp, bounces, accel=1, ext=0):
    gs = S.graphics_settings(W, H, spp=spp, bounces=bounces, ext_flags=ext)
    ctx = P.DeviceContext(0)
    g = P.Scene(ctx, scene)
    r = P.Renderer(ctx, g, W, H, with_f32=True)
    ctx.reset_counters()
    t = time.time(); r.render(gs); ctx.sync(); dt = time.time() - t
    out = P.textures_to_numpy(r.textures)
    c = ctx.counters()
    t = time.time()
    gb, rays, f32 = orc.render(scene, gs, accel_mode=accel, want_f32=True, layouts=L)
    dtc = time.time() - t
    print(f"{scene.name} {W}x{H} spp{spp} b{bounces}: gpu rays {c.PrimaryRays}+{c.SecondaryRays} in {dt*1e3:.1f} ms; oracle rays {rays} in {dtc:.2f}s")
    for k in ("Position", "FlatNormal", "GeometricNormal", "BaseColorMetalness", "NormalRoughness", "IOR", "LinearDepth", "NormalizedDepth", "MotionVector"):
        a, b = out[k], gb[k]
        eq = np.array_equal(a, b) if a.dtype.kind != 'f' else np.array_equal(a.view(np.uint32), b.view(np.uint32))
        nd = int((a != b).any(-1).sum()) if not eq else 0
        print(f"   {k:20s} identical={eq} differing px={nd}")
    st = ge.compare_radiance(out["RadianceF32"], f32)
    print("   radiance", st, "fp16 identical:", np.array_equal(out["Radiance"], gb["Radiance"]))
    ctx.close()
    return st

if __name__ == "__main__":
    run(S.cornell_box(aspect=16/9, variant="ggx", glass_sphere=True), 256, 144, 4, 8)
    run(S.cornell_box(aspect=16/9, variant="diffuse", has_normals=False), 256, 144, 1, 2, ext=1)
    run(S.instanced_grid(n=20), 256, 144, 2, 4)
    run(S.sponza_scale(n_side=60), 256, 144, 2, 4)
