import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.load_package()
import dxpbrt_amd.layouts as L, dxpbrt_amd.ptamd as P, dxpbrt_amd.scenes as S
orc = ge.load_oracle()
W, H = 128, 72
scene = S.dynamic_scene(aspect=W / H)
bar = scene.nodes[2].meshes[0]
ctx = P.DeviceContext(0)
g = P.Scene(ctx, scene)
r = P.Renderer(ctx, g, W, H, with_f32=True)
for step, (angle, lift) in enumerate([(25.0, 0.0), (-40.0, 0.15)]):
    pose = S.bar_pose(angle, lift)
    g.SkinSkeletalMeshes(bar, pose)
    tr = np.ascontiguousarray(pose, np.float32)
    orc.lib().or_skin_mesh(bar.skeletal_vertices.ctypes.data, tr.ctypes.data, bar.vertices.ctypes.data, bar.motion_vectors.ctypes.data, len(bar.vertices))
    g.UpdateAccelerationStructures(2)
    gs = S.graphics_settings(W, H, spp=2, bounces=4, frame_index=step)
    r.render(gs); ctx.sync()
    out = P.textures_to_numpy(r.textures)
    gb, rays, f32 = orc.render(scene, gs, accel_mode=0, want_f32=True, layouts=L)
    for k in ("Position", "FlatNormal", "GeometricNormal", "NormalRoughness"):
        d = (out[k] != gb[k]).any(-1)
        print(step, k, "diff px", int(d.sum()))
        for y, x in np.argwhere(d)[:3]:
            print("   ", (x, y), out[k][y, x], gb[k][y, x], "pos", out["Position"][y, x])
