"""Dynamic scenes: GPU skinning (Shaders/SkeletalMeshSkinning.hlsl:28-62), BLAS update + TLAS rebuild
(Source/Scene.ixx:327-345), per-vertex motion vectors in the G-buffer (Shaders/GBufferGeneration.hlsl:62-91)."""
import ctypes as C

import numpy as np
import pytest

import __graft_entry__ as ge


def oracle_skin(oracle, mesh, transforms):
    tr = np.ascontiguousarray(transforms, np.float32)
    oracle.lib().or_skin_mesh(mesh.skeletal_vertices.ctypes.data, tr.ctypes.data, mesh.vertices.ctypes.data,
                              mesh.motion_vectors.ctypes.data, len(mesh.vertices))


def test_oracle_skinning_known_answers(oracle, pkg):
    S = pkg.scenes
    bar = S.skinned_bar()
    rest = bar.vertices.copy()
    oracle_skin(oracle, bar, S.bar_pose(0.0))                   # identity pose: nothing moves, motion = 0
    assert np.array_equal(bar.vertices["Position"], rest["Position"]) and not bar.motion_vectors[:, :3].any()
    assert np.abs(bar.vertices["Normal"].astype(int) - rest["Normal"].astype(int)).max() <= 1     # truncating re-pack
    oracle_skin(oracle, bar, S.bar_pose(90.0))
    top = rest["Position"][:, 1] == 1.0
    assert top.any()
    # fully-weighted top vertices are rotated by 90 degrees about Z: (x, 1, z) -> (-1, x, z)
    assert np.allclose(bar.vertices["Position"][top][:, 0], -1.0, atol=1e-6)
    assert np.allclose(bar.vertices["Position"][top][:, 1], rest["Position"][top][:, 0], atol=1e-6)
    bottom = rest["Position"][:, 1] == 0.0
    assert np.array_equal(bar.vertices["Position"][bottom], rest["Position"][bottom])
    mv = bar.motion_vectors.view(np.float16)[:, :3].astype(np.float32)
    assert np.allclose(mv[top], (rest["Position"] - bar.vertices["Position"])[top], atol=2e-3)   # old - new, stored as half


@pytest.mark.gpu
def test_gpu_skinning_update_and_motion_vectors(gpu, ptamd, oracle, pkg):
    S, L = pkg.scenes, pkg.layouts
    W, H = 128, 72
    scene = S.dynamic_scene(aspect=W / H)
    bar = scene.nodes[2].meshes[0]
    gpu.set_sharding(0, 1, 16)
    g = ptamd.Scene(gpu, scene)
    hv = next(h for m, h, _ in scene.geometry if m is bar)
    hm = int(scene._motion_heap[id(bar)])
    r = ptamd.Renderer(gpu, g, W, H, with_f32=True)
    prev = scene.instance_data["ObjectToWorld"].copy()
    for step, (angle, lift) in enumerate([(25.0, 0.0), (-40.0, 0.15)]):
        pose = S.bar_pose(angle, lift)
        g.SkinSkeletalMeshes(bar, pose)                          # device
        oracle_skin(oracle, bar, pose)                           # host copy, same bytes expected
        gpu.sync()
        assert np.array_equal(g.download(hv, np.uint8), bar.vertices.view(np.uint8).reshape(-1))
        got_mv = g.download(hm, np.uint16).reshape(-1, 4)
        assert np.array_equal(got_mv[:, :3], bar.motion_vectors[:, :3])
        g.UpdateAccelerationStructures(2)                        # BLAS update + TLAS rebuild
        gs = S.graphics_settings(W, H, spp=2, bounces=4, frame_index=step)
        for t in r.textures.values():
            t.zero_()           # miss pixels keep whatever the textures held (as in the reference); the oracle starts from zeros
        gpu.reset_counters(); r.render(gs); gpu.sync()
        out = ptamd.textures_to_numpy(r.textures); c = gpu.counters()
        ref_gb, ref_rays, ref_f32 = oracle.render(scene, gs, accel_mode=0, want_f32=True, layouts=L)
        for k in ("Position", "FlatNormal", "GeometricNormal", "NormalRoughness", "MotionVector", "LinearDepth"):
            a, b = out[k], ref_gb[k]
            if a.dtype.kind == "f":
                a, b = a.view(np.uint32), b.view(np.uint32)
            assert np.array_equal(a, b), (step, k)
        assert c.PrimaryRays + c.SecondaryRays == ref_rays
        assert np.array_equal(out["RadianceF32"].view(np.uint32), ref_f32.view(np.uint32))
        if step == 1:                                            # the bar moved between the frames: its pixels carry motion
            mv = out["MotionVector"].view(np.float16)[..., :2].astype(np.float32)
            assert np.abs(mv).max() > 1.0
    assert np.array_equal(prev, scene.instance_data["ObjectToWorld"])


@pytest.mark.gpu
def test_update_keeps_the_opaque_flag_of_alpha_tested_geometry(gpu, ptamd, oracle, pkg):
    """the PERFORM_UPDATE path recomputes D3D12_RAYTRACING_GEOMETRY_FLAG_OPAQUE from AlphaMode like the build does
    (Scene.ixx:320-324). A blend-mode skinned column must still let rays through after an update: compared with the oracle."""
    S, L = pkg.scenes, pkg.layouts
    W, H = 96, 54
    scene = S.dynamic_scene(aspect=W / H)
    bar = scene.nodes[2].meshes[0]
    bar.material["AlphaMode"] = 2                                # Blend
    bar.material["BaseColor"] = (0.8, 0.5, 0.2, 0.25)            # alpha below 0.5: IsOpaque rejects every candidate
    scene.finalize()
    gpu.set_sharding(0, 1, 16)
    g = ptamd.Scene(gpu, scene)
    r = ptamd.Renderer(gpu, g, W, H, with_f32=True)
    pose = S.bar_pose(30.0, 0.1)
    g.SkinSkeletalMeshes(bar, pose); oracle_skin(oracle, bar, pose); gpu.sync()
    g.UpdateAccelerationStructures(2)
    gs = S.graphics_settings(W, H, spp=2, bounces=4, frame_index=1)
    for t in r.textures.values():
        t.zero_()
    gpu.reset_counters(); r.render(gs); gpu.sync()
    out = ptamd.textures_to_numpy(r.textures); c = gpu.counters()
    ref_gb, ref_rays, ref_f32 = oracle.render(scene, gs, accel_mode=0, want_f32=True, layouts=L)
    assert np.array_equal(out["Position"].view(np.uint32), ref_gb["Position"].view(np.uint32))
    assert c.PrimaryRays + c.SecondaryRays == ref_rays
    assert np.array_equal(out["RadianceF32"].view(np.uint32), ref_f32.view(np.uint32))
    # the column is invisible to primary rays: no pixel's position lies on it (x within the column's 0.24 footprint around its instance)
    opaque = S.dynamic_scene(aspect=W / H); opaque.finalize()
    ref_opaque, _, _ = oracle.render(opaque, gs, accel_mode=0, layouts=L)
    assert not np.array_equal(ref_opaque["Position"], ref_gb["Position"])


@pytest.mark.gpu
def test_dynamic_frames_enqueued_without_synchronisation(gpu, ptamd, oracle, pkg):
    """skin + update + render for several frames back to back with NO host synchronisation in between (what
    bench.py --workload dynamic does). Every frame's skinned vertices (a stream-ordered copy taken right after the skin kernel)
    must be the oracle's for THAT frame's pose: the pinned staging buffer of the joint matrices is a ring guarded by events, so
    a host that runs ahead cannot overwrite a pose the H2D copy has yet to read."""
    import torch
    S = pkg.scenes
    W, H = 64, 36
    scene = S.dynamic_scene(aspect=W / H)
    bar = scene.nodes[2].meshes[0]
    gpu.set_sharding(0, 1, 16)
    g = ptamd.Scene(gpu, scene)
    hv = next(h for m, h, _ in scene.geometry if m is bar)
    r = ptamd.Renderer(gpu, g, W, H)
    gpu.sync()
    poses = [S.bar_pose(7.0 * k - 40.0, 0.01 * k) for k in range(12)]       # more frames than the ring has slots
    snapshots = []
    for k, pose in enumerate(poses):
        g.SkinSkeletalMeshes(bar, pose)
        snapshots.append(g._heap_dev[hv].clone())                          # stream-ordered behind the skin kernel
        g.UpdateAccelerationStructures(2)
        r.render(S.graphics_settings(W, H, spp=1, bounces=2, frame_index=k))
    gpu.sync()
    # a refit used to cost a dynamic scene its normal records for good (drop_tlas put them aside, and the rebuild of an unchanged
    # binding never brought them back)
    assert gpu.accel_stats().NormalRecords == 1
    for k, pose in enumerate(poses):                                         # the oracle skins the same sequence (skinning reads the previous position)
        oracle_skin(oracle, bar, pose)
        assert np.array_equal(snapshots[k].cpu().numpy(), bar.vertices.view(np.uint8).reshape(-1)), f"frame {k} was skinned with another frame's pose"
