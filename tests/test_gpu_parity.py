"""GPU parity tests (-m gpu): the HIP path, called through the C-ABI, against the CPU oracle on the same seeded
inputs, against the committed golden fixtures, and -- at BASELINE.json's full sizes -- through size-independent
properties (determinism, sharding invariance, ray-count bounds, monotonicity).

Tolerance (BASELINE.json north_star): per-pixel L2 radiance error < 1e-3 vs the oracle. Scenes whose whole
arithmetic is bit-pinned (Cornell box: no libm call on the path) are additionally required to be BIT-IDENTICAL;
scenes with the procedural sky go through powf (Color::FromSrgb) and are held to the 1e-3 / 1e-5 tolerances.
"""
import importlib.util
import os

import numpy as np
import pytest

import __graft_entry__ as ge

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
make_golden = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(make_golden)

L2_TOLERANCE = 1e-3          # per-pixel L2 radiance (rms over the image), north_star
SKY_ABS_TOLERANCE = 2e-5     # max per-pixel deviation allowed where powf (procedural sky) is involved


def gpu_render(ptamd, ctx, scene, gs, W, H, sharding=(0, 1, 16), stats=False):
    ctx.set_sharding(*sharding)
    g = ptamd.Scene(ctx, scene)
    r = ptamd.Renderer(ctx, g, W, H, with_f32=True)
    ctx.set_debug_flags(1 if stats else 0)
    ctx.reset_counters()
    r.render(gs)
    ctx.sync()
    c = ctx.counters()
    out = ptamd.textures_to_numpy(r.textures)
    ctx.set_debug_flags(0)
    ctx.set_sharding(0, 1, 16)
    return out, c


GB_KEYS = ("Position", "FlatNormal", "GeometricNormal", "LinearDepth", "NormalizedDepth", "MotionVector",
           "BaseColorMetalness", "NormalRoughness", "IOR")


def assert_gbuffer_identical(out, ref):
    for k in GB_KEYS:
        a, b = out[k], ref[k]
        if a.dtype.kind == "f":
            a, b = a.view(np.uint32), b.view(np.uint32)
        assert np.array_equal(a, b), f"G-buffer texture {k} differs from the oracle"


@pytest.mark.parametrize("name", sorted(make_golden.CASES))
def test_golden_fixtures_bit_exact(name, gpu, ptamd):
    """HIP path vs the committed golden vectors (no oracle code runs in this test)."""
    make, W, H, spp, bounces, ext = make_golden.CASES[name]
    ge.load_package()
    import dxpbrt_amd.scenes as S
    scene = make()
    gs = S.graphics_settings(W, H, spp=spp, bounces=bounces, frame_index=7, ext_flags=ext)
    out, c = gpu_render(ptamd, gpu, scene, gs, W, H)
    g = np.load(os.path.join(GOLD, name + ".npz"))
    assert c.PrimaryRays + c.SecondaryRays == int(g["rays"])
    assert np.array_equal(out["Position"].view(np.uint32), g["position"].view(np.uint32))
    assert np.array_equal(out["FlatNormal"], g["flat_normal"])
    assert np.array_equal(out["NormalRoughness"], g["normal_roughness"])
    assert np.array_equal(out["BaseColorMetalness"], g["base_color_metalness"])
    assert np.array_equal(out["RadianceF32"].view(np.uint32), g["radiance_f32"].view(np.uint32))
    assert np.array_equal(out["Radiance"], g["radiance_f16"])


CORNELL_CASES = [
    # variant, glass, normals, W, H, spp, bounces, rr, ext, jitter, frame
    ("ggx", True, True, 160, 90, 4, 8, True, 0, (0.0, 0.0), 0),
    ("ggx", False, True, 96, 96, 16, 16, True, 0, (0.31, -0.17), 5),        # C4-shaped: 16 spp x 16 bounces
    ("diffuse", False, False, 256, 256, 1, 2, True, 1, (0.0, 0.0), 0),      # C1: Lambertian-only switch, no vertex normals
    ("ggx", True, True, 64, 48, 2, 100, False, 0, (0.0, 0.0), 1),           # max Bounces (MyAppData.h:183), RR off
    ("diffuse", True, True, 33, 17, 3, 5, True, 0, (-0.5, 0.49), 2),        # ragged size (not a multiple of 16 / 64)
]


@pytest.mark.parametrize("case", CORNELL_CASES, ids=lambda c: f"{c[0]}_{c[3]}x{c[4]}_s{c[5]}_b{c[6]}")
def test_cornell_bit_identical_to_oracle(case, gpu, ptamd, oracle, pkg):
    variant, glass, normals, W, H, spp, bounces, rr, ext, jitter, frame = case
    S, L = pkg.scenes, pkg.layouts
    scene = S.cornell_box(aspect=W / H, variant=variant, glass_sphere=glass, has_normals=normals, jitter=jitter)
    gs = S.graphics_settings(W, H, spp=spp, bounces=bounces, frame_index=frame, russian_roulette=rr, ext_flags=ext)
    out, c = gpu_render(ptamd, gpu, scene, gs, W, H)
    ref_gb, ref_rays, ref_f32 = oracle.render(scene, gs, accel_mode=0, want_f32=True, layouts=L)   # brute force: no BVH at all
    assert_gbuffer_identical(out, ref_gb)
    assert c.PrimaryRays == W * H and c.PrimaryRays + c.SecondaryRays == ref_rays
    st = ge.compare_radiance(out["RadianceF32"], ref_f32)
    assert st["rms"] < L2_TOLERANCE
    assert np.array_equal(out["RadianceF32"].view(np.uint32), ref_f32.view(np.uint32)), st
    assert np.array_equal(out["Radiance"], ref_gb["Radiance"])             # the fp16 texture the reference consumer reads


def test_textured_materials_alpha_test_and_environment_maps(gpu, ptamd, oracle, pkg):
    """Rows a4/a5/a9/a10: UV + tangent interpolation, all seven texture slots, normal mapping, alpha-tested
    (non-opaque) geometry inside traversal, cube and lat-long environment textures.
    Constant environment and cube map: no libm on the path -> bit-identical. Lat-long goes through atan2f/acosf
    (Math::ToLatLongCoordinate) -> tolerance."""
    S, L = pkg.scenes, pkg.layouts
    W, H = 128, 72
    for env in (None, "cube", "latlong"):
        scene = S.cornell_box_textured(aspect=W / H, env=env)
        gs = S.graphics_settings(W, H, spp=3, bounces=6, frame_index=2)
        out, c = gpu_render(ptamd, gpu, scene, gs, W, H)
        ref_gb, ref_rays, ref_f32 = oracle.render(scene, gs, accel_mode=0, want_f32=True, layouts=L)
        assert_gbuffer_identical(out, ref_gb)
        assert c.PrimaryRays + c.SecondaryRays == ref_rays
        st = ge.compare_radiance(out["RadianceF32"], ref_f32)
        assert st["rms"] < L2_TOLERANCE
        if env != "latlong":
            assert np.array_equal(out["RadianceF32"].view(np.uint32), ref_f32.view(np.uint32)), st
            assert np.array_equal(out["Radiance"], ref_gb["Radiance"])
        else:
            assert st["max"] < 1e-3, st
    # the alpha-masked lattice really is see-through where alpha < cutoff: primary rays aimed at it differ from an opaque copy
    scene = S.cornell_box_textured(aspect=W / H, env=None)
    scene.camera = S.make_camera((0, -0.6, 0.1), forward=(0, 1, 0.001), up=(0, 0, 1), hfov_deg=60.0, aspect=W / H)   # looking up through the lattice
    gs = S.graphics_settings(W, H, spp=1, bounces=1)
    out, _ = gpu_render(ptamd, gpu, scene, gs, W, H)
    ref_gb, _, _ = oracle.render(scene, gs, accel_mode=0, layouts=L)
    assert_gbuffer_identical(out, ref_gb)
    ys = out["Position"][..., 1]
    assert (np.abs(ys - 0.6) < 1e-3).any() and (ys > 0.9).any()       # some pixels stop at the lattice (y=0.6), others pass to the ceiling / light


def test_denoiser_facing_outputs(gpu, ptamd, oracle, pkg):
    """Raytracing.hlsl:235-239,387-413: which lobe the first bounce of sample 0 took and how far it went, packed for
    DLSS-RR (SpecularHitDistance) or NRD (Diffuse / Specular = indirect radiance + hit distance). The denoisers are out of
    scope; the hand-off textures are not."""
    S, L = pkg.scenes, pkg.layouts
    W, H = 128, 72
    scene = S.cornell_box(aspect=W / H, variant="ggx", glass_sphere=True)
    for denoiser in (L.DENOISER_DLSS_RR, L.DENOISER_NRD_REBLUR, L.DENOISER_NRD_RELAX):
        gs = S.graphics_settings(W, H, spp=2, bounces=5, frame_index=4)
        gs["Denoiser"] = denoiser
        gpu.set_sharding(0, 1, 16)
        g = ptamd.Scene(gpu, scene)
        r = ptamd.Renderer(gpu, g, W, H, with_f32=True, with_denoiser_outputs=True)
        gpu.reset_counters(); r.render(gs); gpu.sync()
        out = ptamd.textures_to_numpy(r.textures); c = gpu.counters()
        ref_gb, ref_rays, ref_f32 = oracle.render(scene, gs, gbuffer_flags=0xFFFFFFFF, accel_mode=0, want_f32=True, layouts=L)   # App.cpp:1223
        assert c.PrimaryRays + c.SecondaryRays == ref_rays
        for k in ("Radiance", "Diffuse", "Specular", "SpecularHitDistance", "DiffuseAlbedo", "SpecularAlbedo"):
            assert np.array_equal(out[k], ref_gb[k]), (denoiser, k)
        # albedo demodulation factors (GBufferGeneration.hlsl:171-186): in [0.01, 1], zero where the primary ray missed
        da = out["DiffuseAlbedo"].view(np.float16).astype(np.float32)[..., :3]; sa = out["SpecularAlbedo"].view(np.float16).astype(np.float32)[..., :3]
        hit = np.isfinite(out["LinearDepth"][..., 0])
        assert hit.any() and (da[hit] >= 0.0099).all() and (da[hit] <= 1.0).all() and (sa[hit] >= 0.0099).all() and (sa[hit] <= 1.0).all()
        if denoiser == L.DENOISER_DLSS_RR:
            d = out["SpecularHitDistance"].view(np.float16).astype(np.float32)
            assert (d > 0).any() and np.isfinite(d).all()
        else:
            dif = out["Diffuse"].view(np.float16).astype(np.float32); spc = out["Specular"].view(np.float16).astype(np.float32)
            assert (dif[..., :3].sum(-1) > 0).any() and (spc[..., :3].sum(-1) > 0).any()
            assert not ((dif[..., :3].sum(-1) > 0) & (spc[..., :3].sum(-1) > 0)).any()        # a pixel is either diffuse or specular
            # NRD modes leave Radiance as the G-buffer wrote it (emission / environment)
            assert np.array_equal(out["Radiance"], ref_gb["Radiance"])


def test_bounces_zero_and_misses(gpu, ptamd, oracle, pkg):
    S, L = pkg.scenes, pkg.layouts
    W, H = 128, 32
    scene = S.cornell_box(aspect=W / H)
    scene.camera = S.make_camera((0, 0, -4.0), hfov_deg=90.0, aspect=W / H)   # rays beside the box miss
    for bounces in (0, 3):
        gs = S.graphics_settings(W, H, spp=2, bounces=bounces)
        out, c = gpu_render(ptamd, gpu, scene, gs, W, H)
        ref_gb, ref_rays, _ = oracle.render(scene, gs, accel_mode=0, layouts=L)
        assert_gbuffer_identical(out, ref_gb)
        assert np.array_equal(out["Radiance"], ref_gb["Radiance"])
        assert c.PrimaryRays + c.SecondaryRays == ref_rays
        miss = ~np.isfinite(out["Position"][..., 3])
        assert miss.any() and np.all(np.isinf(out["Position"][miss]))


def test_sky_instanced_and_large_mesh_within_tolerance(gpu, ptamd, oracle, pkg):
    """Two-level BVH with many instances (C5-shaped) and a multi-geometry BLAS (C3-shaped), procedural sky."""
    S, L = pkg.scenes, pkg.layouts
    W, H = 192, 108
    for scene in (S.instanced_grid(n=24, aspect=W / H), S.sponza_scale(n_side=64, aspect=W / H)):
        gs = S.graphics_settings(W, H, spp=2, bounces=6)
        out, c = gpu_render(ptamd, gpu, scene, gs, W, H)
        ref_gb, ref_rays, ref_f32 = oracle.render(scene, gs, accel_mode=1, want_f32=True, layouts=L)
        assert_gbuffer_identical(out, ref_gb)                   # hits, positions, normals: no libm involved
        assert c.PrimaryRays + c.SecondaryRays == ref_rays      # identical path structure
        st = ge.compare_radiance(out["RadianceF32"], ref_f32)
        assert st["rms"] < L2_TOLERANCE and st["max"] < SKY_ABS_TOLERANCE, st


def test_instances_of_one_bottom_level_with_different_vertex_data(gpu, ptamd, oracle, pkg):
    """The frame's normal records (csrc/pt_shade.hpp ShadeTables) describe a bottom level by the objects of its FIRST instance. ObjectData is
    per (instance, geometry) in the reference (RaytracingHelpers.hlsli:79-85), so another instance of the same bottom level may name another
    vertex buffer -- here: the same positions, other normals. The library has to notice (k_check_shared_geometry) and fetch the vertices of every
    hit through that hit's own object, as the oracle does; with the second buffer dropped again the records come back. Bit-identical both times."""
    S, L = pkg.scenes, pkg.layouts
    W, H = 160, 90
    scene = S.instanced_grid(n=6, aspect=W / H)
    mesh = scene.nodes[0].meshes[0]
    other = mesh.vertices.copy()
    other["Normal"] = other["Normal"][::-1].copy()              # other normals for the same positions
    scene.heap.append(S.HeapItem(other, 0))
    gs = S.graphics_settings(W, H, spp=2, bounces=4, ext_flags=0)
    for swapped in (True, False):
        for k in (3, 17, 30):                                    # three of the 36 instances of node 0 (one object each)
            scene.object_data[int(scene.instance_ids[k])]["MeshDescriptors"]["Vertices"] = len(scene.heap) - 1 if swapped else scene.geometry[0][1]
        out, c = gpu_render(ptamd, gpu, scene, gs, W, H)
        ref_gb, ref_rays, ref_f32 = oracle.render(scene, gs, accel_mode=1, want_f32=True, layouts=L)
        assert_gbuffer_identical(out, ref_gb)
        assert c.PrimaryRays + c.SecondaryRays == ref_rays
        st = ge.compare_radiance(out["RadianceF32"], ref_f32)
        assert st["rms"] < L2_TOLERANCE and st["max"] < SKY_ABS_TOLERANCE, (swapped, st)
        if swapped:
            first = out["NormalRoughness"].copy()
        else:
            assert not np.array_equal(first, out["NormalRoughness"])      # the other normals were really used


def test_object_data_rewritten_in_place_is_resolved_again(gpu, ptamd, oracle, pkg):
    """hit reconstruction reads a resolved-geometry table (buffer pointers, stride, offsets) made when the object data is
    validated. A caller that rewrites MeshDescriptors of the BOUND device array in place announces it with pt_invalidate_object_data; binding
    the same pointer again alone is free and resolves nothing (a host that fills its slots before every Render must not pay a wait per frame).
    After the announcement the next frame must use the new vertex buffers -- and drop / regain the frame's normal records as the instances
    of a bottom level stop / start agreeing on their vertex data (PtAccelStats.NormalRecords)."""
    S, L = pkg.scenes, pkg.layouts
    W, H = 160, 90
    scene = S.instanced_grid(n=6, aspect=W / H)
    mesh = scene.nodes[0].meshes[0]
    other = mesh.vertices.copy()
    other["Normal"] = other["Normal"][::-1].copy()
    scene.heap.append(S.HeapItem(other, 0))
    gs = S.graphics_settings(W, H, spp=2, bounces=4, ext_flags=0)
    gpu.set_sharding(0, 1, 16)
    g = ptamd.Scene(gpu, scene)
    r = ptamd.Renderer(gpu, g, W, H, with_f32=True)

    def frame():
        gpu.reset_counters(); r.render(gs); gpu.sync()
        return ptamd.textures_to_numpy(r.textures), gpu.counters()

    def check(out, c):
        ref_gb, ref_rays, ref_f32 = oracle.render(scene, gs, accel_mode=1, want_f32=True, layouts=L)
        assert_gbuffer_identical(out, ref_gb)
        assert c.PrimaryRays + c.SecondaryRays == ref_rays
        st = ge.compare_radiance(out["RadianceF32"], ref_f32)
        assert st["rms"] < L2_TOLERANCE and st["max"] < SKY_ABS_TOLERANCE, st

    out0, c0 = frame(); check(out0, c0)
    assert gpu.accel_stats().NormalRecords == 1
    for k in (3, 17, 30):
        scene.object_data[int(scene.instance_ids[k])]["MeshDescriptors"]["Vertices"] = len(scene.heap) - 1
    g.object_data.copy_(ptamd.to_device(scene.object_data, g.device))          # in place: same device pointer, same count
    gpu.check(gpu.lib.pt_set_object_data(gpu.handle, g.object_data.data_ptr(), len(scene.object_data)))
    stale, _ = frame()                                                          # rebinding alone resolves nothing: still the old buffers
    assert np.array_equal(stale["NormalRoughness"], out0["NormalRoughness"])
    gpu.invalidate_object_data()
    out1, c1 = frame(); check(out1, c1)
    assert not np.array_equal(out1["NormalRoughness"], out0["NormalRoughness"])
    assert gpu.accel_stats().NormalRecords == 0                                 # three instances name other vertex data: no shared records
    for k in (3, 17, 30):
        scene.object_data[int(scene.instance_ids[k])]["MeshDescriptors"]["Vertices"] = scene.geometry[0][1]
    g.object_data.copy_(ptamd.to_device(scene.object_data, g.device))
    gpu.invalidate_object_data()
    out2, c2 = frame(); check(out2, c2)
    assert gpu.accel_stats().NormalRecords == 1
    assert np.array_equal(out2["RadianceF32"], out0["RadianceF32"])
    g.close()


def test_lbvh_agrees_with_brute_force_on_incoherent_rays(gpu, ptamd, oracle, pkg):
    """LBVH traversal (conservative boxes, tie-break) vs the oracle's brute-force loop over every triangle."""
    S, L = pkg.scenes, pkg.layouts
    W, H = 96, 54
    scene = S.sponza_scale(n_side=40, aspect=W / H)
    scene.scene_data = S.make_scene_data((0.2, 0.3, 0.4, 1.0))   # constant environment: bit-pinned arithmetic
    gs = S.graphics_settings(W, H, spp=2, bounces=8)
    out, c = gpu_render(ptamd, gpu, scene, gs, W, H, stats=True)
    ref_gb, ref_rays, ref_f32 = oracle.render(scene, gs, accel_mode=0, want_f32=True, layouts=L)
    assert_gbuffer_identical(out, ref_gb)
    assert c.PrimaryRays + c.SecondaryRays == ref_rays
    assert np.array_equal(out["RadianceF32"].view(np.uint32), ref_f32.view(np.uint32))
    assert c.NodesVisited > 0 and c.TrianglesTested > 0
    # the hierarchy actually culls: far fewer triangle tests than brute force
    assert c.TrianglesTested < 0.05 * (c.PrimaryRays + c.SecondaryRays) * scene.triangle_count


def test_lbvh_vs_device_brute_force_no_mismatch(gpu, ptamd, pkg):
    """PT_DEBUG_BRUTE_FORCE runs both traversals for every bounce ray on the device and counts disagreements.
    Regression: a reflected ray with an exactly-zero direction component (axis-aligned walls produce them) used to
    be culled by a NaN slab in the box test."""
    S = pkg.scenes
    cases = [(S.cornell_box(aspect=1.0, variant="ggx", jitter=(0.31, -0.17)), 96, 96, 16, 16, 5),
             (S.instanced_grid(n=16, aspect=1.5), 96, 64, 2, 6, 0),
             (S.sponza_scale(n_side=48, aspect=1.5), 96, 64, 2, 6, 0)]
    for scene, W, H, spp, bounces, frame in cases:
        gs = S.graphics_settings(W, H, spp=spp, bounces=bounces, frame_index=frame)
        gpu.set_sharding(0, 1, 16)
        g = ptamd.Scene(gpu, scene)
        r = ptamd.Renderer(gpu, g, W, H)
        gpu.set_debug_flags(2)
        gpu.reset_counters()
        r.render(gs)
        gpu.sync()
        c = gpu.counters()
        gpu.set_debug_flags(0)
        assert c.SecondaryRays > W * H and c.BvhMismatches == 0


def test_edge_case_scenes_bit_identical_to_oracle(gpu, ptamd, oracle, pkg):
    """Inputs a scene loader can hand over: no instances at all, hidden instances (InstanceMask 0, Scene.ixx:365-377),
    mirrored and strongly non-uniform instance transforms (negative determinant: worldToObject, normals and the spawn
    offset must all follow), zero-area and repeated triangles inside a mesh, one mesh node instanced many times."""
    S, L = pkg.scenes, pkg.layouts
    W, H = 96, 64

    def check(scene, spp=2, bounces=5):
        gs = S.graphics_settings(W, H, spp=spp, bounces=bounces, frame_index=7)
        out, c = gpu_render(ptamd, gpu, scene, gs, W, H)
        ref_gb, ref_rays, ref_f32 = oracle.render(scene, gs, accel_mode=0, want_f32=True, layouts=L)
        assert c.PrimaryRays + c.SecondaryRays == ref_rays
        assert_gbuffer_identical(out, ref_gb)
        assert np.array_equal(out["RadianceF32"].view(np.uint32), ref_f32.view(np.uint32))
        return out

    # (a) nothing to hit: every pixel keeps the constant environment colour, no secondary ray is traced
    cam = S.make_camera((0, 0, -2.0), hfov_deg=90.0, aspect=W / H)
    empty = S.Scene([], [], cam, S.make_scene_data((0.25, 0.5, 0.75, 1.0)), name="empty").finalize()
    out = check(empty)
    rad16 = out["Radiance"].view(np.float16)[..., :3].astype(np.float32)          # primary misses keep what the G-buffer pass wrote
    assert np.all(np.isinf(out["Position"][..., 3])) and np.allclose(rad16, (0.25, 0.5, 0.75))

    # (b) hidden instances: the boxes and one wall are in the TLAS with mask 0
    hidden = S.cornell_box(aspect=W / H, variant="ggx")
    for k in (3, 6, 7):
        hidden.objects[k].visible = False
    hidden.finalize()
    check(hidden)

    # (c) mirrored (negative determinant) and 40:1 non-uniform instance transforms
    odd = S.cornell_box(aspect=W / H, variant="ggx", glass_sphere=True)
    odd.objects[6].transform = S.trs((-0.35, -0.4, 0.35), -18.0, (-0.6, 1.2, 0.6))
    odd.objects[7].transform = S.trs((0.35, -0.9, -0.25), 15.0, (0.8, 0.02, 0.8))
    odd.objects[8].transform = S.trs((0.3, -0.2, -0.3), 30.0, (0.25, -0.35, 0.15), pitch_deg=20.0)
    odd.finalize()
    check(odd)

    # (d) degenerate geometry inside a mesh: a zero-area triangle, a collinear one and an exact duplicate of a real one
    deg = S.cornell_box(aspect=W / H, variant="diffuse")
    m = deg.nodes[6].meshes[0]
    extra = np.array([0, 0, 0,  0, 1, 1,  0, 1, 2], m.indices.dtype)           # point, edge, copy of triangle 0
    extra[6:9] = m.indices[0:3]
    m.indices = np.concatenate([m.indices, extra])
    deg.finalize()
    check(deg)

    # (e) one mesh node, many instances (same BLAS behind different transforms and InstanceIDs)
    many = S.cornell_box(aspect=W / H, variant="ggx")
    for k in range(10):
        many.objects.append(S.RenderObject(7, S.trs((-0.8 + 0.17 * k, -0.9 + 0.08 * k, 0.6 - 0.1 * k), 13.0 * k, (0.12, 0.12, 0.12))))
    many.finalize()
    check(many)


def test_pathological_mesh_coincident_centroids_and_diagonal_slivers(gpu, ptamd, oracle, pkg):
    """What a Morton-ordered builder likes least: thousands of triangles whose centroids coincide (one Morton cell: the
    split falls back to the primitive's position in the sorted order) and a long diagonal strip of slivers (boxes that overlap
    everything along the diagonal). The structure must stay within the traversal stack (StackOverflows 0, depth reported), agree
    with the device's brute-force loop on every bounce ray, and the image must be the oracle's brute-force image bit for bit."""
    S, L = pkg.scenes, pkg.layouts
    W, H = 64, 48
    rng = np.random.default_rng(99)
    # (a) 3000 triangles around one centroid: random orientations and sizes, each vertex triple sums to the same point
    n_a = 3000
    c = np.array([0.0, 0.2, 2.0])
    a = rng.standard_normal((n_a, 3)); b = rng.standard_normal((n_a, 3))
    r = (0.05 + 0.6 * rng.random((n_a, 1)))
    v0 = c + r * a; v1 = c + r * b; v2 = c - r * (a + b)                    # v0 + v1 + v2 = 3 c
    pos_a = np.stack([v0, v1, v2], 1).reshape(-1, 3)
    idx_a = np.arange(3 * n_a)
    # (b) a diagonal strip of 1500 sliver quads from (-2,-1,0.5) to (2,1.5,4.5), 1 mm wide
    n_b = 1500
    t = np.linspace(0.0, 1.0, n_b + 1)[:, None]
    p = np.array([-2.0, -1.0, 0.5]) + t * np.array([4.0, 2.5, 4.0])
    off = np.array([0.001, -0.001, 0.0])
    pos_b = np.concatenate([p, p + off], 0)
    i0 = np.arange(n_b)
    idx_b = np.stack([i0, i0 + 1, i0 + n_b + 1, i0 + 1, i0 + n_b + 2, i0 + n_b + 1], -1).reshape(-1)
    mat_a = S.material((0.7, 0.5, 0.3), metallic=0.0, roughness=0.6)
    mat_b = S.material((0.9, 0.9, 0.9), metallic=1.0, roughness=0.2)
    meshes = [S.Mesh(S.make_vertices(pos_a), S.make_indices(idx_a), False, mat_a),
              S.Mesh(S.make_vertices(pos_b), S.make_indices(idx_b), False, mat_b)]
    cam = S.make_camera((0, 0.2, -1.0), hfov_deg=80.0, aspect=W / H)
    scene = S.Scene([S.MeshNode(meshes)], [S.RenderObject(0, S.trs()), S.RenderObject(0, S.trs((0.4, -0.3, 1.0), 40.0, (0.5, 0.5, 0.5)))],
                    cam, S.make_scene_data((0.6, 0.7, 0.9, 1.0)), name="pathological").finalize()
    gs = S.graphics_settings(W, H, spp=2, bounces=4, frame_index=5)

    out, cnt = gpu_render(ptamd, gpu, scene, gs, W, H)
    ref_gb, ref_rays, ref_f32 = oracle.render(scene, gs, accel_mode=0, want_f32=True, layouts=L)      # brute force: no BVH of its own to trust
    assert cnt.StackOverflows == 0
    assert cnt.PrimaryRays + cnt.SecondaryRays == ref_rays and cnt.SecondaryRays > W * H // 4
    assert_gbuffer_identical(out, ref_gb)
    assert np.array_equal(out["RadianceF32"].view(np.uint32), ref_f32.view(np.uint32))

    # the same frame with the device validator: both traversals per bounce ray, disagreements counted
    gpu.set_sharding(0, 1, 16)
    g = ptamd.Scene(gpu, scene)
    r = ptamd.Renderer(gpu, g, W, H)
    st = gpu.accel_stats()
    assert 0 < st.MaxBottomLevelDepth <= 28                                 # 6000 triangles: a balanced wide tree is 4-5 deep; the stack allows 2 * (tlas + blas) + 4 <= 64
    gpu.set_debug_flags(2); gpu.reset_counters()
    r.render(gs); gpu.sync()
    c2 = gpu.counters()
    gpu.set_debug_flags(0)
    g.close()
    assert c2.BvhMismatches == 0 and c2.StackOverflows == 0


def test_degenerate_meshes_build_and_render(gpu, ptamd, oracle, pkg):
    """a mesh of zero-area triangles only (collinear and coincident vertices: every collapse cost is 0, so the cost tables
    see no gain in opening a node) and an instance of it next to ordinary geometry. D3D12 accepts such a mesh; the builder must too
    (node capacity covers the worst case, ties go to more roots), and the image must be the oracle's brute-force image."""
    S, L = pkg.scenes, pkg.layouts
    W, H = 64, 40
    n = 700
    t = np.linspace(-1.0, 1.0, n)[:, None]
    line = np.array([0.0, 0.1, 1.5]) + t * np.array([1.0, 0.2, 0.4])                # n points on one line
    pos = np.concatenate([line, line[:1].repeat(8, 0)], 0)                        # + 8 copies of one point
    i0 = np.arange(n - 2)
    idx = np.concatenate([np.stack([i0, i0 + 1, i0 + 2], -1).reshape(-1), n + np.arange(6)])      # collinear triples + two point-triangles
    degenerate = S.Mesh(S.make_vertices(pos), S.make_indices(idx), False, S.material((0.9, 0.1, 0.1)))
    base = S.cornell_box(aspect=W / H, glass_sphere=True)
    nodes = list(base.nodes) + [S.MeshNode([degenerate])]
    objects = list(base.objects) + [S.RenderObject(len(nodes) - 1, S.trs()), S.RenderObject(len(nodes) - 1, S.trs((0.1, 0.2, 0.0), 30.0, (0.5, 0.5, 0.5)))]
    scene = S.Scene(nodes, objects, base.camera, base.scene_data, name="degenerate").finalize()
    gs = S.graphics_settings(W, H, spp=2, bounces=4, frame_index=2)
    out, cnt = gpu_render(ptamd, gpu, scene, gs, W, H)
    ref_gb, ref_rays, ref_f32 = oracle.render(scene, gs, accel_mode=0, want_f32=True, layouts=L)
    assert cnt.StackOverflows == 0 and cnt.PrimaryRays + cnt.SecondaryRays == ref_rays
    assert_gbuffer_identical(out, ref_gb)
    assert np.array_equal(out["RadianceF32"].view(np.uint32), ref_f32.view(np.uint32))


def test_traversal_schedules_agree(gpu, ptamd, pkg):
    """The three schedules of the bounce-ray traversal (flat instance scan with wave-compacted work items, phase-aligned TLAS
    walk, interleaved TLAS/BLAS) share tri_test / is_better, and a round is either one fused launch (k_round) or the
    k_shade + k_extend2 pair: images and ray counts must be identical bit for bit in every combination."""
    S = pkg.scenes
    W, H = 96, 64
    scenes = [S.cornell_box(aspect=W / H, variant="ggx", glass_sphere=True),       # 9 instances: quads, boxes and a 320-triangle sphere
              S.cornell_box_textured(env=None, aspect=W / H),                      # alpha-tested candidates inside work items
              S.instanced_grid(n=5, aspect=W / H)]                                 # 25 instances: still the flat schedule by default
    for scene in scenes:
        gs = S.graphics_settings(W, H, spp=3, bounces=7, frame_index=2)
        results = []
        for flags in (0, 8, 4, 0x10, 0x18):           # default (fused round), _PHASED, _V1, _UNFUSED_ROUNDS, _UNFUSED_ROUNDS | _PHASED
            gpu.set_sharding(0, 1, 16)
            g = ptamd.Scene(gpu, scene)
            r = ptamd.Renderer(gpu, g, W, H, with_f32=True)
            gpu.set_debug_flags(flags); gpu.reset_counters()
            r.render(gs); gpu.sync()
            c = gpu.counters()
            results.append((ptamd.textures_to_numpy(r.textures)["RadianceF32"].view(np.uint32).copy(), c.SecondaryRays))
            gpu.set_debug_flags(0)
        for img, rays in results[1:]:
            assert rays == results[0][1] and np.array_equal(img, results[0][0])


def test_streaming_and_lockstep_schedules_agree(gpu, ptamd, pkg):
    """Scenes whose traversal copy does not fit LDS take the streaming form of a round by default (persistent traversal lanes that
    pull rays from the sub-queue and hand hits to a batched shade); PT_DEBUG_LOCKSTEP (flat or phased schedule, by instance count),
    _PHASED, _V1 and the two-kernel form are the same arithmetic on other schedules: images and ray counts identical bit for bit,
    with and without the traversal statistics variant."""
    S = pkg.scenes
    W, H = 128, 80
    scenes = [S.sponza_scale(n_side=48, aspect=W / H),                     # 2 instances, one 4.6 k-triangle BLAS: flat when lock-step
              S.instanced_grid(n=24, aspect=W / H)]                        # 578 instances: phased when lock-step
    for scene in scenes:
        scene.scene_data = S.make_scene_data((0.2, 0.3, 0.4, 1.0))
        gs = S.graphics_settings(W, H, spp=3, bounces=7, frame_index=2)
        results = []
        for flags in (0, 0x20, 8, 4, 0x10, 1, 0x21):
            gpu.set_sharding(0, 1, 16)
            g = ptamd.Scene(gpu, scene)
            r = ptamd.Renderer(gpu, g, W, H, with_f32=True)
            gpu.set_debug_flags(flags); gpu.reset_counters()
            r.render(gs); gpu.sync()
            c = gpu.counters()
            assert c.StackOverflows == 0
            if flags & 1:
                assert c.NodesVisited > 0 and c.TrianglesTested > 0
            results.append((ptamd.textures_to_numpy(r.textures)["RadianceF32"].view(np.uint32).copy(), c.SecondaryRays))
            gpu.set_debug_flags(0)
        # pt_set_frames_in_flight only sizes the streaming traversal's grid (512 blocks instead of 1024): same image
        gpu.set_frames_in_flight(3)
        g = ptamd.Scene(gpu, scene)
        r = ptamd.Renderer(gpu, g, W, H, with_f32=True)
        gpu.reset_counters()
        r.render(gs); gpu.sync()
        results.append((ptamd.textures_to_numpy(r.textures)["RadianceF32"].view(np.uint32).copy(), gpu.counters().SecondaryRays))
        gpu.set_frames_in_flight(1)
        for img, rays in results[1:]:
            assert rays == results[0][1] and np.array_equal(img, results[0][0])


def test_round_chains_do_not_change_the_image(gpu, ptamd, pkg):
    """pt_set_round_chains: the rounds of a frame as 1..4 independent chains over groups of sub-queues. On the default stream the chains run
    one after the other (no concurrency: this pins the group arithmetic -- which block serves which sub-queue, the grids of a part of the
    queue); with a caller's stream every chain is a graph on a stream of its own (test_round_chains_on_streams). Fused form (Cornell) and
    streaming form (a mesh beyond LDS), incl. group sizes that do not divide the 32 / 128 sub-queues."""
    S = pkg.scenes
    W, H = 320, 180
    scenes = [S.cornell_box(aspect=W / H, variant="ggx", glass_sphere=True), S.sponza_scale(n_side=48, aspect=W / H)]
    scenes[1].scene_data = S.make_scene_data((0.2, 0.3, 0.4, 1.0))
    gs = S.graphics_settings(W, H, spp=2, bounces=5, frame_index=3)
    try:
        for scene in scenes:
            ref = None
            for n in (1, 2, 3, 4):
                gpu.set_round_chains(n)
                out, c = gpu_render(ptamd, gpu, scene, gs, W, H)
                assert c.StackOverflows == 0
                if ref is None:
                    ref = (out, c)
                else:
                    assert c.SecondaryRays == ref[1].SecondaryRays, n
                    assert np.array_equal(out["RadianceF32"].view(np.uint32), ref[0]["RadianceF32"].view(np.uint32)), n
            gpu.set_round_chains(3)                                  # a shard: fewer tiles than sub-queues in some groups
            part, cp = gpu_render(ptamd, gpu, scene, gs, W, H, sharding=(1, 4, 16))
            gpu.set_round_chains(1)
            part1, cp1 = gpu_render(ptamd, gpu, scene, gs, W, H, sharding=(1, 4, 16))
            assert cp.SecondaryRays == cp1.SecondaryRays and np.array_equal(part["Radiance"], part1["Radiance"])
    finally:
        gpu.set_round_chains(0)


def test_round_chains_on_streams(ptamd, pkg):
    """The product form of the chains: a context on a caller's stream replays every chain as a linear graph on a stream of its own, forked from and
    joined to the caller's stream. Several frames back to back (each frame's preamble must wait for the previous frame's chains; the G-buffer
    pass of the next frame rewrites textures the chains read) must equal the frames rendered with one chain."""
    import torch
    S = pkg.scenes
    W, H = 320, 180
    for scene in (S.cornell_box(aspect=W / H, variant="ggx", glass_sphere=True), S.sponza_scale(n_side=48, aspect=W / H)):
        settings = [S.graphics_settings(W, H, spp=2, bounces=5, frame_index=f) for f in range(4)]
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            ctx = ptamd.DeviceContext(0, stream=stream.cuda_stream)
            try:
                g = ptamd.Scene(ctx, scene)
                r = ptamd.Renderer(ctx, g, W, H, with_f32=True)
                frames = {}
                for n in (1, 3, 2, 4):
                    ctx.set_round_chains(n)
                    outs = []
                    for gs in settings:                               # enqueued without waiting in between
                        r.render(gs)
                        outs.append(r.textures["RadianceF32"].clone())    # stream-ordered behind the frame
                    ctx.sync()
                    frames[n] = [o.cpu().numpy() for o in outs]
                for n in (3, 2, 4):
                    for f in range(len(settings)):
                        assert np.array_equal(frames[n][f].view(np.uint32), frames[1][f].view(np.uint32)), (scene.name, n, f)
            finally:
                ctx.close()


def test_deterministic_and_sharding_invariant(gpu, ptamd, pkg):
    """Run twice: bit-identical. Render as rank r of 3 and of 8: the assembled frame equals the unsharded one
    (RNG seeds and camera rays use global pixel coordinates)."""
    S = pkg.scenes
    ge.load_package()
    import dxpbrt_amd.sharding as SH
    W, H = 160, 90
    scene = S.cornell_box(aspect=W / H, glass_sphere=True)
    gs = S.graphics_settings(W, H, spp=3, bounces=6, frame_index=9)
    a, ca = gpu_render(ptamd, gpu, scene, gs, W, H)
    b, cb = gpu_render(ptamd, gpu, scene, gs, W, H)
    assert np.array_equal(a["Radiance"], b["Radiance"]) and ca.SecondaryRays == cb.SecondaryRays
    for world, band in ((3, 16), (8, 4)):
        pieces, rays = [], 0
        for r in range(world):
            o, c = gpu_render(ptamd, gpu, scene, gs, W, H, sharding=(r, world, band))
            assert o["Radiance"].shape[0] == SH.local_rows(H, r, world, band)
            pieces.append(o["Radiance"]); rays += c.PrimaryRays + c.SecondaryRays
        assert np.array_equal(SH.deinterleave(pieces, H, band), a["Radiance"])
        assert rays == ca.PrimaryRays + ca.SecondaryRays


def test_device_deinterleave(gpu, ptamd):
    import ctypes as C
    import torch
    ge.load_package()
    import dxpbrt_amd.sharding as SH
    H, W, world, band = 45, 24, 4, 8
    rng = np.random.default_rng(0)
    full = rng.integers(0, 65535, (H, W, 4)).astype(np.uint16)
    max_rows = max(SH.local_rows(H, r, world, band) for r in range(world))
    gathered = np.zeros((world, max_rows, W, 4), np.uint16)
    for r in range(world):
        p = SH.extract_local(full, r, world, band)
        gathered[r, :p.shape[0]] = p
    d_g = torch.from_numpy(gathered.view(np.int16)).cuda()
    d_out = torch.zeros((H, W, 4), dtype=torch.int16, device="cuda")
    offs = np.arange(world, dtype=np.uint64) * np.uint64(max_rows * W * 8)
    gpu.check(gpu.lib.pt_deinterleave_bands(gpu.handle, d_out.data_ptr(), d_g.data_ptr(), offs.ctypes.data, world, band, W, H, 8))
    gpu.sync()
    assert np.array_equal(d_out.cpu().numpy().view(np.uint16), full)


def test_gather_bands_one_gpu_plays_every_rank(gpu, ptamd):
    """pt_gather_bands with PT_DEBUG_GATHER_LOCAL_ONLY: rank r's bands go to their rows of the full frame and nothing is exchanged,
    so one GPU can play every rank in turn (root = r): after all of them the frame is complete. Without the flag and without a
    communicator a sharded gather is refused; with one rank it is a plain copy and needs no communicator."""
    import torch
    ge.load_package()
    import dxpbrt_amd.sharding as SH
    rng = np.random.default_rng(1)
    for (H, W, world, band, px) in ((45, 24, 4, 8, 8), (1080, 64, 8, 16, 8), (33, 8, 3, 16, 16)):
        full = rng.integers(0, 2 ** 31, (H, W, px // 4), dtype=np.int64).astype(np.int32)
        d_out = torch.zeros((H, W, px // 4), dtype=torch.int32, device="cuda")
        gpu.set_debug_flags(0x40)
        for r in range(world):
            local = torch.from_numpy(np.ascontiguousarray(SH.extract_local(full, r, world, band))).cuda()
            gpu.set_sharding(r, world, band)
            gpu.gather_bands(local, d_out, W, H, px, root=r)
        gpu.sync()
        assert np.array_equal(d_out.cpu().numpy(), full)
        gpu.set_debug_flags(0)
        gpu.set_sharding(1, world, band)
        with pytest.raises(ptamd.PtError) as e:
            gpu.gather_bands(local, d_out, W, H, px, root=0)
        assert "communicator" in str(e.value)
        gpu.set_sharding(0, 1, band)
        d_one = torch.zeros_like(d_out)
        gpu.gather_bands(torch.from_numpy(full).cuda(), d_one, W, H, px, root=0)
        gpu.sync()
        assert np.array_equal(d_one.cpu().numpy(), full)
    gpu.set_sharding(0, 1, 16)


def test_gather_bands_self_exchange_runs_the_rccl_path(ptamd, pkg):
    """VERDICT r3 item 4b / ADVICE r3: with one GPU per box `exchange = N > 1` never let an ncclSend / ncclRecv execute. Under
    PT_DEBUG_GATHER_SELF_EXCHANGE a world-size-1 communicator carries the rank's OWN bands the way a foreign band travels: ncclSend to itself
    + ncclRecv from itself, one pair per band, all in one group, the receive aimed at the band's rows of the full frame. One GPU plays every
    rank of an 8-rank sharding of a 1080p frame in turn (8-9 messages of 245 KB per rank: the real message sizes), then the
    68 bands of an unsharded frame in ONE group; and three contexts on three streams with a communicator each run their gathers
    concurrently, as bench.py's frames in flight do. Every assembled frame must equal the source bit for bit."""
    import torch
    ge.load_package()
    import dxpbrt_amd.sharding as SH
    rng = np.random.default_rng(7)
    H, W, px, band = 1080, 1920, 8, 16
    full = rng.integers(0, 2 ** 31, (H, W, px // 4), dtype=np.int64).astype(np.int32)
    lanes = []
    try:
        for _ in range(3):
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                ctx = ptamd.DeviceContext(0, stream=st.cuda_stream)
            ctx.comm_init(ptamd.DeviceContext.comm_unique_id(), 0, 1)
            ctx.set_debug_flags(0x80)
            lanes.append((st, ctx))
        # (1) one GPU plays the eight ranks in turn, on lane 0
        st, ctx = lanes[0]
        with torch.cuda.stream(st):
            d_out = torch.zeros((H, W, px // 4), dtype=torch.int32, device="cuda")
            for world in (8, 3):
                d_out.zero_()
                for r in range(world):
                    local = torch.from_numpy(np.ascontiguousarray(SH.extract_local(full, r, world, band))).cuda()
                    ctx.set_sharding(r, world, band)
                    ctx.gather_bands(local, d_out, W, H, px, root=0)
                ctx.sync()
                assert np.array_equal(d_out.cpu().numpy(), full), world
        # (2) three communicators on three streams at once, each moving all 68 bands of the frame in one group, several frames deep
        src = torch.from_numpy(full).cuda()
        outs = [torch.zeros_like(src) for _ in lanes]
        torch.cuda.synchronize()
        for rep in range(4):
            for (st, ctx), o in zip(lanes, outs):
                with torch.cuda.stream(st):
                    ctx.set_sharding(0, 1, band)
                    if rep == 3:
                        o.zero_()
                    ctx.gather_bands(src, o, W, H, px, root=0)
        torch.cuda.synchronize()
        for o in outs:
            assert np.array_equal(o.cpu().numpy(), full)
        # without a communicator the flag is refused
        bare = ptamd.DeviceContext(0)
        bare.set_debug_flags(0x80)
        with pytest.raises(ptamd.PtError, match="world size 1"):
            bare.gather_bands(src, outs[0], W, H, px, root=0)
        bare.close()
    finally:
        for st, ctx in lanes:
            ctx.close()


def test_full_size_properties_c2(gpu, ptamd, pkg):
    """BASELINE configs[1] at full size (1920x1080, 4 spp, 8 bounces): properties that need no oracle."""
    S = pkg.scenes
    W, H = 1920, 1080
    scene = S.cornell_box(aspect=W / H, variant="ggx")
    gs = S.graphics_settings(W, H, spp=4, bounces=8)
    a, ca = gpu_render(ptamd, gpu, scene, gs, W, H)
    rays = ca.PrimaryRays + ca.SecondaryRays
    assert ca.PrimaryRays == W * H and W * H < rays <= W * H * (1 + 4 * 8)
    rad = a["RadianceF32"][..., :3]
    assert np.isfinite(rad).all() and (rad >= 0).all() and 0.05 < rad.mean() < 5.0
    assert np.all(np.isfinite(a["Position"][..., 3]))           # camera inside the opening: every primary ray hits
    # fp16 output texture == rounding of the fp32 value
    assert np.array_equal(a["Radiance"][..., :3], rad.astype(np.float16).view(np.uint16))
    # idempotence: same inputs -> same bits
    b, cb = gpu_render(ptamd, gpu, scene, gs, W, H)
    assert np.array_equal(a["Radiance"], b["Radiance"]) and cb.SecondaryRays == ca.SecondaryRays
    # linearity of the estimator in the emitter: doubling EmissiveStrength doubles every pixel exactly (power of two)
    scene2 = S.cornell_box(aspect=W / H, variant="ggx")
    scene2.nodes[5].meshes[0].material["EmissiveStrength"] = 30.0
    scene2.finalize()
    c2, _ = gpu_render(ptamd, gpu, scene2, gs, W, H)
    assert np.array_equal(c2["RadianceF32"][..., :3], 2.0 * rad)
    # a 2-rank sharded render of the same frame assembles to the same image
    ge.load_package()
    import dxpbrt_amd.sharding as SH
    pieces = [gpu_render(ptamd, gpu, scene, gs, W, H, sharding=(r, 2, 16))[0]["Radiance"] for r in range(2)]
    assert np.array_equal(SH.deinterleave(pieces, H, 16), a["Radiance"])


def test_frames_in_flight_are_independent(gpu, ptamd, pkg):
    """bench.py keeps three frames in flight: one context + HIP stream + hipGraph per lane. Lanes must not see each other:
    every frame, rendered while two others run, equals the same frame rendered alone on the default context."""
    import torch
    S = pkg.scenes
    W, H = 480, 272
    scene = S.cornell_box(aspect=W / H, variant="ggx")
    settings = [S.graphics_settings(W, H, spp=3, bounces=6, frame_index=f) for f in range(3)]
    alone = [gpu_render(ptamd, gpu, scene, gs, W, H)[0]["Radiance"] for gs in settings]
    lanes = []
    for _ in range(3):
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            ctx = ptamd.DeviceContext(0, stream=stream.cuda_stream)
            ctx.set_sharding(0, 1, 16)
            sc = ptamd.Scene(ctx, scene)
            lanes.append((stream, ctx, sc, ptamd.Renderer(ctx, sc, W, H)))
        stream.synchronize()
    try:
        for it in range(4):                                       # graph capture on the first pass, replay afterwards
            for k, (stream, ctx, sc, r) in enumerate(lanes):
                with torch.cuda.stream(stream):
                    r.render(settings[(k + it) % 3])
            torch.cuda.synchronize()
            for k, (stream, ctx, sc, r) in enumerate(lanes):
                got = ptamd.textures_to_numpy(r.textures)["Radiance"]
                assert np.array_equal(got, alone[(k + it) % 3]), (it, k)
    finally:
        for stream, ctx, sc, r in lanes:
            ctx.close()


def test_error_behaviour(ptamd, pkg):
    """status codes in place of the reference's exceptions (Source/RaytracingHelpers.ixx:83-88, ErrorHelpers.ixx)."""
    import ctypes as C
    import torch
    S, L = pkg.scenes, pkg.layouts
    ctx = ptamd.DeviceContext(0)
    try:
        k = np.zeros((), L.GBUFFER_CONSTANTS); k["RenderSize"] = (8, 8)
        t = ptamd.Textures()
        with pytest.raises(ptamd.PtError, match="top-level"):       # render before any acceleration structure
            ctx.check(ctx.lib.pt_gbuffer_render(ctx.handle, C.c_void_p(k.ctypes.data), C.addressof(t)))
        buf = torch.zeros(64, dtype=torch.uint8, device="cuda")
        g = ptamd.GeometryDesc(buf.data_ptr(), 2, 32, buf.data_ptr(), 3, 3, 1, 0)
        bid = C.c_uint64()
        with pytest.raises(ptamd.PtInvalidArgument, match="uint16 or uint32"):
            ctx.check(ctx.lib.pt_build_bottom_level(ctx.handle, C.addressof(g), 1, 0, C.byref(bid)))
        g.IndexStride, g.IndexCount = 2, 4
        with pytest.raises(ptamd.PtInvalidArgument, match="divisible by 3"):
            ctx.check(ctx.lib.pt_build_bottom_level(ctx.handle, C.addressof(g), 1, 0, C.byref(bid)))
        d = ptamd.InstanceDesc(); d.AccelerationStructure = 12345
        with pytest.raises(ptamd.PtInvalidArgument, match="unknown bottom-level"):
            ctx.check(ctx.lib.pt_build_top_level(ctx.handle, C.addressof(d), 1, 0))
        # empty scene: builds, renders, every pixel is a miss with the environment colour
        ctx.check(ctx.lib.pt_build_top_level(ctx.handle, None, 0, 0))
        assert ctx.accel_stats().InstanceCount == 0
    finally:
        ctx.close()


def test_shared_scene_frames_in_flight(gpu, ptamd, pkg):
    """pt_share_scene: three contexts on three streams render from ONE copy of the scene (built by the first). Every frame equals the
    same frame rendered alone; the views hold no structure memory; the owner refuses to rebuild or release while it is viewed."""
    import ctypes as C
    import torch
    S = pkg.scenes
    W, H = 320, 180
    scene = S.sponza_scale(n_side=48, aspect=W / H)              # beyond LDS: the traversal copy is read from memory by all three
    scene.scene_data = S.make_scene_data((0.2, 0.3, 0.4, 1.0))
    settings = [S.graphics_settings(W, H, spp=2, bounces=5, frame_index=f) for f in range(3)]
    alone = [gpu_render(ptamd, gpu, scene, gs, W, H)[0]["Radiance"] for gs in settings]
    lanes = []
    for k in range(3):
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            ctx = ptamd.DeviceContext(0, stream=stream.cuda_stream)
            ctx.set_sharding(0, 1, 16)
            sc = ptamd.Scene(ctx, scene) if k == 0 else ptamd.SharedScene(ctx, lanes[0][2])
            lanes.append((stream, ctx, sc, ptamd.Renderer(ctx, sc, W, H)))
        stream.synchronize()
    try:
        own, view = lanes[0][1].accel_stats(), lanes[1][1].accel_stats()
        assert own.SharedScene == 0 and own.BlobBytes > 0 and view.SharedScene == 1 and view.BlobBytes == 0
        assert view.InstanceCount == own.InstanceCount
        for it in range(3):
            for k, (stream, ctx, sc, r) in enumerate(lanes):
                with torch.cuda.stream(stream):
                    r.render(settings[(k + it) % 3])
            torch.cuda.synchronize()
            for k, (stream, ctx, sc, r) in enumerate(lanes):
                assert np.array_equal(ptamd.textures_to_numpy(r.textures)["Radiance"], alone[(k + it) % 3]), (it, k)
        owner = lanes[0][1]
        with pytest.raises(ptamd.PtInvalidArgument, match="view this context"):
            owner.check(owner.lib.pt_release_bottom_level(owner.handle, lanes[0][2].blas_ids[0]))
        with pytest.raises(ptamd.PtInvalidArgument, match="view this context"):
            lanes[0][2]._build_top_level()
    finally:
        for stream, ctx, sc, r in reversed(lanes):                # views first, then the owner
            ctx.close()


def test_release_of_a_referenced_bottom_level_drops_the_top_level(ptamd, pkg):
    """pt_release_bottom_level used to free arrays the live TLAS still pointed at. Now the TLAS dies with it and a render
    answers PT_ERROR_NOT_READY (status -4) until the next pt_build_top_level."""
    S = pkg.scenes
    W, H = 64, 36
    ctx = ptamd.DeviceContext(0)
    try:
        scene = S.cornell_box(aspect=W / H)
        g = ptamd.Scene(ctx, scene)
        r = ptamd.Renderer(ctx, g, W, H)
        gs = S.graphics_settings(W, H, spp=1, bounces=2)
        r.render(gs); ctx.sync()
        ctx.check(ctx.lib.pt_release_bottom_level(ctx.handle, g.blas_ids[6]))
        with pytest.raises(ptamd.PtError, match="status -4"):
            r.render(gs)
        # an unreferenced bottom level can go without touching the TLAS
        g2 = ptamd.Scene(ctx, scene)                              # builds its own BLASes and a new TLAS
        r2 = ptamd.Renderer(ctx, g2, W, H)
        for bid in g.blas_ids[:6] + g.blas_ids[7:]:
            ctx.check(ctx.lib.pt_release_bottom_level(ctx.handle, bid))
        g.blas_ids = []
        r2.render(gs); ctx.sync()
    finally:
        ctx.close()


def test_static_bottom_levels_live_in_the_traversal_copy_only(ptamd, oracle, pkg):
    """nodes, packets and indices of a static bottom level used to exist twice -- in the arrays its build left and in the
    traversal copy every top-level build assembled from them. The first top-level build that sees a static bottom level now ADOPTS it: copies it
    in, frees the arrays (PtAccelStats.OwnedBottomLevelBytes: what is held outside the copy); later builds find it in place; a change of the
    layout moves the pieces from the old copy to a new one; adopted bottom levels a top level does not name ride along, so that a later one can
    name them again; an updatable bottom level keeps its arrays (the refit writes them)."""
    S, L = pkg.scenes, pkg.layouts
    W, H = 160, 90
    gs = S.graphics_settings(W, H, spp=2, bounces=4)
    ctx = ptamd.DeviceContext(0)
    try:
        a = S.sponza_scale(n_side=40, aspect=W / H)
        a.scene_data = S.make_scene_data((0.2, 0.3, 0.4, 1.0))
        ga = ptamd.Scene(ctx, a)
        st = ctx.accel_stats()
        assert st.OwnedBottomLevelBytes == 0 and st.BlobBytes >= st.NodeBytes + st.TriangleBytes
        ra = ptamd.Renderer(ctx, ga, W, H, with_f32=True)
        ra.render(gs); ctx.sync()
        first = ptamd.textures_to_numpy(ra.textures)
        ref_gb, ref_rays, ref_f32 = oracle.render(a, gs, accel_mode=1, want_f32=True, layouts=L)
        assert np.array_equal(first["RadianceF32"].view(np.uint32), ref_f32.view(np.uint32))
        ga._build_top_level()                                   # the same top level again: everything is in place, nothing moves
        assert ctx.accel_stats().OwnedBottomLevelBytes == 0
        ra.render(gs); ctx.sync()
        assert np.array_equal(ptamd.textures_to_numpy(ra.textures)["RadianceF32"], first["RadianceF32"])
        # a second scene on the same context: its top level names other bottom levels; the first scene's ride along in the copy ...
        b = S.instanced_grid(n=6, aspect=W / H)
        b.scene_data = S.make_scene_data((0.3, 0.3, 0.35, 1.0))
        gb = ptamd.Scene(ctx, b)
        assert ctx.accel_stats().OwnedBottomLevelBytes == 0
        rb = ptamd.Renderer(ctx, gb, W, H, with_f32=True)
        rb.render(gs); ctx.sync()
        refb_gb, refb_rays, refb_f32 = oracle.render(b, gs, accel_mode=1, want_f32=True, layouts=L)
        assert np.array_equal(ptamd.textures_to_numpy(rb.textures)["RadianceF32"].view(np.uint32), refb_f32.view(np.uint32))
        # ... so the first scene's top level can be built again (another layout: its pieces move) and renders the same frame
        import ctypes as C
        ctx.check(ctx.lib.pt_heap_resize(ctx.handle, len(a.heap)))                 # (the second scene had bound its own descriptor table and object data)
        for i, (item, t) in enumerate(zip(a.heap, ga._heap_dev)):
            ctx.check(ctx.lib.pt_heap_set_buffer(ctx.handle, i, C.c_void_p(t.data_ptr()), item.array.nbytes, item.stride))
        ctx.check(ctx.lib.pt_set_object_data(ctx.handle, ga.object_data.data_ptr(), len(a.object_data)))
        ctx.check(ctx.lib.pt_set_instance_data(ctx.handle, ga.instance_data.data_ptr(), len(a.instance_data)))
        ga._build_top_level()
        ra.render(gs); ctx.sync()
        assert np.array_equal(ptamd.textures_to_numpy(ra.textures)["RadianceF32"], first["RadianceF32"])
        gb.close(); ga.close()
        # a dynamic scene: the skinned mesh's bottom level keeps its arrays, the static ones do not
        d = S.dynamic_scene(aspect=W / H)
        gd = ptamd.Scene(ctx, d)
        owned = ctx.accel_stats().OwnedBottomLevelBytes
        bar = d.nodes[2].meshes[0]
        assert 0 < owned <= (bar.indices.size // 3) * (48 + 16 + 80)
        for k in range(3):
            gd.SkinSkeletalMeshes(bar, S.bar_pose(10.0 * k, 0.02 * k))
            gd.UpdateAccelerationStructures(2)
        assert ctx.accel_stats().OwnedBottomLevelBytes == owned
        gd.close()
    finally:
        ctx.close()


def test_scene_inputs_are_validated(ptamd, pkg):
    """indices the kernels dereference are checked once per scene change, with a message naming the culprit."""
    import torch
    S, L = pkg.scenes, pkg.layouts
    W, H = 64, 36
    gs = S.graphics_settings(W, H, spp=1, bounces=2)

    def render_with(mutate, match):
        ctx = ptamd.DeviceContext(0)
        try:
            scene = S.cornell_box_textured(aspect=W / H, env=None)
            g = ptamd.Scene(ctx, scene)
            od = scene.object_data.copy()
            n = mutate(od, len(scene.heap))
            dev = ptamd.to_device(od[:n] if n else od, g.device)
            ctx.check(ctx.lib.pt_set_object_data(ctx.handle, dev.data_ptr(), n if n else len(od)))
            r = ptamd.Renderer(ctx, g, W, H)
            with pytest.raises(ptamd.PtInvalidArgument, match=match):
                r.render(gs)
        finally:
            ctx.close()

    def bad_vertices(od, heap_count):
        od["MeshDescriptors"]["Vertices"][3] = heap_count + 5

    def texture_as_index_buffer(od, heap_count):
        tex = [int(d) for d in od["TextureMapInfoArray"]["Descriptor"].reshape(-1) if d != 0xFFFFFFFF][0]
        od["MeshDescriptors"]["Indices"][2] = tex

    def bad_texture(od, heap_count):
        od["TextureMapInfoArray"]["Descriptor"][1][0] = heap_count

    def too_few_objects(od, heap_count):
        return len(od) - 1

    render_with(bad_vertices, r"ObjectData\[3\]\.MeshDescriptors\.Vertices = \d+ is beyond the descriptor heap")
    render_with(texture_as_index_buffer, r"ObjectData\[2\]\.MeshDescriptors\.Indices = \d+ is not the kind")
    render_with(bad_texture, r"ObjectData\[1\]\.TextureMapInfo\.Descriptor = \d+ is beyond")
    render_with(too_few_objects, r"only \d+ objects are bound")
