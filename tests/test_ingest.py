"""Scene-JSON + glTF ingest: the loader conventions of Source/GLTFHelpers.ixx:142-537,
Source/MyScene.ixx:33-90, Source/JSONConverters.ixx:12-33 and Source/Scene.ixx:195-231."""
import json
import math
import os

import numpy as np
import pytest

import __graft_entry__ as ge


@pytest.fixture(scope="module")
def ingest():
    ge.load_package()
    import dxpbrt_amd.ingest as I
    return I


def world_triangles(scene):
    out = []
    for ro in scene.objects:
        m = np.asarray(ro.transform, np.float64)
        for mesh in scene.nodes[ro.node].meshes:
            p = mesh.vertices["Position"].astype(np.float64)
            w = p @ m[:, :3].T + m[:, 3]
            tri = w[mesh.indices.astype(np.int64).reshape(-1, 3)]
            out.append(np.sort(tri.reshape(-1, 9).round(5), axis=0))
    return np.sort(np.concatenate(out).round(4), axis=0)


def test_json_rotation_and_affine_conventions(ingest):
    I = ingest
    # Yaw about +Y (row-vector, left-handed): +Z forward turns towards +X for positive yaw
    r = I.rotation_from_json({"Yaw": 90, "Pitch": 0, "Roll": 0})
    assert np.allclose(np.array([0, 0, 1, 0]) @ r, [1, 0, 0, 0], atol=1e-12)
    # pitch sign is negated by the converter: positive Pitch looks up
    r = I.rotation_from_json({"Yaw": 0, "Pitch": 30, "Roll": 0})
    assert (np.array([0, 0, 1, 0]) @ r)[1] > 0
    # all-zero angles fall through to the raw quaternion
    s = math.sin(math.pi / 8); c = math.cos(math.pi / 8)
    r = I.rotation_from_json({"X": 0, "Y": s, "Z": 0, "W": c})
    assert np.allclose(r, I.rot_y(math.pi / 4), atol=1e-12)
    # AffineTransform = Scale * Rotation * Translation
    m = I.affine_from_json({"Translation": {"X": 1, "Y": 2, "Z": 3}, "Scale": {"X": 2, "Y": 2, "Z": 2}, "Rotation": {"Yaw": 90}})
    assert np.allclose(np.array([0, 0, 1, 1]) @ m, [3, 2, 3, 1], atol=1e-12)
    assert I.store_float3x4(m).shape == (3, 4) and np.allclose(I.store_float3x4(m)[:, 3], [1, 2, 3])


def test_missing_model_reference_raises(ingest, tmp_path):
    p = tmp_path / "s.json"
    p.write_text(json.dumps({"Models": {}, "RenderObjects": [{"Name": "a", "Model": "nope"}]}))
    with pytest.raises(RuntimeError, match="RenderObject a: Models nope not found"):
        ingest.load_scene_desc(str(p))


def test_roundtrip_geometry_materials_textures(ingest, pkg, tmp_path):
    S = pkg.scenes
    src = S.cornell_box_textured(env=None)
    path = ingest.export_scene(src, str(tmp_path), "cornell")
    dst = ingest.load_scene(path, aspect=16 / 9)
    assert len(dst.objects) == len(src.objects) and dst.triangle_count == src.triangle_count
    assert np.allclose(world_triangles(dst), world_triangles(src), atol=2e-4)
    for a, b in zip(src.object_data, dst.object_data):
        for k in ("BaseColor", "EmissiveStrength", "EmissiveColor", "Metallic", "Roughness", "IOR", "Transmission", "AlphaMode", "AlphaCutoff"):
            assert np.allclose(a["Material"][k], b["Material"][k], atol=1e-6), k
        # same texture slots bound, except the separate Metallic / Roughness maps: glTF (and the reference loader) only
        # know the packed MetallicRoughness texture (GLTFHelpers.ixx:393-399)
        for slot in (0, 1, 4, 5, 6):
            assert (a["TextureMapInfoArray"][slot]["Descriptor"] == 0xFFFFFFFF) == (b["TextureMapInfoArray"][slot]["Descriptor"] == 0xFFFFFFFF)
        assert b["TextureMapInfoArray"][2]["Descriptor"] == 0xFFFFFFFF and b["TextureMapInfoArray"][3]["Descriptor"] == 0xFFFFFFFF
    # base colour / emissive textures are sRGB, the others linear
    fmts = {slot: dst.heap[int(od["TextureMapInfoArray"][k]["Descriptor"])].fmt
            for od in dst.object_data for k, slot in enumerate(S.TEX_SLOTS) if od["TextureMapInfoArray"][k]["Descriptor"] != 0xFFFFFFFF}
    assert fmts["BaseColor"] == S.FMT_RGBA8_UNORM_SRGB and fmts["EmissiveColor"] == S.FMT_RGBA8_UNORM_SRGB
    assert fmts["Normal"] == S.FMT_RGBA8_UNORM and fmts["MetallicRoughness"] == S.FMT_RGBA8_UNORM
    # tangents are (re)computed whenever NORMAL + TEXCOORD_0 exist; instance transforms carry the Z flip (det < 0)
    assert all(m.has_tangents == (m.has_normals and m.has_uv[0]) for n in dst.nodes for m in n.meshes)
    assert all(np.linalg.det(np.asarray(o.transform)[:, :3]) < 0 for o in dst.objects)
    # index order is reversed: primitive k of the loaded mesh is file triangle N-1-k with its vertices swapped
    m_src, m_dst = src.nodes[6].meshes[0], dst.nodes[6].meshes[0]
    assert np.array_equal(m_dst.indices, m_src.indices) and m_dst.indices.dtype == np.uint16     # export reversed once, loader reversed back
    assert np.allclose(m_dst.vertices["Position"][:, 2], -m_src.vertices["Position"][:, 2])


def test_loaded_scene_renders_like_the_procedural_one(ingest, pkg, oracle, tmp_path):
    S, L = pkg.scenes, pkg.layouts
    src = S.cornell_box(aspect=1.0, variant="ggx", glass_sphere=True)
    dst = ingest.load_scene(ingest.export_scene(src, str(tmp_path), "c"), aspect=1.0)
    gs = S.graphics_settings(48, 48, spp=8, bounces=6)
    a = oracle.render(src, gs, accel_mode=1, want_f32=True, layouts=L)
    b = oracle.render(dst, gs, accel_mode=1, want_f32=True, layouts=L)
    # same geometry and materials, different vertex order / mirrored object space => not bit-identical, but the primary
    # hits coincide and the images agree statistically
    assert np.allclose(a[0]["Position"][..., :3], b[0]["Position"][..., :3], atol=1e-4)
    assert abs(a[2].mean() - b[2].mean()) < 0.08 * a[2].mean()


@pytest.mark.gpu
def test_gpu_matches_oracle_on_ingested_scene(ingest, pkg, oracle, gpu, ptamd, tmp_path):
    S, L = pkg.scenes, pkg.layouts
    src = S.cornell_box_textured(env=None)
    dst = ingest.load_scene(ingest.export_scene(src, str(tmp_path), "t"), aspect=96 / 54)
    W, H = 96, 54
    gs = S.graphics_settings(W, H, spp=2, bounces=6)
    gpu.set_sharding(0, 1, 16)
    g = ptamd.Scene(gpu, dst)
    r = ptamd.Renderer(gpu, g, W, H, with_f32=True)
    gpu.reset_counters(); r.render(gs); gpu.sync()
    out = ptamd.textures_to_numpy(r.textures); c = gpu.counters()
    ref_gb, ref_rays, ref_f32 = oracle.render(dst, gs, accel_mode=0, want_f32=True, layouts=L)
    assert c.PrimaryRays + c.SecondaryRays == ref_rays
    assert np.array_equal(out["Position"].view(np.uint32), ref_gb["Position"].view(np.uint32))
    assert np.array_equal(out["RadianceF32"].view(np.uint32), ref_f32.view(np.uint32))


def test_hand_written_fixture_matches_hand_derived_arrays(pkg):
    """an ingest check with no exporter in the loop. tests/golden/ingest/{fixture.gltf,scene.json} are written by
    tests/golden/make_ingest_fixture.py with struct + json only; every expected array below is derived BY HAND from the reference:

      indices   GLTFHelpers.ixx:179   slot count-1-i <- index i (order AND winding reversed); :183-188 R16_UINT iff count <= 65535,
                                      whatever the component type in the file (the 3-index triangle is UNSIGNED_INT there)
      vertices  positions as they are (the Z flip lives in the instance transform), normals SNORM16 (Vertex.ixx:17-22)
      instances Scene.ixx:199-214     world = GlobalTransform * Scale(1,1,-1) * (Scale * Rotation * Translation), row vectors; for a point
                                      (x, y, z) of the quad under RenderObject "a":  node chain (1,2,3)+2*RotY90 -> (2z+1, 2y+2, -2x+3),
                                      Z flip -> (2z+1, 2y+2, 2x-3), Yaw 90 (x' = z, z' = -x) -> (2x-3, 2y+2, -2z-1), +(10,0,0) -> (2x+7, 2y+2, -2z-1)
      rotations JSONConverters.ixx:18-26: {Yaw,Pitch,Roll} non-zero -> CreateFromYawPitchRoll(yaw, -pitch, -roll); all zero -> raw quaternion
                                      ((0,0,sin45,cos45) = 90 degrees about Z: (x, y, z) -> (-y, x, z))
    """
    import dxpbrt_amd.ingest as I
    import os
    L = pkg.layouts
    sc = I.load_scene(os.path.join(os.path.dirname(__file__), "golden", "ingest", "scene.json"), aspect=1.0)
    assert len(sc.nodes) == 2 and len(sc.objects) == 4
    quad, tri = sc.nodes[0].meshes[0], sc.nodes[1].meshes[0]
    # ---- index buffers
    assert quad.indices.dtype == np.uint16 and quad.indices.tolist() == [3, 2, 0, 2, 1, 0]
    assert tri.indices.dtype == np.uint16 and tri.indices.tolist() == [2, 1, 0]
    # ---- vertices
    assert quad.vertices["Position"].tolist() == [[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]]
    assert quad.vertices["Normal"].tolist() == [[0, 0, 32767]] * 4 and quad.has_normals and not quad.has_tangents
    assert tri.vertices["Position"].tolist() == [[0, 0, 0], [2, 0, 0], [0, 0, 2]] and not tri.has_normals
    # ---- materials (GLTFHelpers.ixx:330-368; no material -> Material() defaults, App.cpp:1044)
    m = quad.material
    assert np.allclose(m["BaseColor"], (0.8, 0.2, 0.1, 1.0)) and float(m["EmissiveStrength"]) == 3.0 and np.allclose(m["EmissiveColor"], (1.0, 0.5, 0.25))
    assert float(m["Metallic"]) == 0.25 and float(m["Roughness"]) == 0.5 and np.isclose(float(m["IOR"]), 1.33) and float(m["Transmission"]) == 0.75
    assert int(m["AlphaMode"]) == 1 and np.isclose(float(m["AlphaCutoff"]), 0.3)
    assert tri.material is None
    od = sc.object_data
    assert np.allclose(od["Material"]["BaseColor"][1], (0, 0, 0, 1)) and float(od["Material"]["IOR"][1]) == 1.5 and float(od["Material"]["Roughness"][1]) == 0.5
    assert int(od["VertexDesc"]["Stride"][0]) == 32 and int(od["VertexDesc"]["Normal"][0]) == 12
    assert int(od["VertexDesc"]["Normal"][1]) == 0xFFFFFFFF and int(od["VertexDesc"]["Tangent"][0]) == 0xFFFFFFFF
    # ---- instances: order (render object major, mesh nodes in scene order), ids, masks, transforms
    assert sc.instance_data["FirstGeometryIndex"].tolist() == [0, 1, 2, 3] and list(sc.instance_ids) == [0, 1, 2, 3]
    assert list(sc.instance_masks) == [255, 255, 0, 0] and list(sc.instance_blas) == [0, 1, 0, 1]
    expect = np.array([
        [[2, 0, 0, 7], [0, 2, 0, 2], [0, 0, -2, -1]],            # a / quad : (2x+7, 2y+2, -2z-1)
        [[0, 0, -1, 9.5], [0, 1, 0, 0], [-1, 0, 0, 1]],          # a / tri  : (9.5-z, y, 1-x)
        [[0, -2, 0, -2], [0, 0, 2, 6], [2, 0, 0, -3]],           # b / quad : (-2y-2, 2z+6, 2x-3)
        [[0, -1, 0, 0], [1, 0, 0, 4], [0, 0, -1, -0.5]],         # b / tri  : (-y, x+4, -z-0.5)
    ], np.float32)
    got = sc.instance_data["ObjectToWorld"].reshape(4, 3, 4)
    assert np.allclose(got, expect, atol=1e-6), got
    assert np.allclose(sc.instance_data["PreviousObjectToWorld"].reshape(4, 3, 4), expect, atol=1e-6)
    # ---- camera (Yaw 90: forward (0,0,1) -> (1,0,0), right (1,0,0) -> (0,0,-1)) and environment
    cam = sc.camera
    assert np.allclose(cam["Position"], (0, 1, -5)) and np.allclose(cam["ForwardDirection"], (1, 0, 0), atol=1e-6)
    assert np.allclose(cam["RightDirection"] / np.linalg.norm(cam["RightDirection"]), (0, 0, -1), atol=1e-6)
    assert np.allclose(cam["UpDirection"] / np.linalg.norm(cam["UpDirection"]), (0, 1, 0), atol=1e-6)
    assert np.allclose(sc.scene_data["EnvironmentLightColor"], (0.1, 0.2, 0.3, 1.0))


def test_glb_container_loads_like_the_gltf(pkg):
    """The binary container (.glb: header, JSON chunk, BIN chunk; hand-packed by tests/golden/make_ingest_fixture.py with struct only)
    gives the same scene as the .gltf with its base64 buffer: fastgltf::Parser::loadGltf reads both (Source/GLTFHelpers.ixx:53-57)."""
    import dxpbrt_amd.ingest as I
    import os
    d = os.path.join(os.path.dirname(__file__), "golden", "ingest")
    raw = open(os.path.join(d, "fixture.glb"), "rb").read()
    assert raw[:4] == b"glTF" and int.from_bytes(raw[4:8], "little") == 2 and int.from_bytes(raw[8:12], "little") == len(raw) and len(raw) % 4 == 0
    a = I.load_scene(os.path.join(d, "scene.json"), aspect=1.0)
    b = I.load_scene(os.path.join(d, "scene_glb.json"), aspect=1.0)
    assert len(a.nodes) == len(b.nodes) and len(a.objects) == len(b.objects)
    for na, nb in zip(a.nodes, b.nodes):
        for ma, mb in zip(na.meshes, nb.meshes):
            assert ma.vertices.tobytes() == mb.vertices.tobytes() and ma.indices.tobytes() == mb.indices.tobytes()
            assert (ma.material is None) == (mb.material is None) and (ma.material is None or ma.material.tobytes() == mb.material.tobytes())
    assert a.object_data.tobytes() == b.object_data.tobytes() and a.instance_data.tobytes() == b.instance_data.tobytes()
