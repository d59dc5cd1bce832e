"""Oracle vs the committed golden fixtures (tests/golden/, produced by make_golden.py from the oracle itself:
self-consistency pins -- the reference has no golden vectors) and oracle-internal invariants."""
import importlib.util
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")
_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
make_golden = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(make_golden)


@pytest.mark.parametrize("name", sorted(make_golden.CASES))
def test_oracle_matches_golden(name, oracle):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    scene, gs, gb, rays, f32 = make_golden.render_case(name, oracle)
    assert rays == int(g["rays"])
    assert np.array_equal(f32.view(np.uint32), g["radiance_f32"].view(np.uint32))
    assert np.array_equal(gb["Radiance"], g["radiance_f16"])
    assert np.array_equal(gb["Position"].view(np.uint32), g["position"].view(np.uint32))
    assert np.array_equal(gb["FlatNormal"], g["flat_normal"])
    assert np.array_equal(gb["NormalRoughness"], g["normal_roughness"])
    assert np.array_equal(gb["BaseColorMetalness"], g["base_color_metalness"])


def test_brute_force_equals_bvh(oracle, pkg):
    """closest-hit tie-break makes the result independent of traversal order: mode 0 == mode 1 bit for bit."""
    S, L = pkg.scenes, pkg.layouts
    for scene in (S.cornell_box(aspect=1.5, glass_sphere=True), S.instanced_grid(n=6, aspect=1.5), S.sponza_scale(n_side=24, aspect=1.5)):
        gs = S.graphics_settings(48, 32, spp=2, bounces=5)
        a = oracle.render(scene, gs, accel_mode=0, want_f32=True, layouts=L)
        b = oracle.render(scene, gs, accel_mode=1, want_f32=True, layouts=L)
        assert a[1] == b[1]
        assert np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))
        for k in ("Position", "FlatNormal", "NormalRoughness", "Radiance"):
            assert np.array_equal(a[0][k], b[0][k])


def test_estimator_structure(oracle, pkg):
    S, L = pkg.scenes, pkg.layouts
    scene = S.cornell_box(aspect=1.0)
    W = H = 40
    # Bounces == 0: the path tracer is not dispatched, the image is the G-buffer radiance (App.cpp:1277)
    gb0, rays0, _ = oracle.render(scene, S.graphics_settings(W, H, spp=4, bounces=0), layouts=L)
    assert rays0 == W * H
    rad0 = gb0["Radiance"].view(np.float16).astype(np.float32)
    lit = rad0[..., 0] > 0
    assert lit.any() and np.all(rad0[lit][:, :3] == 15.0)       # only the emissive quad, strength 15
    # rays <= W*H*(1 + spp*Bounces); more bounces trace more rays; image stays finite and non-negative
    prev = 0
    for b in (1, 2, 4, 8):
        gb, rays, f32 = oracle.render(scene, S.graphics_settings(W, H, spp=2, bounces=b), want_f32=True, layouts=L)
        assert W * H < rays <= W * H * (1 + 2 * b) and rays > prev
        prev = rays
        assert np.isfinite(f32).all() and (f32 >= 0).all()
    # primary-miss pixels keep the environment colour written by the G-buffer pass (Raytracing.hlsl:241-252)
    open_scene = S.cornell_box(aspect=4.0)
    open_scene.camera = S.make_camera((0, 0, -4.0), hfov_deg=90.0, aspect=4.0)   # far back: rays left/right of the box miss
    gbm, _, f32m = oracle.render(open_scene, S.graphics_settings(64, 16, spp=1, bounces=2), want_f32=True, layouts=L)
    miss = ~np.isfinite(gbm["Position"][..., 3])
    assert miss.any() and np.all(gbm["Radiance"][miss] == 0) and np.all(f32m[miss] == 0)


def test_frame_index_and_jitter_change_the_image(oracle, pkg):
    S, L = pkg.scenes, pkg.layouts
    scene = S.cornell_box(aspect=1.0)
    a = oracle.render(scene, S.graphics_settings(32, 32, spp=1, bounces=4, frame_index=0), want_f32=True, layouts=L)[2]
    b = oracle.render(scene, S.graphics_settings(32, 32, spp=1, bounces=4, frame_index=1), want_f32=True, layouts=L)[2]
    assert not np.array_equal(a, b)
    a2 = oracle.render(scene, S.graphics_settings(32, 32, spp=1, bounces=4, frame_index=0), want_f32=True, layouts=L)[2]
    assert np.array_equal(a, a2)                                 # deterministic, thread-count independent
    a1t = oracle.render(scene, S.graphics_settings(32, 32, spp=1, bounces=4, frame_index=0), want_f32=True, layouts=L, threads=1)[2]
    assert np.array_equal(a, a1t)


def test_samples_are_sequential_per_pixel(oracle, pkg):
    """All SPP samples of a pixel draw from ONE RNG stream in order (Raytracing.hlsl:108,191): the 2-spp image
    is not the average of two independently seeded 1-spp images."""
    S, L = pkg.scenes, pkg.layouts
    scene = S.cornell_box(aspect=1.0)
    s2 = oracle.render(scene, S.graphics_settings(24, 24, spp=2, bounces=3, frame_index=0), want_f32=True, layouts=L)[2]
    s1 = oracle.render(scene, S.graphics_settings(24, 24, spp=1, bounces=3, frame_index=0), want_f32=True, layouts=L)[2]
    s1b = oracle.render(scene, S.graphics_settings(24, 24, spp=1, bounces=3, frame_index=1), want_f32=True, layouts=L)[2]
    assert not np.allclose(s2, 0.5 * (s1 + s1b))
