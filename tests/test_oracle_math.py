"""CPU tests of the oracle's arithmetic (oracle/pt_oracle.c): analytic known answers and properties.

The reference ships no tests; these follow its section-4 consequence: BSDF PDF
normalisation / energy bounds, ray-triangle known answers, octahedral + SNORM round trips, RNG vectors.
"""
import ctypes as C
import math
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")
FP = C.POINTER(C.c_float)


def fa(*v):
    return np.array(v, np.float32)


def fp(a):
    return a.ctypes.data_as(FP)


def test_sincos_accuracy_and_golden(oracle):
    lib = oracle.lib()
    s, c = C.c_float(), C.c_float()
    worst = 0.0
    for u in np.linspace(0, 1, 4097, dtype=np.float32):
        lib.or_sincos_2pi(float(u), C.byref(s), C.byref(c))
        worst = max(worst, abs(s.value - math.sin(2 * math.pi * float(u))), abs(c.value - math.cos(2 * math.pi * float(u))))
    assert worst < 2.5e-7
    g = np.load(os.path.join(GOLD, "math_vectors.npz"))
    for u, (gs, gc) in zip(g["sincos_u"], g["sincos"]):
        lib.or_sincos_2pi(float(u), C.byref(s), C.byref(c))
        assert s.value == gs and c.value == gc          # bit-exact pin


def test_rng_golden_and_range(oracle):
    lib = oracle.lib()
    g = np.load(os.path.join(GOLD, "math_vectors.npz"))
    st = C.c_uint32(lib.or_rng_init(3, 5, 11))
    assert st.value == int(g["rng_seed"])
    got = np.array([lib.or_rng_float(C.byref(st)) for _ in range(16)], np.float32)
    assert np.array_equal(got, g["rng_floats"])
    # distinct pixels / frames give distinct streams; values in [0,1)
    seeds = {lib.or_rng_init(x, y, f) for x in range(16) for y in range(16) for f in range(4)}
    assert len(seeds) == 16 * 16 * 4
    st = C.c_uint32(lib.or_rng_init(100, 200, 0))
    v = np.array([lib.or_rng_float(C.byref(st)) for _ in range(20000)])
    assert v.min() >= 0.0 and v.max() < 1.0 and abs(v.mean() - 0.5) < 0.01


def test_f16_conversion_matches_ieee(oracle):
    lib = oracle.lib()
    rng = np.random.default_rng(1)
    vals = np.concatenate([rng.standard_normal(2000).astype(np.float32) * 10.0 ** rng.integers(-9, 6, 2000),
                           fa(0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e9, -1e9, 5.96e-8, 2.98e-8, 2.9802325e-08, 6.1e-5, np.inf, -np.inf)])
    for v in vals.astype(np.float32):
        h = lib.or_f32_to_f16(float(v))
        assert h == int(np.float32(v).astype(np.float16).view(np.uint16)), v
        assert lib.or_f16_to_f32(h) == float(np.uint16(h).view(np.float16).astype(np.float32)) or math.isnan(lib.or_f16_to_f32(h))
    assert lib.or_f32_to_f16(float("nan")) & 0x7C00 == 0x7C00


def test_snorm_unorm_rules(oracle):
    lib = oracle.lib()
    assert lib.or_f32_to_snorm16(1.0) == 32767 and lib.or_f32_to_snorm16(-1.0) == -32767
    assert lib.or_f32_to_snorm16(2.0) == 32767 and lib.or_f32_to_snorm16(float("nan")) == 0
    assert lib.or_snorm16_to_f32(-32768) == -1.0 and lib.or_snorm16_to_f32(32767) == 1.0
    assert lib.or_f32_to_unorm8(0.5) == 128 and lib.or_f32_to_unorm8(1.5) == 255 and lib.or_f32_to_unorm8(-1) == 0
    for q in range(-32767, 32768, 257):                       # decode -> encode is the identity
        assert lib.or_f32_to_snorm16(lib.or_snorm16_to_f32(q)) == q


def test_octahedral_roundtrip(oracle):
    lib = oracle.lib()
    rng = np.random.default_rng(2)
    n = rng.standard_normal((500, 3)).astype(np.float32)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    e, d = np.zeros(2, np.float32), np.zeros(3, np.float32)
    for v in list(n) + [fa(0, 0, 1), fa(0, 0, -1), fa(1, 0, 0), fa(0, -1, 0)]:
        v = np.ascontiguousarray(v)
        lib.or_oct_encode(fp(v), fp(e))
        assert np.all(np.abs(e) <= 1.0 + 1e-6)
        lib.or_oct_decode(fp(e), fp(d))
        assert np.allclose(d, v, atol=2e-6)


def test_ray_triangle_known_answers(oracle):
    lib = oracle.lib()
    t, u, v = C.c_float(), C.c_float(), C.c_float()
    v0, v1, v2 = fa(0, 0, 2), fa(1, 0, 2), fa(0, 1, 2)
    o, d = fa(0.25, 0.25, 0), fa(0, 0, 1)
    assert lib.or_ray_triangle(fp(o), fp(d), 0.0, 1e30, fp(v0), fp(v1), fp(v2), C.byref(t), C.byref(u), C.byref(v)) == 1
    assert t.value == 2.0 and u.value == 0.25 and v.value == 0.25      # u weights v1, v weights v2 (DXR)
    # no face culling: the flipped winding hits too
    assert lib.or_ray_triangle(fp(o), fp(d), 0.0, 1e30, fp(v0), fp(v2), fp(v1), C.byref(t), C.byref(u), C.byref(v)) == 1
    # exclusive ray extents (TMin < t < TMax)
    assert lib.or_ray_triangle(fp(o), fp(d), 2.0, 1e30, fp(v0), fp(v1), fp(v2), C.byref(t), C.byref(u), C.byref(v)) == 0
    assert lib.or_ray_triangle(fp(o), fp(d), 0.0, 2.0, fp(v0), fp(v1), fp(v2), C.byref(t), C.byref(u), C.byref(v)) == 0
    # behind the origin / outside / parallel
    assert lib.or_ray_triangle(fp(o), fp(fa(0, 0, -1)), 0.0, 1e30, fp(v0), fp(v1), fp(v2), C.byref(t), C.byref(u), C.byref(v)) == 0
    assert lib.or_ray_triangle(fp(fa(0.8, 0.8, 0)), fp(d), 0.0, 1e30, fp(v0), fp(v1), fp(v2), C.byref(t), C.byref(u), C.byref(v)) == 0
    assert lib.or_ray_triangle(fp(o), fp(fa(1, 0, 0)), 0.0, 1e30, fp(v0), fp(v1), fp(v2), C.byref(t), C.byref(u), C.byref(v)) == 0
    # watertight shared edge: a ray through the common edge of two triangles hits at least one of them
    a, b, c_, dd = fa(0, 0, 1), fa(1, 0, 1), fa(1, 1, 1), fa(0, 1, 1)
    rng = np.random.default_rng(3)
    for _ in range(300):
        s = np.float32(rng.random())
        p = (a + (c_ - a) * s).astype(np.float32)             # point on the diagonal a-c
        org = fa(rng.random() * 2 - 0.5, rng.random() * 2 - 0.5, -1)
        dr = (p - org).astype(np.float32)
        h1 = lib.or_ray_triangle(fp(org), fp(dr), 0.0, 1e30, fp(a), fp(b), fp(c_), C.byref(t), C.byref(u), C.byref(v))
        h2 = lib.or_ray_triangle(fp(org), fp(dr), 0.0, 1e30, fp(a), fp(c_), fp(dd), C.byref(t), C.byref(u), C.byref(v))
        assert h1 or h2


def _bsdf(oracle, mat, front, ng, ns, V, rnd, ext=0):
    lib = oracle.lib()
    L_, f, w = np.zeros(3, np.float32), np.zeros(3, np.float32), np.zeros(3, np.float32)
    lobe, pdf = C.c_int(), C.c_float()
    ok = lib.or_bsdf_sample(fp(fa(*mat)), front, fp(fa(*ng)), fp(fa(*ns)), fp(fa(*V)), fp(fa(*rnd)), ext,
                            fp(L_), C.byref(lobe), C.byref(pdf), fp(f), fp(w))
    return ok, L_.copy(), lobe.value, pdf.value, f.copy(), w.copy()


def test_lobe_weights_and_selection(oracle):
    # default dielectric wall: F0 = 0.04 -> both reflection lobes present, weights sum to 1 (BxDF.hlsli:184-196)
    ok, L_, lobe, pdf, f, w = _bsdf(oracle, (0.73, 0.73, 0.73, 0, 0.5, 1.5, 0), 1, (0, 0, 1), (0, 0, 1), (0, 0.6, 0.8), (0.99, 0.3, 0.3, 0.5))
    assert abs(w.sum() - 1) < 1e-6 and 0.05 <= w[1] <= 0.95 and w[2] == 0
    assert lobe == 0                                            # u >= w[T]+w[S] -> diffuse (FindLobe :198-212)
    ok, L_, lobe, *_ = _bsdf(oracle, (0.73, 0.73, 0.73, 0, 0.5, 1.5, 0), 1, (0, 0, 1), (0, 0, 1), (0, 0.6, 0.8), (0.001, 0.3, 0.3, 0.5))
    assert lobe == 1
    # pure metal: diffuse probability 0 is not clamped (only values strictly inside (0,1) are, :29-33)
    *_, w = _bsdf(oracle, (0.9, 0.9, 0.9, 1.0, 0.3, 1.5, 0), 1, (0, 0, 1), (0, 0, 1), (0, 0.6, 0.8), (0.5, 0.3, 0.3, 0.5))
    assert w[0] == 0 and w[1] == 1
    # transmission weight = Transmission * (1 - Metallic), picked first
    ok, L_, lobe, pdf, f, w = _bsdf(oracle, (1, 1, 1, 0, 0.05, 1.5, 1.0), 1, (0, 0, 1), (0, 0, 1), (0, 0.6, 0.8), (0.5, 0.3, 0.3, 0.9))
    assert w[2] == 1 and lobe == 2 and ok == 1
    # Lambertian-only switch (config C1)
    ok, L_, lobe, pdf, f, w = _bsdf(oracle, (0.5, 0.5, 0.5, 0, 0.5, 1.5, 0), 1, (0, 0, 1), (0, 0, 1), (0, 0, 1), (0.01, 0.3, 0.3, 0.5), ext=1)
    assert tuple(w) == (1, 0, 0) and lobe == 0
    assert np.allclose(f / pdf, 0.5, rtol=1e-5)                # f/pdf = albedo for cosine sampling with 1/pi


def test_transmission_throughput_is_base_color(oracle):
    # pdf = NoL, f = NoL * BaseColor -> throughput factor exactly BaseColor
    rng = np.random.default_rng(4)
    for _ in range(50):
        rnd = tuple(rng.random(4).astype(np.float32))
        ok, L_, lobe, pdf, f, w = _bsdf(oracle, (0.9, 0.8, 0.7, 0, 0.1, 1.5, 1.0), 1, (0, 0, 1), (0, 0, 1), (0.3, 0.1, 0.9486833), rnd)
        assert ok == 1 and lobe == 2 and pdf > 0
        assert np.allclose(f / pdf, (0.9, 0.8, 0.7), rtol=1e-5)
        assert abs(np.linalg.norm(L_) - 1) < 1e-4


def test_diffuse_pdf_integrates_to_one_and_energy_bound(oracle):
    # E[ 1/pdf * pdf ] over the hemisphere via uniform sampling of the cosine lobe's own estimator:
    # the estimator f/pdf must stay bounded (<= ~1.06 for Burley) and its mean is the albedo-weighted reflectance.
    rng = np.random.default_rng(5)
    vals = []
    for _ in range(2000):
        rnd = (0.999, rng.random(), rng.random(), 0.5)         # force the diffuse lobe
        ok, L_, lobe, pdf, f, w = _bsdf(oracle, (1, 1, 1, 0, 0.5, 1.5, 0), 1, (0, 0, 1), (0, 0, 1), (0, 0.6, 0.8), rnd)
        assert lobe == 0
        if ok and pdf > 0:
            assert L_[2] > 0 and abs(pdf / w[0] - L_[2] / math.pi) < 1e-5     # cosine pdf
            vals.append((f / pdf)[0] * w[0])                    # undo the lobe-probability weighting -> Burley * pi
    m = float(np.mean(vals))
    assert 0.7 < m < 1.15


def test_specular_vndf_weight_bounded(oracle):
    # with VNDF sampling f/pdf = F * G2/G1 <= 1 per channel (times lobe-weight ratio 1/w_s)
    rng = np.random.default_rng(6)
    for rough in (0.05, 0.3, 0.8):
        for _ in range(300):
            rnd = (0.0, rng.random(), rng.random(), 0.5)       # force specular on a metal
            th = rng.random() * 1.4
            V = (math.sin(th), 0.0, math.cos(th))
            ok, L_, lobe, pdf, f, w = _bsdf(oracle, (1, 1, 1, 1.0, rough, 1.5, 0), 1, (0, 0, 1), (0, 0, 1), V, rnd)
            assert lobe == 1
            if ok and pdf > 0:
                assert np.all(f / pdf * w[1] <= 1.0 + 1e-3) and np.all(f >= 0)


def test_env_term_limits(oracle):
    lib = oracle.lib()
    out = np.zeros(3, np.float32)
    lib.or_env_term_rtg(fp(fa(1, 1, 1)), 1.0, 0.0, fp(out))   # perfect mirror, normal incidence: ~1
    assert np.all(out > 0.95)
    lib.or_env_term_rtg(fp(fa(0.04, 0.04, 0.04)), 1.0, 0.5, fp(out))
    assert np.all((out > 0.02) & (out < 0.1))
    lib.or_env_term_rtg(fp(fa(0.04, 0.04, 0.04)), 0.05, 0.5, fp(out))     # grazing: Fresnel lifts it
    assert np.all(out > 0.1)


def test_safe_spawn_point(oracle):
    lib = oracle.lib()
    v = fa(0, 0, 0, 1, 0, 0, 0, 1, 0)
    o2w = fa(2, 0, 0, 5, 0, 2, 0, -1, 0, 0, 2, 3)              # scale 2 + translate
    w2o = np.zeros(12, np.float32)
    lib.or_invert_3x4(fp(o2w), fp(w2o))
    op, wp, on, wn = (np.zeros(3, np.float32) for _ in range(4))
    off = C.c_float()
    lib.or_safe_spawn(fp(v), fp(fa(0.25, 0.5)), fp(o2w), fp(w2o), fp(op), fp(wp), fp(on), fp(wn), C.byref(off))
    assert np.allclose(op, (0.25, 0.5, 0)) and np.allclose(wp, (5.5, 0, 3))
    assert np.allclose(on, (0, 0, 1)) and np.allclose(wn, (0, 0, 1))
    assert 0 < off.value < 1e-5                                  # a few ulps of the coordinates involved
    # the offset origin really leaves the surface: re-intersecting from it along +n misses, along -n hits
    t, u, vv = C.c_float(), C.c_float(), C.c_float()
    org = (wp + wn * off.value).astype(np.float32)
    wv = [fa(5, -1, 3), fa(7, -1, 3), fa(5, 1, 3)]
    assert lib.or_ray_triangle(fp(org), fp(fa(0, 0, 1)), 0.0, 1e30, fp(wv[0]), fp(wv[1]), fp(wv[2]), C.byref(t), C.byref(u), C.byref(vv)) == 0
    assert lib.or_ray_triangle(fp(org), fp(fa(0, 0, -1)), 0.0, 1e30, fp(wv[0]), fp(wv[1]), fp(wv[2]), C.byref(t), C.byref(u), C.byref(vv)) == 1


def test_invert_3x4(oracle, pkg):
    lib = oracle.lib()
    m = pkg.scenes.trs((-0.35, -0.4, 0.35), -18.0, (0.6, 1.2, 0.6), pitch_deg=7.0).reshape(-1).copy()
    out = np.zeros(12, np.float32)
    lib.or_invert_3x4(fp(m), fp(out))
    M = np.vstack([m.reshape(3, 4), [0, 0, 0, 1]]).astype(np.float64)
    W = np.vstack([out.reshape(3, 4), [0, 0, 0, 1]]).astype(np.float64)
    assert np.allclose(M @ W, np.eye(4), atol=1e-6)


def test_division_by_a_constant_through_double_is_the_ieee_fp32_quotient():
    """csrc/pt_math.hpp PT_DIV_CONST: the kernels evaluate x / c for compile-time constants c (32767, 255, pi) as
    (float)((double)x * (1.0 / (double)c)) -- 3 instructions instead of the ~10 of the fp32 division expansion. The
    spec (and the oracle) say IEEE fp32 division; the two agree for every input: exhaustively for the integer cases the
    texture / vertex decoders produce, and on 10^8 random floats (any sign, any exponent, denormal quotients) for pi."""
    def mismatches(x, c):
        x = x.astype(np.float32); c32 = np.float32(c)
        with np.errstate(all="ignore"):
            a = (x / c32).astype(np.float32)
            b = (x.astype(np.float64) * (np.float64(1.0) / np.float64(c32))).astype(np.float32)
        return int(((a.view(np.uint32) != b.view(np.uint32)) & ~(np.isnan(a) & np.isnan(b))).sum())
    assert mismatches(np.arange(-32768, 32768), 32767.0) == 0          # SNORM16 decode (vertex normals, G-buffer normals)
    assert mismatches(np.arange(0, 256), 255.0) == 0                   # UNORM8 decode
    rng = np.random.default_rng(1)
    pi32 = np.float32(3.14159265358979323846)
    for _ in range(10):
        bits = rng.integers(0, 2 ** 32, size=5_000_000, dtype=np.uint64).astype(np.uint32)
        x = bits.view(np.float32)
        assert mismatches(x[np.isfinite(x)], pi32) == 0
    assert mismatches(rng.random(50_000_000).astype(np.float32) * 4, pi32) == 0
    assert mismatches((rng.random(2_000_000) * 1e-37).astype(np.float32), pi32) == 0     # denormal quotients
