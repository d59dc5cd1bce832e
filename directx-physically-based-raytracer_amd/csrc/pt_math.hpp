// pt_math.hpp -- fp32 device arithmetic of the path tracer (gfx950).
//
// Two groups of functions:
//   * restatements of the reference's own HLSL helpers, cited file:line;
//   * "[MathLib]" functions: the reference calls NVIDIA-RTX/MathLib (ml.hlsli), an un-vendored
//     submodule that is absent from the reference tree. They are implemented here
//     from the published algorithms MathLib cites; DESIGN.md "Arithmetic spec" lists each one.
//
// Arithmetic rules: compiled with -ffp-contract=off, so the compiler fuses nothing; the
// only fused operations are the explicit mad() / __builtin_fmaf calls: sums of products (dot, matrix rows,
// cross, interpolation, polynomials) and the reference's `precise` mad code.
// Division and sqrt are IEEE correctly rounded (hipcc default). No libm transcendental is used on
// the Cornell-box path: sin/cos(2*pi*u) come from sincos_2pi() below.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pt {

struct v3 { float x, y, z; };

#define PT_DEV __device__ __forceinline__

PT_DEV v3 V3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_DEV v3 V3(const float* p) { return V3(p[0], p[1], p[2]); }
PT_DEV v3 operator+(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_DEV v3 operator-(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_DEV v3 operator*(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_DEV v3 operator*(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
PT_DEV v3 operator-(v3 a) { return V3(-a.x, -a.y, -a.z); }
PT_DEV v3 vabs(v3 a) { return V3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
// Sums of products (arithmetic spec): the first product is rounded, every further term is one fused multiply-add,
// terms taken left to right. mad(a, b, c) = a * b + c with one rounding.
PT_DEV float mad(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
PT_DEV float sop3(float a0, float b0, float a1, float b1, float a2, float b2) { return mad(a2, b2, mad(a1, b1, a0 * b0)); }                     // a0 b0 + a1 b1 + a2 b2
PT_DEV float sop3t(float a0, float b0, float a1, float b1, float a2, float b2, float t) { return mad(a2, b2, mad(a1, b1, mad(a0, b0, t))); }   // ... + t (affine row)
PT_DEV float dot(v3 a, v3 b) { return sop3(a.x, b.x, a.y, b.y, a.z, b.z); }
PT_DEV v3 cross(v3 a, v3 b) { return V3(mad(a.y, b.z, -(a.z * b.y)), mad(a.z, b.x, -(a.x * b.z)), mad(a.x, b.y, -(a.y * b.x))); }
PT_DEV v3 madd(v3 a, float s, v3 b) { return V3(mad(a.x, s, b.x), mad(a.y, s, b.y), mad(a.z, s, b.z)); }                                         // a * s + b
PT_DEV v3 madd(v3 a, v3 s, v3 b) { return V3(mad(a.x, s.x, b.x), mad(a.y, s.y, b.y), mad(a.z, s.z, b.z)); }                                      // a * s + b, per component
// Vertex::Interpolate (Vertex.hlsli:63-72): a0 + (a1 - a0) * u + (a2 - a0) * v
PT_DEV float interp1(float a0, float a1, float a2, float u, float v) { return mad(a2 - a0, v, mad(a1 - a0, u, a0)); }
PT_DEV v3 interp3(v3 a0, v3 a1, v3 a2, float u, float v) { return V3(interp1(a0.x, a1.x, a2.x, u, v), interp1(a0.y, a1.y, a2.y, u, v), interp1(a0.z, a1.z, a2.z, u, v)); }
PT_DEV v3 normalize(v3 v) { float inv = 1.0f / sqrtf(dot(v, v)); return v * inv; }
PT_DEV float saturate(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
PT_DEV float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
PT_DEV bool finite3(v3 v) { return isfinite(v.x) && isfinite(v.y) && isfinite(v.z); }
PT_DEV float comp(v3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

constexpr float kPi = 3.14159265358979323846f;
constexpr float kFltMin = 1.17549435e-38f;

// ---- [MathLib] scalar helpers -------------------------------------------------------------
PT_DEV float ml_sign(float x) { return x >= 0.0f ? 1.0f : -1.0f; }        // step(0,x)*2-1, never 0
PT_DEV float ml_sqrt01(float x) { return sqrtf(saturate(x)); }
PT_DEV float ml_positive_rcp(float x) { return 1.0f / fmaxf(x, kFltMin); }
PT_DEV float ml_pow5_01(float x) { x = saturate(x); float x2 = x * x; return x2 * x2 * x; }
PT_DEV float ml_luminance(v3 c) { return dot(c, V3(0.2990f, 0.5870f, 0.1140f)); }   // BT.601

// sin(2 pi u), cos(2 pi u): quadrant reduction + fixed-order Taylor polynomials (|err| < 1e-7).
PT_DEV void sincos_2pi(float u, float& s, float& c)
{
    float a = u * 4.0f;
    float k = floorf(a + 0.5f);
    float r = a - k;
    float x = r * 1.57079632679489662f;
    float x2 = x * x;
    float sp = 2.75573192e-6f;
    sp = mad(sp, x2, -1.98412698e-4f);
    sp = mad(sp, x2, 8.33333333e-3f);
    sp = mad(sp, x2, -1.66666667e-1f);
    sp = mad(sp, x2, 1.0f);
    sp = sp * x;
    float cp = -2.75573192e-7f;
    cp = mad(cp, x2, 2.48015873e-5f);
    cp = mad(cp, x2, -1.38888889e-3f);
    cp = mad(cp, x2, 4.16666667e-2f);
    cp = mad(cp, x2, -0.5f);
    cp = mad(cp, x2, 1.0f);
    int q = ((int)k) & 3;
    s = q == 0 ? sp : (q == 1 ? cp : (q == 2 ? -sp : -cp));
    c = q == 0 ? cp : (q == 1 ? -sp : (q == 2 ? -cp : sp));
}

// ---- [MathLib] Rng::Hash ------------------------------------------------------------------
PT_DEV uint32_t ml_hash(uint32_t x)
{
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
PT_DEV uint32_t rng_init(uint32_t px, uint32_t py, uint32_t frame)       // Raytracing.hlsl:108
{
    uint32_t seed = ml_hash(frame + 0x035F9F29u);
    uint32_t v = (px << 16) | (py & 0xFFFFu);
    return seed ^ (ml_hash(v) + 0x9E3779B9u + (seed << 6) + (seed >> 2));
}
PT_DEV float rng_float(uint32_t& st)
{
    st = st * 1664525u + 1013904223u;
    return (float)(ml_hash(st) >> 8) * (1.0f / 16777216.0f);
}

// ---- packing (DXGI typed-store conversion rules, D3D11.3 functional spec 3.2.3) -----------
PT_DEV uint16_t f32_to_f16(float f)          // round-to-nearest-even, overflow -> inf, NaN kept
{
    _Float16 h = (_Float16)f;                // v_cvt_f16_f32, RNE
    return __builtin_bit_cast(uint16_t, h);
}
PT_DEV float f16_to_f32(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
PT_DEV int16_t f32_to_snorm16(float f)
{
    if (!(f == f)) return 0;
    f = clampf(f, -1.0f, 1.0f) * 32767.0f;
    f = f >= 0.0f ? f + 0.5f : f - 0.5f;
    return (int16_t)(int32_t)f;
}
// x / c for a compile-time constant c, bit-identical to the IEEE fp32 division the arithmetic spec asks for, in 3 instructions
// (cvt, v_mul_f64, cvt) instead of the ~10 of the division expansion: the quotient of two 24-bit significands is never closer
// than 2^-49 (relative) to an fp32 rounding boundary, and x * RN64(1 / c) evaluated in double is within 2^-52 of it.
// tests/test_oracle_math.py checks the identity exhaustively for the integer inputs and on 10^8 random floats.
#define PT_DIV_CONST(x, c) ((float)((double)(x) * (1.0 / (double)(float)(c))))

PT_DEV float snorm16_to_f32(int16_t v) { return v == -32768 ? -1.0f : PT_DIV_CONST((float)v, 32767.0f); }
PT_DEV uint8_t f32_to_unorm8(float f)
{
    if (!(f == f)) return 0;
    f = saturate(f) * 255.0f + 0.5f;
    return (uint8_t)(int32_t)f;
}
PT_DEV float unorm8_to_f32(uint8_t v) { return PT_DIV_CONST((float)v, 255.0f); }
PT_DEV float unpack_r16_snorm(int16_t v) { return fmaxf(PT_DIV_CONST((float)v, 32767.0f), -1.0f); }   // Shaders/Packing.hlsli:8-11

// [MathLib] Packing::EncodeUnitVector / DecodeUnitVector (signed octahedral)
PT_DEV void oct_encode(v3 n, float& ex, float& ey)
{
    float s = fabsf(n.x) + fabsf(n.y) + fabsf(n.z);
    float x = n.x / s, y = n.y / s, z = n.z / s;
    if (!(z >= 0.0f)) {
        float wx = (1.0f - fabsf(y)) * ml_sign(x);
        float wy = (1.0f - fabsf(x)) * ml_sign(y);
        x = wx; y = wy;
    }
    ex = x; ey = y;
}
PT_DEV v3 oct_decode(float px, float py)
{
    v3 n = V3(px, py, 1.0f - fabsf(px) - fabsf(py));
    float t = saturate(-n.z);
    n.x -= t * ml_sign(n.x);
    n.y -= t * ml_sign(n.y);
    return normalize(n);
}

// ---- [MathLib] Geometry -------------------------------------------------------------------
struct basis3 { v3 T, B, N; };
PT_DEV basis3 ml_get_basis(v3 N)              // branchless ONB (Duff et al., JCGT 2017)
{
    float sz = ml_sign(N.z);
    float a = 1.0f / (sz + N.z);
    float ya = N.y * a;
    float b = N.x * ya;
    float c = N.x * sz;
    basis3 r;
    r.T = V3(c * N.x * a - 1.0f, sz * b, c);
    r.B = V3(b, N.y * ya - sz, N.y);
    r.N = N;
    return r;
}
PT_DEV v3 rotate_vector(const basis3& m, v3 v) { return V3(dot(m.T, v), dot(m.B, v), dot(m.N, v)); }
PT_DEV v3 rotate_vector_inv(const basis3& m, v3 v)
{
    return V3(sop3(m.T.x, v.x, m.B.x, v.y, m.N.x, v.z),
              sop3(m.T.y, v.x, m.B.y, v.y, m.N.y, v.z),
              sop3(m.T.z, v.x, m.B.z, v.y, m.N.z, v.z));
}

// ---- [MathLib] ImportanceSampling / BRDF ---------------------------------------------------
PT_DEV v3 ml_cosine_get_ray(float u0, float u1)
{
    float s, c; sincos_2pi(u0, s, c);
    float cosT = ml_sqrt01(u1);
    float sinT = ml_sqrt01(mad(-cosT, cosT, 1.0f));
    return V3(sinT * c, sinT * s, cosT);
}
PT_DEV float ml_cosine_pdf(float NoL) { return PT_DIV_CONST(NoL, kPi); }

PT_DEV float ml_distribution_ggx(float roughness, float NoH)
{
    float m = roughness * roughness;
    float m2 = m * m;
    float t = mad(mad(NoH, m2, -NoH), NoH, 1.0f);
    float a = m / t;
    return PT_DIV_CONST(a * a, kPi);
}
PT_DEV float ml_geometry_term_mod(float roughness, float NoL, float NoV)
{
    float m = roughness * roughness;
    float m2 = m * m;
    float a = NoL * ml_sqrt01(mad(mad(-m2, NoV, NoV), NoV, m2));
    float b = NoV * ml_sqrt01(mad(mad(-m2, NoL, NoL), NoL, m2));
    return 0.5f * ml_positive_rcp(a + b);
}
PT_DEV v3 ml_fresnel_schlick(v3 F0, float VoH)
{
    float f = ml_pow5_01(1.0f - VoH);
    return V3(mad(1.0f - F0.x, f, F0.x), mad(1.0f - F0.y, f, F0.y), mad(1.0f - F0.z, f, F0.z));
}
PT_DEV float ml_fresnel_dielectric(float eta, float VoN)
{
    float saSq = eta * eta * mad(-VoN, VoN, 1.0f);
    float ca = ml_sqrt01(1.0f - saSq);
    float Rs = mad(eta, VoN, -ca) * ml_positive_rcp(mad(eta, VoN, ca));
    float Rp = mad(eta, ca, -VoN) * ml_positive_rcp(mad(eta, ca, VoN));
    return 0.5f * mad(Rp, Rp, Rs * Rs);
}
PT_DEV float ml_diffuse_burley(float roughness, float NoL, float NoV, float VoH)
{
    float f = mad(2.0f * VoH * VoH, roughness, -0.5f);
    float FdV = mad(f, ml_pow5_01(1.0f - NoV), 1.0f);
    float FdL = mad(f, ml_pow5_01(1.0f - NoL), 1.0f);
    return PT_DIV_CONST(FdV * FdL, kPi);
}
PT_DEV v3 ml_env_term_rtg(v3 F0, float NoV, float roughness)     // RTG ch.32 rational fit
{
    float m = roughness * roughness;
    float X1 = NoV, X2 = NoV * NoV, X3 = NoV * X2;
    float Y1 = m, Y3 = m * (m * m);
    float b0 = mad(-1.28514f, X1, 0.99044f);
    float b1 = mad(-0.755907f, X1, 1.29678f);
    float bn = mad(b1, Y1, b0);
    float c0 = mad(59.4188f, X3, mad(2.92338f, X1, 1.0f));
    float c1 = mad(222.592f, X3, mad(-27.0302f, X1, 20.3225f));
    float c2 = mad(316.627f, X3, mad(626.13f, X1, 121.563f));
    float bd = mad(c2, Y3, mad(c1, Y1, c0));
    float bias = bn * ml_positive_rcp(bd);
    float s0 = mad(3.32707f, X1, 0.0365463f);
    float s1 = mad(-9.04756f, X1, 9.0632f);
    float sn = mad(s1, Y1, s0);
    float d0 = mad(-1.36772f, X3, mad(3.59685f, X2, 1.0f));
    float d1 = mad(9.22949f, X3, mad(-16.3174f, X2, 9.04401f));
    float d2 = mad(-20.2123f, X3, mad(19.7886f, X2, 5.56589f));
    float sd = mad(d2, Y3, mad(d1, Y1, d0));
    float scale = sn * ml_positive_rcp(sd);
    return V3(saturate(mad(F0.x, scale, bias)), saturate(mad(F0.y, scale, bias)), saturate(mad(F0.z, scale, bias)));
}
PT_DEV v3 ml_vndf_get_ray(float u0, float u1, float roughness, v3 Vl)     // Dupuy & Benyoub 2023
{
    float m = roughness * roughness;
    v3 Vh = normalize(V3(m * Vl.x, m * Vl.y, Vl.z));
    float s, c; sincos_2pi(u0, s, c);
    float z = mad(1.0f - u1, 1.0f + Vh.z, -Vh.z);
    float sinT = ml_sqrt01(mad(-z, z, 1.0f));
    v3 h = V3(mad(sinT, c, Vh.x), mad(sinT, s, Vh.y), z + Vh.z);
    return normalize(V3(m * h.x, m * h.y, fmaxf(h.z, 0.0f)));
}
PT_DEV float ml_vndf_pdf(v3 Vl, float NoH, float roughness)
{
    float m = roughness * roughness;
    float D = ml_distribution_ggx(roughness, NoH);
    float ax = m * Vl.x, ay = m * Vl.y;
    float len2 = mad(ay, ay, ax * ax);
    float t = sqrtf(mad(Vl.z, Vl.z, len2));
    if (Vl.z >= 0.0f) return D / (2.0f * (Vl.z + t));
    return D * (t - Vl.z) / (2.0f * len2);
}
PT_DEV float ml_from_srgb1(float x)           // procedural sky only; powf => not bit-pinned
{
    x = saturate(x);
    return x >= 0.04045f ? powf(x * (1.0f / 1.055f) + (0.055f / 1.055f), 2.4f) : x * (1.0f / 12.92f);
}

// ---- Shaders/SurfaceVectors.hlsli:5-16 ------------------------------------------------------
struct SurfaceVectors {
    v3 FrontGeometricNormal, ShadingNormal;
    basis3 ShadingBasis;
};
PT_DEV SurfaceVectors surface_vectors(bool isFront, v3 geometricNormal, v3 shadingNormal)
{
    SurfaceVectors sv;
    sv.FrontGeometricNormal = isFront ? geometricNormal : -geometricNormal;
    sv.ShadingNormal = shadingNormal;
    sv.ShadingBasis = ml_get_basis(shadingNormal);
    return sv;
}

// ---- Shaders/BxDF.hlsli ---------------------------------------------------------------------
constexpr float kMinRoughness = 2e-3f;        // BxDF.hlsli:19
enum : int { LOBE_DIFFUSE = 0, LOBE_SPECULAR = 1, LOBE_TRANSMISSION = 2 };
constexpr uint32_t kExtLambertianOnly = 0x1u;

struct BSDFSample {                           // BxDF.hlsli:36-44
    v3 BaseColor; float Metallic; v3 Albedo; float Roughness, IORi, IORo; v3 F0; float Transmission;

    PT_DEV void Initialize(v3 baseColor, float metallic, float roughness, float IOR, float transmission, bool isFrontFace)
    {                                         // BxDF.hlsli:45-67
        BaseColor = baseColor;
        Metallic = metallic;
        Albedo = baseColor * (1.0f - metallic);
        Roughness = fmaxf(kMinRoughness, roughness);
        IORi = 1.0f; IORo = IOR;
        if (!isFrontFace) { IORi = IOR; IORo = 1.0f; }
        float r = (IORi - IORo) / (IORi + IORo);
        float r2 = r * r;
        F0 = V3(mad(metallic, baseColor.x - r2, r2), mad(metallic, baseColor.y - r2, r2), mad(metallic, baseColor.z - r2, r2));
        Transmission = transmission;
    }

    PT_DEV float EstimateDiffuseProbability(float NoV) const   // BxDF.hlsli:21-34
    {
        v3 Fenv = ml_env_term_rtg(F0, NoV, Roughness);
        float diffuse = ml_luminance(Albedo * V3(1.0f - Fenv.x, 1.0f - Fenv.y, 1.0f - Fenv.z));
        float specular = ml_luminance(Fenv);
        float sum = diffuse + specular;
        float p = sum > 0.0f ? diffuse / sum : 1.0f;
        if (0.0f < p && p < 1.0f) return clampf(p, 0.05f, 0.95f);
        return p;
    }

    PT_DEV void ComputeLobeWeights(const SurfaceVectors& sv, v3 V, uint32_t ext, float w[3]) const   // BxDF.hlsli:184-196
    {
        if (ext & kExtLambertianOnly) { w[0] = 1.0f; w[1] = 0.0f; w[2] = 0.0f; return; }
        float NoV = fabsf(dot(sv.ShadingNormal, V));
        float tw = Transmission * (1.0f - Metallic);
        float rw = 1.0f - tw;
        float dw = EstimateDiffuseProbability(NoV);
        float sw = 1.0f - dw;
        w[LOBE_DIFFUSE] = dw * rw;
        w[LOBE_SPECULAR] = sw * rw;
        w[LOBE_TRANSMISSION] = tw;
    }

    PT_DEV static v3 reflect(v3 i, v3 n) { float d = dot(n, i); return madd(n, -(2.0f * d), i); }
    PT_DEV static v3 refract(v3 i, v3 n, float eta)
    {
        float d = dot(n, i);
        float k = mad(-(eta * eta), mad(-d, d, 1.0f), 1.0f);
        if (k < 0.0f) return V3(0.0f, 0.0f, 0.0f);
        float s = mad(eta, d, sqrtf(k));
        return madd(n, -s, i * eta);
    }

    // FindLobe :198-212 + Sample :214-226 (+ :81-86, :110-118, :148-168)
    PT_DEV bool Sample(const SurfaceVectors& sv, v3 V, const float w[3], const float rnd[4], v3& L, int& lobe) const
    {
        lobe = 0;
        {
            float weight = w[2];
            if (rnd[0] < weight) lobe = 2;
            else { weight += w[1]; if (rnd[0] < weight) lobe = 1; }
        }
        if (lobe == LOBE_DIFFUSE) {
            L = rotate_vector_inv(sv.ShadingBasis, ml_cosine_get_ray(rnd[1], rnd[2]));
            return dot(sv.FrontGeometricNormal, L) > 0.0f;
        }
        v3 Vlocal = rotate_vector(sv.ShadingBasis, V);
        v3 H = rotate_vector_inv(sv.ShadingBasis, ml_vndf_get_ray(rnd[1], rnd[2], Roughness, Vlocal));
        if (lobe == LOBE_SPECULAR) {
            L = reflect(-V, H);
            return dot(sv.FrontGeometricNormal, L) > 0.0f;
        }
        float VoH = fabsf(dot(V, H)), eta = IORi / IORo;
        if (eta * eta * (1.0f - VoH * VoH) > 1.0f || rnd[3] < ml_fresnel_dielectric(eta, VoH)) {
            L = reflect(-V, H);
        } else {
            L = refract(-V, H, eta);
            if (!finite3(L)) L = -V;
        }
        return true;
    }

    PT_DEV v3 ComputeHalfVector(const SurfaceVectors& sv, v3 L, v3 V, bool isTransmissive) const   // :228-245
    {
        v3 N = sv.FrontGeometricNormal, H;
        if (isTransmissive && dot(N, L) < 0.0f) {
            H = normalize(L * IORo + V * IORi);
            if (dot(N, H) < 0.0f) H = -H;
        } else {
            H = normalize(L + V);
        }
        return H;
    }

    // all-lobe EvaluatePDF :247-264 and Evaluate :266-285 (the RTXDI-facing overloads), built from the single-lobe forms
    PT_DEV void EvaluateAll(const SurfaceVectors& sv, v3 L, v3 V, const float w[3], float& pdf, v3& diffuse, v3& specular) const
    {
        const float tw = w[LOBE_TRANSMISSION];
        pdf = 0.0f; diffuse = V3(0.0f, 0.0f, 0.0f); specular = V3(0.0f, 0.0f, 0.0f);
        if (tw > 0.0f) { float p; v3 f; EvaluateLobe(sv, L, V, w, LOBE_TRANSMISSION, 0, p, f); pdf = p; specular = f; }
        if (tw < 1.0f && dot(sv.FrontGeometricNormal, L) > 0.0f) {
            float pd, ps; v3 fd, fs;
            EvaluateLobe(sv, L, V, w, LOBE_DIFFUSE, 0, pd, fd);
            EvaluateLobe(sv, L, V, w, LOBE_SPECULAR, 0, ps, fs);
            pdf += pd + ps;
            diffuse = fd;
            specular = specular + fs;
        }
    }

    // single-lobe EvaluatePDF :287-299 and Evaluate :301-315, fused (they share H and the dots)
    PT_DEV void EvaluateLobe(const SurfaceVectors& sv, v3 L, v3 V, const float w[3], int lobe, uint32_t ext,
                             float& pdf, v3& f) const
    {
        const float tw = w[LOBE_TRANSMISSION];
        const v3 H = ComputeHalfVector(sv, L, V, tw > 0.0f);
        const v3 N = sv.ShadingNormal;
        const float lw = w[lobe];
        const float NoL = fabsf(dot(N, L));
        if (lobe == LOBE_TRANSMISSION) {
            pdf = NoL * lw;
            f = (BaseColor * NoL) * tw;
            return;
        }
        const float rw = 1.0f - tw;
        pdf = 0.0f; f = V3(0.0f, 0.0f, 0.0f);
        if (!(dot(sv.FrontGeometricNormal, L) > 0.0f)) return;
        const float NoV = fabsf(dot(N, V)), VoH = fabsf(dot(V, H));
        if (lobe == LOBE_DIFFUSE) {
            pdf = ml_cosine_pdf(NoL) * lw;
            float dterm = (ext & kExtLambertianOnly) ? (1.0f / kPi) : ml_diffuse_burley(Roughness, NoL, NoV, VoH);
            f = ((Albedo * NoL) * dterm) * rw;
            return;
        }
        const float NoH = fabsf(dot(N, H));
        v3 Vlocal = rotate_vector(sv.ShadingBasis, V);
        pdf = ml_vndf_pdf(Vlocal, NoH, Roughness) * lw;
        float D = ml_distribution_ggx(Roughness, NoH);
        float G = ml_geometry_term_mod(Roughness, NoL, NoV);
        v3 F = ml_fresnel_schlick(F0, VoH);
        float k = NoL * D * G;
        f = (F * k) * rw;
    }
};

// ---- Shaders/SelfIntersectionAvoidance.hlsli:39-117, restated --------------------------------
// Method, operation order and error-bound constants: "Solving Self-Intersection Artifacts in DirectX Raytracing",
// Copyright (c) 2023 NVIDIA CORPORATION & AFFILIATES, BSD-3-Clause -- full notice in THIRD_PARTY_NOTICES.md at the repository root.
// M = objectToWorld 3x4, W = worldToObject 3x4 (row-major).
PT_DEV void safe_triangle_spawn_point(v3 v0, v3 v1, v3 v2, float bx, float by, const float* M, const float* W,
                                      v3& objPosition, v3& wldPosition, v3& wldNormal, float& wldOffsetOut)
{
    v3 e1 = v1 - v0, e2 = v2 - v0;
    v3 op = V3(v0.x + __builtin_fmaf(bx, e1.x, by * e2.x), v0.y + __builtin_fmaf(bx, e1.y, by * e2.y),
               v0.z + __builtin_fmaf(bx, e1.z, by * e2.z));
    v3 on = cross(e1, e2);
    v3 wp;
    wp.x = M[3]  + __builtin_fmaf(M[0], op.x, __builtin_fmaf(M[1], op.y, M[2]  * op.z));
    wp.y = M[7]  + __builtin_fmaf(M[4], op.x, __builtin_fmaf(M[5], op.y, M[6]  * op.z));
    wp.z = M[11] + __builtin_fmaf(M[8], op.x, __builtin_fmaf(M[9], op.y, M[10] * op.z));
    v3 wn = V3(W[0] * on.x + W[4] * on.y + W[8]  * on.z,
               W[1] * on.x + W[5] * on.y + W[9]  * on.z,
               W[2] * on.x + W[6] * on.y + W[10] * on.z);
    float wldScale = 1.0f / sqrtf(dot(wn, wn));
    wn = wn * wldScale;

    const float c0 = 5.9604644775390625E-8f;
    const float c1 = 1.788139769587360206060111522674560546875E-7f;
    const float c2 = 1.19209317972490680404007434844970703125E-7f;
    v3 ae1 = vabs(e1), ae2 = vabs(e2);
    v3 ext3 = (ae1 + ae2) + vabs(ae1 - ae2);
    float extent = fmaxf(fmaxf(ext3.x, ext3.y), ext3.z);
    v3 av0 = vabs(v0);
    float ce = c1 * extent;
    v3 objErr = V3(__builtin_fmaf(c0, av0.x, ce), __builtin_fmaf(c0, av0.y, ce), __builtin_fmaf(c0, av0.z, ce));
    v3 aop = vabs(op);
    v3 mo = V3(fabsf(M[0]) * aop.x + fabsf(M[1]) * aop.y + fabsf(M[2])  * aop.z,
               fabsf(M[4]) * aop.x + fabsf(M[5]) * aop.y + fabsf(M[6])  * aop.z,
               fabsf(M[8]) * aop.x + fabsf(M[9]) * aop.y + fabsf(M[10]) * aop.z);
    v3 wldErr = V3(__builtin_fmaf(c1, mo.x, c2 * fabsf(M[3])), __builtin_fmaf(c1, mo.y, c2 * fabsf(M[7])),
                   __builtin_fmaf(c1, mo.z, c2 * fabsf(M[11])));
    v3 awp = vabs(wp);
    v3 wo = V3(fabsf(W[0]) * awp.x + fabsf(W[1]) * awp.y + fabsf(W[2])  * awp.z + fabsf(W[3]),
               fabsf(W[4]) * awp.x + fabsf(W[5]) * awp.y + fabsf(W[6])  * awp.z + fabsf(W[7]),
               fabsf(W[8]) * awp.x + fabsf(W[9]) * awp.y + fabsf(W[10]) * awp.z + fabsf(W[11]));
    objErr = V3(__builtin_fmaf(c2, wo.x, objErr.x), __builtin_fmaf(c2, wo.y, objErr.y), __builtin_fmaf(c2, wo.z, objErr.z));
    float wldOffset = dot(wldErr, vabs(wn));
    float objOffset = dot(objErr, vabs(on));
    wldOffset = __builtin_fmaf(wldScale, objOffset, wldOffset);
    objPosition = op; wldPosition = wp; wldNormal = wn; wldOffsetOut = wldOffset;
}

// HitInfo::GetSafeWorldRayOrigin (Shaders/HitInfo.hlsli:96-99) + OffsetSpawnPoint (:113-117)
PT_DEV v3 safe_world_ray_origin(v3 position, v3 flatNormal, float offset, v3 dir)
{
    float s = ml_sign(dot(dir, flatNormal));
    v3 n = flatNormal * s;
    return V3(__builtin_fmaf(offset, n.x, position.x), __builtin_fmaf(offset, n.y, position.y), __builtin_fmaf(offset, n.z, position.z));
}

} // namespace pt
