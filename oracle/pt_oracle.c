/*
 * pt_oracle.c -- CPU ORACLE (test infrastructure, NOT product code). See pt_oracle.h.
 *
 * Scalar fp32 restatement of the reference path. Build: gcc -O2 -ffp-contract=off
 * (no implicit FMA; every fused multiply-add below is an explicit fmaf()).
 * Arithmetic conventions ("the spec", shared in WORDS with the HIP kernels, not in code):
 *   - sums of products: first product rounded, every further term one fmaf, left to right:
 *     dot(a,b) = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x)); matrix rows alike (affine rows: translation innermost);
 *     cross, Vertex::Interpolate, bilinear filtering and polynomials as DESIGN.md section 1 lists them
 *   - normalize(v) = v * (1/sqrtf(dot(v,v))); rsqrt(x) = 1/sqrtf(x); all divisions IEEE
 *   - HLSL mad() inside `precise` code = fmaf(); every other scalar expression is mul then add
 *   - sin/cos of 2*pi*u come from or_sincos_2pi() (explicit polynomial), never libm
 */
#include "pt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ======================================================================== */
/* small vector helpers                                                      */
/* ======================================================================== */
typedef struct { float x, y, z; } f3;

static inline f3 F3(float x, float y, float z) { f3 r = { x, y, z }; return r; }
static inline f3 ld3(const float* p) { return F3(p[0], p[1], p[2]); }
static inline f3 add3(f3 a, f3 b) { return F3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 sub3(f3 a, f3 b) { return F3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 mul3(f3 a, f3 b) { return F3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline f3 scl3(f3 a, float s) { return F3(a.x * s, a.y * s, a.z * s); }
static inline f3 neg3(f3 a) { return F3(-a.x, -a.y, -a.z); }
static inline f3 abs3(f3 a) { return F3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
/* Sums of products (arithmetic spec): the first product is rounded, every further term is one fused multiply-add (fmaf is exact
 * in C), terms taken left to right. */
static inline float mad(float a, float b, float c) { return fmaf(a, b, c); }
static inline float sop3(float a0, float b0, float a1, float b1, float a2, float b2) { return mad(a2, b2, mad(a1, b1, a0 * b0)); }
static inline float sop3t(float a0, float b0, float a1, float b1, float a2, float b2, float t) { return mad(a2, b2, mad(a1, b1, mad(a0, b0, t))); }
static inline float dot3(f3 a, f3 b) { return sop3(a.x, b.x, a.y, b.y, a.z, b.z); }
static inline f3 cross3(f3 a, f3 b)
{
    return F3(mad(a.y, b.z, -(a.z * b.y)), mad(a.z, b.x, -(a.x * b.z)), mad(a.x, b.y, -(a.y * b.x)));
}
static inline f3 madd3(f3 a, float s, f3 b) { return F3(mad(a.x, s, b.x), mad(a.y, s, b.y), mad(a.z, s, b.z)); }   /* a * s + b */
/* Vertex::Interpolate (Vertex.hlsli:63-72): a0 + (a1 - a0) * u + (a2 - a0) * v */
static inline float interp1(float a0, float a1, float a2, float u, float v) { return mad(a2 - a0, v, mad(a1 - a0, u, a0)); }
static inline f3 interp3(f3 a0, f3 a1, f3 a2, float u, float v) { return F3(interp1(a0.x, a1.x, a2.x, u, v), interp1(a0.y, a1.y, a2.y, u, v), interp1(a0.z, a1.z, a2.z, u, v)); }
static inline f3 normalize3(f3 v) { float inv = 1.0f / sqrtf(dot3(v, v)); return scl3(v, inv); }
static inline float saturatef(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
static inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
static inline int finite3(f3 v) { return isfinite(v.x) && isfinite(v.y) && isfinite(v.z); }
static inline float get3(f3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

/* ======================================================================== */
/* [MathLib spec] scalar helpers                                             */
/* ======================================================================== */
/* Math::Sign = step(0,x)*2-1: never 0, NaN -> -1 */
static inline float ml_sign(float x) { return x >= 0.0f ? 1.0f : -1.0f; }
static inline float ml_sqrt01(float x) { return sqrtf(saturatef(x)); }
static inline float ml_positive_rcp(float x) { return 1.0f / fmaxf(x, FLT_MIN); }
static inline float ml_pow5_01(float x) { x = saturatef(x); float x2 = x * x; return x2 * x2 * x; }
/* Color::Luminance, BT.601 weights (MathLib default) */
static inline float ml_luminance(f3 c) { return dot3(c, F3(0.2990f, 0.5870f, 0.1140f)); }

/* sin(2*pi*u), cos(2*pi*u) by quadrant reduction + Taylor polynomials evaluated in a
 * fixed mul/add order so CPU and GPU agree bit for bit. |abs err| < 1e-7 on [0,1]. */
void or_sincos_2pi(float u, float* s, float* c)
{
    float a = u * 4.0f;
    float k = floorf(a + 0.5f);
    float r = a - k;                         /* [-0.5, 0.5] */
    float x = r * 1.57079632679489662f;      /* * pi/2 */
    float x2 = x * x;
    float sp = 2.75573192e-6f;               /* 1/9! */
    sp = mad(sp, x2, -1.98412698e-4f);          /* -1/7! */
    sp = mad(sp, x2, 8.33333333e-3f);           /* 1/5! */
    sp = mad(sp, x2, -1.66666667e-1f);          /* -1/3! */
    sp = mad(sp, x2, 1.0f);
    sp = sp * x;
    float cp = -2.75573192e-7f;              /* -1/10! */
    cp = mad(cp, x2, 2.48015873e-5f);           /* 1/8! */
    cp = mad(cp, x2, -1.38888889e-3f);          /* -1/6! */
    cp = mad(cp, x2, 4.16666667e-2f);           /* 1/4! */
    cp = mad(cp, x2, -0.5f);
    cp = mad(cp, x2, 1.0f);
    int q = ((int)k) & 3;
    if (q == 0) { *s = sp; *c = cp; }
    else if (q == 1) { *s = cp; *c = -sp; }
    else if (q == 2) { *s = -sp; *c = -cp; }
    else { *s = -cp; *c = sp; }
}

/* ---- [MathLib spec] Rng::Hash ------------------------------------------ */
/* lowbias32 integer hash + boost-style combine for the seed; LCG state advance with a
 * lowbias32 output permutation; float = top 24 bits / 2^24 in [0,1).
 * Draw ORDER on the path is the reference's (Raytracing.hlsl:108,330,351). */
static inline uint32_t ml_hash(uint32_t x)
{
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
static inline uint32_t ml_hash_combine(uint32_t seed, uint32_t v)
{
    return seed ^ (ml_hash(v) + 0x9E3779B9u + (seed << 6) + (seed >> 2));
}
uint32_t or_rng_init(uint32_t px, uint32_t py, uint32_t frame)
{
    return ml_hash_combine(ml_hash(frame + 0x035F9F29u), (px << 16) | (py & 0xFFFFu));
}
static inline uint32_t rng_uint(uint32_t* st)
{
    *st = *st * 1664525u + 1013904223u;
    return ml_hash(*st);
}
float or_rng_float(uint32_t* st) { return (float)(rng_uint(st) >> 8) * (1.0f / 16777216.0f); }

/* ---- packing ------------------------------------------------------------ */
/* IEEE binary32 -> binary16, round-to-nearest-even, overflow -> inf (D3D typed-UAV store rule) */
uint16_t or_f32_to_f16(float f)
{
    uint32_t x; memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7FFFFFFFu;
    if (ax >= 0x7F800000u) return (uint16_t)(sign | (ax > 0x7F800000u ? 0x7E00u : 0x7C00u));
    if (ax >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);             /* rounds to >= 65520 -> inf */
    if (ax < 0x33000001u) return (uint16_t)sign;                          /* <= 2^-25 -> 0 */
    int32_t e = (int32_t)(ax >> 23) - 127;
    uint32_t m = (ax & 0x7FFFFFu) | 0x800000u;
    uint32_t h;
    if (e < -14) {                                                        /* subnormal half */
        uint32_t shift = (uint32_t)(-14 - e) + 13u;                       /* 14..24 */
        uint32_t q = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (q & 1u))) q++;
        h = q;
    } else {
        uint32_t q = m >> 13, rem = m & 0x1FFFu;
        if (rem > 0x1000u || (rem == 0x1000u && (q & 1u))) q++;
        h = ((uint32_t)(e + 15) << 10) + (q - 0x400u);                    /* carry propagates into exponent */
    }
    return (uint16_t)(sign | h);
}
float or_f16_to_f32(uint16_t h)
{
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16, e = (h >> 10) & 0x1Fu, m = h & 0x3FFu, x;
    if (e == 0) {
        if (m == 0) x = sign;
        else { float v = (float)m * 5.9604644775390625e-8f; memcpy(&x, &v, 4); x |= sign; }
    } else if (e == 31) x = sign | 0x7F800000u | (m << 13);
    else x = sign | ((e + 112u) << 23) | (m << 13);
    float f; memcpy(&f, &x, 4); return f;
}
/* D3D11.3 functional spec 3.2.3.x FLOAT -> SNORM/UNORM: NaN -> 0, clamp, scale, +-0.5, truncate */
int16_t or_f32_to_snorm16(float f)
{
    if (!(f == f)) return 0;
    f = clampf(f, -1.0f, 1.0f) * 32767.0f;
    f = f >= 0.0f ? f + 0.5f : f - 0.5f;
    return (int16_t)(int32_t)f;
}
float or_snorm16_to_f32(int16_t v) { return v == -32768 ? -1.0f : (float)v / 32767.0f; }
uint8_t or_f32_to_unorm8(float f)
{
    if (!(f == f)) return 0;
    f = saturatef(f) * 255.0f + 0.5f;
    return (uint8_t)(int32_t)f;
}
static inline float unorm8_to_f32(uint8_t v) { return (float)v / 255.0f; }

/* Shaders/Packing.hlsli:8-11 (vertex attribute decode) */
static inline float unpack_r16_snorm(int16_t v) { return fmaxf((float)v / 32767.0f, -1.0f); }

/* [MathLib spec] Packing::EncodeUnitVector / DecodeUnitVector, signed octahedral */
void or_oct_encode(const float n[3], float out[2])
{
    float s = fabsf(n[0]) + fabsf(n[1]) + fabsf(n[2]);
    float x = n[0] / s, y = n[1] / s, z = n[2] / s;
    if (!(z >= 0.0f)) {
        float wx = (1.0f - fabsf(y)) * ml_sign(x);
        float wy = (1.0f - fabsf(x)) * ml_sign(y);
        x = wx; y = wy;
    }
    out[0] = x; out[1] = y;
}
void or_oct_decode(const float p[2], float out[3])
{
    f3 n = F3(p[0], p[1], 1.0f - fabsf(p[0]) - fabsf(p[1]));
    float t = saturatef(-n.z);
    n.x -= t * ml_sign(n.x);
    n.y -= t * ml_sign(n.y);
    n = normalize3(n);
    out[0] = n.x; out[1] = n.y; out[2] = n.z;
}

/* ======================================================================== */
/* [MathLib spec] BRDF / ImportanceSampling / Geometry                       */
/* ======================================================================== */
typedef struct { f3 T, B, N; } basis3;      /* rows of Geometry::GetBasis */

/* Geometry::GetBasis: branchless ONB ("Building an Orthonormal Basis, Revisited", JCGT 2017),
 * T = (-1,0,0), B = (0,-1,0) for N = +Z */
static basis3 ml_get_basis(f3 N)
{
    float sz = ml_sign(N.z);
    float a = 1.0f / (sz + N.z);
    float ya = N.y * a;
    float b = N.x * ya;
    float c = N.x * sz;
    basis3 r;
    r.T = F3(c * N.x * a - 1.0f, sz * b, c);
    r.B = F3(b, N.y * ya - sz, N.y);
    r.N = N;
    return r;
}
/* Geometry::RotateVector(m, v) = mul(m, v) ; RotateVectorInverse(m, v) = mul(transpose(m), v) */
static inline f3 rotate_vector(basis3 m, f3 v) { return F3(dot3(m.T, v), dot3(m.B, v), dot3(m.N, v)); }
static inline f3 rotate_vector_inv(basis3 m, f3 v)
{
    return F3(sop3(m.T.x, v.x, m.B.x, v.y, m.N.x, v.z),
              sop3(m.T.y, v.x, m.B.y, v.y, m.N.y, v.z),
              sop3(m.T.z, v.x, m.B.z, v.y, m.N.z, v.z));
}

/* ImportanceSampling::Cosine::GetRay / GetPDF */
static f3 ml_cosine_get_ray(float u0, float u1)
{
    float s, c; or_sincos_2pi(u0, &s, &c);
    float cosT = ml_sqrt01(u1);
    float sinT = ml_sqrt01(mad(-cosT, cosT, 1.0f));
    return F3(sinT * c, sinT * s, cosT);
}
static inline float ml_cosine_pdf(float NoL) { return NoL / 3.14159265358979323846f; }

/* BRDF::DistributionTerm: GGX, alpha = roughness^2 */
static float ml_distribution_ggx(float roughness, float NoH)
{
    float m = roughness * roughness;
    float m2 = m * m;
    float t = mad(mad(NoH, m2, -NoH), NoH, 1.0f);
    float a = m / t;
    return a * a / 3.14159265358979323846f;
}
/* BRDF::GeometryTermMod: height-correlated Smith, = G / (4 NoL NoV) */
static float ml_geometry_term_mod(float roughness, float NoL, float NoV)
{
    float m = roughness * roughness;
    float m2 = m * m;
    float a = NoL * ml_sqrt01(mad(mad(-m2, NoV, NoV), NoV, m2));
    float b = NoV * ml_sqrt01(mad(mad(-m2, NoL, NoL), NoL, m2));
    return 0.5f * ml_positive_rcp(a + b);
}
/* BRDF::FresnelTerm: Schlick */
static f3 ml_fresnel_schlick(f3 F0, float VoH)
{
    float f = ml_pow5_01(1.0f - VoH);
    return F3(mad(1.0f - F0.x, f, F0.x), mad(1.0f - F0.y, f, F0.y), mad(1.0f - F0.z, f, F0.z));
}
/* BRDF::FresnelTerm_Dielectric: exact unpolarised Fresnel, eta = n_i / n_t */
static float ml_fresnel_dielectric(float eta, float VoN)
{
    float saSq = eta * eta * mad(-VoN, VoN, 1.0f);
    float ca = ml_sqrt01(1.0f - saSq);
    float Rs = mad(eta, VoN, -ca) * ml_positive_rcp(mad(eta, VoN, ca));
    float Rp = mad(eta, ca, -VoN) * ml_positive_rcp(mad(eta, ca, VoN));
    return 0.5f * mad(Rp, Rp, Rs * Rs);
}
/* BRDF::DiffuseTerm: Burley / Disney */
static float ml_diffuse_burley(float roughness, float NoL, float NoV, float VoH)
{
    float f = mad(2.0f * VoH * VoH, roughness, -0.5f);
    float FdV = mad(f, ml_pow5_01(1.0f - NoV), 1.0f);
    float FdL = mad(f, ml_pow5_01(1.0f - NoL), 1.0f);
    return FdV * FdL / 3.14159265358979323846f;
}
/* BRDF::EnvironmentTerm_Rtg: rational fit of the split-sum integral, Ray Tracing Gems ch. 32 */
void or_env_term_rtg(const float f0[3], float NoV, float roughness, float out[3])
{
    float m = roughness * roughness;
    float X1 = NoV, X2 = NoV * NoV, X3 = NoV * X2;
    float Y1 = m, Y2 = m * m, Y3 = m * Y2;
    /* bias = dot(M1*X.xy, Y.xy) / dot(M2*X.xyw, Y.xyw) */
    float b0 = mad(-1.28514f, X1, 0.99044f);
    float b1 = mad(-0.755907f, X1, 1.29678f);
    float bn = mad(b1, Y1, b0);
    float c0 = mad(59.4188f, X3, mad(2.92338f, X1, 1.0f));
    float c1 = mad(222.592f, X3, mad(-27.0302f, X1, 20.3225f));
    float c2 = mad(316.627f, X3, mad(626.13f, X1, 121.563f));
    float bd = mad(c2, Y3, mad(c1, Y1, c0));
    float bias = bn * ml_positive_rcp(bd);
    /* scale = dot(M3*X.xy, Y.xy) / dot(M4*X.xzw, Y.xyw) */
    float s0 = mad(3.32707f, X1, 0.0365463f);
    float s1 = mad(-9.04756f, X1, 9.0632f);
    float sn = mad(s1, Y1, s0);
    float d0 = mad(-1.36772f, X3, mad(3.59685f, X2, 1.0f));
    float d1 = mad(9.22949f, X3, mad(-16.3174f, X2, 9.04401f));
    float d2 = mad(-20.2123f, X3, mad(19.7886f, X2, 5.56589f));
    float sd = mad(d2, Y3, mad(d1, Y1, d0));
    float scale = sn * ml_positive_rcp(sd);
    (void)Y2;
    for (int i = 0; i < 3; i++) out[i] = saturatef(mad(f0[i], scale, bias));
}

/* ImportanceSampling::VNDF::GetRay: spherical-cap VNDF sampling (Dupuy & Benyoub 2023), returns local H */
static f3 ml_vndf_get_ray(float u0, float u1, float roughness, f3 Vl)
{
    float m = roughness * roughness;
    f3 Vh = normalize3(F3(m * Vl.x, m * Vl.y, Vl.z));
    float s, c; or_sincos_2pi(u0, &s, &c);
    float z = mad(1.0f - u1, 1.0f + Vh.z, -Vh.z);
    float sinT = ml_sqrt01(mad(-z, z, 1.0f));
    f3 h = F3(mad(sinT, c, Vh.x), mad(sinT, s, Vh.y), z + Vh.z);
    return normalize3(F3(m * h.x, m * h.y, fmaxf(h.z, 0.0f)));
}
/* ImportanceSampling::VNDF::GetPDF(Vlocal, NoH, roughness): pdf of L = G1(V) D(H) / (4 NoV)
 * = D / (2 (Vz + sqrt(a^2 (Vx^2+Vy^2) + Vz^2))), with the numerically stable form for Vz < 0 */
static float ml_vndf_pdf(f3 Vl, float NoH, float roughness)
{
    float m = roughness * roughness;
    float D = ml_distribution_ggx(roughness, NoH);
    float ax = m * Vl.x, ay = m * Vl.y;
    float len2 = mad(ay, ay, ax * ax);
    float t = sqrtf(mad(Vl.z, Vl.z, len2));
    if (Vl.z >= 0.0f) return D / (2.0f * (Vl.z + t));
    return D * (t - Vl.z) / (2.0f * len2);
}
/* Color::FromSrgb (procedural sky only) */
static float ml_from_srgb1(float x)
{
    x = saturatef(x);
    return x >= 0.04045f ? powf(x * (1.0f / 1.055f) + (0.055f / 1.055f), 2.4f) : x * (1.0f / 12.92f);
}

/* ======================================================================== */
/* BxDF.hlsli                                                                */
/* ======================================================================== */
typedef struct {                 /* Shaders/SurfaceVectors.hlsli:5-16 */
    f3 FrontGeometricNormal, ShadingNormal;
    basis3 ShadingBasis;
} SurfaceVectors;

static SurfaceVectors surface_vectors(int isFront, f3 geometricNormal, f3 shadingNormal)
{
    SurfaceVectors sv;
    sv.FrontGeometricNormal = isFront ? geometricNormal : neg3(geometricNormal);
    sv.ShadingNormal = shadingNormal;
    sv.ShadingBasis = ml_get_basis(shadingNormal);
    return sv;
}

typedef struct {                 /* Shaders/BxDF.hlsli:36-44 */
    f3 BaseColor; float Metallic; f3 Albedo; float Roughness, IORi, IORo; f3 F0; float Transmission;
} BSDFSample;

#define MIN_ROUGHNESS 2e-3f      /* BxDF.hlsli:19 */
enum { LOBE_DIFFUSE = 0, LOBE_SPECULAR = 1, LOBE_TRANSMISSION = 2 };

/* BxDF.hlsli:45-67 */
static void bsdf_init(BSDFSample* b, f3 baseColor, float metallic, float roughness, float IOR,
                      float transmission, int isFrontFace)
{
    b->BaseColor = baseColor;
    b->Metallic = metallic;
    b->Albedo = scl3(baseColor, 1.0f - metallic);
    b->Roughness = fmaxf(MIN_ROUGHNESS, roughness);
    b->IORi = 1.0f; b->IORo = IOR;
    if (!isFrontFace) { b->IORi = IOR; b->IORo = 1.0f; }
    float r = (b->IORi - b->IORo) / (b->IORi + b->IORo);
    float r2 = r * r;                                   /* pow(x, 2) */
    b->F0 = F3(mad(metallic, baseColor.x - r2, r2), mad(metallic, baseColor.y - r2, r2),
               mad(metallic, baseColor.z - r2, r2));    /* lerp(a,b,t) = fma(t, b-a, a) */
    b->Transmission = transmission;
}

/* BxDF.hlsli:21-34 */
static float estimate_diffuse_probability(f3 albedo, f3 f0, float roughness, float NoV)
{
    float F0[3] = { f0.x, f0.y, f0.z }, Fe[3];
    or_env_term_rtg(F0, NoV, roughness, Fe);
    f3 Fenv = F3(Fe[0], Fe[1], Fe[2]);
    float diffuse = ml_luminance(mul3(albedo, F3(1.0f - Fenv.x, 1.0f - Fenv.y, 1.0f - Fenv.z)));
    float specular = ml_luminance(Fenv);
    float sum = diffuse + specular;
    float p = sum > 0.0f ? diffuse / sum : 1.0f;
    if (0.0f < p && p < 1.0f) return clampf(p, 0.05f, 0.95f);
    return p;
}

/* BxDF.hlsli:184-196 ; OR_EXT_LAMBERTIAN_ONLY forces {1,0,0} (build-side switch, config C1) */
static void compute_lobe_weights(const BSDFSample* b, const SurfaceVectors* sv, f3 V, uint32_t ext, float w[3])
{
    if (ext & OR_EXT_LAMBERTIAN_ONLY) { w[0] = 1.0f; w[1] = 0.0f; w[2] = 0.0f; return; }
    float NoV = fabsf(dot3(sv->ShadingNormal, V));
    float tw = b->Transmission * (1.0f - b->Metallic);
    float rw = 1.0f - tw;
    float dw = estimate_diffuse_probability(b->Albedo, b->F0, b->Roughness, NoV);
    float sw = 1.0f - dw;
    w[LOBE_DIFFUSE] = dw * rw;
    w[LOBE_SPECULAR] = sw * rw;
    w[LOBE_TRANSMISSION] = tw;
}

/* BxDF.hlsli:198-212 */
static int find_lobe(const float w[3], float random)
{
    int lobe = 3; float weight = 0.0f;
    while (--lobe > 0) { weight += w[lobe]; if (random < weight) break; }
    return lobe;
}

static inline f3 reflect3(f3 i, f3 n) { float d = dot3(n, i); return madd3(n, -(2.0f * d), i); }
static inline f3 refract3(f3 i, f3 n, float eta)
{
    float d = dot3(n, i);
    float k = mad(-(eta * eta), mad(-d, d, 1.0f), 1.0f);
    if (k < 0.0f) return F3(0.0f, 0.0f, 0.0f);
    float s = mad(eta, d, sqrtf(k));
    return madd3(n, -s, scl3(i, eta));
}

/* BxDF.hlsli:81-86, 110-118, 148-168, 214-226 */
static int bsdf_sample(const BSDFSample* b, const SurfaceVectors* sv, f3 V, const float w[3],
                       const float rnd[4], f3* L, int* lobeType)
{
    int lobe = find_lobe(w, rnd[0]);
    *lobeType = lobe;
    if (lobe == LOBE_DIFFUSE) {
        *L = rotate_vector_inv(sv->ShadingBasis, ml_cosine_get_ray(rnd[1], rnd[2]));
        return dot3(sv->FrontGeometricNormal, *L) > 0.0f;
    }
    f3 Vlocal = rotate_vector(sv->ShadingBasis, V);
    f3 H = rotate_vector_inv(sv->ShadingBasis, ml_vndf_get_ray(rnd[1], rnd[2], b->Roughness, Vlocal));
    if (lobe == LOBE_SPECULAR) {
        *L = reflect3(neg3(V), H);
        return dot3(sv->FrontGeometricNormal, *L) > 0.0f;
    }
    float VoH = fabsf(dot3(V, H)), eta = b->IORi / b->IORo;
    if (eta * eta * (1.0f - VoH * VoH) > 1.0f || rnd[3] < ml_fresnel_dielectric(eta, VoH)) {
        *L = reflect3(neg3(V), H);
    } else {
        *L = refract3(neg3(V), H, eta);
        if (!finite3(*L)) *L = neg3(V);
    }
    return 1;
}

/* BxDF.hlsli:228-245 */
static f3 compute_half_vector(const BSDFSample* b, const SurfaceVectors* sv, f3 L, f3 V, int isTransmissive)
{
    f3 N = sv->FrontGeometricNormal, H;
    if (isTransmissive && dot3(N, L) < 0.0f) {
        H = normalize3(add3(scl3(L, b->IORo), scl3(V, b->IORi)));
        if (dot3(N, H) < 0.0f) H = neg3(H);
    } else {
        H = normalize3(add3(L, V));
    }
    return H;
}

/* BxDF.hlsli:287-299 (single-lobe PDF) with :88-97, :120-131, :170-175 */
static float bsdf_pdf_lobe(const BSDFSample* b, const SurfaceVectors* sv, f3 L, f3 V, const float w[3], int lobe)
{
    f3 H = compute_half_vector(b, sv, L, V, w[LOBE_TRANSMISSION] > 0.0f);
    f3 N = sv->ShadingNormal;
    float lw = w[lobe];
    if (lobe == LOBE_DIFFUSE) {
        if (dot3(sv->FrontGeometricNormal, L) > 0.0f) return ml_cosine_pdf(fabsf(dot3(N, L))) * lw;
        return 0.0f * lw;
    }
    if (lobe == LOBE_SPECULAR) {
        if (dot3(sv->FrontGeometricNormal, L) > 0.0f) {
            f3 Vlocal = rotate_vector(sv->ShadingBasis, V);
            float NoH = fabsf(dot3(N, H));
            return ml_vndf_pdf(Vlocal, NoH, b->Roughness) * lw;
        }
        return 0.0f * lw;
    }
    return fabsf(dot3(N, L)) * lw;
}

/* BxDF.hlsli:301-315 (single-lobe f) with :99-108, :133-146, :177-182 */
static f3 bsdf_eval_lobe(const BSDFSample* b, const SurfaceVectors* sv, f3 L, f3 V, const float w[3],
                         int lobe, uint32_t ext)
{
    float tw = w[LOBE_TRANSMISSION];
    f3 H = compute_half_vector(b, sv, L, V, tw > 0.0f);
    f3 N = sv->ShadingNormal;
    if (lobe == LOBE_TRANSMISSION) {
        float NoL = fabsf(dot3(N, L));
        return scl3(scl3(b->BaseColor, NoL), tw);       /* (NoL * BaseColor) * tw */
    }
    float rw = 1.0f - tw;
    f3 zero = F3(0.0f, 0.0f, 0.0f);
    if (lobe == LOBE_DIFFUSE) {
        if (dot3(sv->FrontGeometricNormal, L) > 0.0f) {
            float NoL = fabsf(dot3(N, L)), NoV = fabsf(dot3(N, V)), VoH = fabsf(dot3(V, H));
            float dterm = (ext & OR_EXT_LAMBERTIAN_ONLY) ? (1.0f / 3.14159265358979323846f)
                                                         : ml_diffuse_burley(b->Roughness, NoL, NoV, VoH);
            return scl3(scl3(scl3(b->Albedo, NoL), dterm), rw);   /* ((NoL * Albedo) * DiffuseTerm) * rw */
        }
        return zero;
    }
    if (dot3(sv->FrontGeometricNormal, L) > 0.0f) {
        float NoL = fabsf(dot3(N, L)), NoV = fabsf(dot3(N, V)), VoH = fabsf(dot3(V, H)), NoH = fabsf(dot3(N, H));
        float D = ml_distribution_ggx(b->Roughness, NoH);
        float G = ml_geometry_term_mod(b->Roughness, NoL, NoV);
        f3 F = ml_fresnel_schlick(b->F0, VoH);
        float k = NoL * D * G;                          /* NoL * D * Gmod * F */
        return scl3(scl3(F, k), rw);
    }
    return zero;
}

/* all-lobe EvaluatePDF :247-264 and Evaluate :266-285 */
void or_bsdf_evaluate(const float* q, uint32_t count, float* r)
{
    for (uint32_t i = 0; i < count; i++, q += 20, r += 8) {
        BSDFSample b;
        int front = q[7] != 0.0f;
        bsdf_init(&b, ld3(q), q[3], q[4], q[5], q[6], front);
        SurfaceVectors sv = surface_vectors(front, ld3(q + 8), ld3(q + 11));
        f3 V = ld3(q + 14), L = ld3(q + 17);
        float w[3]; compute_lobe_weights(&b, &sv, V, 0, w);
        const float tw = w[LOBE_TRANSMISSION];
        float pdf = 0.0f; f3 dif = F3(0, 0, 0), spc = F3(0, 0, 0);
        if (tw > 0.0f) {
            pdf = bsdf_pdf_lobe(&b, &sv, L, V, w, LOBE_TRANSMISSION);
            spc = bsdf_eval_lobe(&b, &sv, L, V, w, LOBE_TRANSMISSION, 0);
        }
        if (tw < 1.0f && dot3(sv.FrontGeometricNormal, L) > 0.0f) {
            pdf += bsdf_pdf_lobe(&b, &sv, L, V, w, LOBE_DIFFUSE) + bsdf_pdf_lobe(&b, &sv, L, V, w, LOBE_SPECULAR);
            dif = bsdf_eval_lobe(&b, &sv, L, V, w, LOBE_DIFFUSE, 0);
            spc = add3(spc, bsdf_eval_lobe(&b, &sv, L, V, w, LOBE_SPECULAR, 0));
        }
        r[0] = dif.x; r[1] = dif.y; r[2] = dif.z; r[3] = spc.x; r[4] = spc.y; r[5] = spc.z; r[6] = pdf; r[7] = 0.0f;
    }
}

int or_bsdf_sample(const float mat[7], int front_face, const float Ng[3], const float Ns[3],
                   const float Vv[3], const float rnd[4], uint32_t ext_flags,
                   float Lout[3], int* lobe, float* pdf, float f[3], float weights[3])
{
    BSDFSample b;
    bsdf_init(&b, ld3(mat), mat[3], mat[4], mat[5], mat[6], front_face);
    SurfaceVectors sv = surface_vectors(front_face, ld3(Ng), ld3(Ns));
    f3 V = ld3(Vv), L = F3(0, 0, 0);
    compute_lobe_weights(&b, &sv, V, ext_flags, weights);
    int ok = bsdf_sample(&b, &sv, V, weights, rnd, &L, lobe);
    Lout[0] = L.x; Lout[1] = L.y; Lout[2] = L.z;
    *pdf = 0.0f; f[0] = f[1] = f[2] = 0.0f;
    if (ok) {
        *pdf = bsdf_pdf_lobe(&b, &sv, L, V, weights, *lobe);
        f3 fv = bsdf_eval_lobe(&b, &sv, L, V, weights, *lobe, ext_flags);
        f[0] = fv.x; f[1] = fv.y; f[2] = fv.z;
    }
    return ok;
}

/* ======================================================================== */
/* SelfIntersectionAvoidance.hlsli:39-117, HitInfo.hlsli                     */
/* method and constants: Copyright (c) 2023 NVIDIA CORPORATION & AFFILIATES, BSD-3-Clause (THIRD_PARTY_NOTICES.md) */
/* ======================================================================== */
void or_safe_spawn(const float v[9], const float bary[2], const float M[12], const float W[12],
                   float objPosOut[3], float wldPosOut[3], float objNOut[3], float wldNOut[3], float* offset)
{
    f3 v0 = ld3(v), v1 = ld3(v + 3), v2 = ld3(v + 6);
    f3 e1 = sub3(v1, v0), e2 = sub3(v2, v0);
    float bx = bary[0], by = bary[1];
    /* objPosition = v0 + mad(b.x, edge1, mul(b.y, edge2)) */
    f3 op = F3(v0.x + fmaf(bx, e1.x, by * e2.x), v0.y + fmaf(bx, e1.y, by * e2.y), v0.z + fmaf(bx, e1.z, by * e2.z));
    f3 on = cross3(e1, e2);
    f3 wp;
    wp.x = M[3]  + fmaf(M[0], op.x, fmaf(M[1], op.y, M[2]  * op.z));
    wp.y = M[7]  + fmaf(M[4], op.x, fmaf(M[5], op.y, M[6]  * op.z));
    wp.z = M[11] + fmaf(M[8], op.x, fmaf(M[9], op.y, M[10] * op.z));
    /* wldNormal = mul(transpose((float3x3)worldToObject), objNormal) */
    f3 wn = F3(W[0] * on.x + W[4] * on.y + W[8]  * on.z,
               W[1] * on.x + W[5] * on.y + W[9]  * on.z,
               W[2] * on.x + W[6] * on.y + W[10] * on.z);
    float wldScale = 1.0f / sqrtf(dot3(wn, wn));
    wn = scl3(wn, wldScale);

    const float c0 = 5.9604644775390625E-8f;
    const float c1 = 1.788139769587360206060111522674560546875E-7f;
    const float c2 = 1.19209317972490680404007434844970703125E-7f;
    f3 ae1 = abs3(e1), ae2 = abs3(e2);
    f3 ext3 = add3(add3(ae1, ae2), abs3(sub3(ae1, ae2)));
    float extent = fmaxf(fmaxf(ext3.x, ext3.y), ext3.z);
    f3 av0 = abs3(v0);
    f3 objErr = F3(fmaf(c0, av0.x, c1 * extent), fmaf(c0, av0.y, c1 * extent), fmaf(c0, av0.z, c1 * extent));
    f3 aop = abs3(op);
    f3 mo = F3(fabsf(M[0]) * aop.x + fabsf(M[1]) * aop.y + fabsf(M[2])  * aop.z,
               fabsf(M[4]) * aop.x + fabsf(M[5]) * aop.y + fabsf(M[6])  * aop.z,
               fabsf(M[8]) * aop.x + fabsf(M[9]) * aop.y + fabsf(M[10]) * aop.z);
    f3 wldErr = F3(fmaf(c1, mo.x, c2 * fabsf(M[3])), fmaf(c1, mo.y, c2 * fabsf(M[7])), fmaf(c1, mo.z, c2 * fabsf(M[11])));
    f3 awp = abs3(wp);
    f3 wo = F3(fabsf(W[0]) * awp.x + fabsf(W[1]) * awp.y + fabsf(W[2])  * awp.z + fabsf(W[3]),
               fabsf(W[4]) * awp.x + fabsf(W[5]) * awp.y + fabsf(W[6])  * awp.z + fabsf(W[7]),
               fabsf(W[8]) * awp.x + fabsf(W[9]) * awp.y + fabsf(W[10]) * awp.z + fabsf(W[11]));
    objErr = F3(fmaf(c2, wo.x, objErr.x), fmaf(c2, wo.y, objErr.y), fmaf(c2, wo.z, objErr.z));
    float wldOffset = dot3(wldErr, abs3(wn));
    float objOffset = dot3(objErr, abs3(on));
    wldOffset = fmaf(wldScale, objOffset, wldOffset);

    f3 onn = normalize3(on);
    objPosOut[0] = op.x; objPosOut[1] = op.y; objPosOut[2] = op.z;
    wldPosOut[0] = wp.x; wldPosOut[1] = wp.y; wldPosOut[2] = wp.z;
    objNOut[0] = onn.x; objNOut[1] = onn.y; objNOut[2] = onn.z;
    wldNOut[0] = wn.x; wldNOut[1] = wn.y; wldNOut[2] = wn.z;
    *offset = wldOffset;
}

typedef struct {                 /* Shaders/HitInfo.hlsli:7-22 (texture coordinates/tangent: untextured scope) */
    f3 Position, ObjectPosition; float PositionOffset;
    f3 FlatNormal, GeometricNormal, ShadingNormal, Tangent;
    int IsFrontFace;
    float TextureCoordinates[2][2];
    float Bary[2];
    float Distance;
    uint32_t InstanceIndex, ObjectIndex, PrimitiveIndex;
} HitInfo;

/* HitInfo.hlsli:96-99 + SelfIntersectionAvoidance.hlsli:113-117 */
static f3 safe_world_ray_origin(const HitInfo* h, f3 dir)
{
    float s = ml_sign(dot3(dir, h->FlatNormal));
    f3 n = scl3(h->FlatNormal, s);
    float o = h->PositionOffset;
    return F3(fmaf(o, n.x, h->Position.x), fmaf(o, n.y, h->Position.y), fmaf(o, n.z, h->Position.z));
}


/* ======================================================================== */
/* textures (ShadingHelpers.hlsli:53-59 Sample<T>: SampleLevel(g_anisotropicSampler, uv, 0))                 */
/* The root-signature static sampler (Raytracing.hlsl:79 "StaticSampler(s0)") has D3D12's defaults: WRAP      */
/* addressing, and at an explicit LOD the anisotropic filter degenerates to a bilinear tap of mip 0.          */
/* [spec] weights in fp32 (hardware uses 8-bit fixed-point fractions), sRGB decoded per texel before the      */
/* filter through a 256-entry table evaluated in double precision.                                            */
/* ======================================================================== */
static float g_srgb_lut[256];
static int g_srgb_ready = 0;
static void srgb_init(void)
{
    if (g_srgb_ready) return;
    for (int i = 0; i < 256; i++) {
        double c = i / 255.0;
        g_srgb_lut[i] = (float)(c <= 0.04045 ? c / 12.92 : pow((c + 0.055) / 1.055, 2.4));
    }
    g_srgb_ready = 1;
}

static void texel_fetch(const OrHeapEntry* t, uint32_t face, uint32_t x, uint32_t y, float out[4])
{
    const uint32_t W = (uint32_t)(t->Bytes & 0xFFFFFFFFu), H = (uint32_t)(t->Bytes >> 32);
    const size_t idx = ((size_t)face * H + y) * W + x;
    if (t->Stride == OR_FMT_RGBA32_FLOAT) { memcpy(out, (const float*)t->Ptr + 4 * idx, 16); return; }
    const uint8_t* p = (const uint8_t*)t->Ptr + 4 * idx;
    if (t->Stride == OR_FMT_RGBA8_UNORM_SRGB) {
        srgb_init();
        out[0] = g_srgb_lut[p[0]]; out[1] = g_srgb_lut[p[1]]; out[2] = g_srgb_lut[p[2]];
    } else { out[0] = unorm8_to_f32(p[0]); out[1] = unorm8_to_f32(p[1]); out[2] = unorm8_to_f32(p[2]); }
    out[3] = unorm8_to_f32(p[3]);
}

static void bilinear(const OrHeapEntry* t, uint32_t face, float fx, float fy, int wrap, float out[4])
{
    const int W = (int)(t->Bytes & 0xFFFFFFFFu), H = (int)(t->Bytes >> 32);
    float x0f = floorf(fx), y0f = floorf(fy);
    float wx = fx - x0f, wy = fy - y0f;
    int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    if (wrap) {
        x0 = ((x0 % W) + W) % W; x1 = ((x1 % W) + W) % W; y0 = ((y0 % H) + H) % H; y1 = ((y1 % H) + H) % H;
    } else {
        x0 = x0 < 0 ? 0 : (x0 >= W ? W - 1 : x0); x1 = x1 < 0 ? 0 : (x1 >= W ? W - 1 : x1);
        y0 = y0 < 0 ? 0 : (y0 >= H ? H - 1 : y0); y1 = y1 < 0 ? 0 : (y1 >= H ? H - 1 : y1);
    }
    float c00[4], c10[4], c01[4], c11[4];
    texel_fetch(t, face, (uint32_t)x0, (uint32_t)y0, c00); texel_fetch(t, face, (uint32_t)x1, (uint32_t)y0, c10);
    texel_fetch(t, face, (uint32_t)x0, (uint32_t)y1, c01); texel_fetch(t, face, (uint32_t)x1, (uint32_t)y1, c11);
    for (int c = 0; c < 4; c++) {
        float top = mad(c10[c], wx, c00[c] * (1.0f - wx));
        float bot = mad(c11[c], wx, c01[c] * (1.0f - wx));
        out[c] = mad(bot, wy, top * (1.0f - wy));
    }
}

void or_texture_sample(const OrHeapEntry* t, float u, float v, float out[4])
{
    const float W = (float)(uint32_t)(t->Bytes & 0xFFFFFFFFu), H = (float)(uint32_t)(t->Bytes >> 32);
    if (!(u == u) || !isfinite(u)) u = 0.0f;
    if (!(v == v) || !isfinite(v)) v = 0.0f;
    u = u - floorf(u); v = v - floorf(v);                    /* WRAP */
    bilinear(t, 0, u * W - 0.5f, v * H - 0.5f, 1, out);
}

/* TextureCube.SampleLevel: D3D face selection (major axis), bilinear inside the face, clamped at its border */
void or_cube_sample(const OrHeapEntry* t, const float d[3], float out[4])
{
    const float W = (float)(uint32_t)(t->Bytes & 0xFFFFFFFFu), H = (float)(uint32_t)(t->Bytes >> 32);
    float ax = fabsf(d[0]), ay = fabsf(d[1]), az = fabsf(d[2]), sc, tc, ma; uint32_t face;
    if (ax >= ay && ax >= az) { ma = ax; if (d[0] >= 0.0f) { face = 0; sc = -d[2]; tc = -d[1]; } else { face = 1; sc = d[2]; tc = -d[1]; } }
    else if (ay >= az) { ma = ay; if (d[1] >= 0.0f) { face = 2; sc = d[0]; tc = d[2]; } else { face = 3; sc = d[0]; tc = -d[2]; } }
    else { ma = az; if (d[2] >= 0.0f) { face = 4; sc = d[0]; tc = -d[1]; } else { face = 5; sc = -d[0]; tc = -d[1]; } }
    float u = (sc / ma + 1.0f) * 0.5f, v = (tc / ma + 1.0f) * 0.5f;
    bilinear(t, face, u * W - 0.5f, v * H - 0.5f, 0, out);
}

/* Sample<T>(TextureMapInfo, textureCoordinates) ShadingHelpers.hlsli:53-59 */
static void sample_map(const OrHeapEntry* heap, const OrTextureMapInfo* info, const float uv[2][2], float out[4])
{
    const float* c = uv[info->TextureCoordinateIndex & 1u];
    or_texture_sample(&heap[info->Descriptor], c[0], c[1], out);
}

enum { TEX_BaseColor = 0, TEX_EmissiveColor, TEX_Metallic, TEX_Roughness, TEX_MetallicRoughness, TEX_Transmission, TEX_Normal };

/* EvaluateBaseColor ShadingHelpers.hlsli:61-73 */
static void evaluate_base_color(float bc[4], const OrHeapEntry* heap, const OrTextureMapInfo* info, const float uv[2][2])
{
    if ((bc[0] > 0.0f || bc[1] > 0.0f || bc[2] > 0.0f || bc[3] > 0.0f) && info->Descriptor != ~0u) {
        float t[4]; sample_map(heap, info, uv, t);
        for (int c = 0; c < 4; c++) bc[c] *= t[c];
    }
}

/* IsOpaque ShadingHelpers.hlsli:105-115 (the closest-hit overload) */
static int is_opaque(const OrObjectData* od, const OrHeapEntry* heap, const float uv[2][2])
{
    float bc[4]; memcpy(bc, od->Material.BaseColor, 16);
    evaluate_base_color(bc, heap, &od->TextureMapInfoArray[TEX_BaseColor], uv);
    return bc[3] >= od->Material.AlphaCutoff;
}

/* EvaluateMaterial ShadingHelpers.hlsli:161-235; N is the (front-facing) shading normal, T the front tangent */
static OrMaterial evaluate_material(f3* N, f3 T, const OrObjectData* od, const OrHeapEntry* heap, const float uv[2][2])
{
    OrMaterial m = od->Material;
    const OrTextureMapInfo* ti = od->TextureMapInfoArray;
    float t[4];
    evaluate_base_color(m.BaseColor, heap, &ti[TEX_BaseColor], uv);
    f3 em = scl3(ld3(m.EmissiveColor), m.EmissiveStrength);
    if ((em.x > 0.0f || em.y > 0.0f || em.z > 0.0f) && ti[TEX_EmissiveColor].Descriptor != ~0u) {
        sample_map(heap, &ti[TEX_EmissiveColor], uv, t);
        m.EmissiveColor[0] *= t[0]; m.EmissiveColor[1] *= t[1]; m.EmissiveColor[2] *= t[2];
    }
    if (ti[TEX_MetallicRoughness].Descriptor != ~0u) {
        if (m.Metallic > 0.0f || m.Roughness > 0.0f) {
            sample_map(heap, &ti[TEX_MetallicRoughness], uv, t);
            m.Metallic *= t[2]; m.Roughness *= t[1];
        }
    } else {
        if (m.Metallic > 0.0f && ti[TEX_Metallic].Descriptor != ~0u) { sample_map(heap, &ti[TEX_Metallic], uv, t); m.Metallic *= t[0]; }
        if (m.Roughness > 0.0f && ti[TEX_Roughness].Descriptor != ~0u) { sample_map(heap, &ti[TEX_Roughness], uv, t); m.Roughness *= t[0]; }
    }
    if (m.Metallic < 1.0f) {
        if (m.Transmission > 0.0f && ti[TEX_Transmission].Descriptor != ~0u) { sample_map(heap, &ti[TEX_Transmission], uv, t); m.Transmission *= t[0]; }
    }
    if ((T.x != 0.0f || T.y != 0.0f || T.z != 0.0f) && ti[TEX_Normal].Descriptor != ~0u) {        /* PerturbNormal :89-103 */
        sample_map(heap, &ti[TEX_Normal], uv, t);
        /* [MathLib spec] Geometry::UnpackLocalNormal: xy*2-1, z = sqrt(saturate(1 - |xy|^2)) */
        float nx = t[0] * 2.0f - 1.0f, ny = t[1] * 2.0f - 1.0f;
        float nz = ml_sqrt01(1.0f - (nx * nx + ny * ny));
        /* Math::CalculateTBN Math.hlsli:17-21 */
        f3 Tn = normalize3(sub3(T, scl3(*N, dot3(*N, T))));
        f3 B = cross3(*N, Tn);
        f3 r = F3(sop3(Tn.x, nx, B.x, ny, N->x, nz), sop3(Tn.y, nx, B.y, ny, N->y, nz), sop3(Tn.z, nx, B.z, ny, N->z, nz));
        *N = normalize3(r);
    }
    return m;
}

/* GetTextureCoordinates ShadingHelpers.hlsli:32-51 */
static void get_texture_coordinates(const OrObjectData* od, const OrHeapEntry* heap, uint32_t prim, float bu, float bv, float uv[2][2])
{
    const OrHeapEntry* vb = &heap[od->MeshDescriptors.Vertices];
    const OrHeapEntry* ib = &heap[od->MeshDescriptors.Indices];
    for (int i = 0; i < 2; i++) {
        uv[i][0] = uv[i][1] = 0.0f;
        uint32_t off = od->VertexDesc.TextureCoordinates[i];
        if (off == ~0u) continue;
        float a[3][2];
        for (int k = 0; k < 3; k++) {
            uint32_t idx = ib->Stride == 2 ? (uint32_t)((const uint16_t*)ib->Ptr)[3 * prim + k] : ((const uint32_t*)ib->Ptr)[3 * prim + k];
            uint16_t h[2]; memcpy(h, (const uint8_t*)vb->Ptr + (size_t)od->VertexDesc.Stride * idx + off, 4);
            a[k][0] = or_f16_to_f32(h[0]); a[k][1] = or_f16_to_f32(h[1]);
        }
        for (int c = 0; c < 2; c++) uv[i][c] = interp1(a[0][c], a[1][c], a[2][c], bu, bv);
    }
}

/* ======================================================================== */
/* scene + traversal                                                         */
/* ======================================================================== */
typedef struct { f3 v0, v1, v2; uint32_t geom, prim, opaque; } Tri;           /* geom = GeometryIndex within the BLAS */
typedef struct { float lo[3], hi[3]; uint32_t left, right, first, count; } Node; /* count>0 => leaf */

typedef struct {
    Tri* tris; uint32_t n_tris;
    Node* nodes; uint32_t n_nodes;
    float lo[3], hi[3];
} Blas;

struct OrScene {
    Blas* blas; uint32_t n_blas;
    OrInstanceDesc* inst; float* w2o; uint32_t n_inst;   /* w2o: 12 floats per instance */
    Node* tl_nodes; uint32_t n_tl_nodes; uint32_t* tl_index;
    OrObjectData* objects; uint32_t n_objects;
    OrInstanceData* inst_data;
    OrHeapEntry* heap; uint32_t n_heap;
    int accel_mode;
};

/* worldToObject = inverse of the affine 3x4, evaluated in double and rounded once to float.
 * (DXR computes CommittedWorldToObject3x4 itself; this is the build's definition of it.) */
void or_invert_3x4(const float m[12], float out[12])
{
    double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
    double tx = m[3], ty = m[7], tz = m[11];
    double A = e * i - f * h, B = c * h - b * i, C = b * f - c * e;
    double D = f * g - d * i, E = a * i - c * g, F = c * d - a * f;
    double G = d * h - e * g, H = b * g - a * h, I = a * e - b * d;
    double det = a * A + b * D + c * G;
    double r = 1.0 / det;
    double i00 = A * r, i01 = B * r, i02 = C * r, i10 = D * r, i11 = E * r, i12 = F * r, i20 = G * r, i21 = H * r, i22 = I * r;
    out[0] = (float)i00; out[1] = (float)i01; out[2]  = (float)i02; out[3]  = (float)(-(i00 * tx + i01 * ty + i02 * tz));
    out[4] = (float)i10; out[5] = (float)i11; out[6]  = (float)i12; out[7]  = (float)(-(i10 * tx + i11 * ty + i12 * tz));
    out[8] = (float)i20; out[9] = (float)i21; out[10] = (float)i22; out[11] = (float)(-(i20 * tx + i21 * ty + i22 * tz));
}

/* ---- ray / triangle: watertight (Woop, Benthin, Wald 2013), fp32 with fp64 edge fallback -------- */
typedef struct {
    f3 o, d; int kx, ky, kz; float Sx, Sy, Sz;
} RayObj;

static void ray_setup(RayObj* r, f3 o, f3 d)
{
    r->o = o; r->d = d;
    int kz = 0; float m = fabsf(d.x);
    if (fabsf(d.y) > m) { kz = 1; m = fabsf(d.y); }
    if (fabsf(d.z) > m) { kz = 2; }
    int kx = kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    if (get3(d, kz) < 0.0f) { int t = kx; kx = ky; ky = t; }
    r->kx = kx; r->ky = ky; r->kz = kz;
    r->Sz = 1.0f / get3(d, kz);          /* one IEEE division; Sx, Sy by multiplication (spec) */
    r->Sx = get3(d, kx) * r->Sz;
    r->Sy = get3(d, ky) * r->Sz;
}

/* returns 1 and (t,u,v) when the triangle is hit with t in (tmin, +inf); caller applies tmax/tie-break */
static int tri_test(const RayObj* r, f3 v0, f3 v1, f3 v2, float* t, float* u, float* v)
{
    f3 A = sub3(v0, r->o), B = sub3(v1, r->o), C = sub3(v2, r->o);
    float Akz = get3(A, r->kz), Bkz = get3(B, r->kz), Ckz = get3(C, r->kz);
    float Ax = get3(A, r->kx) - r->Sx * Akz, Ay = get3(A, r->ky) - r->Sy * Akz;
    float Bx = get3(B, r->kx) - r->Sx * Bkz, By = get3(B, r->ky) - r->Sy * Bkz;
    float Cx = get3(C, r->kx) - r->Sx * Ckz, Cy = get3(C, r->ky) - r->Sy * Ckz;
    float U = Cx * By - Cy * Bx;
    float V = Ax * Cy - Ay * Cx;
    float W = Bx * Ay - By * Ax;
    if (U == 0.0f || V == 0.0f || W == 0.0f) {
        double CxBy = (double)Cx * (double)By, CyBx = (double)Cy * (double)Bx;
        U = (float)(CxBy - CyBx);
        double AxCy = (double)Ax * (double)Cy, AyCx = (double)Ay * (double)Cx;
        V = (float)(AxCy - AyCx);
        double BxAy = (double)Bx * (double)Ay, ByAx = (double)By * (double)Ax;
        W = (float)(BxAy - ByAx);
    }
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return 0;
    float det = U + V + W;
    if (det == 0.0f) return 0;
    float Az = r->Sz * Akz, Bz = r->Sz * Bkz, Cz = r->Sz * Ckz;
    float T = U * Az + V * Bz + W * Cz;
    float rcp = 1.0f / det;
    *t = T * rcp; *u = V * rcp; *v = W * rcp;
    return 1;
}

int or_ray_triangle(const float o[3], const float d[3], float tmin, float tmax,
                    const float v0[3], const float v1[3], const float v2[3], float* t, float* u, float* v)
{
    RayObj r; ray_setup(&r, ld3(o), ld3(d));
    float tt, uu, vv;
    if (!tri_test(&r, ld3(v0), ld3(v1), ld3(v2), &tt, &uu, &vv)) return 0;
    if (!(tt > tmin && tt < tmax)) return 0;
    *t = tt; *u = uu; *v = vv;
    return 1;
}

typedef struct {
    float t, u, v; uint32_t inst, geom, prim; int hit;
} Committed;

/* closest hit; ties on t resolved by (instance, geometry, primitive) lexicographic order so
 * the result does not depend on traversal order (DXR leaves ties undefined). */
static int is_opaque(const OrObjectData* od, const OrHeapEntry* heap, const float uv[2][2]);
static void get_texture_coordinates(const OrObjectData* od, const OrHeapEntry* heap, uint32_t prim, float bu, float bv, float uv[2][2]);
static int candidate_is_opaque(const OrScene* s, uint32_t inst, uint32_t geom, uint32_t prim, float u, float v);

static inline void commit_candidate(const OrScene* s, int opaque, Committed* c, float tmin, float t, float u, float v,
                                    uint32_t inst, uint32_t geom, uint32_t prim)
{
    if (!(t > tmin)) return;
    int better;
    if (t < c->t) better = 1;
    else if (t == c->t && c->hit) {
        better = inst < c->inst || (inst == c->inst && (geom < c->geom || (geom == c->geom && prim < c->prim)));
    } else better = 0;
    /* CANDIDATE_NON_OPAQUE_TRIANGLE: alpha test before the commit (RaytracingHelpers.hlsli:19-44) */
    if (better && !opaque && !candidate_is_opaque(s, inst, geom, prim, u, v)) better = 0;
    if (better) { c->t = t; c->u = u; c->v = v; c->inst = inst; c->geom = geom; c->prim = prim; c->hit = 1; }
}

/* box test only: clamp |d| away from 0 so that slab planes never become inf - inf = NaN */
static inline float safe_inv1(float d) { float a = fabsf(d) < 1e-20f ? copysignf(1e-20f, d) : d; return 1.0f / a; }

static inline int box_test(const float lo[3], const float hi[3], f3 o, f3 inv, float tmin, float tmax)
{
    float t0x = (lo[0] - o.x) * inv.x, t1x = (hi[0] - o.x) * inv.x;
    float t0y = (lo[1] - o.y) * inv.y, t1y = (hi[1] - o.y) * inv.y;
    float t0z = (lo[2] - o.z) * inv.z, t1z = (hi[2] - o.z) * inv.z;
    float tn = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), tmin));
    float tf = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), tmax));
    return tn <= tf * 1.0000004f;
}

static void blas_intersect(const OrScene* s, const Blas* b, int accel, f3 o, f3 d, float tmin, uint32_t inst, Committed* c)
{
    RayObj r; ray_setup(&r, o, d);
    if (!accel) {
        for (uint32_t i = 0; i < b->n_tris; i++) {
            float t, u, v;
            if (tri_test(&r, b->tris[i].v0, b->tris[i].v1, b->tris[i].v2, &t, &u, &v))
                commit_candidate(s, (int)b->tris[i].opaque, c, tmin, t, u, v, inst, b->tris[i].geom, b->tris[i].prim);
        }
        return;
    }
    if (!b->n_nodes) return;
    f3 inv = F3(safe_inv1(d.x), safe_inv1(d.y), safe_inv1(d.z));
    uint32_t stack[128]; int sp = 0; stack[sp++] = 0;
    while (sp) {
        const Node* n = &b->nodes[stack[--sp]];
        if (!box_test(n->lo, n->hi, o, inv, tmin, c->t)) continue;
        if (n->count) {
            for (uint32_t i = n->first; i < n->first + n->count; i++) {
                float t, u, v;
                if (tri_test(&r, b->tris[i].v0, b->tris[i].v1, b->tris[i].v2, &t, &u, &v))
                    commit_candidate(s, (int)b->tris[i].opaque, c, tmin, t, u, v, inst, b->tris[i].geom, b->tris[i].prim);
            }
        } else { stack[sp++] = n->left; stack[sp++] = n->right; }
    }
}

static inline void instance_intersect(const OrScene* s, uint32_t ii, f3 o, f3 d, float tmin, Committed* c)
{
    const OrInstanceDesc* in = &s->inst[ii];
    if (!(in->InstanceMask & 0xFFu)) return;
    const float* W = &s->w2o[12 * ii];
    f3 oo = F3(sop3t(W[0], o.x, W[1], o.y, W[2], o.z, W[3]), sop3t(W[4], o.x, W[5], o.y, W[6], o.z, W[7]), sop3t(W[8], o.x, W[9], o.y, W[10], o.z, W[11]));
    f3 od = F3(sop3(W[0], d.x, W[1], d.y, W[2], d.z), sop3(W[4], d.x, W[5], d.y, W[6], d.z), sop3(W[8], d.x, W[9], d.y, W[10], d.z));
    blas_intersect(s, &s->blas[in->Blas], s->accel_mode, oo, od, tmin, ii, c);
}

/* TraceRay (RaytracingHelpers.hlsli:7-55) with flags NONE, mask ~0: closest hit, no culling,
 * all geometry treated as opaque (alpha-tested candidates are SURVEY 8f "next"). Triangle hits are
 * accepted for t in (TMin, TMax) exclusive (DXR ray-extents rule for triangles). */
static Committed trace_ray(const OrScene* s, f3 o, f3 d, float tmin, float tmax)
{
    Committed c; memset(&c, 0, sizeof c); c.t = tmax;
    if (!s->accel_mode) {
        for (uint32_t i = 0; i < s->n_inst; i++) instance_intersect(s, i, o, d, tmin, &c);
    } else if (s->n_tl_nodes) {
        f3 inv = F3(safe_inv1(d.x), safe_inv1(d.y), safe_inv1(d.z));
        uint32_t stack[128]; int sp = 0; stack[sp++] = 0;
        while (sp) {
            const Node* n = &s->tl_nodes[stack[--sp]];
            if (!box_test(n->lo, n->hi, o, inv, tmin, c.t)) continue;
            if (n->count) {
                for (uint32_t i = n->first; i < n->first + n->count; i++) instance_intersect(s, s->tl_index[i], o, d, tmin, &c);
            } else { stack[sp++] = n->left; stack[sp++] = n->right; }
        }
    }
    if (c.hit && !(c.t < tmax)) c.hit = 0;
    return c;
}

/* ---- oracle's own BVH (top-down, median split on the widest centroid axis) ------------------ */
typedef struct { float lo[3], hi[3], c[3]; uint32_t id; } BRef;
static int g_sort_axis;   /* build is single-threaded */
static int bref_cmp(const void* a, const void* b)
{
    float x = ((const BRef*)a)->c[g_sort_axis], y = ((const BRef*)b)->c[g_sort_axis];
    if (x < y) return -1; if (x > y) return 1;
    uint32_t ia = ((const BRef*)a)->id, ib = ((const BRef*)b)->id;
    return ia < ib ? -1 : (ia > ib ? 1 : 0);
}
static void pad_box(float lo[3], float hi[3])
{
    for (int k = 0; k < 3; k++) {
        float e = 1e-5f * fmaxf(fabsf(lo[k]), fabsf(hi[k])) + 1e-6f * (hi[k] - lo[k]) + 1e-30f;
        lo[k] -= e; hi[k] += e;
    }
}
static uint32_t build_rec(BRef* refs, uint32_t first, uint32_t count, Node* nodes, uint32_t* n_nodes, uint32_t leaf_max)
{
    uint32_t me = (*n_nodes)++;
    Node* n = &nodes[me];
    float clo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, chi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (int k = 0; k < 3; k++) { n->lo[k] = FLT_MAX; n->hi[k] = -FLT_MAX; }
    for (uint32_t i = first; i < first + count; i++)
        for (int k = 0; k < 3; k++) {
            n->lo[k] = fminf(n->lo[k], refs[i].lo[k]); n->hi[k] = fmaxf(n->hi[k], refs[i].hi[k]);
            clo[k] = fminf(clo[k], refs[i].c[k]); chi[k] = fmaxf(chi[k], refs[i].c[k]);
        }
    pad_box(n->lo, n->hi);
    n->left = n->right = 0; n->first = first; n->count = 0;
    if (count <= leaf_max) { n->count = count; return me; }
    int ax = 0; float w = chi[0] - clo[0];
    if (chi[1] - clo[1] > w) { ax = 1; w = chi[1] - clo[1]; }
    if (chi[2] - clo[2] > w) { ax = 2; }
    g_sort_axis = ax;
    qsort(refs + first, count, sizeof(BRef), bref_cmp);
    uint32_t half = count / 2;
    uint32_t l = build_rec(refs, first, half, nodes, n_nodes, leaf_max);
    uint32_t r = build_rec(refs, first + half, count - half, nodes, n_nodes, leaf_max);
    nodes[me].left = l; nodes[me].right = r;
    return me;
}

static uint32_t load_index(const void* p, uint32_t stride, uint32_t i)
{
    return stride == 2 ? (uint32_t)((const uint16_t*)p)[i] : ((const uint32_t*)p)[i];
}

OrScene* or_scene_create(const OrGeometryDesc* geoms, uint32_t n_geoms,
                         const OrBlasDesc* blas, uint32_t n_blas,
                         const OrInstanceDesc* inst, uint32_t n_inst,
                         const OrObjectData* objects, uint32_t n_objects,
                         const OrInstanceData* inst_data,
                         const OrHeapEntry* heap, uint32_t n_heap, int accel_mode)
{
    (void)n_geoms;
    OrScene* s = (OrScene*)calloc(1, sizeof *s);
    s->accel_mode = accel_mode;
    s->n_blas = n_blas; s->blas = (Blas*)calloc(n_blas ? n_blas : 1, sizeof(Blas));
    for (uint32_t b = 0; b < n_blas; b++) {
        Blas* B = &s->blas[b];
        uint32_t nt = 0;
        for (uint32_t g = 0; g < blas[b].GeometryCount; g++) nt += geoms[blas[b].FirstGeometry + g].IndexCount / 3;
        B->n_tris = nt; B->tris = (Tri*)malloc(sizeof(Tri) * (nt ? nt : 1));
        uint32_t k = 0;
        for (int a = 0; a < 3; a++) { B->lo[a] = FLT_MAX; B->hi[a] = -FLT_MAX; }
        for (uint32_t g = 0; g < blas[b].GeometryCount; g++) {
            const OrGeometryDesc* G = &geoms[blas[b].FirstGeometry + g];
            for (uint32_t p = 0; p < G->IndexCount / 3; p++, k++) {
                Tri* T = &B->tris[k];
                const float* pv[3];
                for (int c = 0; c < 3; c++) {
                    uint32_t idx = load_index(G->Indices, G->IndexStride, 3 * p + c);
                    pv[c] = (const float*)((const uint8_t*)G->Vertices + (size_t)G->VertexStride * idx);
                }
                T->v0 = ld3(pv[0]); T->v1 = ld3(pv[1]); T->v2 = ld3(pv[2]); T->geom = g; T->prim = p; T->opaque = G->Flags & 1u;
                for (int c = 0; c < 3; c++) for (int a = 0; a < 3; a++) {
                    B->lo[a] = fminf(B->lo[a], pv[c][a]); B->hi[a] = fmaxf(B->hi[a], pv[c][a]);
                }
            }
        }
        if (accel_mode && nt) {
            BRef* refs = (BRef*)malloc(sizeof(BRef) * nt);
            for (uint32_t i = 0; i < nt; i++) {
                const Tri* T = &B->tris[i];
                for (int a = 0; a < 3; a++) {
                    float x0 = get3(T->v0, a), x1 = get3(T->v1, a), x2 = get3(T->v2, a);
                    refs[i].lo[a] = fminf(fminf(x0, x1), x2); refs[i].hi[a] = fmaxf(fmaxf(x0, x1), x2);
                    refs[i].c[a] = 0.5f * (refs[i].lo[a] + refs[i].hi[a]);
                }
                refs[i].id = i;
            }
            B->nodes = (Node*)malloc(sizeof(Node) * (2 * nt));
            build_rec(refs, 0, nt, B->nodes, &B->n_nodes, 4);
            Tri* sorted = (Tri*)malloc(sizeof(Tri) * nt);
            for (uint32_t i = 0; i < nt; i++) sorted[i] = B->tris[refs[i].id];
            free(B->tris); B->tris = sorted; free(refs);
        }
    }
    s->n_inst = n_inst;
    s->inst = (OrInstanceDesc*)malloc(sizeof(OrInstanceDesc) * (n_inst ? n_inst : 1));
    memcpy(s->inst, inst, sizeof(OrInstanceDesc) * n_inst);
    s->w2o = (float*)malloc(sizeof(float) * 12 * (n_inst ? n_inst : 1));
    for (uint32_t i = 0; i < n_inst; i++) or_invert_3x4(inst[i].Transform, &s->w2o[12 * i]);
    if (accel_mode && n_inst) {
        BRef* refs = (BRef*)malloc(sizeof(BRef) * n_inst);
        for (uint32_t i = 0; i < n_inst; i++) {
            const Blas* B = &s->blas[inst[i].Blas]; const float* M = inst[i].Transform;
            for (int a = 0; a < 3; a++) { refs[i].lo[a] = FLT_MAX; refs[i].hi[a] = -FLT_MAX; }
            for (int cn = 0; cn < 8; cn++) {
                float x = (cn & 1) ? B->hi[0] : B->lo[0], y = (cn & 2) ? B->hi[1] : B->lo[1], z = (cn & 4) ? B->hi[2] : B->lo[2];
                for (int a = 0; a < 3; a++) {
                    float w = M[4 * a] * x + M[4 * a + 1] * y + M[4 * a + 2] * z + M[4 * a + 3];
                    refs[i].lo[a] = fminf(refs[i].lo[a], w); refs[i].hi[a] = fmaxf(refs[i].hi[a], w);
                }
            }
            pad_box(refs[i].lo, refs[i].hi);
            for (int a = 0; a < 3; a++) refs[i].c[a] = 0.5f * (refs[i].lo[a] + refs[i].hi[a]);
            refs[i].id = i;
        }
        s->tl_nodes = (Node*)malloc(sizeof(Node) * 2 * n_inst);
        build_rec(refs, 0, n_inst, s->tl_nodes, &s->n_tl_nodes, 2);
        s->tl_index = (uint32_t*)malloc(sizeof(uint32_t) * n_inst);
        for (uint32_t i = 0; i < n_inst; i++) s->tl_index[i] = refs[i].id;
        free(refs);
    }
    s->n_objects = n_objects;
    s->objects = (OrObjectData*)malloc(sizeof(OrObjectData) * (n_objects ? n_objects : 1));
    memcpy(s->objects, objects, sizeof(OrObjectData) * n_objects);
    s->inst_data = (OrInstanceData*)malloc(sizeof(OrInstanceData) * (n_inst ? n_inst : 1));
    if (inst_data) memcpy(s->inst_data, inst_data, sizeof(OrInstanceData) * n_inst);
    s->n_heap = n_heap;
    s->heap = (OrHeapEntry*)malloc(sizeof(OrHeapEntry) * (n_heap ? n_heap : 1));
    memcpy(s->heap, heap, sizeof(OrHeapEntry) * n_heap);
    return s;
}

void or_scene_destroy(OrScene* s)
{
    if (!s) return;
    for (uint32_t b = 0; b < s->n_blas; b++) { free(s->blas[b].tris); free(s->blas[b].nodes); }
    free(s->blas); free(s->inst); free(s->w2o); free(s->tl_nodes); free(s->tl_index);
    free(s->objects); free(s->inst_data); free(s->heap); free(s);
}

static int candidate_is_opaque(const OrScene* s, uint32_t inst, uint32_t geom, uint32_t prim, float u, float v)
{
    const OrObjectData* od = &s->objects[s->inst[inst].InstanceID + geom];
    float uv[2][2];
    get_texture_coordinates(od, s->heap, prim, u, v, uv);
    return is_opaque(od, s->heap, uv);
}

/* ======================================================================== */
/* visibility rays: TraceRay<FORCE_NON_OPAQUE | ACCEPT_FIRST_HIT>, IsOpaque (direct lighting overload)          */
/* ======================================================================== */
/* ShadingHelpers.hlsli:117-159: returns 1 when the candidate blocks the ray (commit + end search) */
static int is_opaque_visibility(const OrObjectData* od, const OrHeapEntry* heap, const float uv[2][2], float vis[3])
{
    OrMaterial m = od->Material;
    const OrTextureMapInfo* ti = od->TextureMapInfoArray;
    float t[4];
    evaluate_base_color(m.BaseColor, heap, &ti[TEX_BaseColor], uv);
    if (m.AlphaMode != 0) {
        int ret = m.BaseColor[3] >= m.AlphaCutoff;
        for (int c = 0; c < 3; c++) vis[c] *= ret ? 0.0f : 1.0f;
        return ret;
    }
    if (m.Metallic > 0.0f) {
        if (ti[TEX_MetallicRoughness].Descriptor != ~0u) { sample_map(heap, &ti[TEX_MetallicRoughness], uv, t); m.Metallic *= t[2]; }
        else if (ti[TEX_Metallic].Descriptor != ~0u) { sample_map(heap, &ti[TEX_Metallic], uv, t); m.Metallic *= t[0]; }
        if (m.Metallic == 1.0f) { vis[0] = vis[1] = vis[2] = 0.0f; return 1; }
    }
    if (m.Transmission > 0.0f && ti[TEX_Transmission].Descriptor != ~0u) { sample_map(heap, &ti[TEX_Transmission], uv, t); m.Transmission *= t[0]; }
    for (int c = 0; c < 3; c++) vis[c] *= (1.0f - m.Metallic) * m.BaseColor[c] * m.Transmission;
    return vis[0] == 0.0f && vis[1] == 0.0f && vis[2] == 0.0f;
}

void or_trace_visibility(const OrScene* s, const float* rays, uint32_t count, float* out)
{
    #pragma omp parallel for schedule(dynamic, 64)
    for (int64_t ii = 0; ii < (int64_t)count; ii++) {
        const float* r = rays + 8 * ii;
        f3 o = ld3(r), d = ld3(r + 4); float tmin = r[3], tmax = r[7];
        float vis[3] = { 1.0f, 1.0f, 1.0f }; int committed = 0;
        /* every triangle of every instance (any-hit order is undefined in DXR; brute force, ascending ids) */
        for (uint32_t in = 0; in < s->n_inst && !committed; in++) {
            const OrInstanceDesc* I = &s->inst[in];
            if (!(I->InstanceMask & 0xFFu)) continue;
            const float* W = &s->w2o[12 * in];
            f3 oo = F3(sop3t(W[0], o.x, W[1], o.y, W[2], o.z, W[3]), sop3t(W[4], o.x, W[5], o.y, W[6], o.z, W[7]), sop3t(W[8], o.x, W[9], o.y, W[10], o.z, W[11]));
            f3 od = F3(sop3(W[0], d.x, W[1], d.y, W[2], d.z), sop3(W[4], d.x, W[5], d.y, W[6], d.z), sop3(W[8], d.x, W[9], d.y, W[10], d.z));
            RayObj ro; ray_setup(&ro, oo, od);
            const Blas* B = &s->blas[I->Blas];
            for (uint32_t k = 0; k < B->n_tris && !committed; k++) {
                float t, u, v;
                if (!tri_test(&ro, B->tris[k].v0, B->tris[k].v1, B->tris[k].v2, &t, &u, &v)) continue;
                if (!(t > tmin && t < tmax)) continue;
                const OrObjectData* odt = &s->objects[I->InstanceID + B->tris[k].geom];
                float uv[2][2];
                get_texture_coordinates(odt, s->heap, B->tris[k].prim, u, v, uv);
                if (is_opaque_visibility(odt, s->heap, uv, vis)) committed = 1;
            }
        }
        out[4 * ii] = vis[0]; out[4 * ii + 1] = vis[1]; out[4 * ii + 2] = vis[2]; out[4 * ii + 3] = committed ? 0.0f : 1.0f;
    }
}

/* ======================================================================== */
/* SkeletalMeshSkinning.hlsl:28-62                                           */
/* ======================================================================== */
static inline int16_t pack_r16_snorm(float v) { return (int16_t)(clampf(v, -1.0f, 1.0f) * 32767.0f); }   /* Packing.hlsli:3-6: truncating cast */

void or_skin_mesh(const void* skeletal, const float* transforms, void* vertices, uint16_t* motion, uint32_t count)
{
    for (uint32_t i = 0; i < count; i++) {
        const uint8_t* sv = (const uint8_t*)skeletal + 48 * (size_t)i;
        uint8_t* dv = (uint8_t*)vertices + 32 * (size_t)i;
        float pos[3], wt[4]; int16_t nq[3], tq[3]; uint16_t joints[4];
        memcpy(pos, sv, 12); memcpy(nq, sv + 12, 6); memcpy(tq, sv + 18, 6); memcpy(joints, sv + 24, 8); memcpy(wt, sv + 32, 16);
        const float w[4] = { wt[0], wt[1], wt[2], 1.0f - wt[0] - wt[1] - wt[2] };
        float M[12] = { 0 };
        for (int j = 0; j < 4; j++) for (int k = 0; k < 12; k++) M[k] = mad(w[j], transforms[12 * (size_t)joints[j] + k], M[k]);
        f3 p = F3(sop3t(M[0], pos[0], M[1], pos[1], M[2], pos[2], M[3]), sop3t(M[4], pos[0], M[5], pos[1], M[6], pos[2], M[7]), sop3t(M[8], pos[0], M[9], pos[1], M[10], pos[2], M[11]));
        float old[3]; memcpy(old, dv, 12);
        f3 mv = F3(old[0] - p.x, old[1] - p.y, old[2] - p.z);
        f3 n = F3(unpack_r16_snorm(nq[0]), unpack_r16_snorm(nq[1]), unpack_r16_snorm(nq[2]));
        f3 t = F3(unpack_r16_snorm(tq[0]), unpack_r16_snorm(tq[1]), unpack_r16_snorm(tq[2]));
        f3 r0 = F3(M[0], M[1], M[2]), r1 = F3(M[4], M[5], M[6]), r2 = F3(M[8], M[9], M[10]);
        /* Math::InverseTranspose Math.hlsli:23-27 */
        f3 v = cross3(r0, r1);
        float d = dot3(v, r2);
        f3 i0 = cross3(r1, r2), i1 = cross3(r2, r0);
        i0 = F3(i0.x / d, i0.y / d, i0.z / d); i1 = F3(i1.x / d, i1.y / d, i1.z / d); f3 i2 = F3(v.x / d, v.y / d, v.z / d);
        f3 nn = normalize3(F3(dot3(i0, n), dot3(i1, n), dot3(i2, n)));
        f3 tt = normalize3(F3(dot3(r0, t), dot3(r1, t), dot3(r2, t)));
        float np[3] = { p.x, p.y, p.z };
        memcpy(dv, np, 12);
        int16_t qn[3] = { pack_r16_snorm(nn.x), pack_r16_snorm(nn.y), pack_r16_snorm(nn.z) };
        int16_t qt[3] = { pack_r16_snorm(tt.x), pack_r16_snorm(tt.y), pack_r16_snorm(tt.z) };
        memcpy(dv + 12, qn, 6); memcpy(dv + 18, qt, 6);
        motion[4 * (size_t)i + 0] = or_f32_to_f16(mv.x); motion[4 * (size_t)i + 1] = or_f32_to_f16(mv.y); motion[4 * (size_t)i + 2] = or_f32_to_f16(mv.z);
    }
}

/* ======================================================================== */
/* CastRay (RaytracingHelpers.hlsli:57-133)                                  */
/* ======================================================================== */
typedef struct { f3 Origin, Direction; float TMin, TMax; } RayDesc;

static int cast_ray(const OrScene* s, const RayDesc* ray, HitInfo* h)
{
    h->Position = add3(ray->Origin, scl3(ray->Direction, 1e8f));
    h->Distance = INFINITY;
    Committed c = trace_ray(s, ray->Origin, ray->Direction, ray->TMin, ray->TMax);
    if (!c.hit) return 0;
    const OrInstanceDesc* in = &s->inst[c.inst];
    h->Distance = c.t;
    h->InstanceIndex = c.inst;
    h->ObjectIndex = in->InstanceID + c.geom;            /* :79 */
    h->PrimitiveIndex = c.prim;
    const OrObjectData* od = &s->objects[h->ObjectIndex];
    const OrHeapEntry* vb = &s->heap[od->MeshDescriptors.Vertices];
    const OrHeapEntry* ib = &s->heap[od->MeshDescriptors.Indices];
    uint32_t idx[3];
    for (int k = 0; k < 3; k++) idx[k] = load_index(ib->Ptr, ib->Stride, 3 * c.prim + k);   /* MeshHelpers.hlsli:5-9 */
    float pos[9];
    const uint8_t* vbase = (const uint8_t*)vb->Ptr;
    uint32_t stride = od->VertexDesc.Stride;
    for (int k = 0; k < 3; k++) memcpy(&pos[3 * k], vbase + (size_t)stride * idx[k], 12); /* Vertex.hlsli:14-24 */
    float bary[2] = { c.u, c.v };
    float op[3], wp[3], on[3], wn[3], off;
    or_safe_spawn(pos, bary, in->Transform, &s->w2o[12 * c.inst], op, wp, on, wn, &off);   /* HitInfo.hlsli:24-35 */
    h->ObjectPosition = ld3(op); h->Position = ld3(wp); h->FlatNormal = ld3(wn); h->PositionOffset = off;
    h->Bary[0] = c.u; h->Bary[1] = c.v;
    if (od->VertexDesc.Normal != ~0u) {                  /* HitInfo.hlsli:52-65 */
        f3 nrm[3];
        for (int k = 0; k < 3; k++) {
            int16_t q[3]; memcpy(q, vbase + (size_t)stride * idx[k] + od->VertexDesc.Normal, 6);
            nrm[k] = F3(unpack_r16_snorm(q[0]), unpack_r16_snorm(q[1]), unpack_r16_snorm(q[2]));
        }
        /* Vertex::Interpolate: a0 + b.x*(a1-a0) + b.y*(a2-a0) */
        f3 n = interp3(nrm[0], nrm[1], nrm[2], c.u, c.v);
        const float* W = &s->w2o[12 * c.inst];
        f3 g = F3(sop3(W[0], n.x, W[4], n.y, W[8], n.z), sop3(W[1], n.x, W[5], n.y, W[9], n.z), sop3(W[2], n.x, W[6], n.y, W[10], n.z));   /* RotateVectorInverse((float3x3)worldToObject, n) */
        h->GeometricNormal = normalize3(g);
    } else {                                             /* HitInfo.hlsli:37-50 */
        h->GeometricNormal = h->FlatNormal;
    }
    h->ShadingNormal = h->GeometricNormal;
    h->IsFrontFace = dot3(h->GeometricNormal, ray->Direction) < 0.0f;
    if (!h->IsFrontFace) h->ShadingNormal = neg3(h->ShadingNormal);
    h->Tangent = F3(0, 0, 0);                            /* :115-122 */
    if (od->VertexDesc.Tangent != ~0u) {
        f3 tg[3];
        for (int k = 0; k < 3; k++) {
            int16_t q[3]; memcpy(q, vbase + (size_t)stride * idx[k] + od->VertexDesc.Tangent, 6);
            tg[k] = F3(unpack_r16_snorm(q[0]), unpack_r16_snorm(q[1]), unpack_r16_snorm(q[2]));
        }
        f3 t = interp3(tg[0], tg[1], tg[2], c.u, c.v);
        const float* M = in->Transform;
        f3 w = F3(sop3(M[0], t.x, M[1], t.y, M[2], t.z), sop3(M[4], t.x, M[5], t.y, M[6], t.z), sop3(M[8], t.x, M[9], t.y, M[10], t.z));
        h->Tangent = normalize3(w);
    }
    get_texture_coordinates(od, s->heap, c.prim, c.u, c.v, h->TextureCoordinates);       /* :124-130 */
    return 1;
}

/* ShadingHelpers.hlsli:11-30 (no environment texture: SURVEY 8f "next") */
static f3 environment_light_color(const OrScene* s, const OrSceneData* sd, f3 dir)
{
    if (sd->EnvironmentLightTextureDescriptor != ~0u) {
        const float* M = sd->EnvironmentLightTransform;
        f3 w = normalize3(F3(sop3(M[0], dir.x, M[1], dir.y, M[2], dir.z), sop3(M[4], dir.x, M[5], dir.y, M[6], dir.z), sop3(M[8], dir.x, M[9], dir.y, M[10], dir.z)));
        const OrHeapEntry* t = &s->heap[sd->EnvironmentLightTextureDescriptor];
        float out[4];
        if (sd->IsEnvironmentLightTextureCubeMap) { float d[3] = { w.x, w.y, w.z }; or_cube_sample(t, d, out); }
        else {                                              /* Math::ToLatLongCoordinate Math.hlsli:29-33 */
            const float Pi = 3.14159265358979323846f;
            or_texture_sample(t, (1.0f + atan2f(w.x, w.z) / Pi) / 2.0f, acosf(w.y) / Pi, out);
        }
        return F3(out[0], out[1], out[2]);
    }
    if (sd->EnvironmentLightColor[3] >= 0.0f) return ld3(sd->EnvironmentLightColor);
    float t = (dir.y + 1.0f) * 0.5f;
    return F3(ml_from_srgb1(1.0f + t * (0.5f - 1.0f)), ml_from_srgb1(1.0f + t * (0.7f - 1.0f)), ml_from_srgb1(1.0f + t * (1.0f - 1.0f)));
}

/* Camera.hlsli:27-41, Math.hlsli:7-15 */
static RayDesc generate_pinhole_ray(const OrCamera* cam, uint32_t px, uint32_t py, uint32_t W, uint32_t H, float uv[2])
{
    float u = ((float)px + 0.5f + cam->Jitter[0]) / (float)W;
    float v = ((float)py + 0.5f + cam->Jitter[1]) / (float)H;
    uv[0] = u; uv[1] = v;
    float nx = u * 2.0f + -1.0f, ny = v * -2.0f + 1.0f;
    f3 R = ld3(cam->RightDirection), U = ld3(cam->UpDirection), F = ld3(cam->ForwardDirection);
    f3 d = F3(mad(ny, U.x, mad(nx, R.x, F.x)), mad(ny, U.y, mad(nx, R.y, F.y)), mad(ny, U.z, mad(nx, R.z, F.z)));
    RayDesc r;
    r.Origin = ld3(cam->Position);
    r.Direction = normalize3(d);
    float invCos = 1.0f / dot3(normalize3(F), r.Direction);
    r.TMin = cam->NearDepth * invCos;
    r.TMax = cam->FarDepth * invCos;
    return r;
}

/* row-vector transform by an XMFLOAT4X4: out_j = p.x*M[0][j] + p.y*M[1][j] + p.z*M[2][j] + M[3][j] */
static void xform4(const float M[16], f3 p, float out[4])
{
    for (int j = 0; j < 4; j++) out[j] = sop3t(p.x, M[j], p.y, M[4 + j], p.z, M[8 + j], M[12 + j]);
}

static inline f3 material_emission(const OrMaterial* m) { return scl3(ld3(m->EmissiveColor), m->EmissiveStrength); }

/* ======================================================================== */
/* GBufferGeneration.hlsl:116-232                                            */
/* ======================================================================== */
uint64_t or_gbuffer_render(const OrScene* s, const OrCamera* cam, const OrSceneData* sd,
                           const OrGBufferConstants* k, const OrGBufferTextures* tx,
                           uint32_t y0, uint32_t y1, int n_threads)
{
    const uint32_t W = k->RenderSize[0], H = k->RenderSize[1], flags = k->Flags;
    if (y1 > H) y1 = H;
    (void)n_threads;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
    /* work items are 64-pixel pieces of a scanline, not whole scanlines: a bounded slab of rows (bench.py cpu_baseline) has fewer rows
     * than a big host has threads. Pixels are independent, the order changes nothing. */
    const int64_t cpr = ((int64_t)W + 63) / 64, nchunks = (y1 > y0 ? (int64_t)(y1 - y0) : 0) * cpr;
    #pragma omp parallel for schedule(dynamic, 1)
    for (int64_t cc = 0; cc < nchunks; cc++) {
        const uint32_t y = y0 + (uint32_t)(cc / cpr), x0 = (uint32_t)(cc % cpr) * 64u, x1 = x0 + 64u < W ? x0 + 64u : W;
        for (uint32_t x = x0; x < x1; x++) {
            size_t pi = (size_t)y * W + x;
            float Position[4] = { INFINITY, INFINITY, INFINITY, INFINITY };
            float LinearDepth = INFINITY, NormalizedDepth = cam->IsNormalizedDepthReversed ? 0.0f : 1.0f;
            float uv[2];
            RayDesc ray = generate_pinhole_ray(cam, x, y, W, H, uv);
            HitInfo h; memset(&h, 0, sizeof h);
            if (cast_ray(s, &ray, &h)) {
                if (flags & OR_GB_Geometry) {
                    Position[0] = h.Position.x; Position[1] = h.Position.y; Position[2] = h.Position.z; Position[3] = h.PositionOffset;
                    float fn[3] = { h.FlatNormal.x, h.FlatNormal.y, h.FlatNormal.z }, gn[3] = { h.GeometricNormal.x, h.GeometricNormal.y, h.GeometricNormal.z }, e[2];
                    if ((flags & OR_GB_FlatNormal) && tx->FlatNormal) {
                        or_oct_encode(fn, e);
                        tx->FlatNormal[2 * pi] = or_f32_to_snorm16(e[0]); tx->FlatNormal[2 * pi + 1] = or_f32_to_snorm16(e[1]);
                    }
                    if ((flags & OR_GB_GeometricNormal) && tx->GeometricNormal) {
                        or_oct_encode(gn, e);
                        tx->GeometricNormal[2 * pi] = or_f32_to_snorm16(e[0]); tx->GeometricNormal[2 * pi + 1] = or_f32_to_snorm16(e[1]);
                    }
                    float proj[4]; xform4(cam->WorldToProjection, h.Position, proj);
                    LinearDepth = proj[3];
                    NormalizedDepth = proj[2] / proj[3];
                    if ((flags & OR_GB_MotionVector) && tx->MotionVector) {
                        /* CalculateMotionVector :62-91; static scenes / no per-vertex motion buffers */
                        f3 prev = h.Position;
                        if (!sd->IsStatic) {
                            const float* P = s->inst_data[h.InstanceIndex].PreviousObjectToWorld; f3 q = h.ObjectPosition;
                            const OrMeshDescriptors* md = &s->objects[h.ObjectIndex].MeshDescriptors;
                            if (md->MotionVectors != ~0u) {                 /* :73-84, StructuredBuffer<float16_t4> */
                                const uint16_t* mvb = (const uint16_t*)s->heap[md->MotionVectors].Ptr;
                                const OrHeapEntry* ib = &s->heap[md->Indices];
                                f3 m3[3];
                                for (int kk = 0; kk < 3; kk++) {
                                    uint32_t vi = load_index(ib->Ptr, ib->Stride, 3 * h.PrimitiveIndex + kk);
                                    m3[kk] = F3(or_f16_to_f32(mvb[4 * vi]), or_f16_to_f32(mvb[4 * vi + 1]), or_f16_to_f32(mvb[4 * vi + 2]));
                                }
                                f3 mi = interp3(m3[0], m3[1], m3[2], h.Bary[0], h.Bary[1]);
                                q = add3(q, mi);
                            }
                            prev = F3(sop3t(P[0], q.x, P[1], q.y, P[2], q.z, P[3]), sop3t(P[4], q.x, P[5], q.y, P[6], q.z, P[7]), sop3t(P[8], q.x, P[9], q.y, P[10], q.z, P[11]));
                        }
                        float clip[4], view[4];
                        xform4(cam->PreviousWorldToProjection, prev, clip);
                        xform4(cam->PreviousWorldToView, prev, view);
                        float su = (clip[0] / clip[3]) * 0.5f + 0.5f, sv = (clip[1] / clip[3]) * -0.5f + 0.5f;
                        tx->MotionVector[4 * pi + 0] = or_f32_to_f16((su - uv[0]) * (float)W);
                        tx->MotionVector[4 * pi + 1] = or_f32_to_f16((sv - uv[1]) * (float)H);
                        tx->MotionVector[4 * pi + 2] = or_f32_to_f16(view[2] - LinearDepth);
                        tx->MotionVector[4 * pi + 3] = 0;
                    }
                }
                BSDFSample bs; memset(&bs, 0, sizeof bs);
                if (flags & OR_GB_Material) {
                    f3 frontT = h.IsFrontFace ? h.Tangent : neg3(h.Tangent);             /* GetFrontTangent */
                    const OrMaterial mm = evaluate_material(&h.ShadingNormal, frontT, &s->objects[h.ObjectIndex], s->heap, h.TextureCoordinates);
                    const OrMaterial* m = &mm;
                    bsdf_init(&bs, ld3(m->BaseColor), m->Metallic, m->Roughness, m->IOR, m->Transmission, h.IsFrontFace);
                    if (tx->BaseColorMetalness) {
                        tx->BaseColorMetalness[4 * pi + 0] = or_f32_to_unorm8(bs.BaseColor.x);
                        tx->BaseColorMetalness[4 * pi + 1] = or_f32_to_unorm8(bs.BaseColor.y);
                        tx->BaseColorMetalness[4 * pi + 2] = or_f32_to_unorm8(bs.BaseColor.z);
                        tx->BaseColorMetalness[4 * pi + 3] = or_f32_to_unorm8(bs.Metallic);
                    }
                    if (flags & OR_GB_Albedo) {
                        /* BSDFSample::EstimateDemodulationFactors (BxDF.hlsli:317-320) -> NRD_MaterialFactors. [NRD spec] NRD is an
                         * un-vendored submodule (version unpinned); restated from its published form:
                         * Fenv = EnvironmentTerm_Rtg(Rf0, |N.V|, roughness); diffuse = (1 - Fenv) * albedo * 0.99 + 0.01;
                         * specular = Fenv * 0.99 + 0.01. float3 stores to RGBA16F write 0 to alpha. */
                        const f3 V = neg3(ray.Direction);
                        const float NoV = fabsf(dot3(h.ShadingNormal, V));
                        float F0[3] = { bs.F0.x, bs.F0.y, bs.F0.z }, Fe[3];
                        or_env_term_rtg(F0, NoV, bs.Roughness, Fe);
                        const float al[3] = { bs.Albedo.x, bs.Albedo.y, bs.Albedo.z };
                        for (int c = 0; c < 3; c++) {
                            if ((flags & OR_GB_DiffuseAlbedo) && tx->DiffuseAlbedo) tx->DiffuseAlbedo[4 * pi + c] = or_f32_to_f16((1.0f - Fe[c]) * al[c] * 0.99f + 0.01f);
                            if ((flags & OR_GB_SpecularAlbedo) && tx->SpecularAlbedo) tx->SpecularAlbedo[4 * pi + c] = or_f32_to_f16(Fe[c] * 0.99f + 0.01f);
                        }
                        if ((flags & OR_GB_DiffuseAlbedo) && tx->DiffuseAlbedo) tx->DiffuseAlbedo[4 * pi + 3] = 0;
                        if ((flags & OR_GB_SpecularAlbedo) && tx->SpecularAlbedo) tx->SpecularAlbedo[4 * pi + 3] = 0;
                    }
                    if (tx->IOR) tx->IOR[pi] = or_f32_to_f16(m->IOR);
                    if (bs.Metallic < 1.0f && tx->Transmission) tx->Transmission[pi] = or_f32_to_unorm8(bs.Transmission);
                    if ((flags & OR_GB_Radiance) && tx->Radiance) {
                        f3 e = material_emission(m);
                        tx->Radiance[4 * pi + 0] = or_f32_to_f16(e.x); tx->Radiance[4 * pi + 1] = or_f32_to_f16(e.y);
                        tx->Radiance[4 * pi + 2] = or_f32_to_f16(e.z); tx->Radiance[4 * pi + 3] = 0;
                    }
                }
                if ((flags & OR_GB_NormalRoughness) && tx->NormalRoughness) {
                    tx->NormalRoughness[4 * pi + 0] = or_f32_to_snorm16(h.ShadingNormal.x);
                    tx->NormalRoughness[4 * pi + 1] = or_f32_to_snorm16(h.ShadingNormal.y);
                    tx->NormalRoughness[4 * pi + 2] = or_f32_to_snorm16(h.ShadingNormal.z);
                    tx->NormalRoughness[4 * pi + 3] = or_f32_to_snorm16((flags & OR_GB_Material) ? bs.Roughness : 0.0f);
                }
            } else {
                if ((flags & OR_GB_MotionVector) && tx->MotionVector) {
                    float proj[4]; xform4(cam->WorldToProjection, h.Position, proj);
                    float clip[4], view[4];
                    xform4(cam->PreviousWorldToProjection, h.Position, clip);
                    xform4(cam->PreviousWorldToView, h.Position, view);
                    float su = (clip[0] / clip[3]) * 0.5f + 0.5f, sv = (clip[1] / clip[3]) * -0.5f + 0.5f;
                    tx->MotionVector[4 * pi + 0] = or_f32_to_f16((su - uv[0]) * (float)W);
                    tx->MotionVector[4 * pi + 1] = or_f32_to_f16((sv - uv[1]) * (float)H);
                    tx->MotionVector[4 * pi + 2] = or_f32_to_f16(view[2] - proj[3]);
                    tx->MotionVector[4 * pi + 3] = 0;
                }
                if ((flags & OR_GB_Radiance) && tx->Radiance) {
                    f3 e = environment_light_color(s, sd, ray.Direction);
                    tx->Radiance[4 * pi + 0] = or_f32_to_f16(e.x); tx->Radiance[4 * pi + 1] = or_f32_to_f16(e.y);
                    tx->Radiance[4 * pi + 2] = or_f32_to_f16(e.z); tx->Radiance[4 * pi + 3] = 0;
                }
            }
            if ((flags & OR_GB_Position) && tx->Position) memcpy(&tx->Position[4 * pi], Position, 16);
            if ((flags & OR_GB_LinearDepth) && tx->LinearDepth) tx->LinearDepth[pi] = LinearDepth;
            if ((flags & OR_GB_NormalizedDepth) && tx->NormalizedDepth) tx->NormalizedDepth[pi] = NormalizedDepth;
        }
    }
    return (uint64_t)W * (y1 > y0 ? y1 - y0 : 0);
}

/* ======================================================================== */
/* Raytracing.hlsl:103-415, DEFAULT permutation, Denoiser::None, DI off      */
/* ======================================================================== */
uint64_t or_raytrace_render(const OrScene* s, const OrCamera* cam, const OrSceneData* sd,
                            const OrGraphicsSettings* gs, const OrGBufferTextures* tx,
                            uint32_t y0, uint32_t y1, int n_threads)
{
    const uint32_t W = gs->RenderSize[0], H = gs->RenderSize[1];
    if (y1 > H) y1 = H;
    uint64_t total_rays = 0;
    (void)n_threads;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
    const int64_t cpr = ((int64_t)W + 63) / 64, nchunks = (y1 > y0 ? (int64_t)(y1 - y0) : 0) * cpr;    /* 64-pixel pieces of a scanline, as in or_gbuffer_render */
    #pragma omp parallel for schedule(dynamic, 1) reduction(+ : total_rays)
    for (int64_t cc = 0; cc < nchunks; cc++) {
        const uint32_t y = y0 + (uint32_t)(cc / cpr), x0 = (uint32_t)(cc % cpr) * 64u, x1 = x0 + 64u < W ? x0 + 64u : W;
        for (uint32_t x = x0; x < x1; x++) {
            size_t pi = (size_t)y * W + x;
            uint32_t rng = or_rng_init(x, y, gs->FrameIndex);                           /* :108 */
            float uv[2];
            RayDesc primaryRay = generate_pinhole_ray(cam, x, y, W, H, uv);            /* :110-126 */
            const float* position = &tx->Position[4 * pi];                              /* :118 */
            f3 primaryRadiance = F3(or_f16_to_f32(tx->Radiance[4 * pi]), or_f16_to_f32(tx->Radiance[4 * pi + 1]), or_f16_to_f32(tx->Radiance[4 * pi + 2]));
            int isPrimaryHit = isfinite(position[3]);                                   /* :127 */
            if (!isPrimaryHit) continue;                                                /* :241-252 (bounce 0 miss: return) */
            HitInfo ph; memset(&ph, 0, sizeof ph);
            BSDFSample pb;
            {
                float nr[4];
                for (int c = 0; c < 4; c++) nr[c] = or_snorm16_to_f32(tx->NormalRoughness[4 * pi + c]);
                float fe[2] = { or_snorm16_to_f32(tx->FlatNormal[2 * pi]), or_snorm16_to_f32(tx->FlatNormal[2 * pi + 1]) };
                float ge[2] = { or_snorm16_to_f32(tx->GeometricNormal[2 * pi]), or_snorm16_to_f32(tx->GeometricNormal[2 * pi + 1]) };
                float fn[3], gn[3];
                or_oct_decode(fe, fn); or_oct_decode(ge, gn);
                ph.Position = ld3(position); ph.PositionOffset = position[3];           /* HitInfo.hlsli:67-79 */
                ph.FlatNormal = ld3(fn); ph.GeometricNormal = ld3(gn); ph.ShadingNormal = ld3(nr);
                ph.IsFrontFace = dot3(ph.GeometricNormal, primaryRay.Direction) < 0.0f;
                f3 dp = sub3(ph.Position, ld3(cam->Position));
                ph.Distance = sqrtf(dot3(dp, dp));                                      /* :138 */
                float bcm[4];
                for (int c = 0; c < 4; c++) bcm[c] = unorm8_to_f32(tx->BaseColorMetalness[4 * pi + c]);
                float ior = or_f16_to_f32(tx->IOR[pi]);
                float tr = bcm[3] < 1.0f ? unorm8_to_f32(tx->Transmission[pi]) : 0.0f;  /* :146 */
                bsdf_init(&pb, ld3(bcm), bcm[3], nr[3], ior, tr, ph.IsFrontFace);
            }
            f3 radiance = F3(0, 0, 0);
            int isDiffuse = 1; float hitDistance = INFINITY;                             /* :188-189 */
            const uint32_t spp = gs->SamplesPerPixel;
            for (uint32_t sample = 0; sample < spp; sample++) {                         /* :191 */
                RayDesc ray = primaryRay;
                int isHit = 1;
                HitInfo hit = ph;
                f3 emission = primaryRadiance;
                BSDFSample bs = pb;
                int lobe = 0;
                f3 L = F3(0, 0, 0), throughput = F3(1, 1, 1), sampleRadiance = F3(0, 0, 0);
                for (uint32_t bounce = 0; bounce <= gs->Bounces; bounce++) {            /* :213 */
                    if (bounce) {
                        ray.Origin = safe_world_ray_origin(&hit, L);                    /* :221-224 */
                        ray.Direction = L; ray.TMin = 0.0f; ray.TMax = INFINITY;
                        isHit = cast_ray(s, &ray, &hit);
                        total_rays++;
                    }
                    if (!sample && bounce == 1) { isDiffuse = lobe == LOBE_DIFFUSE; hitDistance = hit.Distance; }   /* :235-239 */
                    if (!isHit) {                                                       /* :241-259 */
                        f3 env = environment_light_color(s, sd, ray.Direction);
                        sampleRadiance = F3(mad(throughput.x, env.x, sampleRadiance.x), mad(throughput.y, env.y, sampleRadiance.y), mad(throughput.z, env.z, sampleRadiance.z));
                        break;
                    }
                    if (bounce) {                                                       /* :293-304 */
                        f3 frontT = hit.IsFrontFace ? hit.Tangent : neg3(hit.Tangent);
                        const OrMaterial mm = evaluate_material(&hit.ShadingNormal, frontT, &s->objects[hit.ObjectIndex], s->heap, hit.TextureCoordinates);
                        const OrMaterial* m = &mm;
                        emission = material_emission(m);
                        bsdf_init(&bs, ld3(m->BaseColor), m->Metallic, m->Roughness, m->IOR, m->Transmission, hit.IsFrontFace);
                    }
                    sampleRadiance = F3(mad(throughput.x, emission.x, sampleRadiance.x), mad(throughput.y, emission.y, sampleRadiance.y), mad(throughput.z, emission.z, sampleRadiance.z));  /* :320 */
                    SurfaceVectors sv = surface_vectors(hit.IsFrontFace, hit.GeometricNormal, hit.ShadingNormal);
                    f3 V = neg3(ray.Direction);
                    float w[3];
                    compute_lobe_weights(&bs, &sv, V, gs->ExtFlags, w);
                    float rnd[4];
                    for (int c = 0; c < 4; c++) rnd[c] = or_rng_float(&rng);            /* GetFloat4: x,y,z,w in order */
                    if (!bsdf_sample(&bs, &sv, V, w, rnd, &L, &lobe)) break;            /* :330-333 */
                    float pdf = bsdf_pdf_lobe(&bs, &sv, L, V, w, lobe);
                    if (pdf == 0.0f) break;                                             /* :336 */
                    f3 f = bsdf_eval_lobe(&bs, &sv, L, V, w, lobe, gs->ExtFlags);
                    if (f.x == 0.0f && f.y == 0.0f && f.z == 0.0f) break;               /* :342 */
                    { /* :346. float3 / float is ONE IEEE reciprocal and three products (arithmetic spec, DESIGN.md section 1: HLSL does not pin
                       * `currentThroughput / PDF` to three divisions either) */
                        const float ipdf = 1.0f / pdf;
                        throughput = mul3(throughput, F3(f.x * ipdf, f.y * ipdf, f.z * ipdf));
                    }
                    if (gs->IsRussianRouletteEnabled && bounce > 3) {                   /* :348-356 */
                        float p = fmaxf(throughput.x, fmaxf(throughput.y, throughput.z));
                        if (or_rng_float(&rng) >= p) break;
                        { const float ip = 1.0f / p; throughput = F3(throughput.x * ip, throughput.y * ip, throughput.z * ip); }   /* :355, same rule */
                    }
                    if (ml_luminance(throughput) <= gs->ThroughputThreshold) break;     /* :361 */
                }
                radiance = add3(radiance, sampleRadiance);                              /* :372 */
            }
            if (finite3(radiance)) { float n = (float)spp; radiance = F3(radiance.x / n, radiance.y / n, radiance.z / n); }
            else radiance = F3(0, 0, 0);                                                /* :377 */
            if (gs->Denoiser == 2 || gs->Denoiser == 3) {                               /* NRD ReBLUR / ReLAX, :400-413 (direct terms 0: DI off) */
                f3 ind = F3(fmaxf(radiance.x - primaryRadiance.x, 0.0f), fmaxf(radiance.y - primaryRadiance.y, 0.0f), fmaxf(radiance.z - primaryRadiance.z, 0.0f));
                uint16_t packed[4] = { or_f32_to_f16(ind.x), or_f32_to_f16(ind.y), or_f32_to_f16(ind.z), or_f32_to_f16(hitDistance) };
                uint16_t zero[4] = { 0, 0, 0, 0 };
                if (tx->Diffuse) memcpy(&tx->Diffuse[4 * pi], isDiffuse ? packed : zero, 8);
                if (tx->Specular) memcpy(&tx->Specular[4 * pi], isDiffuse ? zero : packed, 8);
                continue;
            }
            if (gs->Denoiser == 1 && !isDiffuse && isfinite(hitDistance) && tx->SpecularHitDistance)    /* DLSS-RR, :395-398 */
                tx->SpecularHitDistance[pi] = or_f32_to_f16(hitDistance);
            tx->Radiance[4 * pi + 0] = or_f32_to_f16(radiance.x);                       /* :385 / :393, RGBA16F store */
            tx->Radiance[4 * pi + 1] = or_f32_to_f16(radiance.y);
            tx->Radiance[4 * pi + 2] = or_f32_to_f16(radiance.z);
            tx->Radiance[4 * pi + 3] = 0;
            if (tx->RadianceF32) {
                tx->RadianceF32[4 * pi + 0] = radiance.x; tx->RadianceF32[4 * pi + 1] = radiance.y;
                tx->RadianceF32[4 * pi + 2] = radiance.z; tx->RadianceF32[4 * pi + 3] = 0.0f;
            }
        }
    }
    return total_rays;
}
