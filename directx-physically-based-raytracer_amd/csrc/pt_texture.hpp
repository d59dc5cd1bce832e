// pt_texture.hpp -- texture taps and material evaluation (device code, gfx950).
//
//   Sample<T> / EvaluateBaseColor / EvaluateTransmission / PerturbNormal / IsOpaque / EvaluateMaterial
//       Shaders/ShadingHelpers.hlsli:53-235
//   GetTextureCoordinates  Shaders/ShadingHelpers.hlsli:32-51
// Every tap is SampleLevel(g_anisotropicSampler, uv, 0) with the root-signature default static sampler
// (Shaders/Raytracing.hlsl:79): WRAP addressing, mip 0, bilinear. [spec] fp32 weights, sRGB decoded per
// texel through a 256-entry table built on the host in double precision (DESIGN.md "Arithmetic spec").
#pragma once
#include "pt_math.hpp"
#include "../../include/ptamd.h"

namespace pt {

struct HeapEntry { const void* ptr; uint64_t bytes; uint32_t stride; uint32_t kind; };   // textures: bytes = w | h << 32, stride = format
enum : uint32_t { kKindBuffer = 0, kKindTexture2D = 1, kKindTextureCube = 2 };
enum : uint32_t { kFmtRGBA8 = 0, kFmtRGBA8Srgb = 1, kFmtRGBA32F = 2 };

struct f4 { float x, y, z, w; };

// Heap pointers come out of a table in memory, so the compiler cannot tell which address space they point into and would emit
// flat_* accesses -- which wait for every outstanding LDS and memory operation of the wave. They are device (global) memory by
// the contract of pt_heap_set_*: say so.
#define PT_GLOBAL_AS __attribute__((address_space(1)))
template <typename T> PT_DEV const PT_GLOBAL_AS T* gptr(const void* p) { return (const PT_GLOBAL_AS T*)p; }

PT_DEV f4 texel_fetch(const HeapEntry& t, const float* srgbLut, uint32_t face, uint32_t x, uint32_t y)
{
    const uint32_t W = (uint32_t)(t.bytes & 0xFFFFFFFFu), H = (uint32_t)(t.bytes >> 32);
    const size_t idx = ((size_t)face * H + y) * W + x;
    f4 o;
    if (t.stride == kFmtRGBA32F) {
        const PT_GLOBAL_AS float* v = gptr<float>(t.ptr) + 4 * idx;
        o.x = v[0]; o.y = v[1]; o.z = v[2]; o.w = v[3];
        return o;
    }
    const uint32_t pw = gptr<uint32_t>(t.ptr)[idx];
    const uchar4 p = make_uchar4((uint8_t)(pw & 0xFFu), (uint8_t)((pw >> 8) & 0xFFu), (uint8_t)((pw >> 16) & 0xFFu), (uint8_t)(pw >> 24));
    if (t.stride == kFmtRGBA8Srgb) { o.x = srgbLut[p.x]; o.y = srgbLut[p.y]; o.z = srgbLut[p.z]; }
    else { o.x = unorm8_to_f32(p.x); o.y = unorm8_to_f32(p.y); o.z = unorm8_to_f32(p.z); }
    o.w = unorm8_to_f32(p.w);
    return o;
}

PT_DEV f4 bilinear(const HeapEntry& t, const float* srgbLut, uint32_t face, float fx, float fy, bool wrap)
{
    const int W = (int)(t.bytes & 0xFFFFFFFFu), H = (int)(t.bytes >> 32);
    const float x0f = floorf(fx), y0f = floorf(fy);
    const float wx = fx - x0f, wy = fy - y0f;
    int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    if (wrap) {
        x0 = ((x0 % W) + W) % W; x1 = ((x1 % W) + W) % W; y0 = ((y0 % H) + H) % H; y1 = ((y1 % H) + H) % H;
    } else {
        x0 = x0 < 0 ? 0 : (x0 >= W ? W - 1 : x0); x1 = x1 < 0 ? 0 : (x1 >= W ? W - 1 : x1);
        y0 = y0 < 0 ? 0 : (y0 >= H ? H - 1 : y0); y1 = y1 < 0 ? 0 : (y1 >= H ? H - 1 : y1);
    }
    const f4 c00 = texel_fetch(t, srgbLut, face, (uint32_t)x0, (uint32_t)y0), c10 = texel_fetch(t, srgbLut, face, (uint32_t)x1, (uint32_t)y0);
    const f4 c01 = texel_fetch(t, srgbLut, face, (uint32_t)x0, (uint32_t)y1), c11 = texel_fetch(t, srgbLut, face, (uint32_t)x1, (uint32_t)y1);
    f4 o;
    { float top = mad(c10.x, wx, c00.x * (1.0f - wx)), bot = mad(c11.x, wx, c01.x * (1.0f - wx)); o.x = mad(bot, wy, top * (1.0f - wy)); }
    { float top = mad(c10.y, wx, c00.y * (1.0f - wx)), bot = mad(c11.y, wx, c01.y * (1.0f - wx)); o.y = mad(bot, wy, top * (1.0f - wy)); }
    { float top = mad(c10.z, wx, c00.z * (1.0f - wx)), bot = mad(c11.z, wx, c01.z * (1.0f - wx)); o.z = mad(bot, wy, top * (1.0f - wy)); }
    { float top = mad(c10.w, wx, c00.w * (1.0f - wx)), bot = mad(c11.w, wx, c01.w * (1.0f - wx)); o.w = mad(bot, wy, top * (1.0f - wy)); }
    return o;
}

PT_DEV f4 texture_sample(const HeapEntry& t, const float* srgbLut, float u, float v)
{
    const float W = (float)(uint32_t)(t.bytes & 0xFFFFFFFFu), H = (float)(uint32_t)(t.bytes >> 32);
    if (!(u == u) || !isfinite(u)) u = 0.0f;
    if (!(v == v) || !isfinite(v)) v = 0.0f;
    u = u - floorf(u); v = v - floorf(v);
    return bilinear(t, srgbLut, 0, u * W - 0.5f, v * H - 0.5f, true);
}

PT_DEV f4 cube_sample(const HeapEntry& t, const float* srgbLut, v3 d)
{
    const float W = (float)(uint32_t)(t.bytes & 0xFFFFFFFFu), H = (float)(uint32_t)(t.bytes >> 32);
    const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    float sc, tc, ma; uint32_t face;
    if (ax >= ay && ax >= az) { ma = ax; if (d.x >= 0.0f) { face = 0; sc = -d.z; tc = -d.y; } else { face = 1; sc = d.z; tc = -d.y; } }
    else if (ay >= az) { ma = ay; if (d.y >= 0.0f) { face = 2; sc = d.x; tc = d.z; } else { face = 3; sc = d.x; tc = -d.z; } }
    else { ma = az; if (d.z >= 0.0f) { face = 4; sc = d.x; tc = -d.y; } else { face = 5; sc = -d.x; tc = -d.y; } }
    const float u = (sc / ma + 1.0f) * 0.5f, v = (tc / ma + 1.0f) * 0.5f;
    return bilinear(t, srgbLut, face, u * W - 0.5f, v * H - 0.5f, false);
}

struct TexCoords { float uv[2][2]; };

PT_DEV f4 sample_map(const HeapEntry* heap, const float* srgbLut, const PtTextureMapInfo& info, const TexCoords& tc)
{
    const float* c = tc.uv[info.TextureCoordinateIndex & 1u];
    return texture_sample(heap[info.Descriptor], srgbLut, c[0], c[1]);
}

// The seven texture slots of an object, resolved once per change of (ObjectData, heap) next to its geometry (k_validate_objects): slot k of
// object o at resolved[o * 7 + k] is a copy of the heap entry TextureMapInfoArray[k].Descriptor names -- ptr == nullptr: no texture -- with
// the texture-coordinate set in `kind`. A hit's texture taps then start one dependent load after the object index is known (resolved
// slot -> texels) instead of two (TextureMapInfo -> heap entry -> texels). A view over either source, so that the code below exists once.
struct TextureSlots {
    const PtTextureMapInfo* info; const HeapEntry* heap; const HeapEntry* resolved;       // resolved != nullptr: use it
    PT_DEV bool has(int k) const { return resolved ? resolved[k].ptr != nullptr : info[k].Descriptor != ~0u; }
    PT_DEV f4 sample(int k, const float* srgbLut, const TexCoords& tc) const
    {
        if (resolved) { const HeapEntry e = resolved[k]; const float* c = tc.uv[e.kind & 1u]; return texture_sample(e, srgbLut, c[0], c[1]); }
        return sample_map(heap, srgbLut, info[k], tc);
    }
};
constexpr uint32_t kTextureSlots = 7;

enum : int { TEX_BaseColor = 0, TEX_EmissiveColor, TEX_Metallic, TEX_Roughness, TEX_MetallicRoughness, TEX_Transmission, TEX_Normal };

PT_DEV uint32_t load_index_tex(const void* ib, uint32_t stride, uint32_t i)
{
    return stride == 2 ? (uint32_t)gptr<uint16_t>(ib)[i] : gptr<uint32_t>(ib)[i];
}

// GetTextureCoordinates, ShadingHelpers.hlsli:32-51
PT_DEV void get_texture_coordinates(const PtObjectData* od, const HeapEntry* heap, uint32_t prim, float bu, float bv, TexCoords& tc)
{
    #pragma unroll
    for (int i = 0; i < 2; i++) {
        tc.uv[i][0] = tc.uv[i][1] = 0.0f;
        const uint32_t off = od->VertexDesc.AttributeOffsets.TextureCoordinates[i];
        if (off == ~0u) continue;
        const HeapEntry vb = heap[od->MeshDescriptors.Vertices], ib = heap[od->MeshDescriptors.Indices];
        float a[3][2];
        for (int k = 0; k < 3; k++) {
            const uint32_t idx = load_index_tex(ib.ptr, ib.stride, 3 * prim + k);
            const PT_GLOBAL_AS uint16_t* h = gptr<uint16_t>((const uint8_t*)vb.ptr + (size_t)od->VertexDesc.Stride * idx + off);
            a[k][0] = f16_to_f32(h[0]); a[k][1] = f16_to_f32(h[1]);
        }
        for (int c = 0; c < 2; c++) tc.uv[i][c] = interp1(a[0][c], a[1][c], a[2][c], bu, bv);
    }
}

// IsOpaque (closest-hit overload), ShadingHelpers.hlsli:105-115
PT_DEV bool is_opaque(const PtObjectData* od, const HeapEntry* heap, const float* srgbLut, const TexCoords& tc, const HeapEntry* resolved = nullptr)
{
    float a = od->Material.BaseColor[3];
    const TextureSlots ts{ od->TextureMapInfoArray, heap, resolved };
    const float* bc = od->Material.BaseColor;
    if ((bc[0] > 0.0f || bc[1] > 0.0f || bc[2] > 0.0f || bc[3] > 0.0f) && ts.has(TEX_BaseColor)) a *= ts.sample(TEX_BaseColor, srgbLut, tc).w;
    return a >= od->Material.AlphaCutoff;
}

// IsOpaque (direct-lighting overload with coloured visibility), ShadingHelpers.hlsli:117-159: true = the candidate blocks the ray
PT_DEV bool is_opaque_visibility(const PtObjectData* od, const HeapEntry* heap, const float* srgbLut, const TexCoords& tc, v3& vis)
{
    PtMaterial m = od->Material;
    const PtTextureMapInfo* ti = od->TextureMapInfoArray;
    if ((m.BaseColor[0] > 0.0f || m.BaseColor[1] > 0.0f || m.BaseColor[2] > 0.0f || m.BaseColor[3] > 0.0f) && ti[TEX_BaseColor].Descriptor != ~0u) {
        const f4 t = sample_map(heap, srgbLut, ti[TEX_BaseColor], tc);
        m.BaseColor[0] *= t.x; m.BaseColor[1] *= t.y; m.BaseColor[2] *= t.z; m.BaseColor[3] *= t.w;
    }
    if (m.AlphaMode != 0) {
        const bool ret = m.BaseColor[3] >= m.AlphaCutoff;
        vis = vis * (ret ? 0.0f : 1.0f);
        return ret;
    }
    if (m.Metallic > 0.0f) {
        if (ti[TEX_MetallicRoughness].Descriptor != ~0u) m.Metallic *= sample_map(heap, srgbLut, ti[TEX_MetallicRoughness], tc).z;
        else if (ti[TEX_Metallic].Descriptor != ~0u) m.Metallic *= sample_map(heap, srgbLut, ti[TEX_Metallic], tc).x;
        if (m.Metallic == 1.0f) { vis = V3(0.0f, 0.0f, 0.0f); return true; }
    }
    if (m.Transmission > 0.0f && ti[TEX_Transmission].Descriptor != ~0u) m.Transmission *= sample_map(heap, srgbLut, ti[TEX_Transmission], tc).x;
    vis = V3(vis.x * ((1.0f - m.Metallic) * m.BaseColor[0] * m.Transmission), vis.y * ((1.0f - m.Metallic) * m.BaseColor[1] * m.Transmission),
             vis.z * ((1.0f - m.Metallic) * m.BaseColor[2] * m.Transmission));
    return vis.x == 0.0f && vis.y == 0.0f && vis.z == 0.0f;
}

// EvaluateMaterial, ShadingHelpers.hlsli:161-235. N: shading normal (in/out), T: front tangent.
PT_DEV PtMaterial evaluate_material(v3& N, v3 T, const PtObjectData* od, const HeapEntry* heap, const float* srgbLut, const TexCoords& tc, const HeapEntry* resolved = nullptr)
{
    PtMaterial m = od->Material;
    const TextureSlots ts{ od->TextureMapInfoArray, heap, resolved };
    if ((m.BaseColor[0] > 0.0f || m.BaseColor[1] > 0.0f || m.BaseColor[2] > 0.0f || m.BaseColor[3] > 0.0f) && ts.has(TEX_BaseColor)) {
        const f4 t = ts.sample(TEX_BaseColor, srgbLut, tc);
        m.BaseColor[0] *= t.x; m.BaseColor[1] *= t.y; m.BaseColor[2] *= t.z; m.BaseColor[3] *= t.w;
    }
    const v3 em = V3(m.EmissiveColor) * m.EmissiveStrength;
    if ((em.x > 0.0f || em.y > 0.0f || em.z > 0.0f) && ts.has(TEX_EmissiveColor)) {
        const f4 t = ts.sample(TEX_EmissiveColor, srgbLut, tc);
        m.EmissiveColor[0] *= t.x; m.EmissiveColor[1] *= t.y; m.EmissiveColor[2] *= t.z;
    }
    if (ts.has(TEX_MetallicRoughness)) {
        if (m.Metallic > 0.0f || m.Roughness > 0.0f) {
            const f4 t = ts.sample(TEX_MetallicRoughness, srgbLut, tc);
            m.Metallic *= t.z; m.Roughness *= t.y;
        }
    } else {
        if (m.Metallic > 0.0f && ts.has(TEX_Metallic)) m.Metallic *= ts.sample(TEX_Metallic, srgbLut, tc).x;
        if (m.Roughness > 0.0f && ts.has(TEX_Roughness)) m.Roughness *= ts.sample(TEX_Roughness, srgbLut, tc).x;
    }
    if (m.Metallic < 1.0f) {
        if (m.Transmission > 0.0f && ts.has(TEX_Transmission)) m.Transmission *= ts.sample(TEX_Transmission, srgbLut, tc).x;
    }
    if ((T.x != 0.0f || T.y != 0.0f || T.z != 0.0f) && ts.has(TEX_Normal)) {        // PerturbNormal :89-103
        const f4 t = ts.sample(TEX_Normal, srgbLut, tc);
        const float nx = t.x * 2.0f - 1.0f, ny = t.y * 2.0f - 1.0f;                                 // [MathLib] Geometry::UnpackLocalNormal
        const float nz = ml_sqrt01(1.0f - (nx * nx + ny * ny));
        const v3 Tn = normalize(T - N * dot(N, T));                                                 // Math::CalculateTBN, Math.hlsli:17-21
        const v3 B = cross(N, Tn);
        N = normalize(V3(sop3(Tn.x, nx, B.x, ny, N.x, nz), sop3(Tn.y, nx, B.y, ny, N.y, nz), sop3(Tn.z, nx, B.z, ny, N.z, nz)));
    }
    return m;
}

} // namespace pt
