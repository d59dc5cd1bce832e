// pt_kernels.hip -- the render kernels (gfx950): primary-ray G-buffer and the wavefront path tracer.
//
//   k_gbuffer   <- Shaders/GBufferGeneration.hlsl:116-232 (main) + CastRay, Shaders/RaytracingHelpers.hlsli:57-133
//   k_round (and its two-kernel form k_pt_init / k_shade / k_extend2)  <- Shaders/Raytracing.hlsl:103-415 (RayGeneration,
//                                      DEFAULT permutation, DI off), as wavefront rounds over a path queue:
//        extend  : closest-hit traversal of every queued ray (CastRay's TraceRay part), schedules in pt_trace2.hpp
//        shade   : material + BSDF sample + Russian roulette for the vertex a path sits on
//                  (Raytracing.hlsl:241-364), emits the next ray, regenerates the pixel's next sample in
//                  place when a path ends (RNG state carried over, Raytracing.hlsl:108,191), compacts the
//                  survivors with wave64 ballot + prefix popcount into the output queue
//        k_round : both halves for one tile in one launch, the hit never leaves registers (the product path)
//   k_visibility / k_bsdf_evaluate  <- the shadow-ray TraceRay and all-lobe BSDF evaluate of the direct-lighting bridge
// MFMA is not used: there is no dense contraction on this path.
#include "pt_internal.hpp"

#include <cstring>

namespace pt {

// ---------------------------------------------------------------------------------------------
// shared device helpers
// ---------------------------------------------------------------------------------------------
constexpr int kLdsStackDepth = 12;      // node-group stack entries (8 B) per lane kept in LDS (24 KB per 256-thread block)

struct RayDesc { v3 o, d; float tmin, tmax; };

// Camera::GeneratePinholeRay (Shaders/Camera.hlsli:27-41), Math::CalculateUV/NDC (Shaders/Math.hlsli:7-15)
PT_DEV RayDesc generate_pinhole_ray(const PtCamera& cam, uint32_t px, uint32_t py, uint32_t W, uint32_t H, float& u, float& v)
{
    u = ((float)px + 0.5f + cam.Jitter[0]) / (float)W;
    v = ((float)py + 0.5f + cam.Jitter[1]) / (float)H;
    float nx = u * 2.0f + -1.0f, ny = v * -2.0f + 1.0f;
    v3 R = V3(cam.RightDirection), U = V3(cam.UpDirection), F = V3(cam.ForwardDirection);
    v3 d = V3(mad(ny, U.x, mad(nx, R.x, F.x)), mad(ny, U.y, mad(nx, R.y, F.y)), mad(ny, U.z, mad(nx, R.z, F.z)));
    RayDesc r;
    r.o = V3(cam.Position);
    r.d = normalize(d);
    float invCos = 1.0f / dot(normalize(F), r.d);
    r.tmin = cam.NearDepth * invCos;
    r.tmax = cam.FarDepth * invCos;
    return r;
}

PT_DEV uint32_t global_row(const FrameView& fv, uint32_t localRow)
{
    uint32_t band = localRow / fv.bandHeight, within = localRow - band * fv.bandHeight;
    return (band * fv.rankCount + fv.rankIndex) * fv.bandHeight + within;
}

// GetEnvironmentLightColor (Shaders/ShadingHelpers.hlsli:11-30)
PT_DEV v3 environment_light_color(const SceneView& sv, const PtSceneData& sd, v3 dir)
{
    if (sd.EnvironmentLightTextureDescriptor != ~0u) {
        const float* M = sd.EnvironmentLightTransform;
        const v3 w = normalize(V3(sop3(M[0], dir.x, M[1], dir.y, M[2], dir.z), sop3(M[4], dir.x, M[5], dir.y, M[6], dir.z), sop3(M[8], dir.x, M[9], dir.y, M[10], dir.z)));
        const HeapEntry t = sv.heap[sd.EnvironmentLightTextureDescriptor];
        f4 c;
        if (sd.IsEnvironmentLightTextureCubeMap) c = cube_sample(t, sv.srgbLut, w);
        else c = texture_sample(t, sv.srgbLut, (1.0f + atan2f(w.x, w.z) / kPi) / 2.0f, acosf(w.y) / kPi);   // Math::ToLatLongCoordinate, Math.hlsli:29-33
        return V3(c.x, c.y, c.z);
    }
    if (sd.EnvironmentLightColor[3] >= 0.0f) return V3(sd.EnvironmentLightColor);
    float t = (dir.y + 1.0f) * 0.5f;
    return V3(ml_from_srgb1(1.0f + t * (0.5f - 1.0f)), ml_from_srgb1(1.0f + t * (0.7f - 1.0f)), ml_from_srgb1(1.0f + t * (1.0f - 1.0f)));
}

// row-vector transform by an XMFLOAT4X4 (HLSL mul(M, float4(p,1)) on the column-major view of it)
PT_DEV void xform4(const float* M, v3 p, float out[4])
{
    for (int j = 0; j < 4; j++) out[j] = sop3t(p.x, M[j], p.y, M[4 + j], p.z, M[8 + j], M[12 + j]);
}

struct SurfaceHit {               // the part of HitInfo (Shaders/HitInfo.hlsli:7-22) this path consumes
    v3 Position, ObjectPosition; float PositionOffset;
    v3 FlatNormal, GeometricNormal, ShadingNormal, Tangent;
    bool IsFrontFace;
    TexCoords TextureCoordinates;
    uint32_t InstanceIndex, ObjectIndex, PrimitiveIndex;
};

PT_DEV AlphaContext alpha_context(const SceneView& sv)
{
    AlphaContext ac; ac.objects = sv.objects; ac.heap = sv.heap; ac.srgbLut = sv.srgbLut; ac.instances = sv.accel.instances;
    return ac;
}

PT_DEV uint32_t load_index_dev(const void* ib, uint32_t stride, uint32_t i)      // MeshHelpers.hlsli:5-9 (typed R16/R32 buffer)
{
    return stride == 2 ? (uint32_t)gptr<uint16_t>(ib)[i] : gptr<uint32_t>(ib)[i];
}

// Hit reconstruction half of CastRay (Shaders/RaytracingHelpers.hlsli:73-131) + HitInfo::Initialize
// (Shaders/HitInfo.hlsli:24-65). Positions come from the BLAS triangle packet (bit-identical copies of the
// vertex-buffer positions); normals from the vertex buffer through the descriptor heap.
// TEXTURED = false (no texture descriptor exists in the heap): tangents, UVs and the TextureMapInfo half of
// ObjectData are never fetched.
// What hit reconstruction needs from the acceleration structure: the instance's two transforms, its InstanceID and the
// triangle packet. Two sources with identical contents: the TLAS / BLAS arrays (k_gbuffer, k_shade) or the compact scene
// blob, which the fused round kernel already holds in LDS for small scenes (two dependent HBM round trips less per hit).
struct HitGeometry { float M[12], W[12]; uint32_t instanceID; TriPacket tp; uint32_t vi[3]; };      // vi: the triangle's vertex indices

template <bool LDS>
PT_DEV HitGeometry load_hit_geometry(const BlobReader<LDS>& blob, const BlobView& bv, uint32_t inst, uint32_t triSlot)
{
    const uint32_t ia = bv.instOff16 + inst * kInst16;
    const f4v w0 = blob.ld(ia), w1 = blob.ld(ia + 1), w2 = blob.ld(ia + 2), b1 = blob.ld(ia + 4), mk = blob.ld(ia + 5);
    const f4v m0 = blob.ld(ia + 6), m1 = blob.ld(ia + 7), m2 = blob.ld(ia + 8);
    const uint32_t ta = bv.triOff16 + (__float_as_uint(b1.w) + triSlot) * kTri16;
    const f4v pa = blob.ld(ta), pb = blob.ld(ta + 1), pc = blob.ld(ta + 2);
    const f4v ix = blob.ld(bv.idxOff16 + __float_as_uint(b1.w) + triSlot);
    HitGeometry g;
    g.vi[0] = __float_as_uint(ix.x); g.vi[1] = __float_as_uint(ix.y); g.vi[2] = __float_as_uint(ix.z);
    g.tp.a = make_float4(pa.x, pa.y, pa.z, pa.w); g.tp.b = make_float4(pb.x, pb.y, pb.z, pb.w); g.tp.c = make_float4(pc.x, pc.y, pc.z, pc.w);
    g.instanceID = __float_as_uint(mk.z);
    g.W[0] = w0.x; g.W[1] = w0.y; g.W[2] = w0.z; g.W[3] = w0.w; g.W[4] = w1.x; g.W[5] = w1.y; g.W[6] = w1.z; g.W[7] = w1.w;
    g.W[8] = w2.x; g.W[9] = w2.y; g.W[10] = w2.z; g.W[11] = w2.w;
    g.M[0] = m0.x; g.M[1] = m0.y; g.M[2] = m0.z; g.M[3] = m0.w; g.M[4] = m1.x; g.M[5] = m1.y; g.M[6] = m1.z; g.M[7] = m1.w;
    g.M[8] = m2.x; g.M[9] = m2.y; g.M[10] = m2.z; g.M[11] = m2.w;
    return g;
}

template <bool TEXTURED>
PT_DEV void reconstruct_hit(const SceneView& sv, const HitGeometry& hg, uint32_t inst, float bu, float bv, v3 rayDir, SurfaceHit& h)
{
    const TriPacket& tp = hg.tp;
    const uint32_t geom = __float_as_uint(tp.a.w), prim = __float_as_uint(tp.b.w);
    h.InstanceIndex = inst;
    h.ObjectIndex = hg.instanceID + geom;                  // RaytracingHelpers.hlsli:79
    h.PrimitiveIndex = prim;
    const float* M = hg.M; const float* W = hg.W;
    safe_triangle_spawn_point(V3(tp.a.x, tp.a.y, tp.a.z), V3(tp.b.x, tp.b.y, tp.b.z), V3(tp.c.x, tp.c.y, tp.c.z), bu, bv, M, W,
                              h.ObjectPosition, h.Position, h.FlatNormal, h.PositionOffset);
    // Vertex attributes: the object's resolved geometry (ONE fetch: buffer pointers, stride, offsets -- instead of object record ->
    // descriptor table) and the triangle's vertex indices, which came with the hit geometry (no index-buffer fetch): two dependent
    // loads from hit to normals where the reference's chain (RaytracingHelpers.hlsli:82-105) has four.
    const ShadeGeom sg = sv.shadeGeom[h.ObjectIndex];
    const uint32_t nOff = sg.nOff;
    if (nOff != ~0u) {                                     // HitInfo.hlsli:52-65
        v3 nrm[3];
        #pragma unroll
        for (int k = 0; k < 3; k++) {
            const PT_GLOBAL_AS int16_t* q = gptr<int16_t>(sg.vb + (size_t)sg.stride * hg.vi[k] + nOff);
            nrm[k] = V3(unpack_r16_snorm(q[0]), unpack_r16_snorm(q[1]), unpack_r16_snorm(q[2]));
        }
        v3 n = interp3(nrm[0], nrm[1], nrm[2], bu, bv);          // Vertex::Interpolate, Vertex.hlsli:63-72
        v3 g = V3(sop3(W[0], n.x, W[4], n.y, W[8], n.z), sop3(W[1], n.x, W[5], n.y, W[9], n.z), sop3(W[2], n.x, W[6], n.y, W[10], n.z));
        h.GeometricNormal = normalize(g);
    } else {                                               // HitInfo.hlsli:37-50
        h.GeometricNormal = h.FlatNormal;
    }
    h.ShadingNormal = h.GeometricNormal;
    h.IsFrontFace = dot(h.GeometricNormal, rayDir) < 0.0f;
    if (!h.IsFrontFace) h.ShadingNormal = -h.ShadingNormal;
    h.Tangent = V3(0.0f, 0.0f, 0.0f);                      // RaytracingHelpers.hlsli:115-122
    if (!TEXTURED) return;
    const uint32_t tOff = sg.tOff;
    if (tOff != ~0u) {
        v3 tg[3];
        #pragma unroll
        for (int k = 0; k < 3; k++) {
            const PT_GLOBAL_AS int16_t* q = gptr<int16_t>(sg.vb + (size_t)sg.stride * hg.vi[k] + tOff);
            tg[k] = V3(unpack_r16_snorm(q[0]), unpack_r16_snorm(q[1]), unpack_r16_snorm(q[2]));
        }
        const v3 t = interp3(tg[0], tg[1], tg[2], bu, bv);
        h.Tangent = normalize(V3(sop3(M[0], t.x, M[1], t.y, M[2], t.z), sop3(M[4], t.x, M[5], t.y, M[6], t.z), sop3(M[8], t.x, M[9], t.y, M[10], t.z)));
    }
    const PtObjectData* od = &sv.objects[h.ObjectIndex];
    get_texture_coordinates(od, sv.heap, prim, bu, bv, h.TextureCoordinates);      // :124-130
}

PT_DEV v3 material_emission(const PtMaterial& m) { return V3(m.EmissiveColor) * m.EmissiveStrength; }

template <bool TEXTURED>
PT_DEV PtMaterial surface_material(const SceneView& sv, SurfaceHit& h)
{
    if (!TEXTURED) return sv.objects[h.ObjectIndex].Material;           // EvaluateMaterial with every Descriptor == ~0u
    return evaluate_material(h.ShadingNormal, h.IsFrontFace ? h.Tangent : -h.Tangent, &sv.objects[h.ObjectIndex], sv.heap, sv.srgbLut,
                             h.TextureCoordinates);                       // ShadingHelpers.hlsli:161-235
}

// ---------------------------------------------------------------------------------------------
// G-buffer (Shaders/GBufferGeneration.hlsl:116-232). One thread per local pixel, 16x16 tiles.
// ---------------------------------------------------------------------------------------------
// MODE 0: interleaved TLAS/BLAS walk over the acceleration-structure arrays (any scene). MODE 1 / 2: the flat schedule of
// pt_trace2.hpp over the scene blob (<= kFlatInstances instances), blob staged in LDS / read from memory.
template <bool STATS, bool TEXTURED, int MODE>
__global__ __launch_bounds__(256) void k_gbuffer(SceneView sv, FrameView fv, PtCamera cam, PtSceneData sd, uint32_t flags, PtTextures tx,
                                                 BlobView bv, DeviceCounters* counters)
{
    // a wave covers an 8 x 8 pixel square of the block's 16 x 16 tile (not a 16 x 4 strip): its primary rays stay together longer in the tree
    const uint32_t wv = threadIdx.x >> 6, ln = threadIdx.x & 63u;
    const uint32_t x = blockIdx.x * 16 + (wv & 1u) * 8u + (ln & 7u), ly = blockIdx.y * 16 + (wv >> 1) * 8u + (ln >> 3);
    const bool valid = x < fv.width && ly < fv.localRows;
    if (MODE == 0 && !valid) return;                           // no barrier below in this mode: early exit is safe
    const uint32_t y = global_row(fv, valid ? ly : 0u);
    const size_t pi = (size_t)ly * fv.width + x;

    float4 Position = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
    float LinearDepth = INFINITY, NormalizedDepth = cam.IsNormalizedDepthReversed ? 0.0f : 1.0f;
    float u, v;
    const RayDesc ray = generate_pinhole_ray(cam, valid ? x : 0u, y, fv.width, fv.height, u, v);
    TraceStats st; st.nodes = 0; st.tris = 0; st.overflow = 0;
    Hit hit;
    HitGeometry hg;
    if constexpr (MODE == 0) {
        __shared__ uint2 ldsStack[kLdsStackDepth * 256];
        uint2 spill[kStackSize - kLdsStackDepth];
        GroupStack<kLdsStackDepth> stack; stack.init((PT_LDS_AS void*)ldsStack, spill);
        BlobReader<false> blob; blob.p = bv.base;
        hit = trace_single<STATS, false, false>(blob, bv, alpha_context(sv), ray.o, ray.d, ray.tmin, ray.tmax, stack, &st, nullptr);
        if (hit.inst != ~0u) hg = load_hit_geometry<false>(blob, bv, hit.inst, hit.slot);
    } else {
        extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
        constexpr bool LDS = MODE == 1;
        BlobReader<LDS> blob;
        if constexpr (LDS) {
            f4v* dst = (f4v*)(smem + kFlatLdsFixed);
            const uint32_t n16 = bv.bytes / 16u;
            for (uint32_t k = threadIdx.x; k < n16; k += 256u) dst[k] = bv.base[k];
            __syncthreads();
            blob.p = (const PT_LDS_AS f4v*)(smem + kFlatLdsFixed);
        } else {
            blob.p = bv.base;
        }
        unsigned char* ldsWave = smem + (uint32_t)kStackLdsFlat * 256u * 8u + (threadIdx.x >> 6) * kFlatWaveLds;
        // pixels outside the frame carry an empty ray interval: they hit nothing but their lanes still serve work items
        hit = trace_closest_flat<STATS, LDS>(blob, bv, alpha_context(sv), ray.o, ray.d, valid ? ray.tmin : 1.0f, valid ? ray.tmax : 0.0f,
                                             (PT_LDS_AS void*)smem, ldsWave, &st);
        if (!valid) return;
        if (hit.inst != ~0u) hg = load_hit_geometry<LDS>(blob, bv, hit.inst, hit.slot);
    }
    if (STATS) { atomicAdd(&counters->nodesVisited, (unsigned long long)st.nodes); atomicAdd(&counters->trianglesTested, (unsigned long long)st.tris); }
    if (st.overflow) atomicAdd(&counters->stackOverflows, st.overflow);
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) atomicAdd(&counters->primaryRays, (unsigned long long)fv.width * fv.localRows);

    if (hit.inst != ~0u) {
        SurfaceHit h;
        reconstruct_hit<TEXTURED>(sv, hg, hit.inst, hit.u, hit.v, ray.d, h);
        if (flags & PT_GB_Geometry) {
            Position = make_float4(h.Position.x, h.Position.y, h.Position.z, h.PositionOffset);
            float ex, ey;
            if ((flags & PT_GB_FlatNormal) && tx.FlatNormal) {
                oct_encode(h.FlatNormal, ex, ey);
                ((short2*)tx.FlatNormal)[pi] = make_short2(f32_to_snorm16(ex), f32_to_snorm16(ey));
            }
            if ((flags & PT_GB_GeometricNormal) && tx.GeometricNormal) {
                oct_encode(h.GeometricNormal, ex, ey);
                ((short2*)tx.GeometricNormal)[pi] = make_short2(f32_to_snorm16(ex), f32_to_snorm16(ey));
            }
            float proj[4]; xform4(cam.WorldToProjection, h.Position, proj);
            LinearDepth = proj[3];
            NormalizedDepth = proj[2] / proj[3];
            if ((flags & PT_GB_MotionVector) && tx.MotionVector) {          // CalculateMotionVector :62-91 (no per-vertex motion buffers)
                v3 prev = h.Position;
                if (!sd.IsStatic && sv.instanceData) {
                    const float* P = sv.instanceData[h.InstanceIndex].PreviousObjectToWorld; v3 q = h.ObjectPosition;
                    const PtMeshDescriptors md = sv.objects[h.ObjectIndex].MeshDescriptors;
                    if (md.MotionVectors != ~0u) {              // :73-84, StructuredBuffer<float16_t4>
                        const PT_GLOBAL_AS uint16_t* mvb = gptr<uint16_t>(sv.heap[md.MotionVectors].ptr);
                        const HeapEntry ib = sv.heap[md.Indices];
                        v3 m3[3];
                        for (int kk = 0; kk < 3; kk++) {
                            const uint32_t vi = load_index_dev(ib.ptr, ib.stride, 3 * h.PrimitiveIndex + kk);
                            m3[kk] = V3(f16_to_f32(mvb[4 * (size_t)vi]), f16_to_f32(mvb[4 * (size_t)vi + 1]), f16_to_f32(mvb[4 * (size_t)vi + 2]));
                        }
                        q = q + interp3(m3[0], m3[1], m3[2], hit.u, hit.v);
                    }
                    prev = V3(sop3t(P[0], q.x, P[1], q.y, P[2], q.z, P[3]), sop3t(P[4], q.x, P[5], q.y, P[6], q.z, P[7]), sop3t(P[8], q.x, P[9], q.y, P[10], q.z, P[11]));
                }
                float clip[4], view[4];
                xform4(cam.PreviousWorldToProjection, prev, clip);
                xform4(cam.PreviousWorldToView, prev, view);
                float su = (clip[0] / clip[3]) * 0.5f + 0.5f, svv = (clip[1] / clip[3]) * -0.5f + 0.5f;
                ushort4 mv = make_ushort4(f32_to_f16((su - u) * (float)fv.width), f32_to_f16((svv - v) * (float)fv.height), f32_to_f16(view[2] - LinearDepth), 0);
                ((ushort4*)tx.MotionVector)[pi] = mv;
            }
        }
        BSDFSample bs; bs.Roughness = 0.0f;
        if (flags & PT_GB_Material) {
            const PtMaterial m = surface_material<TEXTURED>(sv, h);
            bs.Initialize(V3(m.BaseColor), m.Metallic, m.Roughness, m.IOR, m.Transmission, h.IsFrontFace);
            if (tx.BaseColorMetalness)
                ((uchar4*)tx.BaseColorMetalness)[pi] = make_uchar4(f32_to_unorm8(bs.BaseColor.x), f32_to_unorm8(bs.BaseColor.y), f32_to_unorm8(bs.BaseColor.z), f32_to_unorm8(bs.Metallic));
            if (flags & PT_GB_Albedo) {        // EstimateDemodulationFactors (BxDF.hlsli:317-320) = NRD_MaterialFactors, see DESIGN.md [NRD spec]
                const float NoV = fabsf(dot(h.ShadingNormal, -ray.d));
                const v3 Fe = ml_env_term_rtg(bs.F0, NoV, bs.Roughness);
                if ((flags & PT_GB_DiffuseAlbedo) && tx.DiffuseAlbedo)
                    ((ushort4*)tx.DiffuseAlbedo)[pi] = make_ushort4(f32_to_f16((1.0f - Fe.x) * bs.Albedo.x * 0.99f + 0.01f), f32_to_f16((1.0f - Fe.y) * bs.Albedo.y * 0.99f + 0.01f),
                                                                     f32_to_f16((1.0f - Fe.z) * bs.Albedo.z * 0.99f + 0.01f), 0);
                if ((flags & PT_GB_SpecularAlbedo) && tx.SpecularAlbedo)
                    ((ushort4*)tx.SpecularAlbedo)[pi] = make_ushort4(f32_to_f16(Fe.x * 0.99f + 0.01f), f32_to_f16(Fe.y * 0.99f + 0.01f), f32_to_f16(Fe.z * 0.99f + 0.01f), 0);
            }
            if (tx.IOR) ((uint16_t*)tx.IOR)[pi] = f32_to_f16(m.IOR);
            if (bs.Metallic < 1.0f && tx.Transmission) ((uint8_t*)tx.Transmission)[pi] = f32_to_unorm8(bs.Transmission);
            if ((flags & PT_GB_Radiance) && tx.Radiance) {
                v3 e = material_emission(m);
                ((ushort4*)tx.Radiance)[pi] = make_ushort4(f32_to_f16(e.x), f32_to_f16(e.y), f32_to_f16(e.z), 0);
            }
        }
        if ((flags & PT_GB_NormalRoughness) && tx.NormalRoughness) {
            float r = (flags & PT_GB_Material) ? bs.Roughness : 0.0f;
            ((short4*)tx.NormalRoughness)[pi] = make_short4(f32_to_snorm16(h.ShadingNormal.x), f32_to_snorm16(h.ShadingNormal.y), f32_to_snorm16(h.ShadingNormal.z), f32_to_snorm16(r));
        }
    } else {
        if ((flags & PT_GB_MotionVector) && tx.MotionVector) {
            v3 far = ray.o + ray.d * 1e8f;                                  // CastRay miss position, RaytracingHelpers.hlsli:64
            float proj[4], clip[4], view[4];
            xform4(cam.WorldToProjection, far, proj);
            xform4(cam.PreviousWorldToProjection, far, clip);
            xform4(cam.PreviousWorldToView, far, view);
            float su = (clip[0] / clip[3]) * 0.5f + 0.5f, svv = (clip[1] / clip[3]) * -0.5f + 0.5f;
            ((ushort4*)tx.MotionVector)[pi] = make_ushort4(f32_to_f16((su - u) * (float)fv.width), f32_to_f16((svv - v) * (float)fv.height), f32_to_f16(view[2] - proj[3]), 0);
        }
        if ((flags & PT_GB_Radiance) && tx.Radiance) {
            v3 e = environment_light_color(sv, sd, ray.d);
            ((ushort4*)tx.Radiance)[pi] = make_ushort4(f32_to_f16(e.x), f32_to_f16(e.y), f32_to_f16(e.z), 0);
        }
    }
    if ((flags & PT_GB_Position) && tx.Position) ((float4*)tx.Position)[pi] = Position;
    if ((flags & PT_GB_LinearDepth) && tx.LinearDepth) ((float*)tx.LinearDepth)[pi] = LinearDepth;
    if ((flags & PT_GB_NormalizedDepth) && tx.NormalizedDepth) ((float*)tx.NormalizedDepth)[pi] = NormalizedDepth;
}

// ---------------------------------------------------------------------------------------------
// wavefront path tracer
// ---------------------------------------------------------------------------------------------
// Queue geometry. The path queue is cut into kSubQueues independent sub-queues (segments of segCap entries,
// each with its own counter): a single returning atomic on one word saturates near 88 M/s on MI355X, which
// at one atomic per wave made compaction the bottleneck of the whole frame. Pixel tile t (256 pixels) is
// dealt to sub-queue t % kSubQueues, a path never leaves its sub-queue, so a segment can never overflow and
// every sub-queue samples the whole image (balanced). One atomic per 256-thread block and tile.
constexpr uint32_t kSubQueues = 32;
constexpr uint32_t kCountStride = 3u * kSubQueues;      // per round: entries traced | fresh | cursor of the streaming form, one word per sub-queue

// block-wide stream compaction: wave64 ballot + prefix popcount inside each wave, wave totals through LDS,
// ONE atomicAdd per block. Every thread of the block must call it. lds: 8 words.
PT_DEV uint32_t block_reserve(bool alive, uint32_t* counter, uint32_t* lds)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long m = wave_ballot(alive);
    const uint32_t prefix = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) lds[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t total = lds[0] + lds[1] + lds[2] + lds[3];
        lds[4] = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    uint32_t base = lds[4];
    for (uint32_t w = 0; w < wave; w++) base += lds[w];
    __syncthreads();
    return base + prefix;
}

// Two compactions at once (survivors -> traced region, restarts -> fresh region): one barrier sequence instead of two, and
// the two returning atomics are issued by different waves, so their round trips (~1 us each) overlap. lds: 16 words; the caller hands in
// two sets in turn, so no barrier is needed behind the last read (a wave reaches the set again only through both barriers of the call in
// between, which every wave joins after it has finished this one): two barriers per tile instead of three, C2 +1.1 %, C3 +1.7 %.
PT_DEV void block_reserve2(bool a, bool b, uint32_t* counterA, uint32_t* counterB, uint32_t* lds, uint32_t& slotA, uint32_t& slotB)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const unsigned long long ma = wave_ballot(a), mb = wave_ballot(b);
    if (lane == 0) { lds[wave] = (uint32_t)__popcll(ma); lds[4 + wave] = (uint32_t)__popcll(mb); }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t total = lds[0] + lds[1] + lds[2] + lds[3];
        lds[8] = total ? atomicAdd(counterA, total) : 0u;
    }
    if (threadIdx.x == 64) {
        const uint32_t total = lds[4] + lds[5] + lds[6] + lds[7];
        lds[9] = total ? atomicAdd(counterB, total) : 0u;
    }
    __syncthreads();
    uint32_t baseA = lds[8], baseB = lds[9];
    for (uint32_t w = 0; w < wave; w++) { baseA += lds[w]; baseB += lds[4 + w]; }
    slotA = baseA + (uint32_t)__popcll(ma & lt);
    slotB = baseB + (uint32_t)__popcll(mb & lt);
}

// Per-frame constants (camera, scene data, settings) live in a device buffer written in stream order by
// k_set_constants, not in kernel arguments: the frame's launch sequence can then be captured once into a
// hipGraph and replayed for every frame (FrameIndex, jitter, ... change without touching the graph).
__global__ void k_set_constants(FrameConstants v, FrameConstants* dst, uint32_t* queueCounts, uint32_t countWords)
{
    const uint32_t* s = (const uint32_t*)&v; uint32_t* d = (uint32_t*)dst;
    for (uint32_t i = threadIdx.x; i < sizeof(FrameConstants) / 4; i += blockDim.x) d[i] = s[i];
    for (uint32_t i = threadIdx.x; i < countWords; i += blockDim.x) queueCounts[i] = 0u;      // the per-round queue counters of this frame
}

// Queue regions. Every sub-queue segment holds two kinds of entries, grown from its two ends:
//   traced  [0, nT)                 paths whose ray has been traced by k_extend: state + ray + hit record
//   fresh   (segCap-1 ... segCap-nF] paths about to start a sample at the primary surface: state only
// k_shade consumes both kinds in two separate tile loops (no divergence between "reconstruct a hit" and "decode the
// G-buffer"), and writes both kinds. A path that ends a sample with samples left re-enters as fresh, carrying its
// RNG state (all samples of a pixel draw from one stream, Raytracing.hlsl:108,191).

// enqueue every primary-hit pixel as fresh (Raytracing.hlsl:106-127; a primary miss keeps the G-buffer radiance, :241-252)
// It also gathers what bounce 0 of every sample re-reads from the G-buffer (Raytracing.hlsl:118-148: eight textures, 47 bytes) into
// ONE 48-byte record per pixel, bit for bit: rec0 = Position | rec1 = NormalRoughness, FlatNormal, GeometricNormal | rec2 =
// BaseColorMetalness, Radiance, IOR + Transmission. Here pixels are read in order (coalesced); the fresh entries of later rounds hold
// pixels in whatever order their paths ended, and eight scattered loads per lane, each using 1..16 bytes of its cache line, were the
// most expensive part of a round (profiles/r03_round_prof_*.txt: 30 % of a wave's time in the fresh tiles).
__global__ __launch_bounds__(256) void k_pt_init(FrameView fv, const FrameConstants* __restrict__ fc, PtTextures tx, PathQueue q, float2* aux, uint32_t segCap, uint32_t* countFresh,
                                                 uint4* __restrict__ primary)
{
    __shared__ uint32_t lds[8];
    const PtGraphicsSettings& gs = fc->gs;
    const uint32_t npix = fv.width * fv.localRows;
    const uint32_t sq = blockIdx.x % kSubQueues, bq = blockIdx.x / kSubQueues, nbq = gridDim.x / kSubQueues;
    for (uint32_t j = bq; (j * kSubQueues + sq) * 256u < npix; j += nbq) {
        const uint32_t p = (j * kSubQueues + sq) * 256u + threadIdx.x;
        bool alive = false;
        if (p < npix) alive = isfinite(((const float*)tx.Position)[4 * (size_t)p + 3]);
        const uint32_t slot = sq * segCap + (segCap - 1u - block_reserve(alive, &countFresh[sq], lds));
        if (alive) {
            const uint32_t x = p % fv.width, y = global_row(fv, p / fv.width);
            q.s0[slot] = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(p));
            q.s1[slot] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(rng_init(x, y, gs.FrameIndex)));
            q.s2[slot] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0u));            // sample 0, bounce 0
            if (aux) aux[p] = make_float2(INFINITY, 1.0f);                               // hitDistance = inf, isDiffuse = true (:188-189)
            const uint2 nr = ((const uint2*)tx.NormalRoughness)[p], rad = ((const uint2*)tx.Radiance)[p];
            primary[3 * (size_t)p] = ((const uint4*)tx.Position)[p];
            primary[3 * (size_t)p + 1] = make_uint4(nr.x, nr.y, ((const uint32_t*)tx.FlatNormal)[p], ((const uint32_t*)tx.GeometricNormal)[p]);
            primary[3 * (size_t)p + 2] = make_uint4(((const uint32_t*)tx.BaseColorMetalness)[p], rad.x, rad.y,
                                                    (uint32_t)((const uint16_t*)tx.IOR)[p] | ((uint32_t)((const uint8_t*)tx.Transmission)[p] << 16));
        }
    }
}

struct PathRegs { v3 thr, srad, rsum; uint32_t pixel, rng, sample, bounce; };

PT_DEV PathRegs load_path(const PathQueue& q, uint32_t i)
{
    const float4 a = q.s0[i], b = q.s1[i], c = q.s2[i];
    PathRegs p;
    p.thr = V3(a.x, a.y, a.z); p.pixel = __float_as_uint(a.w);
    p.srad = V3(b.x, b.y, b.z); p.rng = __float_as_uint(b.w);
    p.rsum = V3(c.x, c.y, c.z);
    const uint32_t cnt = __float_as_uint(c.w);
    p.sample = cnt >> 16; p.bounce = cnt & 0xFFFFu;
    return p;
}
PT_DEV void store_path(const PathQueue& q, uint32_t i, const PathRegs& p)
{
    q.s0[i] = make_float4(p.thr.x, p.thr.y, p.thr.z, __uint_as_float(p.pixel));
    q.s1[i] = make_float4(p.srad.x, p.srad.y, p.srad.z, __uint_as_float(p.rng));
    q.s2[i] = make_float4(p.rsum.x, p.rsum.y, p.rsum.z, __uint_as_float((p.sample << 16) | p.bounce));
}

// One iteration body of the bounce loop after the surface is known (Raytracing.hlsl:320-364): emission, lobe
// weights, BSDF sample, throughput update, Russian roulette, luminance cut-off. Returns true when the path goes on
// with the ray (newO, newD); false ends the sample. RNG draws happen exactly as in the reference, also on the last
// iteration (bounce == Bounces), which samples but never traces (:213).
PT_DEV bool scatter(const PtGraphicsSettings& gs, PathRegs& p, const SurfaceHit& h, const BSDFSample& bs, v3 emission, v3 rayDir, v3& newO, v3& newD, int& lobe)
{
    p.srad = madd(p.thr, emission, p.srad);                       // :320
    const SurfaceVectors svec = surface_vectors(h.IsFrontFace, h.GeometricNormal, h.ShadingNormal);
    const v3 V = -rayDir;
    float w[3]; bs.ComputeLobeWeights(svec, V, gs.ExtFlags, w);
    float rnd[4];
    rnd[0] = rng_float(p.rng); rnd[1] = rng_float(p.rng); rnd[2] = rng_float(p.rng); rnd[3] = rng_float(p.rng);   // GetFloat4, :330
    v3 L;
    if (!bs.Sample(svec, V, w, rnd, L, lobe)) return false;
    float pdf; v3 f;
    bs.EvaluateLobe(svec, L, V, w, lobe, gs.ExtFlags, pdf, f);
    if (pdf == 0.0f || (f.x == 0.0f && f.y == 0.0f && f.z == 0.0f)) return false;             // :336,342
    p.thr = p.thr * V3(f.x / pdf, f.y / pdf, f.z / pdf);                                      // :346
    if (gs.IsRussianRouletteEnabled && p.bounce > 3) {                                        // :348-356
        const float prob = fmaxf(p.thr.x, fmaxf(p.thr.y, p.thr.z));
        if (rng_float(p.rng) >= prob) return false;
        p.thr = V3(p.thr.x / prob, p.thr.y / prob, p.thr.z / prob);
    }
    if (ml_luminance(p.thr) <= gs.ThroughputThreshold) return false;                          // :361
    if (!(p.bounce < gs.Bounces)) return false;                                               // loop bound, :213
    newO = safe_world_ray_origin(h.Position, h.FlatNormal, h.PositionOffset, L);              // :221
    newD = L;
    p.bounce++;
    return true;
}

// sample ended: accumulate, start the next sample of the pixel or finish the pixel (Raytracing.hlsl:372-413)
PT_DEV bool end_sample(const PtGraphicsSettings& gs, const PtTextures& tx, const float2* aux, PathRegs& p)
{
    p.rsum = p.rsum + p.srad;                                    // :372
    p.sample++;
    if (p.sample < gs.SamplesPerPixel) { p.thr = V3(1, 1, 1); p.srad = V3(0, 0, 0); p.bounce = 0; return true; }
    v3 out = V3(0, 0, 0);
    if (finite3(p.rsum)) { const float ns = (float)gs.SamplesPerPixel; out = V3(p.rsum.x / ns, p.rsum.y / ns, p.rsum.z / ns); }   // :377
    if (gs.Denoiser == PT_DENOISER_NRD_REBLUR || gs.Denoiser == PT_DENOISER_NRD_RELAX) {     // :400-413, direct terms are 0 (DI off)
        const ushort4 rad = ((const ushort4*)tx.Radiance)[p.pixel];                          // primaryRadiance (G-buffer emission)
        const v3 ind = V3(fmaxf(out.x - f16_to_f32(rad.x), 0.0f), fmaxf(out.y - f16_to_f32(rad.y), 0.0f), fmaxf(out.z - f16_to_f32(rad.z), 0.0f));
        const float2 a = aux[p.pixel];
        const ushort4 packed = make_ushort4(f32_to_f16(ind.x), f32_to_f16(ind.y), f32_to_f16(ind.z), f32_to_f16(a.x));
        const ushort4 zero = make_ushort4(0, 0, 0, 0);
        const bool isDiffuse = a.y != 0.0f;
        if (tx.Diffuse) ((ushort4*)tx.Diffuse)[p.pixel] = isDiffuse ? packed : zero;
        if (tx.Specular) ((ushort4*)tx.Specular)[p.pixel] = isDiffuse ? zero : packed;
        return false;
    }
    if (gs.Denoiser == PT_DENOISER_DLSS_RAY_RECONSTRUCTION && tx.SpecularHitDistance) {      // :395-398
        const float2 a = aux[p.pixel];
        if (a.y == 0.0f && isfinite(a.x)) ((uint16_t*)tx.SpecularHitDistance)[p.pixel] = f32_to_f16(a.x);
    }
    ((ushort4*)tx.Radiance)[p.pixel] = make_ushort4(f32_to_f16(out.x), f32_to_f16(out.y), f32_to_f16(out.z), 0);     // :385 / :393
    if (tx.RadianceF32) ((float4*)tx.RadianceF32)[p.pixel] = make_float4(out.x, out.y, out.z, 0.0f);
    return false;
}

// ---- the bodies of k_shade, shared with the fused round kernel k_round ----------------------------------------
// A traced path at its hit (or miss) of bounce >= 1, Raytracing.hlsl:219-304. hit = (instance, triangle slot, u, v).
template <bool LDS> struct GeometryFromBlob {                // ... out of the scene blob (LDS-resident when LDS)
    const BlobReader<LDS>& blob; const BlobView& bv;
    PT_DEV HitGeometry load(uint32_t inst, uint32_t slot) const { return load_hit_geometry<LDS>(blob, bv, inst, slot); }
};

template <bool TEXTURED, typename GEOMETRY>
PT_DEV void shade_traced(const SceneView& sv, const GEOMETRY& geometry, const PtSceneData& sd, const PtGraphicsSettings& gs, const PtTextures& tx, float2* aux,
                         PathRegs& p, uint4 hr, float hitT, v3 rayDir, bool& toTraced, bool& toFresh, v3& newO, v3& newD, RoundProf* prof = nullptr)
{
    bool goes = false; int lobe = 0;
    if (aux && p.sample == 0 && p.bounce == 1) aux[p.pixel].x = hr.x == ~0u ? INFINITY : hitT;        // hitDistance, :235-239
    if (hr.x == ~0u) {                                       // :241-259
        p.srad = madd(p.thr, environment_light_color(sv, sd, rayDir), p.srad);
    } else {                                                 // :293-304
        SurfaceHit h;
        reconstruct_hit<TEXTURED>(sv, geometry.load(hr.x, hr.y), hr.x, __uint_as_float(hr.z), __uint_as_float(hr.w), rayDir, h);
        PT_PROF_MARK(prof, 5);
        const PtMaterial m = surface_material<TEXTURED>(sv, h);
        BSDFSample bs;
        bs.Initialize(V3(m.BaseColor), m.Metallic, m.Roughness, m.IOR, m.Transmission, h.IsFrontFace);
        goes = scatter(gs, p, h, bs, material_emission(m), rayDir, newO, newD, lobe);
    }
    if (goes) toTraced = true;
    else toFresh = end_sample(gs, tx, aux, p);
}

// A fresh path: bounce 0 on the primary surface rebuilt from the G-buffer, Raytracing.hlsl:118-148,193-198
PT_DEV void shade_fresh(const FrameView& fv, const PtCamera& cam, const PtGraphicsSettings& gs, const PtTextures& tx, float2* aux, const uint4* __restrict__ primary,
                        PathRegs& p, bool& toTraced, bool& toFresh, v3& newO, v3& newD, RoundProf* prof = nullptr)
{
    const uint32_t pixel = p.pixel;
    const uint32_t px = pixel % fv.width, py = global_row(fv, pixel / fv.width);
    float uu, vv;
    const RayDesc primaryRay = generate_pinhole_ray(cam, px, py, fv.width, fv.height, uu, vv);   // :110-126
    const v3 rayDir = primaryRay.d;
    const uint4 r0 = primary[3 * (size_t)pixel], r1 = primary[3 * (size_t)pixel + 1], r2 = primary[3 * (size_t)pixel + 2];
    PT_PROF_WAIT(); PT_PROF_MARK(prof, 13);
    const float4 pos = make_float4(__uint_as_float(r0.x), __uint_as_float(r0.y), __uint_as_float(r0.z), __uint_as_float(r0.w));
    const short4 nr = make_short4((short)(r1.x & 0xFFFFu), (short)(r1.x >> 16), (short)(r1.y & 0xFFFFu), (short)(r1.y >> 16));
    const short2 fe = make_short2((short)(r1.z & 0xFFFFu), (short)(r1.z >> 16)), ge = make_short2((short)(r1.w & 0xFFFFu), (short)(r1.w >> 16));
    const uchar4 bcm = make_uchar4((unsigned char)(r2.x & 0xFFu), (unsigned char)((r2.x >> 8) & 0xFFu), (unsigned char)((r2.x >> 16) & 0xFFu), (unsigned char)(r2.x >> 24));
    const ushort4 rad = make_ushort4((unsigned short)(r2.y & 0xFFFFu), (unsigned short)(r2.y >> 16), (unsigned short)(r2.z & 0xFFFFu), (unsigned short)(r2.z >> 16));
    SurfaceHit h;
    h.Position = V3(pos.x, pos.y, pos.z); h.PositionOffset = pos.w;                  // HitInfo.hlsli:67-79
    h.FlatNormal = oct_decode(snorm16_to_f32(fe.x), snorm16_to_f32(fe.y));
    h.GeometricNormal = oct_decode(snorm16_to_f32(ge.x), snorm16_to_f32(ge.y));
    h.ShadingNormal = V3(snorm16_to_f32(nr.x), snorm16_to_f32(nr.y), snorm16_to_f32(nr.z));
    h.IsFrontFace = dot(h.GeometricNormal, rayDir) < 0.0f;
    const v3 emission = V3(f16_to_f32(rad.x), f16_to_f32(rad.y), f16_to_f32(rad.z));           // :119,197
    const float metal = unorm8_to_f32(bcm.w);
    const float ior = f16_to_f32((uint16_t)(r2.w & 0xFFFFu));
    const float tr = metal < 1.0f ? unorm8_to_f32((uint8_t)((r2.w >> 16) & 0xFFu)) : 0.0f;       // :146
    BSDFSample bs;
    bs.Initialize(V3(unorm8_to_f32(bcm.x), unorm8_to_f32(bcm.y), unorm8_to_f32(bcm.z)), metal, snorm16_to_f32(nr.w), ior, tr, h.IsFrontFace);
    int lobe = 0;
    const bool first = p.sample == 0;
    if (scatter(gs, p, h, bs, emission, rayDir, newO, newD, lobe)) {
        toTraced = true;
        if (aux && first) aux[p.pixel].y = lobe == LOBE_DIFFUSE ? 1.0f : 0.0f;       // isDiffuse of the lobe sampled at bounce 0, :237
    } else toFresh = end_sample(gs, tx, aux, p);
}

// compaction + stores of one tile: survivors to the traced region (state + ray), restarts to the fresh region (state).
// Block-wide on purpose. Measured on C2 (round 3): a wave-level reservation (one atomic per wave, no barrier, waves free to drift)
// runs 20 % SLOWER -- 8.6 against 10.8 Grays/s. A block appends the survivors of 256 neighbouring pixels as one run, so a tile of the
// next round is made of two or three such runs; with 64-entry runs appended in arrival order the neighbourhoods dissolve four times as
// fast, and coherent tiles are what keeps the item lists of the traversal balanced and the loads of the shading half on few cache lines.
PT_DEV void emit_tile(const PathQueue& qout, uint32_t seg, uint32_t segCap, uint32_t* countTraced, uint32_t* countFresh, uint32_t* lds,
                      bool toTraced, bool toFresh, const PathRegs& p, v3 newO, v3 newD)
{
    uint32_t st, sf;
    block_reserve2(toTraced, toFresh, countTraced, countFresh, lds, st, sf);
    if (toTraced) {
        store_path(qout, seg + st, p);
        qout.r0[seg + st] = make_float4(newO.x, newO.y, newO.z, 0.0f);                    // TMin = 0, :223
        qout.r1[seg + st] = make_float4(newD.x, newD.y, newD.z, INFINITY);                // TMax = inf, :224
    }
    if (toFresh) store_path(qout, seg + (segCap - 1u - sf), p);
}

// counts: [0..kSubQueues) traced, [kSubQueues..2*kSubQueues) fresh
// k_shade<false> wants 132 VGPRs, one more than four waves per SIMD allow; held to 128 it spills nothing and the fourth wave is worth
// +1.8 % on C3 and +0.7 % on C5 (the kernel waits on its 268 B per ray, not on issue slots)
template <bool TEXTURED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_shade(SceneView sv, FrameView fv, const FrameConstants* __restrict__ fc, PtTextures tx,
                                               PathQueue qin, PathQueue qout, float2* aux, uint32_t segCap, const uint32_t* countIn, uint32_t* countOut,
                                               const uint4* __restrict__ primary, BlobView bv)
{
    BlobReader<false> blob; blob.p = bv.base;
    __shared__ uint32_t lds[32];                                  // two sets of reservation words, taken in turn: a fast wave may enter the next tile's reservation while a slow one still reads this tile's
    uint32_t emits = 0;
    const PtCamera& cam = fc->cam; const PtSceneData& sd = fc->sd; const PtGraphicsSettings& gs = fc->gs;
    const uint32_t sq = blockIdx.x % kSubQueues, bq = blockIdx.x / kSubQueues, nbq = gridDim.x / kSubQueues;
    const uint32_t nT = countIn[sq], nF = countIn[kSubQueues + sq];
    const uint32_t seg = sq * segCap;

    for (uint32_t tile = bq; tile * 256u < nT; tile += nbq) {                // traced entries: hit records left by k_extend
        const uint32_t local = tile * 256u + threadIdx.x;
        const uint32_t i = seg + local;
        bool toTraced = false, toFresh = false;
        PathRegs p; v3 newO = V3(0, 0, 0), newD = V3(0, 0, 1);
        if (local < nT) {
            p = load_path(qin, i);
            const uint4 hr = qin.hit[i];
            const float4 rd = qin.r1[i];                                     // k_extend left t in r1.w (denoiser modes)
            shade_traced<TEXTURED>(sv, GeometryFromBlob<false>{ blob, bv }, sd, gs, tx, aux, p, hr, rd.w, V3(rd.x, rd.y, rd.z), toTraced, toFresh, newO, newD);
        }
        emit_tile(qout, seg, segCap, &countOut[sq], &countOut[kSubQueues + sq], lds + ((emits++ & 1u) << 4), toTraced, toFresh, p, newO, newD);
    }
    for (uint32_t tile = bq; tile * 256u < nF; tile += nbq) {                // fresh entries
        const uint32_t local = tile * 256u + threadIdx.x;
        bool toTraced = false, toFresh = false;
        PathRegs p; v3 newO = V3(0, 0, 0), newD = V3(0, 0, 1);
        if (local < nF) {
            p = load_path(qin, seg + (segCap - 1u - local));
            shade_fresh(fv, cam, gs, tx, aux, primary, p, toTraced, toFresh, newO, newD);
        }
        emit_tile(qout, seg, segCap, &countOut[sq], &countOut[kSubQueues + sq], lds + ((emits++ & 1u) << 4), toTraced, toFresh, p, newO, newD);
    }
}

__global__ __launch_bounds__(256) void k_extend_brute(AccelView av, BlobView bv, AlphaContext ac, PathQueue q, uint32_t segCap, const uint32_t* count, DeviceCounters* counters)
{
    const uint32_t sq = blockIdx.x % kSubQueues, bq = blockIdx.x / kSubQueues, nbq = gridDim.x / kSubQueues;
    const uint32_t n = count[sq];
    if (bq == 0 && threadIdx.x == 0) atomicAdd(&counters->secondaryRays, (unsigned long long)n);
    __shared__ uint2 ldsStack[kLdsStackDepth * 256];
    uint2 spill[kStackSize - kLdsStackDepth];
    GroupStack<kLdsStackDepth> stack; stack.init((PT_LDS_AS void*)ldsStack, spill);
    BlobReader<false> blob; blob.p = bv.base;
    TraceStats st; st.nodes = 0; st.tris = 0; st.overflow = 0;
    for (uint32_t local = bq * 256u + threadIdx.x; local < n; local += nbq * 256u) {
        const uint32_t i = sq * segCap + local;
        const float4 o = q.r0[i], d = q.r1[i];
        const Hit h = trace_brute_force(av, ac, V3(o.x, o.y, o.z), V3(d.x, d.y, d.z), o.w, d.w);
        const Hit b = trace_single<false, false, false>(blob, bv, ac, V3(o.x, o.y, o.z), V3(d.x, d.y, d.z), o.w, d.w, stack, &st, nullptr);
        if (b.inst != h.inst || (h.inst != ~0u && (b.slot != h.slot || b.u != h.u || b.v != h.v))) {
            if (atomicAdd(&counters->mismatchCount, 1u) == 0u) {
                float* m = counters->mismatchRay;
                m[0] = o.x; m[1] = o.y; m[2] = o.z; m[3] = o.w; m[4] = d.x; m[5] = d.y; m[6] = d.z; m[7] = d.w;
                m[8] = __uint_as_float(b.inst); m[9] = __uint_as_float(b.slot); m[10] = b.t; m[11] = 0.0f;
                m[12] = __uint_as_float(h.inst); m[13] = __uint_as_float(h.slot); m[14] = h.t; m[15] = 0.0f;
            }
        }
        q.hit[i] = make_uint4(h.inst, h.slot, __float_as_uint(h.u), __float_as_uint(h.v));
    }
    if (st.overflow) atomicAdd(&counters->stackOverflows, st.overflow);
}

// Traversal kernel (phase-aligned schedule, pt_trace2.hpp). Dynamic LDS: traversal stack (kStackLds entries
// per lane) | candidate lists (kCandidates per lane) | the scene blob when it fits (LDS = true).
constexpr int kStackLds2 = 8;                  // TLAS + BLAS node groups (8 B) share this stack in the phased schedule; deeper entries spill
// phased schedule: stack | candidate lists | per-wave work-item exchange
constexpr uint32_t kStackLds2Bytes = (uint32_t)kStackLds2 * 256u * 8u;
constexpr uint32_t kExtendLdsFixed = kStackLds2Bytes + (uint32_t)kCandidates * 256u * 4u + 4u * kPhasedWaveLds;
constexpr uint32_t kBlobLdsMax = 40u * 1024u;

template <bool STATS, bool LDS, bool WRITE_T = false, bool FLAT = false>
__global__ __launch_bounds__(256) void k_extend2(BlobView bv, AlphaContext ac, PathQueue q, uint32_t segCap, const uint32_t* count, DeviceCounters* counters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr uint32_t kFixed = FLAT ? kFlatLdsFixed : kExtendLdsFixed;
    const uint32_t sq = blockIdx.x % kSubQueues, bq = blockIdx.x / kSubQueues, nbq = gridDim.x / kSubQueues;
    const uint32_t n = count[sq];
    if (bq == 0 && threadIdx.x == 0) atomicAdd(&counters->secondaryRays, (unsigned long long)n);
    if (bq * 256u >= n) return;                                   // block-uniform: nothing to do, skip the staging
    PT_LDS_AS void* ldsStack = (PT_LDS_AS void*)smem;
    BlobReader<LDS> blob;
    if constexpr (LDS) {
        f4v* dst = (f4v*)(smem + kFixed);
        const uint32_t n16 = bv.bytes / 16u;
        for (uint32_t i = threadIdx.x; i < n16; i += 256u) dst[i] = bv.base[i];
        __syncthreads();
        blob.p = (const PT_LDS_AS f4v*)(smem + kFixed);
    } else {
        blob.p = bv.base;
    }
    TraceStats st; st.nodes = 0; st.tris = 0; st.overflow = 0;
    if constexpr (FLAT) {
        // whole waves enter the traversal (lanes past the end carry a ray that can hit nothing): its lanes trade work items
        unsigned char* ldsWave = smem + (uint32_t)kStackLdsFlat * 256u * 8u + (threadIdx.x >> 6) * kFlatWaveLds;
        for (uint32_t base = bq * 256u; base < n; base += nbq * 256u) {
            const uint32_t local = base + threadIdx.x;
            const bool valid = local < n;
            const uint32_t i = sq * segCap + (valid ? local : base);
            float4 o = q.r0[i], d = q.r1[i];
            if (!valid) { o.w = 1.0f; d.w = 0.0f; }               // empty interval: the scan selects no instance
            const Hit h = trace_closest_flat<STATS, LDS>(blob, bv, ac, V3(o.x, o.y, o.z), V3(d.x, d.y, d.z), o.w, d.w, ldsStack, ldsWave, &st);
            if (valid) {
                q.hit[i] = make_uint4(h.inst, h.slot, __float_as_uint(h.u), __float_as_uint(h.v));
                if (WRITE_T) q.r1[i].w = h.t;                      // CommittedRayT for the denoiser hit-distance outputs
            }
        }
    } else {
        uint32_t* ldsCand = (uint32_t*)(smem + kStackLds2Bytes);
        unsigned char* ldsWave = smem + kStackLds2Bytes + (uint32_t)kCandidates * 256u * 4u + (threadIdx.x >> 6) * kPhasedWaveLds;
        for (uint32_t base = bq * 256u; base < n; base += nbq * 256u) {     // whole waves again: phase B trades work items
            const uint32_t local = base + threadIdx.x;
            const bool valid = local < n;
            const uint32_t i = sq * segCap + (valid ? local : base);
            float4 o = q.r0[i], d = q.r1[i];
            if (!valid) { o.w = 1.0f; d.w = 0.0f; }
            const Hit h = trace_closest_v2<STATS, LDS, kStackLds2>(blob, bv, ac, V3(o.x, o.y, o.z), V3(d.x, d.y, d.z), o.w, d.w, ldsStack, ldsCand, ldsWave, &st);
            if (valid) {
                q.hit[i] = make_uint4(h.inst, h.slot, __float_as_uint(h.u), __float_as_uint(h.v));
                if (WRITE_T) q.r1[i].w = h.t;
            }
        }
    }
    if (STATS) { atomicAdd(&counters->nodesVisited, (unsigned long long)st.nodes); atomicAdd(&counters->trianglesTested, (unsigned long long)st.tris); }
    if (st.overflow) atomicAdd(&counters->stackOverflows, st.overflow);
}

// One whole round in ONE launch: a block traces the rays of its tile and shades the same entries right away -- the hit stays in
// registers (no hit record through HBM, no second count read, no grid-wide drain between the two halves) -- then shades its
// fresh tiles. Same device functions as k_extend2 / k_shade, so the arithmetic and the queue protocol are unchanged; the
// frame needs spp * (Bounces + 1) + 1 launches instead of twice as many, which is what the launch-bound regimes (tail
// rounds, 1/8-frame shards of the multi-GPU run) are made of.
// 4 waves per SIMD = 128 VGPRs: fits without a spill (csrc/Makefile: -fno-slp-vectorize) and matches the 4 blocks per CU
// the LDS footprint allows.
// Everything a round needs, in device memory: the kernel takes ONE pointer. By-value kernel arguments are all loaded into SGPRs at
// kernel entry and stay live to their last use -- ~100 scalar registers for these structs, 60-70 of them spilled to VGPR lanes
// (v_writelane / v_readlane on the VALU pipe the kernel is bound by). Behind a pointer each field is an s_load where it is used.
struct RoundArgs {
    SceneView sv; FrameView fv; PtTextures tx; BlobView bv; PathQueue qin, qout;
    const FrameConstants* fc; float2* aux; const uint32_t* countIn; uint32_t* countOut; DeviceCounters* counters; uint32_t segCap, _pad;
    const uint4* primary;
};

template <bool TEXTURED, bool LDS, bool FLAT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_round(const RoundArgs* __restrict__ A)
{
    const SceneView& sv = A->sv; const FrameView& fv = A->fv; const PtTextures& tx = A->tx; const BlobView& bv = A->bv;
    const PathQueue& qin = A->qin; const PathQueue& qout = A->qout;
    const FrameConstants* __restrict__ fc = A->fc; float2* aux = A->aux; const uint32_t segCap = A->segCap;
    const uint32_t* countIn = A->countIn; uint32_t* countOut = A->countOut; DeviceCounters* counters = A->counters;
    const uint4* __restrict__ primary = A->primary;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ uint32_t lds[32];                                  // two sets of reservation words, taken in turn: a fast wave may enter the next tile's reservation while a slow one still reads this tile's
    uint32_t emits = 0;
    constexpr uint32_t kFixed = FLAT ? kFlatLdsFixed : kExtendLdsFixed;
    const PtCamera& cam = fc->cam; const PtSceneData& sd = fc->sd; const PtGraphicsSettings& gs = fc->gs;
    const uint32_t sq = blockIdx.x % kSubQueues, bq = blockIdx.x / kSubQueues, nbq = gridDim.x / kSubQueues;
    const uint32_t nT = countIn[sq], nF = countIn[kSubQueues + sq];
    const uint32_t seg = sq * segCap;
    if (bq == 0 && threadIdx.x == 0) atomicAdd(&counters->secondaryRays, (unsigned long long)nT);

#ifdef PT_ROUND_PROF
    RoundProf profData; RoundProf* prof = &profData;
    for (int k = 0; k < 16; k++) profData.acc[k] = 0;
    profData.last = __builtin_readcyclecounter();
#else
    RoundProf* prof = nullptr;
#endif
    if (bq * 256u < nT) {                                        // block-uniform
        PT_LDS_AS void* ldsStack = (PT_LDS_AS void*)smem;
        BlobReader<LDS> blob;
        if constexpr (LDS) {
            f4v* dst = (f4v*)(smem + kFixed);
            const uint32_t n16 = bv.bytes / 16u;
            for (uint32_t k = threadIdx.x; k < n16; k += 256u) dst[k] = bv.base[k];
            __syncthreads();
            blob.p = (const PT_LDS_AS f4v*)(smem + kFixed);
        } else {
            blob.p = bv.base;
        }
        AlphaContext ac; ac.objects = sv.objects; ac.heap = sv.heap; ac.srgbLut = sv.srgbLut; ac.instances = sv.accel.instances;
        TraceStats st; st.nodes = 0; st.tris = 0; st.overflow = 0;
        for (uint32_t base = bq * 256u; base < nT; base += nbq * 256u) {
            const uint32_t local = base + threadIdx.x;
            const bool valid = local < nT;
            const uint32_t i = seg + (valid ? local : base);
            float4 o = qin.r0[i], d = qin.r1[i];
            PathRegs p = load_path(qin, i);                       // issued before the traversal: its latency hides behind it (registers are
                                                                  // plentiful here, the kernel's budget is set by the shading half)
            if (!valid) { o.w = 1.0f; d.w = 0.0f; }               // empty interval: hits nothing, but the lane still serves work items
            PT_PROF_MARK(prof, 0);
            Hit h;
            if constexpr (FLAT) {
                unsigned char* ldsWave = smem + (uint32_t)kStackLdsFlat * 256u * 8u + (threadIdx.x >> 6) * kFlatWaveLds;
                h = trace_closest_flat<false, LDS>(blob, bv, ac, V3(o.x, o.y, o.z), V3(d.x, d.y, d.z), o.w, d.w, ldsStack, ldsWave, &st, prof);
            } else {
                uint32_t* ldsCand = (uint32_t*)(smem + kStackLds2Bytes);
                unsigned char* ldsWave = smem + kStackLds2Bytes + (uint32_t)kCandidates * 256u * 4u + (threadIdx.x >> 6) * kPhasedWaveLds;
                h = trace_closest_v2<false, LDS, kStackLds2>(blob, bv, ac, V3(o.x, o.y, o.z), V3(d.x, d.y, d.z), o.w, d.w, ldsStack, ldsCand, ldsWave, &st);
            }
            bool toTraced = false, toFresh = false;
            v3 newO = V3(0, 0, 0), newD = V3(0, 0, 1);
            if (valid) {
                shade_traced<TEXTURED>(sv, GeometryFromBlob<LDS>{ blob, bv }, sd, gs, tx, aux, p, make_uint4(h.inst, h.slot, __float_as_uint(h.u), __float_as_uint(h.v)), h.t,
                                       V3(d.x, d.y, d.z), toTraced, toFresh, newO, newD, prof);
            }
            PT_PROF_MARK(prof, 6);
            emit_tile(qout, seg, segCap, &countOut[sq], &countOut[kSubQueues + sq], lds + ((emits++ & 1u) << 4), toTraced, toFresh, p, newO, newD);
            PT_PROF_MARK(prof, 7);
#ifdef PT_ROUND_PROF
            prof->acc[11] += 1u;
#endif
        }
        if (st.overflow) atomicAdd(&counters->stackOverflows, st.overflow);
    }
    for (uint32_t tile = bq; tile * 256u < nF; tile += nbq) {
        const uint32_t local = tile * 256u + threadIdx.x;
        bool toTraced = false, toFresh = false;
        PathRegs p; v3 newO = V3(0, 0, 0), newD = V3(0, 0, 1);
        if (local < nF) {
            p = load_path(qin, seg + (segCap - 1u - local));
            PT_PROF_WAIT(); PT_PROF_MARK(prof, 12);
            shade_fresh(fv, cam, gs, tx, aux, primary, p, toTraced, toFresh, newO, newD, prof);
        }
        PT_PROF_MARK(prof, 14);
        emit_tile(qout, seg, segCap, &countOut[sq], &countOut[kSubQueues + sq], lds + ((emits++ & 1u) << 4), toTraced, toFresh, p, newO, newD);
        PT_PROF_MARK(prof, 15);
    }
#ifdef PT_ROUND_PROF
    // developer build only (tools/round_prof.py): per-wave section clocks (in units of 64 cycles) and item tallies through the mismatch record
    PT_PROF_MARK(prof, 8);
    if ((threadIdx.x & 63u) == 0u) for (int k = 0; k < 16; k++) atomicAdd((unsigned int*)&counters->mismatchRay[k], (k < 9 || k >= 12) ? (prof->acc[k] >> 6) : prof->acc[k]);
#endif
}

// ---------------------------------------------------------------------------------------------
// Streaming traversal for scenes whose blob lives in HBM / L2 (C3: one 250 k-triangle BLAS, C5: 10 k instances).
// The lock-step forms above trace one tile of rays per wave and wait for the slowest lane: on incoherent bounce rays in a big
// BVH (8..150 node visits per ray) a wave spent 8 of 9 issue slots on idle lanes (PMC, profiles/r02_b_c3: 540 VALU
// wave-instructions per ray against ~66 at full lanes). Here a wave is a set of 64 persistent traversal lanes:
//   refill   idle lanes take the next rays of the sub-queue (one atomic on the sub-queue's cursor per refill, consecutive
//            entries for consecutive idle lanes: coalesced reads) -- a lane that finishes early does not wait for its neighbours
//   walk     kStreamSteps steps of the one-ray two-level walk (trace_single's state machine, one stack per lane in LDS)
//   harvest  finished lanes write their hit record; the shading half (k_shade) runs as its own launch, full lanes
// Same arithmetic, same tie-break: the image is bit-identical to the other schedules
// (tests/test_gpu_parity.py::test_streaming_and_lockstep_schedules_agree).
constexpr int kStreamStackLds = 12;                   // stack entries per lane in LDS (24 KB per block); deeper ones go to scratch. 6: -3 %, 8: -1 %, 16: +0.5 %
constexpr uint32_t kStreamSteps = 6;                  // walk steps between two harvests
constexpr uint32_t kStreamRefillMin = 12;             // idle lanes worth a refill
constexpr uint32_t kStreamMinLanes = 8;               // a section (node visit / triangle test / instance entry) runs in a step when at least this many lanes ...
constexpr uint32_t kStreamShareShift = 1;             // ... and at least (lanes of the busiest section >> this) wait for it; the busiest always runs
constexpr uint32_t kStreamGridShared = 512, kStreamGridAlone = 1024;    // workgroups of a launch: other frames in flight on this GPU / the frame alone
constexpr uint32_t kStreamLdsStack = (uint32_t)kStreamStackLds * 256u * 8u;
constexpr bool kStreamTriPairs = true;                // two triangles of a leaf group per step

// The traversal half alone, same streaming walk: hits go to the queue's hit records (16 B per ray through HBM, nothing next to
// the latency it buys back: without the shading half's registers the kernel holds more waves per SIMD).
template <bool STATS, bool WRITE_T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_extend_stream(BlobView bv, AlphaContext ac, PathQueue q, uint32_t segCap,
                                               const uint32_t* count, uint32_t* cursor, DeviceCounters* counters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t sq = blockIdx.x % kSubQueues, bq = blockIdx.x / kSubQueues;
    const uint32_t nT = count[sq];
    const uint32_t seg = sq * segCap;
    if (bq == 0 && threadIdx.x == 0) atomicAdd(&counters->secondaryRays, (unsigned long long)nT);
    if (!nT) return;
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long ltMask = (1ull << lane) - 1ull;
    BlobReader<false> blob; blob.p = bv.base;
    uint2 spill[kStackSize - kStreamStackLds];
    GroupStack<kStreamStackLds> stack; stack.init((PT_LDS_AS void*)smem, spill);
    v3 wo = V3(0, 0, 0), wd = V3(0, 0, 1); float wtmax = 0.0f;      // the lane's ray in world space
    constexpr uint32_t kMarker = 0xFFFFFFFFu;
    TraceStats st; st.nodes = 0; st.tris = 0; st.overflow = 0;
    uint32_t qi = ~0u, curInst = ~0u, nodeBase16 = bv.nodeOff16, triBase16 = 0;
    float tmin = 0.0f;
    BoxRay br; br.o = V3(0, 0, 0); br.idir = V3(1, 1, 1); br.octinv4 = 0;
    RaySetup rs; rs.c1 = rs.c2 = false; rs.Sx = rs.Sy = rs.Sz = 0.0f;
    Hit h; h.t = 0.0f; h.u = h.v = 0.0f; h.inst = ~0u; h.geom = h.prim = h.slot = 0;
    uint2 G = make_uint2(0u, 0u), T = make_uint2(0u, 0u);
    bool exhausted = false;                                  // wave-uniform
    const bool oneInstance = bv.instCount == 1u;
    uint32_t rayNodes = 0;
#ifdef PT_STREAM_PROF
    uint32_t prof[13] = { 0 };
#endif
    while (true) {
        const unsigned long long busy = wave_ballot(qi != ~0u);
#ifdef PT_STREAM_PROF
        prof[12]++;
#endif
        if (exhausted && !busy) break;
        {
            const unsigned long long idle = ~busy;
            const uint32_t nIdle = (uint32_t)__popcll(idle);
            if (!exhausted && (nIdle >= kStreamRefillMin || !busy)) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&cursor[sq], nIdle);
                base = (uint32_t)__shfl((int)base, 0);
                if (base + nIdle >= nT) exhausted = true;
#ifdef PT_STREAM_PROF
                prof[10]++; prof[11] += base < nT ? min(nIdle, nT - base) : 0u;
#endif
                if (qi == ~0u) {
                    const uint32_t e = base + (uint32_t)__popcll(idle & ltMask);
                    if (e < nT) {
                        const float4 o = q.r0[seg + e], d = q.r1[seg + e];
                        wo = V3(o.x, o.y, o.z); wd = V3(d.x, d.y, d.z); wtmax = d.w;
                        qi = e; tmin = o.w; curInst = ~0u; nodeBase16 = bv.nodeOff16;
                        br = box_ray(V3(o.x, o.y, o.z), V3(d.x, d.y, d.z));
                        h.t = d.w; h.u = h.v = 0.0f; h.inst = ~0u; h.geom = h.prim = h.slot = 0;
                        G = root_node_group(oneInstance); T = root_tri_group(oneInstance, 1u);
                        stack.sp = 0;
                        if (STATS) rayNodes = st.nodes;
                    }
                }
            }
        }
        bool finished = false;
        #pragma unroll 1
        for (uint32_t step = 0; step < kStreamSteps; step++) {
            const bool live = qi != ~0u && !finished;
            const bool top = curInst == ~0u;
            const bool wantNode = live && G.y > 0x00FFFFFFu;
            const bool leaf = live && T.y != 0u;
            // which sections run this step: the one most lanes wait for, and any other with enough lanes of its own
            const uint32_t nNode = (uint32_t)__popcll(wave_ballot(wantNode)), nTri = (uint32_t)__popcll(wave_ballot(leaf && !top)),
                           nEnter = (uint32_t)__popcll(wave_ballot(leaf && top));
            const uint32_t most = max(nNode, max(nTri, nEnter));
            const uint32_t lim = max(kStreamMinLanes, most >> kStreamShareShift);
            const bool doNode = nNode >= lim || nNode == most, doTri = nTri >= lim || nTri == most, doEnter = nEnter >= lim || nEnter == most;
#ifdef PT_STREAM_PROF
            {
                const uint32_t nLive = (uint32_t)__popcll(wave_ballot(live));
                prof[0]++; prof[7] += nLive;
                if (doNode && nNode) { prof[1]++; prof[2] += nNode; }
                if (doTri && nTri) { prof[3]++; prof[4] += nTri; }
                if (doEnter && nEnter) { prof[5]++; prof[6] += nEnter; }
                if (exhausted) { prof[8]++; prof[9] += nLive; }
            }
#endif
            // What this lane does in the step. The walk is bound by the latency of its steps (decide -> fetch -> test -> pop is one
            // dependent chain), not by the instructions in them, so a triangle step takes two triangles of the leaf group at once (the second
            // one's record goes where a node's last units would): C3 +4.6 %, C5 +2.6 %.
            bool aN = false, aT = false, aT2 = false, aE = false;
            uint32_t addrN = 0, addrT = 0, addrT2 = 0, item = 0, item2 = 0;
            if (leaf && top) {
                if (doEnter) { aE = true; item = T.x + (uint32_t)__builtin_ctz(T.y); T.y &= T.y - 1u; addrN = bv.enterOff16 + item * kInst16; }
            } else if (leaf && doTri) {
                aT = true; item = T.x + (uint32_t)__builtin_ctz(T.y); T.y &= T.y - 1u; addrT = triBase16 + item * kTri16;
            }
            if (!aE && wantNode && doNode && !aT) {
                if (T.y) { stack.push(T); T.y = 0u; }                     // postpone (the rest of) the leaf group: the visit brings a new one
                aN = true;
                const uint32_t bit = 31u - (uint32_t)__builtin_clz(G.y);
                G.y &= ~(1u << bit);
                if (G.y > 0x00FFFFFFu) stack.push(G);
                const uint32_t slot = (bit - 24u) ^ (br.octinv4 & 7u);
                addrN = nodeBase16 + (G.x + (uint32_t)__builtin_popcount(G.y & 0xFFu & ~(0xFFFFFFFFu << slot))) * kNode16;
            }
            if (kStreamTriPairs && aT && !aN && T.y) { aT2 = true; item2 = T.x + (uint32_t)__builtin_ctz(T.y); T.y &= T.y - 1u; addrT2 = triBase16 + item2 * kTri16; }
            // ---- all loads of the step (a wave-cooperative gather through LDS -- neighbouring lanes fetching neighbouring 16-byte
            // units of one record -- was tried here and lost 30 %). Registers of lanes that do not load stay undefined and are not
            // read: no zero fill; one address per record, immediate offsets. A0..A4: node, or instance record, or (A2..A4) the second
            // triangle; B0..B2: the triangle, or (B0) the last unit of an instance record.
            f4v A0 = undefined_f4v(), A1 = undefined_f4v(), A2 = undefined_f4v(), A3 = undefined_f4v(), A4 = undefined_f4v();
            f4v B0 = undefined_f4v(), B1 = undefined_f4v(), B2 = undefined_f4v(), C0 = undefined_f4v();
            {
                const f4v* recN = blob.p + addrN; const f4v* recT = blob.p + addrT; const f4v* recT2 = blob.p + addrT2;
                if (aN) { A0 = recN[0]; A1 = recN[1]; A2 = recN[2]; A3 = recN[3]; A4 = recN[4]; }
                if (aE) { B0 = recN[0]; B1 = recN[1]; B2 = recN[2]; C0 = recN[3]; A0 = recN[4]; A1 = recN[5]; A2 = recN[6]; A3 = recN[7]; A4 = recN[8]; }   // entry record: transform | bases | root node
                if (aT) { B0 = recT[0]; B1 = recT[1]; B2 = recT[2]; }
                if (aT2) { A2 = recT2[0]; A3 = recT2[1]; A4 = recT2[2]; }
            }
            // ---- sections
            if (aT) {
                if (STATS) st.tris++;
                float t, u, v;
                if (tri_test(rs, br.o, V3(B0.x, B0.y, B0.z), V3(B1.x, B1.y, B1.z), V3(B2.x, B2.y, B2.z), t, u, v))
                    commit_candidate(ac, __float_as_uint(B2.w), h, tmin, t, u, v, curInst, __float_as_uint(B0.w), __float_as_uint(B1.w), item);
            }
            if (aT2) {
                if (STATS) st.tris++;
                float t, u, v;
                if (tri_test(rs, br.o, V3(A2.x, A2.y, A2.z), V3(A3.x, A3.y, A3.z), V3(A4.x, A4.y, A4.z), t, u, v))
                    commit_candidate(ac, __float_as_uint(A4.w), h, tmin, t, u, v, curInst, __float_as_uint(A2.w), __float_as_uint(A3.w), item2);
            }
            if (aN) {
                if (STATS) st.nodes++;
                const uint32_t hits = wide_node_hits(A0, A1, A2, A3, A4, br, tmin, h.t);
                G = make_uint2(__float_as_uint(A1.x), (hits & 0xFF000000u) | (__float_as_uint(A0.w) >> 24));
                T = make_uint2(__float_as_uint(A1.y), hits & 0x00FFFFFFu);
            }
            if (aE) {                                                    // enter the instance (or skip it: hidden / empty) and visit the root of its BLAS
                const uint32_t cm = __float_as_uint(C0.z), ntri = cm & 0x00FFFFFFu;
                if ((cm >> 24) && ntri != 0u) {
                    const v3 ro = V3(sop3t(B0.x, wo.x, B0.y, wo.y, B0.z, wo.z, B0.w), sop3t(B1.x, wo.x, B1.y, wo.y, B1.z, wo.z, B1.w), sop3t(B2.x, wo.x, B2.y, wo.y, B2.z, wo.z, B2.w));
                    const v3 rd = V3(sop3(B0.x, wd.x, B0.y, wd.y, B0.z, wd.z), sop3(B1.x, wd.x, B1.y, wd.y, B1.z, wd.z), sop3(B2.x, wd.x, B2.y, wd.y, B2.z, wd.z));
                    rs = ray_setup(rd);
                    br = box_ray(ro, rd);
                    nodeBase16 = bv.nodeOff16 + __float_as_uint(C0.x) * kNode16;
                    triBase16 = bv.triOff16 + __float_as_uint(C0.y) * kTri16;
                    stack.push(G); stack.push(T); stack.push(make_uint2(kMarker, 0u));
                    curInst = __float_as_uint(C0.w);
                    if (blas_single_leaf(ntri)) { G = root_node_group(true); T = root_tri_group(true, ntri); }
                    else {                                               // the root node came with the record: one step less per instance
                        if (STATS) st.nodes++;
                        const uint32_t hits = wide_node_hits(A0, A1, A2, A3, A4, br, tmin, h.t);
                        G = make_uint2(__float_as_uint(A1.x), (hits & 0xFF000000u) | (__float_as_uint(A0.w) >> 24));
                        T = make_uint2(__float_as_uint(A1.y), hits & 0x00FFFFFFu);
                    }
                }
            }
            // ---- tail: a lane with nothing at hand pops (LDS), or its ray is done
            if (live && !T.y && G.y <= 0x00FFFFFFu) {
                if (stack.sp > 0) {
                    const uint2 e = stack.pop();
                    if (e.x == kMarker && e.y == 0u) {                    // leave the BLAS: back to the world-space ray
                        if (stack.overflow) { finished = true; stack.sp = 0; }
                        else {
                            T = stack.pop(); G = stack.pop();
                                    br = box_ray(wo, wd); nodeBase16 = bv.nodeOff16; curInst = ~0u;
                        }
                    } else if (e.y > 0x00FFFFFFu) G = e;
                    else T = e;                                           // a postponed leaf group of the current level
                } else finished = true;
            }
        }
        if (finished) {
            const bool hit = h.inst != ~0u && h.t < wtmax;
            if (STATS) atomicMax(&counters->maxNodesPerRay, st.nodes - rayNodes);
            q.hit[seg + qi] = make_uint4(hit ? h.inst : ~0u, h.slot, __float_as_uint(h.u), __float_as_uint(h.v));
            if (WRITE_T) q.r1[seg + qi].w = h.t;
            qi = ~0u;
        }
    }
    if (STATS) { atomicAdd(&counters->nodesVisited, (unsigned long long)st.nodes); atomicAdd(&counters->trianglesTested, (unsigned long long)st.tris); }
    if (st.overflow + stack.overflow) atomicAdd(&counters->stackOverflows, st.overflow + stack.overflow);
#ifdef PT_STREAM_PROF
    // developer build only (tools/stream_prof.py): per-wave tallies of section executions and the lanes in them, through the mismatch record
    if (lane == 0) for (int i = 0; i < 13; i++) atomicAdd((unsigned int*)&counters->mismatchRay[i], prof[i]);
#endif
}

template <bool STATS>
__global__ __launch_bounds__(256) void k_extend(BlobView bv, AlphaContext ac, PathQueue q, uint32_t segCap, const uint32_t* count, DeviceCounters* counters)
{
    const uint32_t sq = blockIdx.x % kSubQueues, bq = blockIdx.x / kSubQueues, nbq = gridDim.x / kSubQueues;
    const uint32_t n = count[sq];
    if (bq == 0 && threadIdx.x == 0) atomicAdd(&counters->secondaryRays, (unsigned long long)n);
    __shared__ uint2 ldsStack[kLdsStackDepth * 256];
    uint2 spill[kStackSize - kLdsStackDepth];
    GroupStack<kLdsStackDepth> stack; stack.init((PT_LDS_AS void*)ldsStack, spill);
    BlobReader<false> blob; blob.p = bv.base;
    TraceStats st; st.nodes = 0; st.tris = 0; st.overflow = 0;
    for (uint32_t local = bq * 256u + threadIdx.x; local < n; local += nbq * 256u) {
        const uint32_t i = sq * segCap + local;
        const float4 o = q.r0[i], d = q.r1[i];
        const Hit h = trace_single<STATS, false, false>(blob, bv, ac, V3(o.x, o.y, o.z), V3(d.x, d.y, d.z), o.w, d.w, stack, &st, nullptr);
        q.hit[i] = make_uint4(h.inst, h.slot, __float_as_uint(h.u), __float_as_uint(h.v));
    }
    if (STATS) { atomicAdd(&counters->nodesVisited, (unsigned long long)st.nodes); atomicAdd(&counters->trianglesTested, (unsigned long long)st.tris); }
    if (st.overflow) atomicAdd(&counters->stackOverflows, st.overflow);
}

static uint32_t persistent_grid(Context& c);

// batch entry points for direct-lighting style consumers (RTXDI bridge shape), see include/ptamd.h
__global__ __launch_bounds__(256) void k_visibility(BlobView bv, AlphaContext ac, const float4* __restrict__ rays, uint32_t count, float4* __restrict__ out,
                                                    DeviceCounters* counters)
{
    __shared__ uint2 ldsStack[kLdsStackDepth * 256];
    uint2 spill[kStackSize - kLdsStackDepth];
    GroupStack<kLdsStackDepth> stack; stack.init((PT_LDS_AS void*)ldsStack, spill);
    BlobReader<false> blob; blob.p = bv.base;
    TraceStats st; st.nodes = 0; st.tris = 0; st.overflow = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        const float4 o = rays[2 * (size_t)i], d = rays[2 * (size_t)i + 1];
        v3 vis;
        const Hit h = trace_single<false, false, true>(blob, bv, ac, V3(o.x, o.y, o.z), V3(d.x, d.y, d.z), o.w, d.w, stack, &st, &vis);
        out[i] = make_float4(vis.x, vis.y, vis.z, h.inst == ~0u ? 1.0f : 0.0f);
    }
    if (st.overflow) atomicAdd(&counters->stackOverflows, st.overflow);
}

// developer aid (pt_debug_trace_ray): one ray through trace_single with a step log
__global__ __launch_bounds__(256) void k_debug_trace(BlobView bv, AlphaContext ac, float4 o, float4 d, uint32_t* log, uint32_t logCap)
{
    __shared__ uint2 ldsStack[kLdsStackDepth * 256];
    uint2 spill[kStackSize - kLdsStackDepth];
    GroupStack<kLdsStackDepth> stack; stack.init((PT_LDS_AS void*)ldsStack, spill);
    BlobReader<false> blob; blob.p = bv.base;
    TraceStats st; st.nodes = 0; st.tris = 0; st.overflow = 0;
    // lane 0 traces the ray and logs; the other lanes trace neighbours of it (rotated a little more per lane) without a log, so that
    // the logged walk runs under the divergence of a real launch
    const float a = 0.02f * (float)threadIdx.x, ca = cosf(a), sa = sinf(a);
    const v3 dd = V3(d.x * ca - d.z * sa, d.y, d.x * sa + d.z * ca);
    trace_single<false, false, false, GroupStack<kLdsStackDepth>, true>(blob, bv, ac, V3(o.x, o.y, o.z), dd, o.w, d.w, stack, &st, nullptr, log, threadIdx.x == 0 ? logCap : 0u);
}

hipError_t launch_debug_trace(Context& c, const SceneView& sv, const float* ray8, uint32_t* devLog, uint32_t logCap)
{
    AlphaContext ac; ac.objects = sv.objects; ac.heap = sv.heap; ac.srgbLut = sv.srgbLut; ac.instances = sv.accel.instances;
    k_debug_trace<<<1, 256, 0, c.stream>>>(c.blob, ac, make_float4(ray8[0], ray8[1], ray8[2], ray8[3]), make_float4(ray8[4], ray8[5], ray8[6], ray8[7]), devLog, logCap);
    return hipGetLastError();
}

// scene-input validation (pt_api.hip validate_scene): every descriptor index ObjectData carries must name a heap entry of the
// right kind. out: error member (1 Vertices, 2 Indices, 3 MotionVectors, 4 TextureMapInfo) | object | descriptor | 1 = wrong kind
__global__ void k_validate_objects(const PtObjectData* __restrict__ objects, uint32_t count, const HeapEntry* __restrict__ heap, uint32_t heapCount, uint32_t* out,
                                   ShadeGeom* __restrict__ shadeGeom)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const PtObjectData* od = &objects[i];
    uint32_t err = 0, desc = 0, kind = 0;
    const uint32_t md[3] = { od->MeshDescriptors.Vertices, od->MeshDescriptors.Indices, od->MeshDescriptors.MotionVectors };
    for (uint32_t k = 0; k < 3 && !err; k++) {
        if (md[k] == ~0u) continue;                       // absent (a mesh that is never hit needs none; a hit dereferences Vertices / Indices only when normals exist)
        if (md[k] >= heapCount) { err = k + 1; desc = md[k]; }
        else if (heap[md[k]].kind != kKindBuffer) { err = k + 1; desc = md[k]; kind = 1; }
    }
    for (uint32_t k = 0; k < 7 && !err; k++) {
        const uint32_t d = od->TextureMapInfoArray[k].Descriptor;
        if (d == ~0u) continue;
        if (d >= heapCount) { err = 4; desc = d; }
        else if (heap[d].kind != kKindTexture2D) { err = 4; desc = d; kind = 1; }
    }
    if (err && atomicCAS(&out[0], 0u, err) == 0u) { out[1] = i; out[2] = desc; out[3] = kind; }
    // the resolved geometry of the object (ShadeGeom): an object without vertex / index buffers has no vertex attributes to fetch
    ShadeGeom sg; sg.vb = nullptr; sg.ib = nullptr; sg.stride = 0; sg.ibStride = 0; sg.nOff = ~0u; sg.tOff = ~0u;
    if (!err && md[0] != ~0u && md[1] != ~0u) {
        sg.vb = (const uint8_t*)heap[md[0]].ptr; sg.ib = heap[md[1]].ptr; sg.stride = od->VertexDesc.Stride; sg.ibStride = heap[md[1]].stride;
        sg.nOff = od->VertexDesc.AttributeOffsets.Normal; sg.tOff = od->VertexDesc.AttributeOffsets.Tangent;
    }
    shadeGeom[i] = sg;
}

hipError_t launch_validate_objects(hipStream_t stream, const PtObjectData* objects, uint32_t count, const HeapEntry* heap, uint32_t heapCount, uint32_t* out, ShadeGeom* shadeGeom)
{
    if (count) k_validate_objects<<<(count + 255) / 256, 256, 0, stream>>>(objects, count, heap, heapCount, out, shadeGeom);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_bsdf_evaluate(const float* __restrict__ q, uint32_t count, float* __restrict__ r)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float* a = q + 20 * (size_t)i;
    BSDFSample b;
    const bool front = a[7] != 0.0f;
    b.Initialize(V3(a), a[3], a[4], a[5], a[6], front);
    const SurfaceVectors sv = surface_vectors(front, V3(a + 8), V3(a + 11));
    const v3 V = V3(a + 14), L = V3(a + 17);
    float w[3]; b.ComputeLobeWeights(sv, V, 0, w);
    float pdf; v3 dif, spc;
    b.EvaluateAll(sv, L, V, w, pdf, dif, spc);
    float* o = r + 8 * (size_t)i;
    o[0] = dif.x; o[1] = dif.y; o[2] = dif.z; o[3] = spc.x; o[4] = spc.y; o[5] = spc.z; o[6] = pdf; o[7] = 0.0f;
}

hipError_t launch_visibility(Context& c, const SceneView& sv, const void* rays, uint32_t count, void* out)
{
    if (!count) return hipSuccess;
    AlphaContext ac; ac.objects = sv.objects; ac.heap = sv.heap; ac.srgbLut = sv.srgbLut; ac.instances = sv.accel.instances;
    k_visibility<<<persistent_grid(c), 256, 0, c.stream>>>(c.blob, ac, (const float4*)rays, count, (float4*)out, c.counters);
    return hipGetLastError();
}

hipError_t launch_bsdf_evaluate(hipStream_t stream, const float* q, uint32_t count, float* r)
{
    if (!count) return hipSuccess;
    k_bsdf_evaluate<<<(count + 255) / 256, 256, 0, stream>>>(q, count, r);
    return hipGetLastError();
}


// dst[y][x] <- gathered per-rank band buffers (PtSharding layout)
struct RankOffsets { uint64_t v[64]; };              // by value in the kernel arguments: no allocation, no copy, no sync

__global__ void k_deinterleave(uint8_t* dst, const uint8_t* src, RankOffsets rankOffsets, uint32_t rankCount, uint32_t bandHeight,
                               uint32_t width, uint32_t height, uint32_t pixelBytes)
{
    const uint64_t rowBytes = (uint64_t)width * pixelBytes, chunks = rowBytes / 4;
    const uint64_t total = chunks * height;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t y = (uint32_t)(t / chunks); const uint64_t cx = t % chunks;
        const uint32_t band = y / bandHeight, rank = band % rankCount, localRow = (band / rankCount) * bandHeight + (y - band * bandHeight);
        ((uint32_t*)(dst + (uint64_t)y * rowBytes))[cx] = ((const uint32_t*)(src + rankOffsets.v[rank] + (uint64_t)localRow * rowBytes))[cx];
    }
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
static uint32_t persistent_grid(Context& c)
{
    if (!c.persistentGrid) {                                        // asked once: hipGetDeviceProperties is a slow host call
        hipDeviceProp_t p;
        const uint32_t g = hipGetDeviceProperties(&p, c.device) == hipSuccess ? (uint32_t)p.multiProcessorCount * 8u : 1024u;   // 1536..4096 blocks perform alike on C2
        c.persistentGrid = (g + kSubQueues - 1) / kSubQueues * kSubQueues;                                                        // whole number of blocks per sub-queue
    }
    return c.persistentGrid;
}

hipError_t launch_gbuffer(Context& c, const SceneView& sv, const FrameView& fv, uint32_t flags, const PtTextures& tx)
{
    if (fv.localRows == 0 || fv.width == 0) return hipSuccess;
    dim3 grid((fv.width + 15) / 16, (fv.localRows + 15) / 16);
    const bool stats = (c.debugFlags & PT_DEBUG_TRAVERSAL_STATS) != 0;
    const bool flat = c.blob.base && c.blob.instCount <= kFlatInstances &&
                      !(c.debugFlags & (PT_DEBUG_TRAVERSAL_V1 | PT_DEBUG_TRAVERSAL_PHASED | PT_DEBUG_BRUTE_FORCE));
    const int mode = !flat ? 0 : (c.blob.bytes <= kBlobLdsMax ? 1 : 2);
    const uint32_t smem = mode == 0 ? 0u : kFlatLdsFixed + (mode == 1 ? c.blob.bytes : 0u);
    #define PT_GB(S, T, M) k_gbuffer<S, T, M><<<grid, 256, smem, c.stream>>>(sv, fv, c.camera, c.sceneData, flags, tx, c.blob, c.counters)
    #define PT_GB_M(S, T) do { if (mode == 0) PT_GB(S, T, 0); else if (mode == 1) PT_GB(S, T, 1); else PT_GB(S, T, 2); } while (0)
    #define PT_GB_T(S) do { if (c.heapHasTextures) PT_GB_M(S, true); else PT_GB_M(S, false); } while (0)
    if (stats) PT_GB_T(true); else PT_GB_T(false);
    #undef PT_GB_T
    #undef PT_GB_M
    #undef PT_GB
    return hipGetLastError();
}

static hipError_t ensure_queues(Context& c, uint32_t capacity, uint32_t iterations)
{
    hipError_t e;
    if (capacity > c.queueCapacity) {
        for (int k = 0; k < 2; k++) {
            PathQueue& q = c.queue[k];
            void** ptrs[6] = { (void**)&q.s0, (void**)&q.s1, (void**)&q.s2, (void**)&q.r0, (void**)&q.r1, (void**)&q.hit };
            for (auto pp : ptrs) { if (*pp) hipFree(*pp); *pp = nullptr; if ((e = hipMalloc(pp, (size_t)capacity * 16)) != hipSuccess) return e; }
        }
        c.queueCapacity = capacity;
    }
    if (capacity > c.primaryCapacity) {                             // 48 bytes per local pixel (capacity >= pixels)
        if (c.primaryRecords) hipFree(c.primaryRecords);
        c.primaryRecords = nullptr; c.primaryCapacity = 0;
        if ((e = hipMalloc((void**)&c.primaryRecords, (size_t)capacity * 48)) != hipSuccess) return e;
        c.primaryCapacity = capacity;
    }
    if (iterations > c.queueCountsCap) {
        if (c.queueCounts) hipFree(c.queueCounts);
        c.queueCountsCap = iterations;
        if ((e = hipMalloc((void**)&c.queueCounts, sizeof(uint32_t) * c.queueCountsCap)) != hipSuccess) return e;
    }
    return hipSuccess;
}

static void timing_begin(Context& c, std::vector<hipEvent_t>& ev, uint32_t k)
{
    if (!c.timing) return;
    while (ev.size() < 2 * (size_t)(k + 1)) { hipEvent_t e; hipEventCreate(&e); ev.push_back(e); }
    hipEventRecord(ev[2 * k], c.stream);
}
static void timing_end(Context& c, std::vector<hipEvent_t>& ev, uint32_t k) { if (c.timing) hipEventRecord(ev[2 * k + 1], c.stream); }

// the launch sequence of one frame after k_set_constants (which also zeroes the queue counters): k_pt_init, then the rounds
static hipError_t enqueue_frame(Context& c, const SceneView& sv, const FrameView& fv, const PtTextures& tx, uint32_t rounds, uint32_t segCap, uint32_t grid)
{
    const uint32_t cstride = kCountStride;                                     // traced + fresh counters + the streaming form's cursor, per round
    float2* aux = c.settings.Denoiser != PT_DENOISER_NONE ? c.pixelAux : nullptr;
    k_pt_init<<<grid, 256, 0, c.stream>>>(fv, c.frameConstants, tx, c.queue[0], aux, segCap, &c.queueCounts[kSubQueues], c.primaryRecords);
    const bool stats = (c.debugFlags & PT_DEBUG_TRAVERSAL_STATS) != 0;
    AlphaContext ac; ac.objects = sv.objects; ac.heap = sv.heap; ac.srgbLut = sv.srgbLut; ac.instances = sv.accel.instances;
    {
        const bool lds = c.blob.bytes <= kBlobLdsMax;
        const bool flat = c.blob.instCount <= kFlatInstances && !(c.debugFlags & PT_DEBUG_TRAVERSAL_PHASED);
        const uint32_t smem = (flat ? kFlatLdsFixed : kExtendLdsFixed) + (lds ? c.blob.bytes : 0u);
        // a scene that does not fit LDS: the streaming form (persistent traversal lanes with ray replacement)
        const uint32_t lockStep = PT_DEBUG_LOCKSTEP | PT_DEBUG_TRAVERSAL_PHASED | PT_DEBUG_BRUTE_FORCE | PT_DEBUG_TRAVERSAL_V1 | PT_DEBUG_UNFUSED_ROUNDS;
        if (!lds && !(c.debugFlags & lockStep)) {
            const bool wt = aux != nullptr;
            for (uint32_t r = 0; r <= rounds; r++) {
                PathQueue& qin = c.queue[r & 1]; PathQueue& qout = c.queue[(r + 1) & 1];
                uint32_t* cin = &c.queueCounts[r * cstride]; uint32_t* cout = &c.queueCounts[(r + 1) * cstride];
                timing_begin(c, c.evShade, c.nShade);
                if (c.heapHasTextures) k_shade<true><<<grid, 256, 0, c.stream>>>(sv, fv, c.frameConstants, tx, qin, qout, aux, segCap, cin, cout, c.primaryRecords, c.blob);
                else k_shade<false><<<grid, 256, 0, c.stream>>>(sv, fv, c.frameConstants, tx, qin, qout, aux, segCap, cin, cout, c.primaryRecords, c.blob);
                timing_end(c, c.evShade, c.nShade); c.nShade++;
                if (r == rounds) break;
                timing_begin(c, c.evExtend, c.nExtend);
                // Fewer, longer-lived waves than the other kernels: a wave only keeps its lanes busy if it refills them many times, and with
                // 8192 waves a 600 k-ray round gives each wave one batch of 64 (lane use then is mean / longest walk of the batch). Alone on
                // the GPU 1024 blocks are best (C3 1.42 -> 1.44 Grays/s); with other frames in flight on other streams, which fill the SIMD
                // slots a small grid leaves, 512 (C3 2.06 -> 2.22, C5 1.62 -> 1.89; 256: 2.02 / 1.80).
                const uint32_t sgrid = std::max(kSubQueues, std::min(grid, c.framesInFlight > 1 ? kStreamGridShared : kStreamGridAlone));
                #define PT_XS(S, W) k_extend_stream<S, W><<<sgrid, 256, kStreamLdsStack, c.stream>>>(c.blob, ac, qout, segCap, cout, cout + 2u * kSubQueues, c.counters)
                if (stats) { if (wt) PT_XS(true, true); else PT_XS(true, false); } else { if (wt) PT_XS(false, true); else PT_XS(false, false); }
                #undef PT_XS
                timing_end(c, c.evExtend, c.nExtend); c.nExtend++;
            }
            return hipGetLastError();
        }
        // fused rounds: everything except the validation / statistics variants, which keep the two-kernel form
        const uint32_t pairOnly = PT_DEBUG_TRAVERSAL_STATS | PT_DEBUG_BRUTE_FORCE | PT_DEBUG_TRAVERSAL_V1 | PT_DEBUG_UNFUSED_ROUNDS;
        if (!(c.debugFlags & pairOnly)) {
            for (uint32_t r = 0; r <= rounds; r++) {                        // queues and counters of round r: in its argument block (launch_raytrace)
                timing_begin(c, c.evRound, c.nRound);
                #define PT_ROUND(T, L, F) k_round<T, L, F><<<grid, 256, smem, c.stream>>>(c.roundArgs + r)
                #define PT_ROUND_F(T, L) do { if (flat) PT_ROUND(T, L, true); else PT_ROUND(T, L, false); } while (0)
                #define PT_ROUND_L(T) do { if (lds) PT_ROUND_F(T, true); else PT_ROUND_F(T, false); } while (0)
                if (c.heapHasTextures) PT_ROUND_L(true); else PT_ROUND_L(false);
                #undef PT_ROUND_L
                #undef PT_ROUND_F
                #undef PT_ROUND
                timing_end(c, c.evRound, c.nRound); c.nRound++;
            }
            return hipGetLastError();
        }
    }
    for (uint32_t r = 0; r <= rounds; r++) {
        PathQueue& qin = c.queue[r & 1]; PathQueue& qout = c.queue[(r + 1) & 1];
        uint32_t* cin = &c.queueCounts[r * cstride]; uint32_t* cout = &c.queueCounts[(r + 1) * cstride];
        timing_begin(c, c.evShade, c.nShade);
        if (c.heapHasTextures) k_shade<true><<<grid, 256, 0, c.stream>>>(sv, fv, c.frameConstants, tx, qin, qout, aux, segCap, cin, cout, c.primaryRecords, c.blob);
        else k_shade<false><<<grid, 256, 0, c.stream>>>(sv, fv, c.frameConstants, tx, qin, qout, aux, segCap, cin, cout, c.primaryRecords, c.blob);
        timing_end(c, c.evShade, c.nShade); c.nShade++;
        if (r == rounds) break;
        timing_begin(c, c.evExtend, c.nExtend);
        const bool lds = c.blob.bytes <= kBlobLdsMax;
        const bool flat = c.blob.instCount <= kFlatInstances && !(c.debugFlags & PT_DEBUG_TRAVERSAL_PHASED);
        const uint32_t smem = (flat ? kFlatLdsFixed : kExtendLdsFixed) + (lds ? c.blob.bytes : 0u);
        if (c.debugFlags & PT_DEBUG_BRUTE_FORCE) k_extend_brute<<<grid, 256, 0, c.stream>>>(sv.accel, c.blob, ac, qout, segCap, cout, c.counters);
        else if (c.debugFlags & PT_DEBUG_TRAVERSAL_V1) {
            if (stats) k_extend<true><<<grid, 256, 0, c.stream>>>(c.blob, ac, qout, segCap, cout, c.counters);
            else k_extend<false><<<grid, 256, 0, c.stream>>>(c.blob, ac, qout, segCap, cout, c.counters);
        } else {
            const bool wt = aux != nullptr;                            // denoiser modes need CommittedRayT
            #define PT_EXT2(S, L, W, F) k_extend2<S, L, W, F><<<grid, 256, smem, c.stream>>>(c.blob, ac, qout, segCap, cout, c.counters)
            #define PT_EXT2_F(S, L, W) do { if (flat) PT_EXT2(S, L, W, true); else PT_EXT2(S, L, W, false); } while (0)
            #define PT_EXT2_W(S, L) do { if (wt) PT_EXT2_F(S, L, true); else PT_EXT2_F(S, L, false); } while (0)
            #define PT_EXT2_L(S) do { if (lds) PT_EXT2_W(S, true); else PT_EXT2_W(S, false); } while (0)
            if (stats) PT_EXT2_L(true); else PT_EXT2_L(false);
            #undef PT_EXT2_L
            #undef PT_EXT2_W
            #undef PT_EXT2_F
            #undef PT_EXT2
        }
        timing_end(c, c.evExtend, c.nExtend); c.nExtend++;
    }
    return hipGetLastError();
}

template <typename T> static void key_add(std::string& k, const T& v) { k.append((const char*)&v, sizeof v); }

hipError_t launch_raytrace(Context& c, const SceneView& sv, const FrameView& fv, const PtTextures& tx)
{
    const PtGraphicsSettings& gs = c.settings;
    const uint32_t npix = fv.width * fv.localRows;
    c.lastIterations = 0;
    if (npix == 0 || gs.SamplesPerPixel == 0) return hipSuccess;
    // a round = one k_shade + one k_extend. Per sample a path spends one round as "fresh" (bounce 0, no ray) and at
    // most Bounces rounds as "traced": spp * (Bounces + 1) rounds empty every queue.
    const uint32_t rounds = gs.SamplesPerPixel * (gs.Bounces + 1u);
    const uint32_t tiles = (npix + 255u) / 256u;
    const uint32_t segCap = (tiles + kSubQueues - 1) / kSubQueues * 256u;     // entries per sub-queue segment
    hipError_t e = ensure_queues(c, segCap * kSubQueues, (rounds + 2) * kCountStride);
    if (e != hipSuccess) return e;
    if (!c.frameConstants && (e = hipMalloc((void**)&c.frameConstants, sizeof(FrameConstants))) != hipSuccess) return e;
    if (gs.Denoiser != PT_DENOISER_NONE && npix > c.pixelAuxCapacity) {
        if (c.pixelAux) hipFree(c.pixelAux);
        c.pixelAux = nullptr; c.pixelAuxCapacity = 0;
        if ((e = hipMalloc((void**)&c.pixelAux, (size_t)npix * sizeof(float2))) != hipSuccess) return e;
        c.pixelAuxCapacity = npix;
    }
    FrameConstants fc; fc.cam = c.camera; fc.sd = c.sceneData; fc.gs = c.settings;
    k_set_constants<<<1, 256, 0, c.stream>>>(fc, c.frameConstants, c.queueCounts, (rounds + 2u) * kCountStride);
    // persistent grid, but never more blocks than the queue has tiles: surplus blocks only cost dispatch slots and LDS that
    // a concurrent frame's kernels (other streams) could use -- this matters for small shards (1/8 of a 1080p frame = 1013 tiles)
    const uint32_t grid = std::min(persistent_grid(c), (tiles + kSubQueues - 1) / kSubQueues * kSubQueues);
    c.lastIterations = rounds + 1;

    // everything the launch sequence depends on: the key of the per-round argument blocks and of the captured graph
    std::string key;
    key_add(key, sv); key_add(key, fv); key_add(key, tx); key_add(key, rounds); key_add(key, segCap); key_add(key, grid);
    key_add(key, c.queue[0]); key_add(key, c.queue[1]); key_add(key, c.queueCounts); key_add(key, c.blob); key_add(key, c.heapHasTextures);
    key_add(key, c.frameConstants); key_add(key, c.primaryRecords); key_add(key, c.stream); key_add(key, c.pixelAux); key_add(key, gs.Denoiser); key_add(key, c.debugFlags);
    key_add(key, c.framesInFlight);
    if (key != c.roundArgsKey || !c.roundArgs) {                              // k_round's argument blocks, one per round (device memory)
        if (rounds + 1 > c.roundArgsCap) {
            if (c.roundArgs) hipFree(c.roundArgs);
            c.roundArgs = nullptr; c.roundArgsCap = 0;
            if ((e = hipMalloc((void**)&c.roundArgs, sizeof(RoundArgs) * (rounds + 1))) != hipSuccess) return e;
            c.roundArgsCap = rounds + 1;
        }
        std::vector<RoundArgs> host(rounds + 1);
        float2* aux = gs.Denoiser != PT_DENOISER_NONE ? c.pixelAux : nullptr;
        for (uint32_t r = 0; r <= rounds; r++) {
            RoundArgs& a = host[r];
            std::memset(&a, 0, sizeof a);
            a.sv = sv; a.fv = fv; a.tx = tx; a.bv = c.blob; a.qin = c.queue[r & 1]; a.qout = c.queue[(r + 1) & 1];
            a.fc = c.frameConstants; a.aux = aux; a.countIn = &c.queueCounts[r * kCountStride]; a.countOut = &c.queueCounts[(r + 1) * kCountStride];
            a.counters = c.counters; a.segCap = segCap; a.primary = c.primaryRecords;
        }
        if ((e = hipMemcpyAsync(c.roundArgs, host.data(), sizeof(RoundArgs) * (rounds + 1), hipMemcpyHostToDevice, c.stream)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(c.stream)) != hipSuccess) return e;    // once per change of the scene / frame geometry, never per frame
        c.roundArgsKey = key;
    }

    // hipGraph replay: launch-bound frames (small shards, tail rounds) cost ~75 launches; a replay is one submission.
    const bool graphable = c.stream != nullptr && !c.timing && !c.disableGraphs &&
                           (c.debugFlags & ~(PT_DEBUG_UNFUSED_ROUNDS | PT_DEBUG_TRAVERSAL_PHASED | PT_DEBUG_LOCKSTEP)) == 0;   // counters / validation variants launch directly
    if (graphable) {
        if (key != c.graphKey || !c.graphExec) {
            if (c.graphExec) { hipGraphExecDestroy(c.graphExec); c.graphExec = nullptr; }
            c.graphKey.clear();
            hipGraph_t graph = nullptr;
            e = hipStreamBeginCapture(c.stream, hipStreamCaptureModeRelaxed);
            if (e == hipSuccess) {
                hipError_t e2 = enqueue_frame(c, sv, fv, tx, rounds, segCap, grid);
                e = hipStreamEndCapture(c.stream, &graph);
                if (e == hipSuccess) e = e2;
            }
            if (e == hipSuccess) e = hipGraphInstantiate(&c.graphExec, graph, nullptr, nullptr, 0);
            if (graph) hipGraphDestroy(graph);
            if (e != hipSuccess) { c.graphExec = nullptr; c.disableGraphs = true; (void)hipGetLastError(); }
            else c.graphKey = key;
        }
        if (c.graphExec) return hipGraphLaunch(c.graphExec, c.stream);
    }
    return enqueue_frame(c, sv, fv, tx, rounds, segCap, grid);
}

hipError_t launch_deinterleave(hipStream_t stream, void* dst, const void* src, const uint64_t* rankOffsetsHost, uint32_t rankCount,
                               uint32_t bandHeight, uint32_t width, uint32_t height, uint32_t pixelBytes)
{
    RankOffsets ro;
    for (uint32_t r = 0; r < 64; r++) ro.v[r] = r < rankCount ? rankOffsetsHost[r] : 0;
    k_deinterleave<<<1024, 256, 0, stream>>>((uint8_t*)dst, (const uint8_t*)src, ro, rankCount, bandHeight, width, height, pixelBytes);
    return hipGetLastError();
}

} // namespace pt
