// pt_shade.hpp -- device functions shared by the render kernels of pt_kernels.hip (G-buffer, fused round) and pt_stream.hip (the two
// kernels of a round for scenes beyond LDS): camera ray, environment light, hit reconstruction (Shaders/RaytracingHelpers.hlsli:73-131,
// Shaders/HitInfo.hlsli:24-65), the block-wide compaction, and the bodies of the shading half: one iteration of the bounce loop
// (Shaders/Raytracing.hlsl:219-364), end of a sample (:372-413), bounce 0 from the primary-surface record (:118-148,193-198).
#pragma once
#include "pt_internal.hpp"

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
    AlphaContext ac; ac.objects = sv.objects; ac.heap = sv.heap; ac.srgbLut = sv.srgbLut; ac.instances = sv.accel.instances; ac.shadeTex = sv.shadeTex;
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
struct HitGeometry { float M[12], W[12]; uint32_t instanceID; TriPacket tp; uint32_t vi[3]; uint32_t triIndex; };      // vi: the triangle's vertex indices; triIndex: its packet in the traversal copy

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
    g.triIndex = __float_as_uint(b1.w) + triSlot;
    g.vi[0] = __float_as_uint(ix.x); g.vi[1] = __float_as_uint(ix.y); g.vi[2] = __float_as_uint(ix.z);
    g.tp.a = make_float4(pa.x, pa.y, pa.z, pa.w); g.tp.b = make_float4(pb.x, pb.y, pb.z, pb.w); g.tp.c = make_float4(pc.x, pc.y, pc.z, pc.w);
    g.instanceID = __float_as_uint(mk.z);
    g.W[0] = w0.x; g.W[1] = w0.y; g.W[2] = w0.z; g.W[3] = w0.w; g.W[4] = w1.x; g.W[5] = w1.y; g.W[6] = w1.z; g.W[7] = w1.w;
    g.W[8] = w2.x; g.W[9] = w2.y; g.W[10] = w2.z; g.W[11] = w2.w;
    g.M[0] = m0.x; g.M[1] = m0.y; g.M[2] = m0.z; g.M[3] = m0.w; g.M[4] = m1.x; g.M[5] = m1.y; g.M[6] = m1.z; g.M[7] = m1.w;
    g.M[8] = m2.x; g.M[9] = m2.y; g.M[10] = m2.z; g.M[11] = m2.w;
    return g;
}

// Object table of a small scene staged in LDS by the fused round kernel (kObjLds16 units of 16 B per object: ShadeGeom | Material): the
// two records every hit looks up by object index -- the resolved geometry, which the normal fetch depends on, and the material -- then
// cost an LDS read instead of a round trip to L2. nullptr: read them where the caller keeps them.
constexpr uint32_t kObjLds16 = (sizeof(ShadeGeom) + sizeof(PtMaterial)) / 16u;
static_assert(sizeof(ShadeGeom) == 32 && sizeof(PtMaterial) == 64 && offsetof(PtObjectData, Material) % 16 == 0, "layout");
typedef const PT_LDS_AS f4v* ObjectTableLds;
// Vertex normals per triangle packet, copied from the caller's vertex buffers at the START OF EVERY FRAME by k_capture_normals (so they are
// as live as a fetch at hit time, frame by frame): the three normals of a hit are then one record fetch by packet index -- 20 bytes, A: nine
// snorm16 minus the last, B: the last -- instead of resolved geometry -> three vertex fetches; for a small scene the fused round kernel keeps
// the records in LDS, and no load of a traced tile's shading goes to memory at all. All null: fetch from the vertex buffers at the hit.
struct ShadeTables {
    ObjectTableLds objects = nullptr;
    const PT_LDS_AS f4v* recALds = nullptr; const PT_LDS_AS uint32_t* recBLds = nullptr;
    const uint4* recA = nullptr; const uint32_t* recB = nullptr;
};

template <bool TEXTURED>
PT_DEV void reconstruct_hit(const SceneView& sv, const HitGeometry& hg, uint32_t inst, float bu, float bv, v3 rayDir, SurfaceHit& h, const ShadeTables& tables = ShadeTables())
{
    const ObjectTableLds objLds = tables.objects;
    const TriPacket& tp = hg.tp;
    const uint32_t geom = __float_as_uint(tp.a.w), prim = __float_as_uint(tp.b.w);
    h.InstanceIndex = inst;
    h.ObjectIndex = hg.instanceID + geom;                  // RaytracingHelpers.hlsli:79
    h.PrimitiveIndex = prim;
    const float* M = hg.M; const float* W = hg.W;
    safe_triangle_spawn_point(V3(tp.a.x, tp.a.y, tp.a.z), V3(tp.b.x, tp.b.y, tp.b.z), V3(tp.c.x, tp.c.y, tp.c.z), bu, bv, M, W,
                              h.ObjectPosition, h.Position, h.FlatNormal, h.PositionOffset);
    // Vertex attributes. With the frame's normal records (ShadeTables) the three normals are ONE fetch by packet index; without them: the
    // object's resolved geometry (one fetch: buffer pointers, stride, offsets -- instead of object record -> descriptor table) and the
    // triangle's vertex indices, which came with the hit geometry (no index-buffer fetch), then the vertices -- two dependent loads from
    // hit to normals where the reference's chain (RaytracingHelpers.hlsli:82-105) has four.
    const bool records = tables.recALds != nullptr || tables.recA != nullptr;
    uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0, w4 = 0;
    if (records) {
        if (tables.recALds) { const f4v a = tables.recALds[hg.triIndex]; w0 = __float_as_uint(a.x); w1 = __float_as_uint(a.y); w2 = __float_as_uint(a.z); w3 = __float_as_uint(a.w); w4 = tables.recBLds[hg.triIndex]; }
        else { const uint4 a = tables.recA[hg.triIndex]; w0 = a.x; w1 = a.y; w2 = a.z; w3 = a.w; w4 = tables.recB[hg.triIndex]; }
    }
    ShadeGeom sg; sg.vb = nullptr; sg.stride = 0; sg.nOff = ~0u; sg.tOff = ~0u; sg.uvOff[0] = sg.uvOff[1] = ~0u; sg._pad = 0;
    if (!records || TEXTURED) {
        if (objLds) {
            const f4v a = objLds[h.ObjectIndex * kObjLds16], b = objLds[h.ObjectIndex * kObjLds16 + 1u];
            f4v* d = (f4v*)&sg; d[0] = a; d[1] = b;
        } else sg = sv.shadeGeom[h.ObjectIndex];
    }
    const uint32_t nOff = sg.nOff;
    if (records ? (w4 >> 16) != 0u : nOff != ~0u) {        // HitInfo.hlsli:52-65 (a record's upper half of B: the mesh has normals)
        v3 nrm[3];
        if (records) {
            nrm[0] = V3(unpack_r16_snorm((int16_t)(w0 & 0xFFFFu)), unpack_r16_snorm((int16_t)(w0 >> 16)), unpack_r16_snorm((int16_t)(w1 & 0xFFFFu)));
            nrm[1] = V3(unpack_r16_snorm((int16_t)(w1 >> 16)), unpack_r16_snorm((int16_t)(w2 & 0xFFFFu)), unpack_r16_snorm((int16_t)(w2 >> 16)));
            nrm[2] = V3(unpack_r16_snorm((int16_t)(w3 & 0xFFFFu)), unpack_r16_snorm((int16_t)(w3 >> 16)), unpack_r16_snorm((int16_t)(w4 & 0xFFFFu)));
        } else {
            #pragma unroll
            for (int k = 0; k < 3; k++) {
                const PT_GLOBAL_AS int16_t* q = gptr<int16_t>(sg.vb + (size_t)sg.stride * hg.vi[k] + nOff);
                nrm[k] = V3(unpack_r16_snorm(q[0]), unpack_r16_snorm(q[1]), unpack_r16_snorm(q[2]));
            }
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
    // GetTextureCoordinates (ShadingHelpers.hlsli:32-51, called at RaytracingHelpers.hlsli:124-130) from the resolved geometry and the
    // triangle's indices at hand: the same three vertices, fetched in the dependency level of the tangents instead of behind
    // object record -> two heap entries -> three index loads
#ifdef PT_AB_OLD_TEXTURE_PATH
    get_texture_coordinates(&sv.objects[h.ObjectIndex], sv.heap, prim, bu, bv, h.TextureCoordinates); return;      // A/B builds: the path of rounds 1-3
#endif
    #pragma unroll
    for (int i = 0; i < 2; i++) {
        h.TextureCoordinates.uv[i][0] = h.TextureCoordinates.uv[i][1] = 0.0f;
        const uint32_t off = sg.uvOff[i];
        if (off == ~0u) continue;
        float a[3][2];
        #pragma unroll
        for (int k = 0; k < 3; k++) {
            const PT_GLOBAL_AS uint16_t* q = gptr<uint16_t>(sg.vb + (size_t)sg.stride * hg.vi[k] + off);
            a[k][0] = f16_to_f32(q[0]); a[k][1] = f16_to_f32(q[1]);
        }
        for (int c = 0; c < 2; c++) h.TextureCoordinates.uv[i][c] = interp1(a[0][c], a[1][c], a[2][c], bu, bv);
    }
}

PT_DEV v3 material_emission(const PtMaterial& m) { return V3(m.EmissiveColor) * m.EmissiveStrength; }

template <bool TEXTURED>
PT_DEV PtMaterial surface_material(const SceneView& sv, SurfaceHit& h, ObjectTableLds objLds = nullptr)
{
    if (!TEXTURED) {                                                     // EvaluateMaterial with every Descriptor == ~0u
        if (!objLds) return sv.objects[h.ObjectIndex].Material;
        PtMaterial m; f4v* d = (f4v*)&m;
        #pragma unroll
        for (uint32_t k = 0; k < 4u; k++) d[k] = objLds[h.ObjectIndex * kObjLds16 + 2u + k];
        return m;
    }
    return evaluate_material(h.ShadingNormal, h.IsFrontFace ? h.Tangent : -h.Tangent, &sv.objects[h.ObjectIndex], sv.heap, sv.srgbLut,
#ifdef PT_AB_OLD_TEXTURE_PATH
                             h.TextureCoordinates, nullptr);
#else
                             h.TextureCoordinates, sv.shadeTex ? sv.shadeTex + (size_t)h.ObjectIndex * kTextureSlots : nullptr);   // ShadingHelpers.hlsli:161-235
#endif
}

// ---------------------------------------------------------------------------------------------
// wavefront path tracer
// ---------------------------------------------------------------------------------------------
// Queue geometry. The path queue is cut into nsq = 1 << sqShift independent sub-queues (segments of segCap entries,
// each with its own counter): a single returning atomic on one word saturates near 88 M/s on MI355X, which
// at one atomic per wave made compaction the bottleneck of the whole frame. Pixel tile t (256 pixels) is
// dealt to sub-queue t % nsq, a path never leaves its sub-queue, so a segment can never overflow and
// every sub-queue samples the whole image (balanced). One atomic per 256-thread block and tile.

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
    { const float ipdf = 1.0f / pdf; p.thr = p.thr * V3(f.x * ipdf, f.y * ipdf, f.z * ipdf); }   // :346 (float3 / float = the vector times ONE reciprocal: arithmetic spec)
    if (gs.IsRussianRouletteEnabled && p.bounce > 3) {                                        // :348-356
        const float prob = fmaxf(p.thr.x, fmaxf(p.thr.y, p.thr.z));
        if (rng_float(p.rng) >= prob) return false;
        { const float iprob = 1.0f / prob; p.thr = V3(p.thr.x * iprob, p.thr.y * iprob, p.thr.z * iprob); }
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
    const BlobReader<LDS>& blob; const BlobView& bv; ShadeTables tables = ShadeTables();
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
        reconstruct_hit<TEXTURED>(sv, geometry.load(hr.x, hr.y), hr.x, __uint_as_float(hr.z), __uint_as_float(hr.w), rayDir, h, geometry.tables);
        PT_PROF_MARK(prof, 5);
        const PtMaterial m = surface_material<TEXTURED>(sv, h, geometry.tables.objects);
        BSDFSample bs;
        bs.Initialize(V3(m.BaseColor), m.Metallic, m.Roughness, m.IOR, m.Transmission, h.IsFrontFace);
        goes = scatter(gs, p, h, bs, material_emission(m), rayDir, newO, newD, lobe);
    }
    if (goes) toTraced = true;
    else toFresh = end_sample(gs, tx, aux, p);
}

// A fresh path: bounce 0 on the primary surface rebuilt from the G-buffer, Raytracing.hlsl:118-148,193-198
// (r0, r1, r2: the pixel's primary-surface record)
PT_DEV void shade_fresh_record(const FrameView& fv, const PtCamera& cam, const PtGraphicsSettings& gs, const PtTextures& tx, float2* aux, uint4 r0, uint4 r1, uint4 r2,
                               PathRegs& p, bool& toTraced, bool& toFresh, v3& newO, v3& newD)
{
    const uint32_t pixel = p.pixel;
    const uint32_t px = pixel % fv.width, py = global_row(fv, pixel / fv.width);
    float uu, vv;
    const RayDesc primaryRay = generate_pinhole_ray(cam, px, py, fv.width, fv.height, uu, vv);   // :110-126
    const v3 rayDir = primaryRay.d;
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

PT_DEV void shade_fresh(const FrameView& fv, const PtCamera& cam, const PtGraphicsSettings& gs, const PtTextures& tx, float2* aux, const uint4* __restrict__ primary,
                        PathRegs& p, bool& toTraced, bool& toFresh, v3& newO, v3& newD, RoundProf* prof = nullptr)
{
    const uint32_t pixel = p.pixel;
    const uint4 r0 = primary[3 * (size_t)pixel], r1 = primary[3 * (size_t)pixel + 1], r2 = primary[3 * (size_t)pixel + 2];
    PT_PROF_WAIT(); PT_PROF_MARK(prof, 13);
    shade_fresh_record(fv, cam, gs, tx, aux, r0, r1, r2, p, toTraced, toFresh, newO, newD);
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

} // namespace pt
