// pt_kernels.hip -- the render kernels (gfx950): primary-ray G-buffer and the wavefront path tracer.
//
//   k_gbuffer   <- Shaders/GBufferGeneration.hlsl:116-232 (main) + CastRay, Shaders/RaytracingHelpers.hlsli:57-133
//   k_round (and its two-kernel form k_pt_init / k_shade [pt_stream.hip] / k_extend2)  <- Shaders/Raytracing.hlsl:103-415 (RayGeneration,
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

#include "pt_shade.hpp"

namespace pt {

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
// wavefront path tracer: per-frame set-up kernels, validation / lock-step traversal variants, the fused round kernel, host launchers
// ---------------------------------------------------------------------------------------------
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
                                                 uint4* __restrict__ primary, uint32_t sqShift)
{
    __shared__ uint32_t lds[8];
    const PtGraphicsSettings& gs = fc->gs;
    const uint32_t npix = fv.width * fv.localRows;
    const uint32_t nsq = 1u << sqShift, sq = blockIdx.x & (nsq - 1u), bq = blockIdx.x >> sqShift, nbq = gridDim.x >> sqShift;
    for (uint32_t j = bq; (j * nsq + sq) * 256u < npix; j += nbq) {
        const uint32_t p = (j * nsq + sq) * 256u + threadIdx.x;
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

// k_pt_init and round 0 in ONE launch (the product paths): round 0 has fresh entries only, one per primary-hit pixel, in the order k_pt_init
// wrote them -- so the block that gathers a tile of pixels into primary-surface records shades their first bounce right away, out of
// the registers that hold the record, and emits the survivors as round 0 would have. The fresh state (48 B written, 48 B read), the record
// read-back (48 B) and one launch per frame go away; values and draws are those of k_pt_init + shade_fresh.
__global__ __launch_bounds__(256) void k_pt_first(FrameView fv, const FrameConstants* __restrict__ fc, PtTextures tx, PathQueue qout, float2* aux,
                                                                                             uint32_t segCap, uint32_t* countOut, uint4* __restrict__ primary, uint32_t sqShift)
{
    __shared__ uint32_t lds[32];
    uint32_t emits = 0;
    const PtCamera& cam = fc->cam; const PtGraphicsSettings& gs = fc->gs;
    const uint32_t npix = fv.width * fv.localRows;
    const uint32_t nsq = 1u << sqShift, sq = blockIdx.x & (nsq - 1u), bq = blockIdx.x >> sqShift, nbq = gridDim.x >> sqShift;
    const uint32_t seg = sq * segCap;
    for (uint32_t j = bq; (j * nsq + sq) * 256u < npix; j += nbq) {
        const uint32_t pix = (j * nsq + sq) * 256u + threadIdx.x;
        bool alive = false;
        if (pix < npix) alive = isfinite(((const float*)tx.Position)[4 * (size_t)pix + 3]);
        bool toTraced = false, toFresh = false;
        PathRegs p; p.thr = V3(1.0f, 1.0f, 1.0f); p.srad = V3(0, 0, 0); p.rsum = V3(0, 0, 0); p.pixel = pix; p.rng = 0; p.sample = 0; p.bounce = 0;
        v3 newO = V3(0, 0, 0), newD = V3(0, 0, 1);
        if (alive) {
            const uint32_t x = pix % fv.width, y = global_row(fv, pix / fv.width);
            p.rng = rng_init(x, y, gs.FrameIndex);
            if (aux) aux[pix] = make_float2(INFINITY, 1.0f);                             // hitDistance = inf, isDiffuse = true (:188-189)
            const uint2 nr = ((const uint2*)tx.NormalRoughness)[pix], rad = ((const uint2*)tx.Radiance)[pix];
            const uint4 r0 = ((const uint4*)tx.Position)[pix];
            const uint4 r1 = make_uint4(nr.x, nr.y, ((const uint32_t*)tx.FlatNormal)[pix], ((const uint32_t*)tx.GeometricNormal)[pix]);
            const uint4 r2 = make_uint4(((const uint32_t*)tx.BaseColorMetalness)[pix], rad.x, rad.y,
                                        (uint32_t)((const uint16_t*)tx.IOR)[pix] | ((uint32_t)((const uint8_t*)tx.Transmission)[pix] << 16));
            primary[3 * (size_t)pix] = r0; primary[3 * (size_t)pix + 1] = r1; primary[3 * (size_t)pix + 2] = r2;
            shade_fresh_record(fv, cam, gs, tx, aux, r0, r1, r2, p, toTraced, toFresh, newO, newD);
        }
        emit_tile(qout, seg, segCap, &countOut[sq], &countOut[nsq + sq], lds + ((emits++ & 1u) << 4), toTraced, toFresh, p, newO, newD);
    }
}

__global__ __launch_bounds__(256) void k_extend_brute(AccelView av, BlobView bv, AlphaContext ac, PathQueue q, uint32_t segCap, const uint32_t* count, DeviceCounters* counters, uint32_t sqShift)
{
    const uint32_t nsq = 1u << sqShift, sq = blockIdx.x & (nsq - 1u), bq = blockIdx.x >> sqShift, nbq = gridDim.x >> sqShift;
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
__global__ __launch_bounds__(256) void k_extend2(BlobView bv, AlphaContext ac, PathQueue q, uint32_t segCap, const uint32_t* count, DeviceCounters* counters, uint32_t sqShift)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr uint32_t kFixed = FLAT ? kFlatLdsFixed : kExtendLdsFixed;
    const uint32_t nsq = 1u << sqShift, sq = blockIdx.x & (nsq - 1u), bq = blockIdx.x >> sqShift, nbq = gridDim.x >> sqShift;
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
    const FrameConstants* fc; float2* aux; const uint32_t* countIn; uint32_t* countOut; DeviceCounters* counters; uint32_t segCap;
    uint32_t objectsInLds;                   // objects whose resolved geometry + material are staged behind the blob (0: none)
    const uint4* recA; const uint32_t* recB; // the frame's normal records (null: none)
    uint32_t recordsInLds, sqShift;          // ... and how many of them are staged behind the object table (all or none); log2 of the number of sub-queues
    const uint4* primary;
};

template <bool TEXTURED, bool LDS, bool FLAT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_round(const RoundArgs* __restrict__ A, uint32_t sqBase, uint32_t sqCount)
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
    const uint32_t sqShift = A->sqShift;
    const uint32_t nsq = 1u << sqShift, bq = blockIdx.x / sqCount, sq = sqBase + (blockIdx.x - bq * sqCount), nbq = gridDim.x / sqCount;   // this launch serves sub-queues [sqBase, sqBase + sqCount): one chain of the frame
    const uint32_t nT = countIn[sq], nF = countIn[nsq + sq];
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
        ShadeTables tables;
        tables.recA = A->recA; tables.recB = A->recB;
        if constexpr (LDS) {
            f4v* dst = (f4v*)(smem + kFixed);
            const uint32_t n16 = bv.bytes / 16u;
            for (uint32_t k = threadIdx.x; k < n16; k += 256u) dst[k] = bv.base[k];
            const uint32_t nobj = A->objectsInLds;                // behind the blob: the object table (pt_shade.hpp ObjectTableLds)
            for (uint32_t k = threadIdx.x; k < nobj * kObjLds16; k += 256u) {
                const uint32_t o = k / kObjLds16, part = k - o * kObjLds16;
                dst[n16 + k] = part < 2u ? ((const f4v*)&sv.shadeGeom[o])[part] : ((const f4v*)&sv.objects[o].Material)[part - 2u];
            }
            const uint32_t nrec = A->recordsInLds, recBase = n16 + nobj * kObjLds16;       // then the normal records: A x nrec | B x nrec
            for (uint32_t k = threadIdx.x; k < nrec; k += 256u) {
                const uint4 a = A->recA[k];
                dst[recBase + k] = (f4v){ __uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w) };
                ((uint32_t*)(dst + recBase + nrec))[k] = A->recB[k];
            }
            __syncthreads();
            blob.p = (const PT_LDS_AS f4v*)(smem + kFixed);
            if (nobj) tables.objects = (ObjectTableLds)(smem + kFixed) + n16;
            if (nrec) { tables.recALds = (const PT_LDS_AS f4v*)(smem + kFixed) + recBase; tables.recBLds = (const PT_LDS_AS uint32_t*)((const PT_LDS_AS f4v*)(smem + kFixed) + recBase + nrec); }
        } else {
            blob.p = bv.base;
        }
        AlphaContext ac; ac.objects = sv.objects; ac.heap = sv.heap; ac.srgbLut = sv.srgbLut; ac.instances = sv.accel.instances; ac.shadeTex = sv.shadeTex;
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
                shade_traced<TEXTURED>(sv, GeometryFromBlob<LDS>{ blob, bv, tables }, sd, gs, tx, aux, p, make_uint4(h.inst, h.slot, __float_as_uint(h.u), __float_as_uint(h.v)), h.t,
                                       V3(d.x, d.y, d.z), toTraced, toFresh, newO, newD, prof);
            }
            PT_PROF_MARK(prof, 6);
            emit_tile(qout, seg, segCap, &countOut[sq], &countOut[nsq + sq], lds + ((emits++ & 1u) << 4), toTraced, toFresh, p, newO, newD);
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
        emit_tile(qout, seg, segCap, &countOut[sq], &countOut[nsq + sq], lds + ((emits++ & 1u) << 4), toTraced, toFresh, p, newO, newD);
        PT_PROF_MARK(prof, 15);
    }
#ifdef PT_ROUND_PROF
    // developer build only (tools/round_prof.py): per-wave section clocks (in units of 64 cycles) and item tallies through the mismatch record
    PT_PROF_MARK(prof, 8);
    if ((threadIdx.x & 63u) == 0u) for (int k = 0; k < 16; k++) atomicAdd((unsigned int*)&counters->mismatchRay[k], (k < 9 || k >= 12) ? (prof->acc[k] >> 6) : prof->acc[k]);
#endif
}

template <bool STATS>
__global__ __launch_bounds__(256) void k_extend(BlobView bv, AlphaContext ac, PathQueue q, uint32_t segCap, const uint32_t* count, DeviceCounters* counters, uint32_t sqShift)
{
    const uint32_t nsq = 1u << sqShift, sq = blockIdx.x & (nsq - 1u), bq = blockIdx.x >> sqShift, nbq = gridDim.x >> sqShift;
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
    AlphaContext ac; ac.objects = sv.objects; ac.heap = sv.heap; ac.srgbLut = sv.srgbLut; ac.instances = sv.accel.instances; ac.shadeTex = sv.shadeTex;
    k_debug_trace<<<1, 256, 0, c.stream>>>(c.blob, ac, make_float4(ray8[0], ray8[1], ray8[2], ray8[3]), make_float4(ray8[4], ray8[5], ray8[6], ray8[7]), devLog, logCap);
    return hipGetLastError();
}

// scene-input validation (pt_api.hip validate_scene): every descriptor index ObjectData carries must name a heap entry of the
// right kind. out: error member (1 Vertices, 2 Indices, 3 MotionVectors, 4 TextureMapInfo) | object | descriptor | 1 = wrong kind
__global__ void k_validate_objects(const PtObjectData* __restrict__ objects, uint32_t count, const HeapEntry* __restrict__ heap, uint32_t heapCount, uint32_t* out,
                                   ShadeGeom* __restrict__ shadeGeom, HeapEntry* __restrict__ shadeTex)
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
    for (uint32_t k = 0; k < kTextureSlots; k++) {         // the resolved texture slots (pt_texture.hpp TextureSlots): a copy of the heap entry, the coordinate set in `kind`
        HeapEntry e; e.ptr = nullptr; e.bytes = 0; e.stride = 0; e.kind = 0;
        const uint32_t d = od->TextureMapInfoArray[k].Descriptor;
        if (!err && d != ~0u) { e = heap[d]; e.kind = od->TextureMapInfoArray[k].TextureCoordinateIndex & 1u; }
        shadeTex[(size_t)i * kTextureSlots + k] = e;
    }
    if (err && atomicCAS(&out[0], 0u, err) == 0u) { out[1] = i; out[2] = desc; out[3] = kind; }
    // the resolved geometry of the object (ShadeGeom): an object without vertex / index buffers has no vertex attributes to fetch
    ShadeGeom sg; sg.vb = nullptr; sg.stride = 0; sg.nOff = ~0u; sg.tOff = ~0u; sg.uvOff[0] = sg.uvOff[1] = ~0u; sg._pad = 0;
    if (!err && md[0] != ~0u && md[1] != ~0u) {
        sg.vb = (const uint8_t*)heap[md[0]].ptr; sg.stride = od->VertexDesc.Stride;
        sg.nOff = od->VertexDesc.AttributeOffsets.Normal; sg.tOff = od->VertexDesc.AttributeOffsets.Tangent;
        sg.uvOff[0] = od->VertexDesc.AttributeOffsets.TextureCoordinates[0]; sg.uvOff[1] = od->VertexDesc.AttributeOffsets.TextureCoordinates[1];
    }
    shadeGeom[i] = sg;
}

hipError_t launch_validate_objects(hipStream_t stream, const PtObjectData* objects, uint32_t count, const HeapEntry* heap, uint32_t heapCount, uint32_t* out, ShadeGeom* shadeGeom, HeapEntry* shadeTex)
{
    if (count) k_validate_objects<<<(count + 255) / 256, 256, 0, stream>>>(objects, count, heap, heapCount, out, shadeGeom, shadeTex);
    return hipGetLastError();
}

// The frame's normal records (pt_shade.hpp ShadeTables): one per triangle packet of the traversal copy, read from the vertex buffers the
// objects name NOW. A bottom level is described by the objects of the first instance that refers to it (BlasEntry::objectBase);
// k_check_shared_geometry, run with the validation, makes sure every other instance of it resolves to the same vertex data -- if one
// does not, no records are kept and hits fetch their vertices themselves. grid.y: bottom levels of the top-level build's table.
__global__ __launch_bounds__(256) void k_capture_normals(const BlasEntry* __restrict__ table, const ShadeGeom* __restrict__ shadeGeom, uint4* __restrict__ recA, uint32_t* __restrict__ recB)
{
    const BlasEntry e = table[blockIdx.y];
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < e.triCount; k += gridDim.x * 256u) {
        const uint32_t geom = __float_as_uint(e.tris[k].a.w);
        const ShadeGeom sg = shadeGeom[e.objectBase + geom];
        uint32_t n[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 }, has = 0;
        if (sg.vb && sg.nOff != ~0u) {
            const uint4 ix = e.idx[k];
            const uint32_t vi[3] = { ix.x, ix.y, ix.z };
            has = 1u;
            #pragma unroll
            for (int v = 0; v < 3; v++) {
                const PT_GLOBAL_AS uint16_t* q = gptr<uint16_t>(sg.vb + (size_t)sg.stride * vi[v] + sg.nOff);
                n[3 * v] = q[0]; n[3 * v + 1] = q[1]; n[3 * v + 2] = q[2];
            }
        }
        recA[e.triBase + k] = make_uint4(n[0] | (n[1] << 16), n[2] | (n[3] << 16), n[4] | (n[5] << 16), n[6] | (n[7] << 16));
        recB[e.triBase + k] = n[8] | (has << 16);
    }
}

__global__ void k_check_shared_geometry(const InstanceSource* __restrict__ src, const BlasEntry* __restrict__ table, uint32_t n, const ShadeGeom* __restrict__ shadeGeom, uint32_t* out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t id = src[i].instanceID, base = table[src[i].blasSlot].objectBase, ng = table[src[i].blasSlot].geometryCount;
    if (id == base) return;
    for (uint32_t g = 0; g < ng; g++) {
        const ShadeGeom a = shadeGeom[id + g], b = shadeGeom[base + g];
        if (a.vb != b.vb || a.stride != b.stride || a.nOff != b.nOff) { out[4] = 1u; return; }
    }
}

hipError_t launch_check_shared_geometry(hipStream_t stream, const InstanceSource* src, const BlasEntry* table, uint32_t n, const ShadeGeom* shadeGeom, uint32_t* out)
{
    if (n) k_check_shared_geometry<<<(n + 255) / 256, 256, 0, stream>>>(src, table, n, shadeGeom, out);
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
    AlphaContext ac; ac.objects = sv.objects; ac.heap = sv.heap; ac.srgbLut = sv.srgbLut; ac.instances = sv.accel.instances; ac.shadeTex = sv.shadeTex;
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
        c.persistentGrid = (g + kSubQueuesMax - 1) / kSubQueuesMax * kSubQueuesMax;                                                        // whole number of blocks per sub-queue
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

// Objects whose resolved geometry + material k_round stages in LDS behind the blob: all of them, if that does not cost the kernel its fourth
// workgroup per CU (160 KB / 4, the kernel's static words and the 512-byte allocation granule counted); otherwise none.
static uint32_t lds_bytes_of_records(uint32_t n) { return (n * 20u + 15u) / 16u * 16u; }
uint32_t round_objects_in_lds(const Context& c, uint32_t objectCount, bool haveShadeGeom)
{
    if (c.blob.bytes > kBlobLdsMax || !objectCount || !haveShadeGeom) return 0u;
    const bool flat = c.blob.instCount <= kFlatInstances && !(c.debugFlags & PT_DEBUG_TRAVERSAL_PHASED);
    const uint32_t bytes = (flat ? kFlatLdsFixed : kExtendLdsFixed) + c.blob.bytes + objectCount * kObjLds16 * 16u + 128u;
    return (bytes + 511u) / 512u * 512u <= 160u * 1024u / 4u ? objectCount : 0u;
}
static uint32_t round_objects_in_lds(const Context& c, const SceneView& sv) { return round_objects_in_lds(c, sv.objectCount, sv.shadeGeom != nullptr); }
// ... and the frame's normal records behind them, under the same rule
uint32_t round_records_in_lds(const Context& c, uint32_t objectCount, bool haveShadeGeom)
{
    const uint32_t nobj = round_objects_in_lds(c, objectCount, haveShadeGeom);
    if (!nobj || !normal_records_usable(c)) return 0u;
    const bool flat = c.blob.instCount <= kFlatInstances && !(c.debugFlags & PT_DEBUG_TRAVERSAL_PHASED);
    const uint32_t bytes = (flat ? kFlatLdsFixed : kExtendLdsFixed) + c.blob.bytes + nobj * kObjLds16 * 16u + lds_bytes_of_records(c.blob.triCount) + 128u;
    return (bytes + 511u) / 512u * 512u <= 160u * 1024u / 4u ? c.blob.triCount : 0u;
}
static uint32_t round_records_in_lds(const Context& c, const SceneView& sv) { return round_records_in_lds(c, sv.objectCount, sv.shadeGeom != nullptr); }

// ---- the launch sequence of one frame after k_set_constants (which also zeroes the queue counters) ----------------------------------
// preamble: the frame's normal records and k_pt_first (= k_pt_init + round 0), on the context's stream;
// then the rounds -- as `chains` independent chains of launches, one per group of sub-queues. A path never leaves its sub-queue, so from
// k_pt_first on the groups need nothing from each other: chain g takes sub-queues [sqBase, sqBase + sqCount) through all the rounds.
// chains > 1 (launch_raytrace): every chain is a linear hipGraph of its own, replayed on a stream of its own behind an event recorded after the
// preamble, and the context's stream waits for all of them -- the tail of one chain's launch (a few long walks, a few blocks) overlaps the other
// chains' launches. That is what several frames in flight do BETWEEN frames, done inside ONE frame: what a renderer that presents one frame at a
// time needs (the reference: Source/App.cpp:167). The gain is small, and the reason is instructive: the chains are statistically identical, so
// their launches start and drain together -- every chain still pays (rounds x the longest walk of a round), which is the critical path of a
// frame; frames in flight hide it because their phases differ. Plain streams and events, as frames in flight use them (parallel branches inside
// ONE captured graph were measured too: the same times on C2).
#ifndef PT_AB_CHAINS
#define PT_AB_CHAINS 3
#endif
struct FrameForm { bool streaming, fused, first; };
static FrameForm frame_form(const Context& c)
{
    const uint32_t lockStepFlags = PT_DEBUG_LOCKSTEP | PT_DEBUG_TRAVERSAL_PHASED | PT_DEBUG_BRUTE_FORCE | PT_DEBUG_TRAVERSAL_V1 | PT_DEBUG_UNFUSED_ROUNDS;
    const uint32_t pairOnlyFlags = PT_DEBUG_TRAVERSAL_STATS | PT_DEBUG_BRUTE_FORCE | PT_DEBUG_TRAVERSAL_V1 | PT_DEBUG_UNFUSED_ROUNDS;
    FrameForm f;
    f.streaming = c.blob.bytes > kBlobLdsMax && !(c.debugFlags & lockStepFlags);   // a scene that does not fit LDS: persistent traversal lanes with ray replacement
    f.fused = !f.streaming && !(c.debugFlags & pairOnlyFlags);                     // fused rounds: everything except the validation / statistics variants
    f.first = f.streaming || f.fused;                                              // the product paths start with k_pt_first; the validation variants keep k_pt_init and round 0 apart
    return f;
}
static uint32_t frame_chains(const Context& c, bool ownStreams)
{
    // on streams of their own: the library's choice unless the caller made one; direct launches (per-launch events, the default stream): only
    // what the caller asked for, one chain after the other on the context's stream
    // The library's choice, from the A/B runs in profiles/r04_ab/frame_chains.jsonl: three chains for the streaming form when the frame has the GPU
    // to itself (C3 +2 %, C5 +7.5 %, c3t +2 % per frame), one otherwise -- the fused round kernel gains nothing (C2 -1 %), and with other frames
    // in flight the extra streams only crowd the three hardware queues the runtime exposes (C3 -40 %; four chains: -35 % even alone).
#ifdef PT_AB_CHAINS_FORCE
    const uint32_t choice = PT_AB_CHAINS_FORCE;                   // A/B builds: this many chains whatever the form
#else
    const uint32_t choice = (frame_form(c).streaming && c.framesInFlight <= 1) ? (uint32_t)PT_AB_CHAINS : 1u;
#endif
    const uint32_t want = c.chains ? c.chains : (ownStreams ? choice : 1u);
    return std::max(1u, std::min({ want, Context::kMaxChains, 1u << c.sqShift }));
}

static hipError_t enqueue_preamble(Context& c, const SceneView& sv, const FrameView& fv, const PtTextures& tx, uint32_t segCap, uint32_t grid)
{
    const uint32_t nsq = 1u << c.sqShift, cstride = 3u * nsq;                  // traced + fresh counters + the streaming form's cursor, per round
    float2* aux = c.settings.Denoiser != PT_DENOISER_NONE ? c.pixelAux : nullptr;
    if (normal_records_usable(c))                          // the frame's normal records, from the vertex buffers as they are now
        k_capture_normals<<<dim3(std::min((c.blasTableMaxTris + 255u) / 256u, 64u), c.blasTableCount), 256, 0, c.stream>>>(c.blasTableDev, sv.shadeGeom, c.shadeRecA, c.shadeRecB);
    if (frame_form(c).first) k_pt_first<<<grid, 256, 0, c.stream>>>(fv, c.frameConstants, tx, c.queue[1], aux, segCap, &c.queueCounts[cstride], c.primaryRecords, c.sqShift);
    else k_pt_init<<<grid, 256, 0, c.stream>>>(fv, c.frameConstants, tx, c.queue[0], aux, segCap, &c.queueCounts[nsq], c.primaryRecords, c.sqShift);
    return hipGetLastError();
}

// chain g of `chains`: the rounds of its sub-queues, on stream s (product forms only: frame_form(c).first)
static hipError_t enqueue_chain(Context& c, const SceneView& sv, const FrameView& fv, const PtTextures& tx, uint32_t rounds, uint32_t segCap, uint32_t grid,
                                uint32_t g, uint32_t chains, hipStream_t s)
{
    const uint32_t nsq = 1u << c.sqShift, cstride = 3u * nsq;
    float2* aux = c.settings.Denoiser != PT_DENOISER_NONE ? c.pixelAux : nullptr;
    const FrameForm form = frame_form(c);
    const bool stats = (c.debugFlags & PT_DEBUG_TRAVERSAL_STATS) != 0;
    AlphaContext ac; ac.objects = sv.objects; ac.heap = sv.heap; ac.srgbLut = sv.srgbLut; ac.instances = sv.accel.instances; ac.shadeTex = sv.shadeTex;
    const uint32_t perSq = std::max(1u, grid / nsq);                           // blocks per sub-queue
    const uint32_t sqBase = (uint32_t)((uint64_t)nsq * g / chains), sqCount = (uint32_t)((uint64_t)nsq * (g + 1) / chains) - sqBase;
    if (form.streaming) {                                                      // round 0's shading half ran inside k_pt_first
        const bool wt = aux != nullptr;
        for (uint32_t r = 0; r <= rounds; r++) {
            PathQueue& qin = c.queue[r & 1]; PathQueue& qout = c.queue[(r + 1) & 1];
            uint32_t* cin = &c.queueCounts[r * cstride]; uint32_t* cout = &c.queueCounts[(r + 1) * cstride];
            if (r > 0) {
                timing_begin(c, c.evShade, c.nShade);
                launch_shade(c, sv, fv, tx, qin, qout, aux, segCap, cin, cout, perSq * sqCount, s, sqBase, sqCount);
                timing_end(c, c.evShade, c.nShade); c.nShade++;
            }
            if (r == rounds) break;
            timing_begin(c, c.evExtend, c.nExtend);
            launch_extend_stream(c, ac, qout, segCap, cout, cout + 2u * nsq, grid, stats, wt, s, sqBase, sqCount);
            timing_end(c, c.evExtend, c.nExtend); c.nExtend++;
        }
        return hipGetLastError();
    }
    const bool lds = c.blob.bytes <= kBlobLdsMax;
    const bool flat = c.blob.instCount <= kFlatInstances && !(c.debugFlags & PT_DEBUG_TRAVERSAL_PHASED);
    const uint32_t smem = (flat ? kFlatLdsFixed : kExtendLdsFixed) + (lds ? c.blob.bytes : 0u) + round_objects_in_lds(c, sv) * kObjLds16 * 16u + lds_bytes_of_records(round_records_in_lds(c, sv));
    for (uint32_t r = 1; r <= rounds; r++) {                        // queues and counters of round r: in its argument block (launch_raytrace); round 0 ran inside k_pt_first
        timing_begin(c, c.evRound, c.nRound);
        #define PT_ROUND(T, L, F) k_round<T, L, F><<<perSq * sqCount, 256, smem, s>>>(c.roundArgs + r, sqBase, sqCount)
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

// the validation / statistics variants: two kernels per round over all sub-queues, on the context's stream
static hipError_t enqueue_validation_rounds(Context& c, const SceneView& sv, const FrameView& fv, const PtTextures& tx, uint32_t rounds, uint32_t segCap, uint32_t grid)
{
    const uint32_t nsq = 1u << c.sqShift, cstride = 3u * nsq;
    float2* aux = c.settings.Denoiser != PT_DENOISER_NONE ? c.pixelAux : nullptr;
    const bool stats = (c.debugFlags & PT_DEBUG_TRAVERSAL_STATS) != 0;
    AlphaContext ac; ac.objects = sv.objects; ac.heap = sv.heap; ac.srgbLut = sv.srgbLut; ac.instances = sv.accel.instances; ac.shadeTex = sv.shadeTex;
    for (uint32_t r = 0; r <= rounds; r++) {
        PathQueue& qin = c.queue[r & 1]; PathQueue& qout = c.queue[(r + 1) & 1];
        uint32_t* cin = &c.queueCounts[r * cstride]; uint32_t* cout = &c.queueCounts[(r + 1) * cstride];
        timing_begin(c, c.evShade, c.nShade);
        launch_shade(c, sv, fv, tx, qin, qout, aux, segCap, cin, cout, grid, c.stream, 0u, nsq);
        timing_end(c, c.evShade, c.nShade); c.nShade++;
        if (r == rounds) break;
        timing_begin(c, c.evExtend, c.nExtend);
        const bool lds = c.blob.bytes <= kBlobLdsMax;
        const bool flat = c.blob.instCount <= kFlatInstances && !(c.debugFlags & PT_DEBUG_TRAVERSAL_PHASED);
        const uint32_t smem = (flat ? kFlatLdsFixed : kExtendLdsFixed) + (lds ? c.blob.bytes : 0u);
        if (c.debugFlags & PT_DEBUG_BRUTE_FORCE) k_extend_brute<<<grid, 256, 0, c.stream>>>(sv.accel, c.blob, ac, qout, segCap, cout, c.counters, c.sqShift);
        else if (c.debugFlags & PT_DEBUG_TRAVERSAL_V1) {
            if (stats) k_extend<true><<<grid, 256, 0, c.stream>>>(c.blob, ac, qout, segCap, cout, c.counters, c.sqShift);
            else k_extend<false><<<grid, 256, 0, c.stream>>>(c.blob, ac, qout, segCap, cout, c.counters, c.sqShift);
        } else {
            const bool wt = aux != nullptr;                            // denoiser modes need CommittedRayT
            #define PT_EXT2(S, L, W, F) k_extend2<S, L, W, F><<<grid, 256, smem, c.stream>>>(c.blob, ac, qout, segCap, cout, c.counters, c.sqShift)
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

// the whole frame on the context's stream: preamble, then the chains one after the other (or the validation rounds)
static hipError_t enqueue_frame(Context& c, const SceneView& sv, const FrameView& fv, const PtTextures& tx, uint32_t rounds, uint32_t segCap, uint32_t grid, uint32_t chains)
{
    hipError_t e = enqueue_preamble(c, sv, fv, tx, segCap, grid);
    if (e != hipSuccess) return e;
    if (!frame_form(c).first) return enqueue_validation_rounds(c, sv, fv, tx, rounds, segCap, grid);
    for (uint32_t g = 0; g < chains && e == hipSuccess; g++) e = enqueue_chain(c, sv, fv, tx, rounds, segCap, grid, g, chains, c.stream);
    return e;
}

template <typename T> static void key_add(std::string& k, const T& v) { k.append((const char*)&v, sizeof v); }

hipError_t launch_raytrace(Context& c, const SceneView& sv, const FrameView& fv, const PtTextures& tx)
{
    const PtGraphicsSettings& gs = c.settings;
    const uint32_t npix = fv.width * fv.localRows;
    c.lastIterations = 0;
    if (c.normalsShared && c.blasTableDev && c.blob.triCount > c.shadeRecCap) {       // the frame's normal records: 20 B per triangle packet (grow-only)
        hipError_t ea = hipStreamSynchronize(c.stream);
        if (ea != hipSuccess) return ea;
        if (c.shadeRecA) hipFree(c.shadeRecA);
        if (c.shadeRecB) hipFree(c.shadeRecB);
        c.shadeRecA = nullptr; c.shadeRecB = nullptr; c.shadeRecCap = 0;
        if ((ea = hipMalloc((void**)&c.shadeRecA, sizeof(uint4) * (size_t)c.blob.triCount)) != hipSuccess) return ea;
        if ((ea = hipMalloc((void**)&c.shadeRecB, sizeof(uint32_t) * (size_t)c.blob.triCount)) != hipSuccess) return ea;
        c.shadeRecCap = c.blob.triCount;
    }
    if (npix == 0 || gs.SamplesPerPixel == 0) return hipSuccess;
    // a round = one k_shade + one k_extend. Per sample a path spends one round as "fresh" (bounce 0, no ray) and at
    // most Bounces rounds as "traced": spp * (Bounces + 1) rounds empty every queue.
    const uint32_t rounds = gs.SamplesPerPixel * (gs.Bounces + 1u);
    const uint32_t tiles = (npix + 255u) / 256u;
    c.sqShift = c.blob.bytes <= kBlobLdsMax ? kSubQueueShiftFused : kSubQueueShiftStream;          // (pt_internal.hpp: who likes how many sub-queues)
    const uint32_t nsq = 1u << c.sqShift, cstride = 3u * nsq;
    const uint32_t segCap = (tiles + nsq - 1) / nsq * 256u;                   // entries per sub-queue segment
    hipError_t e = ensure_queues(c, segCap * nsq, (rounds + 2) * cstride);
    if (e != hipSuccess) return e;
    if (!c.frameConstants && (e = hipMalloc((void**)&c.frameConstants, sizeof(FrameConstants))) != hipSuccess) return e;
    if (gs.Denoiser != PT_DENOISER_NONE && npix > c.pixelAuxCapacity) {
        if (c.pixelAux) hipFree(c.pixelAux);
        c.pixelAux = nullptr; c.pixelAuxCapacity = 0;
        if ((e = hipMalloc((void**)&c.pixelAux, (size_t)npix * sizeof(float2))) != hipSuccess) return e;
        c.pixelAuxCapacity = npix;
    }
    FrameConstants fc; fc.cam = c.camera; fc.sd = c.sceneData; fc.gs = c.settings;
    k_set_constants<<<1, 256, 0, c.stream>>>(fc, c.frameConstants, c.queueCounts, (rounds + 2u) * cstride);
    // persistent grid, but never more blocks than the queue has tiles: surplus blocks only cost dispatch slots and LDS that
    // a concurrent frame's kernels (other streams) could use -- this matters for small shards (1/8 of a 1080p frame = 1013 tiles)
    const uint32_t grid = std::min(persistent_grid(c), (tiles + nsq - 1) / nsq * nsq);
    c.lastIterations = rounds + 1;

    // everything the launch sequence depends on: the key of the per-round argument blocks and of the captured graph
    std::string key;
    key_add(key, sv); key_add(key, fv); key_add(key, tx); key_add(key, rounds); key_add(key, segCap); key_add(key, grid);
    key_add(key, c.queue[0]); key_add(key, c.queue[1]); key_add(key, c.queueCounts); key_add(key, c.blob); key_add(key, c.heapHasTextures);
    key_add(key, c.frameConstants); key_add(key, c.primaryRecords); key_add(key, c.stream); key_add(key, c.pixelAux); key_add(key, gs.Denoiser); key_add(key, c.debugFlags);
    key_add(key, c.framesInFlight); key_add(key, c.sqShift); key_add(key, c.chains);
    key_add(key, c.shadeRecA); key_add(key, c.blasTableDev); key_add(key, c.blasTableCount); key_add(key, c.blasTableMaxTris); key_add(key, normal_records_usable(c));
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
            a.fc = c.frameConstants; a.aux = aux; a.countIn = &c.queueCounts[r * cstride]; a.countOut = &c.queueCounts[(r + 1) * cstride]; a.sqShift = c.sqShift;
            a.counters = c.counters; a.segCap = segCap; a.primary = c.primaryRecords; a.objectsInLds = round_objects_in_lds(c, sv);
            if (normal_records_usable(c)) { a.recA = c.shadeRecA; a.recB = c.shadeRecB; a.recordsInLds = round_records_in_lds(c, sv); }
        }
        if ((e = hipMemcpyAsync(c.roundArgs, host.data(), sizeof(RoundArgs) * (rounds + 1), hipMemcpyHostToDevice, c.stream)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(c.stream)) != hipSuccess) return e;    // once per change of the scene / frame geometry, never per frame
        c.roundArgsKey = key;
    }

    // hipGraph replay: launch-bound frames (small shards, tail rounds) cost ~75 launches; a replay is one submission.
    const bool graphable = c.stream != nullptr && !c.timing && !c.disableGraphs &&
                           (c.debugFlags & ~(PT_DEBUG_UNFUSED_ROUNDS | PT_DEBUG_TRAVERSAL_PHASED | PT_DEBUG_LOCKSTEP | PT_DEBUG_GATHER_LOCAL_ONLY | PT_DEBUG_GATHER_SELF_EXCHANGE)) == 0;   // counters / validation variants launch directly
    auto capture = [&](hipStream_t s, hipGraphExec_t& exec, auto&& body) -> hipError_t {          // one linear graph from what `body` enqueues on s
        if (exec) { hipStreamSynchronize(c.stream); hipGraphExecDestroy(exec); exec = nullptr; }   // its last replay may still be running (once per change of scene / frame geometry)
        hipGraph_t graph = nullptr;
        hipError_t ce = hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
        if (ce == hipSuccess) {
            const hipError_t e2 = body();
            ce = hipStreamEndCapture(s, &graph);
            if (ce == hipSuccess) ce = e2;
        }
        if (ce == hipSuccess) ce = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (graph) hipGraphDestroy(graph);
        if (ce != hipSuccess) { exec = nullptr; (void)hipGetLastError(); }
        return ce;
    };
    const uint32_t chains = frame_form(c).first ? frame_chains(c, graphable) : 1u;
    if (graphable && chains > 1) {
        // every chain a linear graph on a stream of its own; the preamble is launched directly (two kernels)
        if (!c.chainFork && (e = hipEventCreateWithFlags(&c.chainFork, hipEventDisableTiming)) != hipSuccess) return e;
        for (uint32_t g = 1; g < chains; g++) {
            if (!c.chainStream[g - 1] && (e = hipStreamCreateWithFlags(&c.chainStream[g - 1], hipStreamNonBlocking)) != hipSuccess) return e;
            if (!c.chainJoin[g - 1] && (e = hipEventCreateWithFlags(&c.chainJoin[g - 1], hipEventDisableTiming)) != hipSuccess) return e;
        }
        if (key != c.chainGraphKey) {
            c.chainGraphKey.clear();
            for (uint32_t g = 0; g < chains; g++) {
                hipStream_t s = g == 0 ? c.stream : c.chainStream[g - 1];
                if (capture(s, c.chainGraph[g], [&] { return enqueue_chain(c, sv, fv, tx, rounds, segCap, grid, g, chains, s); }) != hipSuccess) { c.disableGraphs = true; break; }
            }
            if (!c.disableGraphs) c.chainGraphKey = key;
        }
        if (!c.disableGraphs) {
            if ((e = enqueue_preamble(c, sv, fv, tx, segCap, grid)) != hipSuccess) return e;
            if ((e = hipEventRecord(c.chainFork, c.stream)) != hipSuccess) return e;
            for (uint32_t g = 1; g < chains; g++) {
                hipStream_t s = c.chainStream[g - 1];
                if ((e = hipStreamWaitEvent(s, c.chainFork, 0)) != hipSuccess) return e;
                if ((e = hipGraphLaunch(c.chainGraph[g], s)) != hipSuccess) return e;
                if ((e = hipEventRecord(c.chainJoin[g - 1], s)) != hipSuccess) return e;
            }
            if ((e = hipGraphLaunch(c.chainGraph[0], c.stream)) != hipSuccess) return e;
            for (uint32_t g = 1; g < chains; g++) if ((e = hipStreamWaitEvent(c.stream, c.chainJoin[g - 1], 0)) != hipSuccess) return e;
            return hipSuccess;
        }
    } else if (graphable) {
        if (key != c.graphKey || !c.graphExec) {
            c.graphKey.clear();
            if (capture(c.stream, c.graphExec, [&] { return enqueue_frame(c, sv, fv, tx, rounds, segCap, grid, 1u); }) != hipSuccess) c.disableGraphs = true;
            else c.graphKey = key;
        }
        if (c.graphExec) return hipGraphLaunch(c.graphExec, c.stream);
    }
    return enqueue_frame(c, sv, fv, tx, rounds, segCap, grid, frame_form(c).first ? frame_chains(c, false) : 1u);
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
