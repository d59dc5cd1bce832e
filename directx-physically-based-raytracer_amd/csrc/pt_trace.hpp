// pt_trace.hpp -- acceleration-structure formats, traversal stack, wide-node box test and the watertight
// ray/triangle test (device code, gfx950).
//
// Replaces the DXR hardware traversal behind RayQuery::TraceRayInline/Proceed that the reference
// calls from TraceRay (Shaders/RaytracingHelpers.hlsli:7-55) with ray flags NONE, instance mask
// ~0, RAY_FLAG_SKIP_PROCEDURAL_PRIMITIVES: closest hit, no face culling. No RT hardware and no
// HIP-RT: boxes and triangles are plain fp32 VALU work.
//
// Node format: compressed 8-wide BVH node, 80 B = 5 x 16 B (the layout of Ylitie, Karras, Laine, "Efficient
// Incoherent Ray Traversal on GPUs Through Compressed Wide BVHs", HPG 2017 -- one fetch per visit, child boxes quantised
// to 8 bits against the node's fp32 origin and a power-of-two scale per axis, rounded outward):
//   u[0]  origin.x origin.y origin.z | ex, ey, ez, imask     e*: biased fp32 exponent of the axis scale, imask: bit s = slot s
//                                                              holds an internal node
//   u[1]  childBase | triBase | meta[0..3] | meta[4..7]       internal child of slot s = node childBase + popcount(imask below s)
//   u[2]  qlo.x[0..7] | qlo.y[0..7]                            meta: 0 = empty slot; internal: 001sssss with sssss = 24 + s;
//   u[3]  qlo.z[0..7] | qhi.x[0..7]                            leaf: the top 3 bits are the unary triangle count (001, 011, 111),
//   u[4]  qhi.y[0..7] | qhi.z[0..7]                            the low 5 bits the first triangle relative to triBase (< 24)
// Children sit in the slot that the octant-order heuristic of the paper gives them, so "slot xor ray octant" is the
// front-to-back visiting order and no distance sort is needed. The same node serves the TLAS: its "triangles" are then
// entries of the instance order list.
// Triangle packet (48 B): v0.xyz geom | v1.xyz prim | v2.xyz flags  -- object-space positions; the packets of one node's
// leaf children are contiguous from triBase.
#pragma once
#include "pt_texture.hpp"

namespace pt {

// The lanes of the wave for which the predicate holds. HIP's __ballot goes through an integer compare of a materialised 0/1
// (v_cndmask + v_cmp per call); the builtin takes the condition mask as it is.
__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }


struct alignas(16) WideNode {
    float origin[3]; uint32_t expImask;
    uint32_t childBase, triBase, meta[2];
    uint32_t qlox[2], qloy[2];
    uint32_t qloz[2], qhix[2];
    uint32_t qhiy[2], qhiz[2];
};
struct alignas(16) TriPacket { float4 a, b, c; };        // a.w = geom (bits), b.w = prim (bits), c.w = flags
static_assert(sizeof(WideNode) == 80 && sizeof(TriPacket) == 48, "layout");

struct alignas(16) InstanceRecord {                       // 128 B, shading-side view of one TLAS instance (API order)
    float worldToObject[12];
    float objectToWorld[12];
    const WideNode* nodes;                                // BLAS node pool (root = 0)
    const TriPacket* tris;
    uint32_t instanceID;                                  // D3D12 InstanceID = FirstGeometryIndex
    uint32_t mask;
    uint32_t triCount;                                    // triangles in the BLAS (debug brute-force traversal)
    uint32_t blasSlot;                                    // row of the bottom-level table of the build that made this record
};
static_assert(sizeof(InstanceRecord) == 128, "layout");

struct AccelView {
    const InstanceRecord* instances;
    uint32_t instanceCount;
};

struct Hit {
    float t, u, v;
    uint32_t inst, geom, prim;                            // inst == ~0u : miss
    uint32_t slot;                                        // index of the triangle packet inside its BLAS
};

struct TraceStats { uint32_t nodes, tris, overflow; };

// Traversal stack of node groups (8 B each: child base | hits << 24 | imask). One entry at most per level of the wide
// tree, so kStackSize bounds TLAS depth + BLAS depth + the three entries of an instance transition; the builders refuse
// a structure that does not fit (pt_api.hip), and a push that would still overflow is counted, never silent.
constexpr int kStackSize = 64;
constexpr uint32_t kMaxLeafTris = 3;                      // the unary count of a leaf slot has 3 bits
// Triangles per BLAS leaf, a function of the BLAS size so that builder and traversal agree without storing it.
__host__ __device__ inline uint32_t blas_leaf_tris(uint32_t triCount) { return triCount <= 32u ? 2u : 1u; }
__host__ __device__ inline bool blas_single_leaf(uint32_t triCount) { return triCount <= blas_leaf_tris(triCount); }

// A traversal state is two groups: G = (child base, hits << 24 | imask) of the node whose children are being visited, and
// T = (triangle base, 24-bit mask) of triangles still to test. A tree of one leaf has no node worth a visit: it starts as
// a triangle group over packets [0, triCount); any other tree starts at node 0 (hit bit 31 with imask 0 selects child base + 0).
PT_DEV uint2 root_node_group(bool singleLeaf) { return singleLeaf ? make_uint2(0u, 0u) : make_uint2(0u, 0x80000000u); }
PT_DEV uint2 root_tri_group(bool singleLeaf, uint32_t count) { return singleLeaf ? make_uint2(0u, (1u << count) - 1u) : make_uint2(0u, 0u); }

#define PT_LDS_AS __attribute__((address_space(3)))
typedef uint32_t u2v __attribute__((ext_vector_type(2)));

// The first LDS_DEPTH entries of a lane live in LDS (entry d of thread t at lds[d * 256 + t]: ds_write_b64 / ds_read_b64,
// conflict-free), deeper ones in a private array. The LDS part is addressed through an LDS pointer on purpose: through a generic
// pointer every push and pop is a flat_* access, which makes the wave wait for ALL its outstanding memory operations (flat
// accesses complete out of order: vmcnt(0) and lgkmcnt(0)) -- including the node fetch it was meant to overlap with.
template <int LDS_DEPTH>
struct GroupStack {
    PT_LDS_AS u2v* lds; uint2* spill; int sp; uint32_t overflow;
    PT_DEV void init(PT_LDS_AS void* ldsBase, uint2* spillBase) { lds = (PT_LDS_AS u2v*)ldsBase + threadIdx.x; spill = spillBase; sp = 0; overflow = 0; }
    PT_DEV void push(uint2 v)
    {
        if (sp < LDS_DEPTH) lds[sp * 256] = (u2v){ v.x, v.y };
        else if (sp < kStackSize) spill[sp - LDS_DEPTH] = v;
        else { overflow++; return; }                       // counted in PtCounters.StackOverflows (the builders make this unreachable)
        sp++;
    }
    PT_DEV uint2 pop()
    {
        sp--;
        if (sp < LDS_DEPTH) { const u2v e = lds[sp * 256]; return make_uint2(e.x, e.y); }
        return spill[sp - LDS_DEPTH];
    }
};

// ---------------------------------------------------------------------------------------------
// ray / wide-node test
// ---------------------------------------------------------------------------------------------
PT_DEV float safe_inv1(float d)
{
    // A zero (or denormal-small) direction component would turn the slab planes into inf - inf = NaN,
    // and max(-inf, NaN) = -inf then culls a box the ray is inside of. Clamp |d| for the BOX test only
    // (the triangle test uses the true direction): planes become +-huge finite values with the right signs.
    const float a = fabsf(d) < 1e-20f ? copysignf(1e-20f, d) : d;
    return __builtin_amdgcn_rcpf(a);      // 1 ulp v_rcp_f32: the box test is conservative (padded boxes), not part of the spec
}
PT_DEV v3 safe_inv(v3 d) { return V3(safe_inv1(d.x), safe_inv1(d.y), safe_inv1(d.z)); }

// what the box test needs from a ray in the current space
struct BoxRay { v3 o, idir; uint32_t octinv4; };
PT_DEV BoxRay box_ray(v3 o, v3 d)
{
    BoxRay r; r.o = o; r.idir = safe_inv(d);
    // the paper's inverted octant, replicated into 4 bytes: slot s of a ray with this octant is visited in the order s ^ octinv, highest first
    r.octinv4 = (r.idir.x < 0.0f ? 0u : 0x04040404u) | (r.idir.y < 0.0f ? 0u : 0x02020202u) | (r.idir.z < 0.0f ? 0u : 0x01010101u);
    return r;
}

typedef float f4v __attribute__((ext_vector_type(4)));
// Four registers with no defined content and no instruction spent on them: for values that only the lanes which load them go on
// to read (a plain uninitialised variable is filled with zeros by the compiler where control flow merges).
PT_DEV f4v undefined_f4v()
{
    float a, b, c, d;
    asm volatile("" : "=v"(a)); asm volatile("" : "=v"(b)); asm volatile("" : "=v"(c)); asm volatile("" : "=v"(d));   // volatile: identical empty statements are not merged into one value (which would be copied around)
    return (f4v){ a, b, c, d };
}
PT_DEV float ubyte_f(uint32_t w, int j) { return (float)((w >> (8 * j)) & 0xFFu); }          // v_cvt_f32_ubyte{0..3}

// Slab test of a ray against the eight quantised child boxes of a node. Returns the paper's hit mask: bit 24 + (s ^ octinv)
// for a hit internal child in slot s, bits [first, first + count) of the low 24 for the triangles of a hit leaf slot.
// Conservative by construction: boxes are padded before quantisation and rounded outward by it, NaN slabs are ignored
// (v_min/v_max drop NaN operands), and every slab plane t = q * a + b is moved outward by a bound on its own rounding error:
// a and b are products with the reciprocal direction of terms as large as the NODE (not the child), so the error of t scales
// with |b| + 255 |a|, whatever the size of the child box -- folded into b, it costs nothing per child.
PT_DEV uint32_t wide_node_hits(f4v n0, f4v n1, f4v n2, f4v n3, f4v n4, const BoxRay& r, float tmin, float tmax)
{
    const uint32_t em = __float_as_uint(n0.w);
    const float ax = __uint_as_float((em & 0xFFu) << 23) * r.idir.x;
    const float ay = __uint_as_float(((em >> 8) & 0xFFu) << 23) * r.idir.y;
    const float az = __uint_as_float(((em >> 16) & 0xFFu) << 23) * r.idir.z;
    const float bx = (n0.x - r.o.x) * r.idir.x, by = (n0.y - r.o.y) * r.idir.y, bz = (n0.z - r.o.z) * r.idir.z;
    constexpr float kUlps = 4.76837158e-7f;                                      // 2^-21: four ulps of the largest term
    const float ex = __builtin_fmaf(fabsf(ax), 255.0f, fabsf(bx)) * kUlps, ey = __builtin_fmaf(fabsf(ay), 255.0f, fabsf(by)) * kUlps,
                ez = __builtin_fmaf(fabsf(az), 255.0f, fabsf(bz)) * kUlps;
    const float bnx = bx - ex, bfx = bx + ex, bny = by - ey, bfy = by + ey, bnz = bz - ez, bfz = bz + ez;
    const bool nx = r.idir.x < 0.0f, ny = r.idir.y < 0.0f, nz = r.idir.z < 0.0f;
    uint32_t hits = 0;
    #pragma unroll
    for (int h = 0; h < 2; h++) {
        const uint32_t meta4 = __float_as_uint(h ? n1.w : n1.z);
        const uint32_t inner4 = (meta4 & (meta4 << 1)) & 0x10101010u;           // both bits 3 and 4: low 5 bits >= 24
        const uint32_t innerMask4 = (inner4 >> 4) * 0xFFu;
        const uint32_t bitIndex4 = (meta4 ^ (r.octinv4 & innerMask4)) & 0x1F1F1F1Fu;
        const uint32_t childBits4 = (meta4 >> 5) & 0x07070707u;
        const uint32_t lox = __float_as_uint(h ? n2.y : n2.x), loy = __float_as_uint(h ? n2.w : n2.z), loz = __float_as_uint(h ? n3.y : n3.x);
        const uint32_t hix = __float_as_uint(h ? n3.w : n3.z), hiy = __float_as_uint(h ? n4.y : n4.x), hiz = __float_as_uint(h ? n4.w : n4.z);
        const uint32_t nearx = nx ? hix : lox, farx = nx ? lox : hix;
        const uint32_t neary = ny ? hiy : loy, fary = ny ? loy : hiy;
        const uint32_t nearz = nz ? hiz : loz, farz = nz ? loz : hiz;
        #pragma unroll
        for (int j = 0; j < 4; j++) {
            const float tnx = __builtin_fmaf(ubyte_f(nearx, j), ax, bnx), tfx = __builtin_fmaf(ubyte_f(farx, j), ax, bfx);
            const float tny = __builtin_fmaf(ubyte_f(neary, j), ay, bny), tfy = __builtin_fmaf(ubyte_f(fary, j), ay, bfy);
            const float tnz = __builtin_fmaf(ubyte_f(nearz, j), az, bnz), tfz = __builtin_fmaf(ubyte_f(farz, j), az, bfz);
            const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, tmin));
            const float tf = fminf(fminf(tfx, tfy), fminf(tfz, tmax));
            const uint32_t bits = (childBits4 >> (8 * j)) & 0xFFu, idx = (bitIndex4 >> (8 * j)) & 0xFFu;
            if (tn <= tf) hits |= bits << idx;
        }
    }
    return hits;
}

// Per-ray constants of the watertight test. kz = dominant axis of the direction (c2: z, else c1: y, else x),
// kx = kz + 1, ky = kz + 2 (mod 3). The paper additionally swaps kx and ky when d[kz] < 0 to keep the winding; that swap
// negates U, V, W, det and T exactly (IEEE negation commutes with every operation used) and therefore changes neither the
// hit decision nor t = T/det, u = V/det, v = W/det: it is omitted here, the results stay bit-identical to the oracle's,
// which keeps the swap.
struct RaySetup { bool c1, c2; float Sx, Sy, Sz; };
PT_DEV float sel_kz(v3 v, const RaySetup& r) { return r.c2 ? v.z : (r.c1 ? v.y : v.x); }
PT_DEV float sel_kx(v3 v, const RaySetup& r) { return r.c2 ? v.x : (r.c1 ? v.z : v.y); }
PT_DEV float sel_ky(v3 v, const RaySetup& r) { return r.c2 ? v.y : (r.c1 ? v.x : v.z); }

// Woop, Benthin, Wald: "Watertight Ray/Triangle Intersection", JCGT 2013 -- per-ray part.
PT_DEV RaySetup ray_setup(v3 d)
{
    RaySetup r;
    const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    r.c1 = ay > ax;
    r.c2 = az > (r.c1 ? ay : ax);
    r.Sz = 1.0f / sel_kz(d, r);          // one IEEE division; Sx, Sy by multiplication (spec, same in the oracle)
    r.Sx = sel_kx(d, r) * r.Sz;
    r.Sy = sel_ky(d, r) * r.Sz;
    return r;
}

// The paper's fp64 fallback, kept out of line: taken only when an edge function is exactly 0 (a ray through an
// edge or vertex). Inlined, hipcc if-converts it and every triangle test pays 18 fp64 instructions.
__device__ __attribute__((noinline)) void tri_edge_fallback(float Ax, float Ay, float Bx, float By, float Cx, float Cy, float& U, float& V, float& W)
{
    double CxBy = (double)Cx * (double)By, CyBx = (double)Cy * (double)Bx;
    U = (float)(CxBy - CyBx);
    double AxCy = (double)Ax * (double)Cy, AyCx = (double)Ay * (double)Cx;
    V = (float)(AxCy - AyCx);
    double BxAy = (double)Bx * (double)Ay, ByAx = (double)By * (double)Ax;
    W = (float)(BxAy - ByAx);
}

// Per-triangle part; fp64 recomputation of the edge functions when one is exactly 0 (paper's
// fallback) keeps shared edges watertight. Returns t,u,v (u weights v1, v weights v2: DXR barycentrics).
typedef float f2p __attribute__((ext_vector_type(2)));       // maps to v_pk_{add,mul}_f32 on gfx950: same IEEE results, half the issue slots
PT_DEV bool tri_test(const RaySetup& r, v3 o, v3 v0, v3 v1, v3 v2, float& t, float& u, float& v)
{
    const f2p oxy = { o.x, o.y };
    const f2p Axy = (f2p){ v0.x, v0.y } - oxy, Bxy = (f2p){ v1.x, v1.y } - oxy, Cxy = (f2p){ v2.x, v2.y } - oxy;
    const v3 A = V3(Axy.x, Axy.y, v0.z - o.z), B = V3(Bxy.x, Bxy.y, v1.z - o.z), C = V3(Cxy.x, Cxy.y, v2.z - o.z);
    float Akz = sel_kz(A, r), Bkz = sel_kz(B, r), Ckz = sel_kz(C, r);
    const f2p S = { r.Sx, r.Sy };
    const f2p a = (f2p){ sel_kx(A, r), sel_ky(A, r) } - S * Akz;       // (Ax, Ay)
    const f2p b = (f2p){ sel_kx(B, r), sel_ky(B, r) } - S * Bkz;
    const f2p c = (f2p){ sel_kx(C, r), sel_ky(C, r) } - S * Ckz;
    const f2p pu = c * b.yx, pv = a * c.yx, pw = b * a.yx;               // (Cx*By, Cy*Bx) ...
    float U = pu.x - pu.y;
    float V = pv.x - pv.y;
    float W = pw.x - pw.y;
    if (__builtin_expect(U == 0.0f || V == 0.0f || W == 0.0f, 0)) tri_edge_fallback(a.x, a.y, b.x, b.y, c.x, c.y, U, V, W);
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return false;
    float det = U + V + W;
    if (det == 0.0f) return false;
    float Az = r.Sz * Akz, Bz = r.Sz * Bkz, Cz = r.Sz * Ckz;
    float T = U * Az + V * Bz + W * Cz;
    float rcp = 1.0f / det;
    t = T * rcp; u = V * rcp; v = W * rcp;
    return true;
}

// Closest hit with ties on t broken by (instance, geometry, primitive) so the result does not
// depend on traversal order. Triangle hits are accepted for t in (tmin, tmax) exclusive.
PT_DEV bool is_better(const Hit& h, float tmin, float t, uint32_t inst, uint32_t geom, uint32_t prim)
{
    if (!(t > tmin)) return false;
    if (t < h.t) return true;
    if (t == h.t && h.inst != ~0u)
        return inst < h.inst || (inst == h.inst && (geom < h.geom || (geom == h.geom && prim < h.prim)));
    return false;
}

// What the non-opaque candidate callback needs (TraceRay, Shaders/RaytracingHelpers.hlsli:19-44): the object
// table, the descriptor heap and the instance table (InstanceID). nullptr members = no alpha-tested geometry.
struct AlphaContext {
    const PtObjectData* objects;
    const HeapEntry* heap;
    const float* srgbLut;
    const struct InstanceRecord* instances;
    const HeapEntry* shadeTex = nullptr;      // the objects' resolved texture slots (pt_texture.hpp TextureSlots), or null
};

__device__ __attribute__((noinline)) bool candidate_is_opaque(const AlphaContext ac, uint32_t inst, uint32_t geom, uint32_t prim, float u, float v)
{
    const PtObjectData* od = &ac.objects[ac.instances[inst].instanceID + geom];
    TexCoords tc;
    get_texture_coordinates(od, ac.heap, prim, u, v, tc);
#ifdef PT_AB_OLD_TEXTURE_PATH
    return is_opaque(od, ac.heap, ac.srgbLut, tc);
#else
    return is_opaque(od, ac.heap, ac.srgbLut, tc, ac.shadeTex ? ac.shadeTex + (size_t)(ac.instances[inst].instanceID + geom) * kTextureSlots : nullptr);
#endif
}

// commit() for a candidate of a geometry without D3D12_RAYTRACING_GEOMETRY_FLAG_OPAQUE: the alpha test runs only
// for candidates that would otherwise be committed (DXR reports candidates inside the current ray interval).
PT_DEV void commit_candidate(const AlphaContext& ac, uint32_t flags, Hit& h, float tmin, float t, float u, float v,
                             uint32_t inst, uint32_t geom, uint32_t prim, uint32_t slot)
{
    if (!is_better(h, tmin, t, inst, geom, prim)) return;
    if (!(flags & PT_GEOMETRY_FLAG_OPAQUE) && !candidate_is_opaque(ac, inst, geom, prim, u, v)) return;
    h.t = t; h.u = u; h.v = v; h.inst = inst; h.geom = geom; h.prim = prim; h.slot = slot;
}

PT_DEV void transform_ray(const float* W, v3 o, v3 d, v3& ro, v3& rd)
{
    ro = V3(sop3t(W[0], o.x, W[1], o.y, W[2], o.z, W[3]), sop3t(W[4], o.x, W[5], o.y, W[6], o.z, W[7]), sop3t(W[8], o.x, W[9], o.y, W[10], o.z, W[11]));
    rd = V3(sop3(W[0], d.x, W[1], d.y, W[2], d.z), sop3(W[4], d.x, W[5], d.y, W[6], d.z), sop3(W[8], d.x, W[9], d.y, W[10], d.z));
}

// Debug / validation traversal (PT_DEBUG_BRUTE_FORCE): every triangle of every instance, no BVH.
PT_DEV Hit trace_brute_force(const AccelView& av, const AlphaContext& ac, v3 o, v3 d, float tmin, float tmax)
{
    Hit h; h.t = tmax; h.u = 0.0f; h.v = 0.0f; h.inst = ~0u; h.geom = 0; h.prim = 0; h.slot = 0;
    for (uint32_t x = 0; x < av.instanceCount; x++) {
        const InstanceRecord* ir = &av.instances[x];
        if (!(ir->mask & 0xFFu)) continue;
        v3 ro, rd;
        transform_ray(ir->worldToObject, o, d, ro, rd);
        const RaySetup rs = ray_setup(rd);
        for (uint32_t i = 0; i < ir->triCount; i++) {
            const TriPacket tp = ir->tris[i];
            float t, u, v;
            if (tri_test(rs, ro, V3(tp.a.x, tp.a.y, tp.a.z), V3(tp.b.x, tp.b.y, tp.b.z), V3(tp.c.x, tp.c.y, tp.c.z), t, u, v))
                commit_candidate(ac, __float_as_uint(tp.c.w), h, tmin, t, u, v, x, __float_as_uint(tp.a.w), __float_as_uint(tp.b.w), i);
        }
    }
    if (h.inst != ~0u && !(h.t < tmax)) h.inst = ~0u;
    return h;
}

} // namespace pt
