// pt_trace.hpp -- two-level BVH traversal + watertight ray/triangle test (device code, gfx950).
//
// Replaces the DXR hardware traversal behind RayQuery::TraceRayInline/Proceed that the reference
// calls from TraceRay (Shaders/RaytracingHelpers.hlsli:7-55) with ray flags NONE, instance mask
// ~0, RAY_FLAG_SKIP_PROCEDURAL_PRIMITIVES: closest hit, no face culling. No RT hardware and no
// HIP-RT: boxes and triangles are plain fp32 VALU work.
//
// Node format (64 B, two child boxes per node, children sorted at traversal time):
//   f4[0] = c0.lo.x c0.hi.x c0.lo.y c0.hi.y     f4[1] = c1.lo.x c1.hi.x c1.lo.y c1.hi.y
//   f4[2] = c0.lo.z c0.hi.z c1.lo.z c1.hi.z     i4[3] = child0 child1 - -
//   child >= 0: node index; child < 0: leaf ~x. BLAS leaf x = first_tri << 3 | (count-1);
//   TLAS leaf x = instance slot. An absent child has an inverted (+inf,-inf) box.
// Triangle packet (48 B): v0.xyz geom | v1.xyz prim | v2.xyz flags  -- object-space positions in
// Morton order so a leaf's triangles are contiguous.
#pragma once
#include "pt_texture.hpp"

namespace pt {

struct alignas(16) BvhNode { float4 c0xy, c1xy, cz; int4 child; };
struct alignas(16) TriPacket { float4 a, b, c; };        // a.w = geom (bits), b.w = prim (bits), c.w = flags
static_assert(sizeof(BvhNode) == 64 && sizeof(TriPacket) == 48, "layout");

struct alignas(16) InstanceRecord {                       // 128 B, traversal + shading view of one TLAS instance
    float worldToObject[12];
    float objectToWorld[12];
    const BvhNode* nodes;                                 // BLAS node pool (root = 0)
    const TriPacket* tris;
    uint32_t instanceID;                                  // D3D12 InstanceID = FirstGeometryIndex
    uint32_t mask;
    uint32_t triCount;                                    // triangles in the BLAS (debug brute-force traversal)
    uint32_t _pad;
};
static_assert(sizeof(InstanceRecord) == 128, "layout");

struct AccelView {
    const BvhNode* tlasNodes;
    const InstanceRecord* instances;
    uint32_t instanceCount;
};

struct Hit {
    float t, u, v;
    uint32_t inst, geom, prim;                            // inst == ~0u : miss
    uint32_t slot;                                        // index of the triangle packet inside its BLAS
};

struct TraceStats { uint32_t nodes, tris; };

constexpr int kEntryDone = 0x7FFFFFFF;
constexpr int kEntryRestore = 0x7FFFFFFE;
constexpr int kStackSize = 96;
// Triangles per BLAS leaf, a function of the BLAS size so that builder and traversal agree without storing it. Measured on
// MI355X (Mrays/s with leaves of 4 / 3 / 2 / 1 triangles): Cornell C2 8.16 / 8.04 / 8.84 / 7.43 G, 250k-triangle C3
// 486 / 521 / 576 / 790 M, 10k-instance C5 1071 / 1025 / 1199 / 1339 M. A software triangle test costs about as much as a
// node test and Morton-ordered neighbours make loose 4-triangle boxes, so big meshes want one triangle per leaf; the tiny
// BLASes of a Cornell-type scene want their quads as ONE leaf (entered directly, no node) and faces kept in pairs.
#ifndef PT_LEAF_RULE
#define PT_LEAF_RULE 0
#endif
__host__ __device__ inline uint32_t blas_leaf_tris(uint32_t triCount)
{
#if PT_LEAF_RULE == 0
    return triCount <= 32u ? 2u : 1u;
#elif PT_LEAF_RULE == 1
    return triCount <= 2u ? 2u : 1u;
#else
    return 2u;
#endif
}
__host__ __device__ inline bool blas_single_leaf(uint32_t triCount) { return triCount <= blas_leaf_tris(triCount); }

// A tree of one leaf has no internal node worth a visit (its root node would list the same leaf twice): traversal
// starts at the leaf itself. BLAS leaf 0 covers packets [0, triCount); the only TLAS leaf is instance 0.
PT_DEV int blas_root_entry(uint32_t triCount) { return blas_single_leaf(triCount) ? ~(int)(triCount - 1u) : 0; }   // triCount >= 1
PT_DEV int tlas_root_entry(uint32_t instCount) { return instCount == 1u ? ~0 : 0; }                            // instCount >= 1

// Per-ray constants of the watertight test. kz = dominant axis of the direction (c2: z, else c1: y, else x),
// kx = kz + 1, ky = kz + 2 (mod 3). The paper additionally swaps kx and ky when d[kz] < 0 to keep the winding; that swap
// negates U, V, W, det and T exactly (IEEE negation commutes with every operation used) and therefore changes neither the
// hit decision nor t = T/det, u = V/det, v = W/det: it is omitted here, the results stay bit-identical to the oracle's,
// which keeps the swap.
struct RaySetup { bool c1, c2; float Sx, Sy, Sz; };
PT_DEV float sel_kz(v3 v, const RaySetup& r) { return r.c2 ? v.z : (r.c1 ? v.y : v.x); }
PT_DEV float sel_kx(v3 v, const RaySetup& r) { return r.c2 ? v.x : (r.c1 ? v.z : v.y); }
PT_DEV float sel_ky(v3 v, const RaySetup& r) { return r.c2 ? v.y : (r.c1 ? v.x : v.z); }

// Traversal stack: the first LDS_DEPTH entries of every lane live in LDS (entry d of thread t at
// lds[d * blockDim + t]: consecutive lanes hit consecutive banks), deeper entries spill to a private array.
template <int LDS_DEPTH>
struct TraversalStack {
    int* lds; int* spill; int sp;
    PT_DEV void init(int* ldsBase, int* spillBase) { lds = ldsBase + threadIdx.x; spill = spillBase; sp = 0; }
    PT_DEV void push(int v)
    {
        if (sp < LDS_DEPTH) lds[sp * 256] = v;
        else if (sp < kStackSize) spill[sp - LDS_DEPTH] = v;
        else return;
        sp++;
    }
    PT_DEV int pop()
    {
        sp--;
        return sp < LDS_DEPTH ? lds[sp * 256] : spill[sp - LDS_DEPTH];
    }
};

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
PT_DEV void commit(Hit& h, float tmin, float t, float u, float v, uint32_t inst, uint32_t geom, uint32_t prim, uint32_t slot)
{
    if (is_better(h, tmin, t, inst, geom, prim)) { h.t = t; h.u = u; h.v = v; h.inst = inst; h.geom = geom; h.prim = prim; h.slot = slot; }
}

// What the non-opaque candidate callback needs (TraceRay, Shaders/RaytracingHelpers.hlsli:19-44): the object
// table, the descriptor heap and the instance table (InstanceID). nullptr members = no alpha-tested geometry.
struct AlphaContext {
    const PtObjectData* objects;
    const HeapEntry* heap;
    const float* srgbLut;
    const struct InstanceRecord* instances;
};

__device__ __attribute__((noinline)) bool candidate_is_opaque(const AlphaContext ac, uint32_t inst, uint32_t geom, uint32_t prim, float u, float v);

// commit() for a candidate of a geometry without D3D12_RAYTRACING_GEOMETRY_FLAG_OPAQUE: the alpha test runs only
// for candidates that would otherwise be committed (DXR reports candidates inside the current ray interval).
PT_DEV void commit_candidate(const AlphaContext& ac, uint32_t flags, Hit& h, float tmin, float t, float u, float v,
                             uint32_t inst, uint32_t geom, uint32_t prim, uint32_t slot)
{
    if (!is_better(h, tmin, t, inst, geom, prim)) return;
    if (!(flags & PT_GEOMETRY_FLAG_OPAQUE) && !candidate_is_opaque(ac, inst, geom, prim, u, v)) return;
    h.t = t; h.u = u; h.v = v; h.inst = inst; h.geom = geom; h.prim = prim; h.slot = slot;
}

// slab test against the two child boxes of a node; conservative: NaN slabs are ignored
// (v_min/v_max drop NaN operands) and tfar is widened by 2 ulp.
PT_DEV void node_test(const BvhNode& n, v3 idir, v3 ood, float tmin, float tmax,
                      bool& hit0, bool& hit1, float& tn0, float& tn1)
{
    float c0lox = __builtin_fmaf(n.c0xy.x, idir.x, -ood.x), c0hix = __builtin_fmaf(n.c0xy.y, idir.x, -ood.x);
    float c0loy = __builtin_fmaf(n.c0xy.z, idir.y, -ood.y), c0hiy = __builtin_fmaf(n.c0xy.w, idir.y, -ood.y);
    float c0loz = __builtin_fmaf(n.cz.x, idir.z, -ood.z),   c0hiz = __builtin_fmaf(n.cz.y, idir.z, -ood.z);
    float c1lox = __builtin_fmaf(n.c1xy.x, idir.x, -ood.x), c1hix = __builtin_fmaf(n.c1xy.y, idir.x, -ood.x);
    float c1loy = __builtin_fmaf(n.c1xy.z, idir.y, -ood.y), c1hiy = __builtin_fmaf(n.c1xy.w, idir.y, -ood.y);
    float c1loz = __builtin_fmaf(n.cz.z, idir.z, -ood.z),   c1hiz = __builtin_fmaf(n.cz.w, idir.z, -ood.z);
    tn0 = fmaxf(fmaxf(fminf(c0lox, c0hix), fminf(c0loy, c0hiy)), fmaxf(fminf(c0loz, c0hiz), tmin));
    float tf0 = fminf(fminf(fmaxf(c0lox, c0hix), fmaxf(c0loy, c0hiy)), fminf(fmaxf(c0loz, c0hiz), tmax));
    tn1 = fmaxf(fmaxf(fminf(c1lox, c1hix), fminf(c1loy, c1hiy)), fmaxf(fminf(c1loz, c1hiz), tmin));
    float tf1 = fminf(fminf(fmaxf(c1lox, c1hix), fmaxf(c1loy, c1hiy)), fminf(fmaxf(c1loz, c1hiz), tmax));
    hit0 = tn0 <= tf0 * 1.0000004f;
    hit1 = tn1 <= tf1 * 1.0000004f;
}

PT_DEV float safe_inv1(float d)
{
    // A zero (or denormal-small) direction component would turn the slab planes into inf - inf = NaN,
    // and max(-inf, NaN) = -inf then culls a box the ray is inside of. Clamp |d| for the BOX test only
    // (the triangle test uses the true direction): planes become +-huge finite values with the right signs.
    const float a = fabsf(d) < 1e-20f ? copysignf(1e-20f, d) : d;
    return __builtin_amdgcn_rcpf(a);      // 1 ulp v_rcp_f32: the box test is conservative (padded boxes), not part of the spec
}
PT_DEV v3 safe_inv(v3 d) { return V3(safe_inv1(d.x), safe_inv1(d.y), safe_inv1(d.z)); }

// TraceRay: closest hit over the two-level structure. stack: per-lane array supplied by the caller.
__device__ __attribute__((noinline)) bool candidate_is_opaque(const AlphaContext ac, uint32_t inst, uint32_t geom, uint32_t prim, float u, float v)
{
    const PtObjectData* od = &ac.objects[ac.instances[inst].instanceID + geom];
    TexCoords tc;
    get_texture_coordinates(od, ac.heap, prim, u, v, tc);
    return is_opaque(od, ac.heap, ac.srgbLut, tc);
}

template <bool STATS, typename STACK>
PT_DEV Hit trace_closest(const AccelView& av, const AlphaContext& ac, v3 o, v3 d, float tmin, float tmax, STACK& stack, TraceStats* stats)
{
    Hit h; h.t = tmax; h.u = 0.0f; h.v = 0.0f; h.inst = ~0u; h.geom = 0; h.prim = 0; h.slot = 0;
    if (av.instanceCount == 0) return h;

    v3 ro = o, rd = d;                               // current-space ray (world, then object)
    v3 idir = safe_inv(rd), ood = ro * idir;
    RaySetup rs; rs.c1 = rs.c2 = false; rs.Sx = rs.Sy = rs.Sz = 0.0f;
    const BvhNode* nodes = av.tlasNodes;
    const TriPacket* tris = nullptr;
    uint32_t curInst = ~0u;
    bool bottom = false;

    stack.sp = 0;
    stack.push(kEntryDone);
    int cur = tlas_root_entry(av.instanceCount);
    while (true) {
        // ---- descend through internal nodes
        while (cur >= 0 && cur < kEntryRestore) {
            const BvhNode n = nodes[cur];
            if (STATS) stats->nodes++;
            bool h0, h1; float t0, t1;
            node_test(n, idir, ood, tmin, h.t, h0, h1, t0, t1);
            if (h0 && h1) {
                int nearc = n.child.x, farc = n.child.y;
                if (t1 < t0) { nearc = n.child.y; farc = n.child.x; }
                stack.push(farc);
                cur = nearc;
            } else if (h0) cur = n.child.x;
            else if (h1) cur = n.child.y;
            else cur = stack.pop();
        }
        if (cur == kEntryDone) break;
        if (cur == kEntryRestore) {                  // leave the BLAS: back to the world-space ray
            ro = o; rd = d; idir = safe_inv(rd); ood = ro * idir;
            nodes = av.tlasNodes; bottom = false;
            cur = stack.pop();
            continue;
        }
        // ---- leaf
        const uint32_t x = (uint32_t)~cur;
        if (!bottom) {
            const InstanceRecord* ir = &av.instances[x];
            if ((ir->mask & 0xFFu) && ir->triCount) {
                const float* W = ir->worldToObject;
                ro = V3(W[0] * o.x + W[1] * o.y + W[2]  * o.z + W[3],
                        W[4] * o.x + W[5] * o.y + W[6]  * o.z + W[7],
                        W[8] * o.x + W[9] * o.y + W[10] * o.z + W[11]);
                rd = V3(W[0] * d.x + W[1] * d.y + W[2]  * d.z,
                        W[4] * d.x + W[5] * d.y + W[6]  * d.z,
                        W[8] * d.x + W[9] * d.y + W[10] * d.z);
                idir = safe_inv(rd); ood = ro * idir;
                rs = ray_setup(rd);
                nodes = ir->nodes; tris = ir->tris; curInst = x; bottom = true;
                stack.push(kEntryRestore);
                cur = blas_root_entry(ir->triCount);
                continue;
            }
        } else {
            const uint32_t first = x >> 3, count = (x & 7u) + 1u;
            for (uint32_t i = 0; i < count; i++) {
                const TriPacket tp = tris[first + i];
                if (STATS) stats->tris++;
                float t, u, v;
                if (tri_test(rs, ro, V3(tp.a.x, tp.a.y, tp.a.z), V3(tp.b.x, tp.b.y, tp.b.z), V3(tp.c.x, tp.c.y, tp.c.z), t, u, v))
                    commit_candidate(ac, __float_as_uint(tp.c.w), h, tmin, t, u, v, curInst, __float_as_uint(tp.a.w), __float_as_uint(tp.b.w), first + i);
            }
        }
        cur = stack.pop();
    }
    if (h.inst != ~0u && !(h.t < tmax)) h.inst = ~0u;
    return h;
}

// TraceRay<RAY_FLAG_FORCE_NON_OPAQUE | RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH> (RaytracingHelpers.hlsli:7-55 with the
// coloured-visibility IsOpaque; the shape RTXDIAppBridge.hlsli:418-439 uses for shadow rays). Every triangle inside
// (tmin, tmax) is a candidate; a blocking one ends the search. Returns true when nothing was committed.
template <typename STACK>
PT_DEV bool trace_visibility(const AccelView& av, const AlphaContext& ac, v3 o, v3 d, float tmin, float tmax, STACK& stack, v3& vis)
{
    vis = V3(1.0f, 1.0f, 1.0f);
    if (av.instanceCount == 0) return true;
    v3 ro = o, rd = d;
    v3 idir = safe_inv(rd), ood = ro * idir;
    RaySetup rs; rs.c1 = rs.c2 = false; rs.Sx = rs.Sy = rs.Sz = 0.0f;
    const BvhNode* nodes = av.tlasNodes;
    const TriPacket* tris = nullptr;
    uint32_t curInst = ~0u;
    bool bottom = false;
    stack.sp = 0;
    stack.push(kEntryDone);
    int cur = tlas_root_entry(av.instanceCount);
    while (true) {
        while (cur >= 0 && cur < kEntryRestore) {
            const BvhNode n = nodes[cur];
            bool h0, h1; float t0, t1;
            node_test(n, idir, ood, tmin, tmax, h0, h1, t0, t1);
            if (h0 && h1) { stack.push(n.child.y); cur = n.child.x; }
            else if (h0) cur = n.child.x;
            else if (h1) cur = n.child.y;
            else cur = stack.pop();
        }
        if (cur == kEntryDone) break;
        if (cur == kEntryRestore) {
            ro = o; rd = d; idir = safe_inv(rd); ood = ro * idir;
            nodes = av.tlasNodes; bottom = false;
            cur = stack.pop();
            continue;
        }
        const uint32_t x = (uint32_t)~cur;
        if (!bottom) {
            const InstanceRecord* ir = &av.instances[x];
            if ((ir->mask & 0xFFu) && ir->triCount) {
                const float* W = ir->worldToObject;
                ro = V3(W[0] * o.x + W[1] * o.y + W[2]  * o.z + W[3], W[4] * o.x + W[5] * o.y + W[6]  * o.z + W[7], W[8] * o.x + W[9] * o.y + W[10] * o.z + W[11]);
                rd = V3(W[0] * d.x + W[1] * d.y + W[2]  * d.z, W[4] * d.x + W[5] * d.y + W[6]  * d.z, W[8] * d.x + W[9] * d.y + W[10] * d.z);
                idir = safe_inv(rd); ood = ro * idir;
                rs = ray_setup(rd);
                nodes = ir->nodes; tris = ir->tris; curInst = x; bottom = true;
                stack.push(kEntryRestore);
                cur = blas_root_entry(ir->triCount);
                continue;
            }
        } else {
            const uint32_t first = x >> 3, count = (x & 7u) + 1u;
            for (uint32_t i = 0; i < count; i++) {
                const TriPacket tp = tris[first + i];
                float t, u, v;
                if (!tri_test(rs, ro, V3(tp.a.x, tp.a.y, tp.a.z), V3(tp.b.x, tp.b.y, tp.b.z), V3(tp.c.x, tp.c.y, tp.c.z), t, u, v)) continue;
                if (!(t > tmin && t < tmax)) continue;
                const uint32_t geom = __float_as_uint(tp.a.w), prim = __float_as_uint(tp.b.w);
                const PtObjectData* od = &ac.objects[ac.instances[curInst].instanceID + geom];
                TexCoords tc;
                get_texture_coordinates(od, ac.heap, prim, u, v, tc);
                if (is_opaque_visibility(od, ac.heap, ac.srgbLut, tc, vis)) return false;
            }
        }
        cur = stack.pop();
    }
    return true;
}

// Debug / validation traversal (PT_DEBUG_BRUTE_FORCE): every triangle of every instance, no BVH.
PT_DEV Hit trace_brute_force(const AccelView& av, const AlphaContext& ac, v3 o, v3 d, float tmin, float tmax)
{
    Hit h; h.t = tmax; h.u = 0.0f; h.v = 0.0f; h.inst = ~0u; h.geom = 0; h.prim = 0; h.slot = 0;
    for (uint32_t x = 0; x < av.instanceCount; x++) {
        const InstanceRecord* ir = &av.instances[x];
        if (!(ir->mask & 0xFFu)) continue;
        const float* W = ir->worldToObject;
        v3 ro = V3(W[0] * o.x + W[1] * o.y + W[2]  * o.z + W[3],
                   W[4] * o.x + W[5] * o.y + W[6]  * o.z + W[7],
                   W[8] * o.x + W[9] * o.y + W[10] * o.z + W[11]);
        v3 rd = V3(W[0] * d.x + W[1] * d.y + W[2]  * d.z,
                   W[4] * d.x + W[5] * d.y + W[6]  * d.z,
                   W[8] * d.x + W[9] * d.y + W[10] * d.z);
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
