// pt_trace2.hpp -- phase-aligned two-level traversal over the compact scene blob (gfx950).
//
// Same closest-hit semantics and arithmetic as pt_trace.hpp (tri_test / commit are shared), different
// schedule, designed for wave64 SIMD efficiency:
//   phase A  every lane walks the TLAS only and collects up to K candidate instances in LDS
//   phase B  candidate k of every lane is processed in lock-step: world-box re-test against the current
//            best t, ray transform + Woop setup (the expensive block runs once per round, not once per
//            straggling lane), then a BLAS-only while-while traversal
// and back to phase A while the TLAS walk is unfinished. With interleaved TLAS/BLAS states (pt_trace.hpp) a
// wave executed the union of {TLAS node, instance entry, BLAS node, leaf, restore} every outer iteration.
//
// The blob is ONE contiguous allocation addressed in 16-byte units so that the identical code reads it from
// HBM/L2 (large scenes) or from LDS (scenes that fit: the block stages the blob once, "LDS-staged node /
// triangle packets"):   [ InstanceT x instCount | BvhNode x nodeCount | TriPacket x triCount ]
// TLAS nodes come first in the node array (root = 0); a BLAS's nodes / packets are contiguous at
// nodeBase / triBase. Node child references stay relative to their own tree.
#pragma once
#include "pt_trace.hpp"

namespace pt {

typedef float f4v __attribute__((ext_vector_type(4)));
#define PT_LDS_AS __attribute__((address_space(3)))

struct alignas(16) InstanceT {        // 96 B = 6 x 16
    float worldToObject[12];
    float boxLo[3]; uint32_t nodeBase;   // padded world AABB of the instance | BLAS root index in the blob node array
    float boxHi[3]; uint32_t triBase;    //                                   | first packet of the BLAS in the blob triangle array
    uint32_t mask, triCount, _pad[2];
};
static_assert(sizeof(InstanceT) == 96, "layout");

struct BlobView {
    const f4v* base;                   // device pointer to the blob
    uint32_t instOff16, nodeOff16, triOff16;   // section starts in 16-byte units
    uint32_t instCount, nodeCount, triCount;
    uint32_t bytes;                    // whole blob
};

constexpr uint32_t kInst16 = 6;        // 16-byte units per instance record
constexpr uint32_t kNode16 = 4;
constexpr uint32_t kTri16 = 3;
constexpr int kCandidates = 8;         // K: candidate instances gathered per phase A

template <bool LDS> struct BlobReader;
template <> struct BlobReader<false> {
    const f4v* p;
    PT_DEV f4v ld(uint32_t i) const { return p[i]; }
};
template <> struct BlobReader<true> {
    const PT_LDS_AS f4v* p;
    PT_DEV f4v ld(uint32_t i) const { return p[i]; }
};

PT_DEV void node_test_v(f4v c0xy, f4v c1xy, f4v cz, v3 idir, v3 ood, float tmin, float tmax, bool& hit0, bool& hit1, float& tn0, float& tn1)
{
    BvhNode n;
    n.c0xy = make_float4(c0xy.x, c0xy.y, c0xy.z, c0xy.w);
    n.c1xy = make_float4(c1xy.x, c1xy.y, c1xy.z, c1xy.w);
    n.cz = make_float4(cz.x, cz.y, cz.z, cz.w);
    node_test(n, idir, ood, tmin, tmax, hit0, hit1, tn0, tn1);
}

// lane-private views of the LDS scratch: entry e of thread t at base[e * 256 + t]
template <bool STATS, bool LDS, int STACK_DEPTH>
PT_DEV Hit trace_closest_v2(const BlobReader<LDS>& blob, const BlobView& bv, const AlphaContext& ac, v3 o, v3 d, float tmin, float tmax,
                            int* ldsStack, uint32_t* ldsCand, TraceStats* stats)
{
    Hit h; h.t = tmax; h.u = 0.0f; h.v = 0.0f; h.inst = ~0u; h.geom = 0; h.prim = 0; h.slot = 0;
    if (bv.instCount == 0) return h;
    int spill[kStackSize - STACK_DEPTH];
    TraversalStack<STACK_DEPTH> stack; stack.init(ldsStack, spill);
    stack.push(kEntryDone);
    uint32_t* cand = ldsCand + threadIdx.x;

    const v3 idir = safe_inv(d), ood = o * idir;
    int cur = tlas_root_entry(bv.instCount);
    bool tlasDone = false;
    while (true) {
        // ---------------- phase A: TLAS walk, collect candidates
        uint32_t nCand = 0;
        while (!tlasDone && nCand < (uint32_t)kCandidates) {
            if (cur >= 0 && cur < kEntryRestore) {
                const uint32_t a = bv.nodeOff16 + (uint32_t)cur * kNode16;
                const f4v n0 = blob.ld(a), n1 = blob.ld(a + 1), n2 = blob.ld(a + 2), n3 = blob.ld(a + 3);
                if (STATS) stats->nodes++;
                bool h0, h1; float t0, t1;
                node_test_v(n0, n1, n2, idir, ood, tmin, h.t, h0, h1, t0, t1);
                const int c0 = __float_as_int(n3.x), c1 = __float_as_int(n3.y);
                if (h0 && h1) {
                    int nearc = c0, farc = c1;
                    if (t1 < t0) { nearc = c1; farc = c0; }
                    stack.push(farc);
                    cur = nearc;
                } else if (h0) cur = c0;
                else if (h1) cur = c1;
                else cur = stack.pop();
            } else if (cur == kEntryDone) {
                tlasDone = true;
            } else {
                cand[nCand * 256] = (uint32_t)~cur;
                nCand++;
                cur = stack.pop();
            }
        }
        // ---------------- phase B: candidates in lock-step
        for (uint32_t k = 0; k < nCand; k++) {
            const uint32_t x = cand[k * 256];
            const uint32_t ia = bv.instOff16 + x * kInst16;
            const f4v b0 = blob.ld(ia + 3), b1 = blob.ld(ia + 4);          // boxLo|nodeBase, boxHi|triBase
            const f4v mk = blob.ld(ia + 5);                                  // mask, triCount, -, -
            const uint32_t ntri = __float_as_uint(mk.y);
            if (!(__float_as_uint(mk.x) & 0xFFu) || ntri == 0u) continue;
            {   // late cull against the current best hit
                const float lx = __builtin_fmaf(b0.x, idir.x, -ood.x), hx = __builtin_fmaf(b1.x, idir.x, -ood.x);
                const float ly = __builtin_fmaf(b0.y, idir.y, -ood.y), hy = __builtin_fmaf(b1.y, idir.y, -ood.y);
                const float lz = __builtin_fmaf(b0.z, idir.z, -ood.z), hz = __builtin_fmaf(b1.z, idir.z, -ood.z);
                const float tn = fmaxf(fmaxf(fminf(lx, hx), fminf(ly, hy)), fmaxf(fminf(lz, hz), tmin));
                const float tf = fminf(fminf(fmaxf(lx, hx), fmaxf(ly, hy)), fminf(fmaxf(lz, hz), h.t));
                if (!(tn <= tf * 1.0000004f)) continue;
            }
            const f4v w0 = blob.ld(ia), w1 = blob.ld(ia + 1), w2 = blob.ld(ia + 2);
            const v3 ro = V3(w0.x * o.x + w0.y * o.y + w0.z * o.z + w0.w,
                             w1.x * o.x + w1.y * o.y + w1.z * o.z + w1.w,
                             w2.x * o.x + w2.y * o.y + w2.z * o.z + w2.w);
            const v3 rd = V3(w0.x * d.x + w0.y * d.y + w0.z * d.z,
                             w1.x * d.x + w1.y * d.y + w1.z * d.z,
                             w2.x * d.x + w2.y * d.y + w2.z * d.z);
            const v3 bidir = safe_inv(rd), bood = ro * bidir;
            const RaySetup rs = ray_setup(rd);
            const uint32_t nodeBase = bv.nodeOff16 + __float_as_uint(b0.w) * kNode16;
            const uint32_t triBase = bv.triOff16 + __float_as_uint(b1.w) * kTri16;
            stack.push(kEntryRestore);
            int c = blas_root_entry(ntri);
            while (true) {
                while (c >= 0 && c < kEntryRestore) {
                    const uint32_t a = nodeBase + (uint32_t)c * kNode16;
                    const f4v n0 = blob.ld(a), n1 = blob.ld(a + 1), n2 = blob.ld(a + 2), n3 = blob.ld(a + 3);
                    if (STATS) stats->nodes++;
                    bool h0, h1; float t0, t1;
                    node_test_v(n0, n1, n2, bidir, bood, tmin, h.t, h0, h1, t0, t1);
                    const int c0 = __float_as_int(n3.x), c1 = __float_as_int(n3.y);
                    if (h0 && h1) {
                        int nearc = c0, farc = c1;
                        if (t1 < t0) { nearc = c1; farc = c0; }
                        stack.push(farc);
                        c = nearc;
                    } else if (h0) c = c0;
                    else if (h1) c = c1;
                    else c = stack.pop();
                }
                if (c == kEntryRestore) break;
                const uint32_t leaf = (uint32_t)~c;
                const uint32_t first = leaf >> 3, count = (leaf & 7u) + 1u;
                for (uint32_t i = 0; i < count; i++) {
                    const uint32_t ta = triBase + (first + i) * kTri16;
                    const f4v pa = blob.ld(ta), pb = blob.ld(ta + 1), pc = blob.ld(ta + 2);
                    if (STATS) stats->tris++;
                    float t, u, v;
                    if (tri_test(rs, ro, V3(pa.x, pa.y, pa.z), V3(pb.x, pb.y, pb.z), V3(pc.x, pc.y, pc.z), t, u, v))
                        commit_candidate(ac, __float_as_uint(pc.w), h, tmin, t, u, v, x, __float_as_uint(pa.w), __float_as_uint(pb.w), first + i);
                }
                c = stack.pop();
            }
        }
        if (tlasDone) break;
    }
    if (h.inst != ~0u && !(h.t < tmax)) h.inst = ~0u;
    return h;
}

} // namespace pt
