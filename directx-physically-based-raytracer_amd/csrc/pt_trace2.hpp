// pt_trace2.hpp -- traversal of the compact scene blob (gfx950): the BLAS work-item routine over compressed wide nodes, the
// phase-aligned two-level schedule, the flat schedule with wave-compacted work items for scenes of at most kFlatInstances
// instances, and the plain one-lane-one-ray two-level walk (G-buffer rays, shadow rays, validation).
//
// Same closest-hit semantics and arithmetic everywhere (tri_test / commit are shared), different schedules, designed for
// wave64 SIMD efficiency. Phase-aligned:
//   phase A  every lane walks the TLAS only and collects up to K candidate instances in LDS
//   phase B  the candidates of the whole wave become work items (ray, instance) that any free lane processes: late cull
//            against the current best t, ray transform + Woop setup, BLAS-only traversal
// and back to phase A while the TLAS walk is unfinished.
//
// The blob is ONE contiguous allocation addressed in 16-byte units so that the identical code reads it from
// HBM/L2 (large scenes) or from LDS (scenes that fit: the block stages the blob once, "LDS-staged node /
// triangle packets"):   [ InstanceT x instCount | WideNode x nodeCount | TriPacket x triCount | InstanceT x instCount in TLAS leaf order | entry records ]
// TLAS nodes come first in the node array (root = 0); a BLAS's nodes / packets are contiguous at nodeBase / triBase. Child
// and triangle references of a node stay relative to their own tree. The TLAS's "triangles" are the instance records of the last
// section: the same records as the first, in the order of the TLAS leaves, so that entering an instance is one fetch.
#pragma once
#include "pt_trace.hpp"

namespace pt {

// Developer build only (-DPT_ROUND_PROF, tools/round_prof.py): wave-clock stamps between the sections of a fused round.
#ifdef PT_ROUND_PROF
struct RoundProf { uint64_t last; uint32_t acc[16]; };
#define PT_PROF_WAIT() __builtin_amdgcn_s_waitcnt(0)
#define PT_PROF_MARK(P, i) do { if (P) { const uint64_t now_ = __builtin_readcyclecounter(); (P)->acc[i] += (uint32_t)(now_ - (P)->last); (P)->last = now_; } } while (0)
#else
struct RoundProf;
#define PT_PROF_MARK(P, i) do { } while (0)
#define PT_PROF_WAIT() do { } while (0)
#endif

struct alignas(16) InstanceT {        // 144 B = 9 x 16
    float worldToObject[12];
    float boxLo[3]; uint32_t nodeBase;   // padded world AABB of the instance | BLAS root index in the blob node array
    float boxHi[3]; uint32_t triBase;    //                                   | first packet of the BLAS in the blob triangle array
    uint32_t mask, triCount, instanceID, instanceIndex;   // instanceIndex: position in the API's instance array (InstanceIndex())
    float objectToWorld[12];             // units 6..8: only the shading half of k_round reads them (hit reconstruction)
};
static_assert(sizeof(InstanceT) == 144, "layout");

struct BlobView {
    const f4v* base;                   // device pointer to the blob
    uint32_t instOff16, nodeOff16, triOff16, leafInstOff16;   // section starts in 16-byte units
    uint32_t idxOff16;                 // vertex indices: one unit per triangle packet, same numbering (i0, i1, i2, -)
    uint32_t enterOff16;               // entry records (9 units per instance, TLAS leaf order): worldToObject | bases, count, mask, index | the BLAS's root node
    uint32_t instCount, nodeCount, triCount;
    uint32_t bytes;                    // whole blob
};

constexpr uint32_t kInst16 = 9;        // 16-byte units per instance record
constexpr uint32_t kNode16 = 5;
constexpr uint32_t kTri16 = 3;
constexpr int kCandidates = 8;         // K: candidate instances gathered per phase A

template <bool LDS> struct BlobReader;
template <> struct BlobReader<false> {
    const f4v* p;
    PT_DEV f4v ld(uint32_t i) const { return p[i]; }
    PT_DEV uint32_t ld32(uint32_t i) const { return ((const uint32_t*)p)[i]; }
};
template <> struct BlobReader<true> {
    const PT_LDS_AS f4v* p;
    PT_DEV f4v ld(uint32_t i) const { return p[i]; }
    PT_DEV uint32_t ld32(uint32_t i) const { return ((const PT_LDS_AS uint32_t*)p)[i]; }
};

// One node visit: pops the highest-priority hit child off G, fetches that node and tests its eight boxes. On return G is the
// group of the fetched node's hit internal children, T the group of its hit triangles; the rest of the old G went on the stack.
template <bool STATS, bool LDS, typename STACK>
PT_DEV void visit_node(const BlobReader<LDS>& blob, uint32_t nodeBase16, const BoxRay& br, float tmin, float tmax, uint2& G, uint2& T,
                       STACK& stack, TraceStats* stats)
{
    const uint32_t bit = 31u - (uint32_t)__builtin_clz(G.y);
    G.y &= ~(1u << bit);
    if (G.y > 0x00FFFFFFu) stack.push(G);
    const uint32_t slot = (bit - 24u) ^ (br.octinv4 & 7u);
    const uint32_t rel = (uint32_t)__builtin_popcount(G.y & 0xFFu & ~(0xFFFFFFFFu << slot));
    const uint32_t a = nodeBase16 + (G.x + rel) * kNode16;
    const f4v n0 = blob.ld(a), n1 = blob.ld(a + 1), n2 = blob.ld(a + 2), n3 = blob.ld(a + 3), n4 = blob.ld(a + 4);
    if (STATS) stats->nodes++;
    const uint32_t hits = wide_node_hits(n0, n1, n2, n3, n4, br, tmin, tmax);
    G = make_uint2(__float_as_uint(n1.x), (hits & 0xFF000000u) | (__float_as_uint(n0.w) >> 24));
    T = make_uint2(__float_as_uint(n1.y), hits & 0x00FFFFFFu);
}

// ---------------------------------------------------------------------------------------------
// A work item = (ray, instance): the unit the wave-compacted schedules below hand to whichever lane is free. The ray comes
// out of the wave's LDS exchange area (rays[2 * lane] = o.xyz tmin, rays[2 * lane + 1] = d.xyz best t so far), the closest
// hit inside the instance goes back as (t, u, v, slot); slot == ~0u: nothing closer than the ray's best t.
// The item-local best starts at the ray's best t with an id that loses every tie, so a candidate at exactly that t is kept
// and the ray's own lane applies the instance order when it merges.
template <bool STATS, bool LDS, bool CULL, typename STACK>
PT_DEV f4v trace_item(const BlobReader<LDS>& blob, const BlobView& bv, const AlphaContext& ac, const f4v* rays, uint32_t src, uint32_t x,
                      STACK& stack, TraceStats* stats)
{
    const f4v r0 = rays[2 * src], r1 = rays[2 * src + 1];
    const v3 io = V3(r0.x, r0.y, r0.z), id = V3(r1.x, r1.y, r1.z);
    const float itmin = r0.w, ibest = r1.w;
    Hit hi; hi.t = ibest; hi.u = 0.0f; hi.v = 0.0f; hi.inst = 0xFFFFFFFEu; hi.geom = ~0u; hi.prim = ~0u; hi.slot = ~0u;
    const uint32_t ia = bv.instOff16 + x * kInst16;
    const f4v b0 = blob.ld(ia + 3), b1 = blob.ld(ia + 4);
    bool enter = true;
    if (CULL) {                      // late cull against the ray's best hit so far
        const v3 iidir = safe_inv(id), iood = io * iidir;
        const float lx = __builtin_fmaf(b0.x, iidir.x, -iood.x), hx = __builtin_fmaf(b1.x, iidir.x, -iood.x);
        const float ly = __builtin_fmaf(b0.y, iidir.y, -iood.y), hy = __builtin_fmaf(b1.y, iidir.y, -iood.y);
        const float lz = __builtin_fmaf(b0.z, iidir.z, -iood.z), hz = __builtin_fmaf(b1.z, iidir.z, -iood.z);
        const float tn = fmaxf(fmaxf(fminf(lx, hx), fminf(ly, hy)), fmaxf(fminf(lz, hz), itmin));
        const float tf = fminf(fminf(fmaxf(lx, hx), fmaxf(ly, hy)), fminf(fmaxf(lz, hz), ibest));
        enter = tn <= tf * 1.0000004f;
    }
    if (enter) {
        const f4v w0 = blob.ld(ia), w1 = blob.ld(ia + 1), w2 = blob.ld(ia + 2);
        const v3 ro = V3(sop3t(w0.x, io.x, w0.y, io.y, w0.z, io.z, w0.w), sop3t(w1.x, io.x, w1.y, io.y, w1.z, io.z, w1.w), sop3t(w2.x, io.x, w2.y, io.y, w2.z, io.z, w2.w));
        const v3 rd = V3(sop3(w0.x, id.x, w0.y, id.y, w0.z, id.z), sop3(w1.x, id.x, w1.y, id.y, w1.z, id.z), sop3(w2.x, id.x, w2.y, id.y, w2.z, id.z));
        const RaySetup rs = ray_setup(rd);
        const BoxRay br = box_ray(ro, rd);
        const uint32_t triBase16 = bv.triOff16 + __float_as_uint(b1.w) * kTri16;
        const uint32_t nodeBase16 = bv.nodeOff16 + __float_as_uint(b0.w) * kNode16;
        const uint32_t ntri = __float_as_uint(blob.ld(ia + 5).y);
        const bool single = blas_single_leaf(ntri);                          // single-leaf BLAS: straight to its triangles
        uint2 G = root_node_group(single), T = root_tri_group(single, ntri);
        const int floor = stack.sp;                                          // the caller's entries below stay untouched
        while (true) {
            if (G.y > 0x00FFFFFFu) visit_node<STATS, LDS>(blob, nodeBase16, br, itmin, hi.t, G, T, stack, stats);
            while (T.y) {
                const uint32_t i = T.x + (uint32_t)__builtin_ctz(T.y);
                T.y &= T.y - 1u;
                const uint32_t ta = triBase16 + i * kTri16;
                const f4v pa = blob.ld(ta), pb = blob.ld(ta + 1), pc = blob.ld(ta + 2);
                if (STATS) stats->tris++;
                float t, u, v;
                if (tri_test(rs, ro, V3(pa.x, pa.y, pa.z), V3(pb.x, pb.y, pb.z), V3(pc.x, pc.y, pc.z), t, u, v))
                    commit_candidate(ac, __float_as_uint(pc.w), hi, itmin, t, u, v, x, __float_as_uint(pa.w), __float_as_uint(pb.w), i);
            }
            if (G.y <= 0x00FFFFFFu) {
                if (stack.sp == floor) break;
                G = stack.pop();
            }
        }
    }
    return (f4v){ hi.t, hi.u, hi.v, __uint_as_float(hi.slot) };
}

// merge of one item result into the ray's best hit: is_better() with the instance as the whole tie-break (two results of a
// ray never share an instance; geometry / primitive ties were settled inside the item)
PT_DEV void merge_item(Hit& h, float tmin, f4v q, uint32_t x)
{
    const uint32_t slot = __float_as_uint(q.w);
    if (slot != ~0u && q.x > tmin && (q.x < h.t || (q.x == h.t && h.inst != ~0u && x < h.inst))) {
        h.t = q.x; h.u = q.y; h.v = q.z; h.inst = x; h.slot = slot;
    }
}

// Phase-aligned schedule. lane-private views of the LDS scratch: entry e of thread t at base[e * 256 + t].
// Phase B hands the collected candidates to the wave as work items (trace_item): lanes with many candidates are helped by
// lanes with few, instead of every lane waiting for the longest list (before: lock-step over candidate k of every lane).
// The WHOLE wave must call this function (lanes without a ray pass an empty interval): its lanes trade work.
constexpr uint32_t kPhasedItems = 128;                                       // items per batch and wave
constexpr uint32_t kPhasedWaveLds = 64u * 32u + kPhasedItems * 16u + kPhasedItems * 4u;

template <bool STATS, bool LDS, int STACK_DEPTH>
PT_DEV Hit trace_closest_v2(const BlobReader<LDS>& blob, const BlobView& bv, const AlphaContext& ac, v3 o, v3 d, float tmin, float tmax,
                            PT_LDS_AS void* ldsStack, uint32_t* ldsCand, unsigned char* ldsWave, TraceStats* stats)
{
    Hit h; h.t = tmax; h.u = 0.0f; h.v = 0.0f; h.inst = ~0u; h.geom = 0; h.prim = 0; h.slot = 0;
    if (bv.instCount == 0) return h;
    uint2 spill[kStackSize - STACK_DEPTH];
    GroupStack<STACK_DEPTH> stack; stack.init(ldsStack, spill);
    uint32_t* cand = ldsCand + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long ltMask = (1ull << lane) - 1ull;
    f4v* rays = (f4v*)ldsWave;                                              // [lane][2]: o.xyz tmin | d.xyz best t
    f4v* results = (f4v*)(ldsWave + 64u * 32u);                             // [item]: t u v slot
    uint32_t* items = (uint32_t*)(ldsWave + 64u * 32u + kPhasedItems * 16u);  // [item]: lane | instance << 8
    rays[2 * lane] = (f4v){ o.x, o.y, o.z, tmin };

    const BoxRay br = box_ray(o, d);
    const bool oneInstance = bv.instCount == 1u;                            // a TLAS of one leaf: straight to the instance
    uint2 G = root_node_group(oneInstance), T = root_tri_group(oneInstance, 1u);
    bool tlasDone = !(tmin <= tmax);                                        // an empty interval (a lane without a ray) has nothing to walk
    while (true) {
        // ---------------- phase A: TLAS walk, collect candidates (the TLAS's "triangles" are instance records in leaf order)
        uint32_t nCand = 0;
        while (!tlasDone && nCand < (uint32_t)kCandidates) {
            if (T.y) {
                const uint32_t i = T.x + (uint32_t)__builtin_ctz(T.y);
                T.y &= T.y - 1u;
                const f4v mk = blob.ld(bv.leafInstOff16 + i * kInst16 + 5);  // mask, triCount, InstanceID, InstanceIndex
                if ((__float_as_uint(mk.x) & 0xFFu) && __float_as_uint(mk.y) != 0u) { cand[nCand * 256] = __float_as_uint(mk.w); nCand++; }
            } else if (G.y > 0x00FFFFFFu) {
                visit_node<STATS, LDS>(blob, bv.nodeOff16, br, tmin, h.t, G, T, stack, stats);
            } else if (stack.sp > 0) {
                G = stack.pop();
            } else {
                tlasDone = true;
            }
        }
        // ---------------- phase B: the wave's candidates as compacted work items, batch by batch
        for (uint32_t k = 0; wave_ballot(nCand > k) != 0ull; ) {                // k: levels (k-th candidate of every lane) done; uniform
            uint32_t total = 0, k2 = k;
            while (true) {
                const unsigned long long bb = wave_ballot(nCand > k2);
                const uint32_t n = (uint32_t)__builtin_popcountll(bb);
                if (n == 0u || total + n > kPhasedItems) break;
                if (nCand > k2) items[total + (uint32_t)__builtin_popcountll(bb & ltMask)] = lane | (cand[k2 * 256] << 8);
                total += n; k2++;
            }
            rays[2 * lane + 1] = (f4v){ d.x, d.y, d.z, h.t };
            __builtin_amdgcn_wave_barrier();
            for (uint32_t j = lane; j < total; j += 64u) {
                const uint32_t it = items[j];
                results[j] = trace_item<STATS, LDS, true>(blob, bv, ac, rays, it & 0xFFu, it >> 8, stack, stats);   // on top of this lane's TLAS stack
            }
            __builtin_amdgcn_wave_barrier();
            uint32_t off = 0;
            for (uint32_t kk = k; kk < k2; kk++) {
                const unsigned long long bb = wave_ballot(nCand > kk);
                if (nCand > kk) merge_item(h, tmin, results[off + (uint32_t)__builtin_popcountll(bb & ltMask)], cand[kk * 256]);
                off += (uint32_t)__builtin_popcountll(bb);
            }
            __builtin_amdgcn_wave_barrier();
            k = k2;
        }
        if (!wave_ballot(!tlasDone)) break;                                    // the wave leaves together: finished lanes keep serving items
    }
    stats->overflow += stack.overflow;
    if (h.inst != ~0u && !(h.t < tmax)) h.inst = ~0u;
    return h;
}

// ---------------------------------------------------------------------------------------------
// Flat variant for scenes with at most kFlatInstances instances (every Cornell-type scene): no TLAS walk.
//   scan     all lanes test the world boxes of ALL instances in the same (uniform) order -- broadcast loads, no stack, no
//            divergence -- and keep two bit masks: meshes and single-leaf instances (blas_single_leaf: e.g. a quad)
//   items    every set bit is a work item (ray, instance). The items of the whole wave -- meshes first, then quads -- are
//            compacted into a list in LDS and lane j processes items j, j + 64, ...: not necessarily of its own ray. A ray
//            that touches two boxes and a wall is served by three lanes at once, a ray that touches one wall lends its lane
//            to a neighbour. An item = (late cull,) ray transform + Woop setup, BLAS-only while-while loop entered at the
//            root (meshes) or directly at the only leaf (quads); its closest hit goes back through LDS
//   merge    the ray's own lane takes the results of its items; ties on t go to the lower instance index (one item per
//            (ray, instance), so geometry / primitive order is already settled inside the item)
// Lanes that hit a 2-triangle wall no longer idle while their neighbours walk a 12-triangle box, and both kinds run with
// full lanes (measured before compaction: 1.9 mesh rounds per wave at 38 % occupancy + 2.4 quad rounds at 54 %; now
// ~128 items in 2 rounds). Same tri_test / is_better arithmetic, so the result is bit-identical to the other schedules.
constexpr uint32_t kFlatInstances = 32;
constexpr int kStackLdsFlat = 4;                                             // BLAS-only group stacks are shallow (one entry per level); deeper entries spill
constexpr uint32_t kFlatItems = 192;                                         // items per batch and wave
// LDS per wave for the exchange: rays 64 x 32 B | results kFlatItems x 16 B | items kFlatItems x 4 B
constexpr uint32_t kFlatWaveLds = 64u * 32u + kFlatItems * 16u + kFlatItems * 4u;
constexpr uint32_t kFlatLdsFixed = (uint32_t)kStackLdsFlat * 256u * 8u + 4u * kFlatWaveLds;

template <bool STATS, bool LDS>
PT_DEV Hit trace_closest_flat(const BlobReader<LDS>& blob, const BlobView& bv, const AlphaContext& ac, v3 o, v3 d, float tmin, float tmax,
                              PT_LDS_AS void* ldsStack, unsigned char* ldsWave, TraceStats* stats, RoundProf* prof = nullptr)
{
    Hit h; h.t = tmax; h.u = 0.0f; h.v = 0.0f; h.inst = ~0u; h.geom = 0; h.prim = 0; h.slot = 0;
    uint32_t quads = 0, meshes = 0;
    {
        const v3 idir = safe_inv(d), ood = o * idir;
        for (uint32_t x = 0; x < bv.instCount; x++) {                       // uniform
            const uint32_t ia = bv.instOff16 + x * kInst16;
            const f4v b0 = blob.ld(ia + 3), b1 = blob.ld(ia + 4), mk = blob.ld(ia + 5);
            const uint32_t ntri = __float_as_uint(mk.y);
            if (!(__float_as_uint(mk.x) & 0xFFu) || ntri == 0u) continue;   // uniform
            const float lx = __builtin_fmaf(b0.x, idir.x, -ood.x), hx = __builtin_fmaf(b1.x, idir.x, -ood.x);
            const float ly = __builtin_fmaf(b0.y, idir.y, -ood.y), hy = __builtin_fmaf(b1.y, idir.y, -ood.y);
            const float lz = __builtin_fmaf(b0.z, idir.z, -ood.z), hz = __builtin_fmaf(b1.z, idir.z, -ood.z);
            const float tn = fmaxf(fmaxf(fminf(lx, hx), fminf(ly, hy)), fmaxf(fminf(lz, hz), tmin));
            const float tf = fminf(fminf(fmaxf(lx, hx), fmaxf(ly, hy)), fminf(fmaxf(lz, hz), tmax));
            const uint32_t bit = (tn <= tf * 1.0000004f) ? (1u << x) : 0u;
            if (blas_single_leaf(ntri)) quads |= bit; else meshes |= bit;   // uniform select
        }
        if (STATS) stats->nodes += (bv.instCount * 2u + 4u) / 5u;           // an instance box is 32 B, a node 80 B: counted by bytes
    }
    PT_PROF_MARK(prof, 1);

    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long ltMask = (1ull << lane) - 1ull;
    f4v* rays = (f4v*)ldsWave;                                              // [lane][2]: o.xyz tmin | d.xyz best t
    f4v* results = (f4v*)(ldsWave + 64u * 32u);                             // [item]: t u v slot
    uint32_t* items = (uint32_t*)(ldsWave + 64u * 32u + kFlatItems * 16u);  // [item]: lane | instance << 8
    const uint32_t nm = (uint32_t)__builtin_popcount(meshes), nq = (uint32_t)__builtin_popcount(quads);
    if (!wave_ballot((nm | nq) != 0u)) return h;
    rays[2 * lane] = (f4v){ o.x, o.y, o.z, tmin };
    uint2 spill[kStackSize - kStackLdsFlat];
    GroupStack<kStackLdsFlat> stack; stack.init(ldsStack, spill);

    uint32_t km = 0, kq = 0;                                                // levels (k-th set bit of every lane) already done; uniform
    bool firstBatch = true;
    while (true) {
        // ---- a batch: whole levels, meshes before quads, while they fit
        uint32_t total = 0, km2 = km, kq2 = kq;
        {
            uint32_t mb = meshes, qb = quads;
            while (true) {
                const unsigned long long bb = wave_ballot(nm > km2);
                const uint32_t n = (uint32_t)__builtin_popcountll(bb);
                if (n == 0u || total + n > kFlatItems) break;
                if (nm > km2) {
                    const uint32_t x = (uint32_t)__builtin_ctz(mb); mb &= mb - 1u;
                    items[total + (uint32_t)__builtin_popcountll(bb & ltMask)] = lane | (x << 8);
                }
                total += n; km2++;
            }
            if (!wave_ballot(nm > km2)) {                                      // every mesh level is in: quads may follow
                while (true) {
                    const unsigned long long bb = wave_ballot(nq > kq2);
                    const uint32_t n = (uint32_t)__builtin_popcountll(bb);
                    if (n == 0u || total + n > kFlatItems) break;
                    if (nq > kq2) {
                        const uint32_t x = (uint32_t)__builtin_ctz(qb); qb &= qb - 1u;
                        items[total + (uint32_t)__builtin_popcountll(bb & ltMask)] = lane | (x << 8);
                    }
                    total += n; kq2++;
                }
            }
        }
        if (total == 0u) break;                                             // nothing left (a level never exceeds 64 <= kFlatItems)
        rays[2 * lane + 1] = (f4v){ d.x, d.y, d.z, h.t };
        __builtin_amdgcn_wave_barrier();
        PT_PROF_MARK(prof, 2);
#ifdef PT_ROUND_PROF
        if (prof) { prof->acc[9] += total; prof->acc[10] += (total + 63u) / 64u; }
#endif

        // ---- process: lane j takes items j, j + 64, ...
        for (uint32_t j = lane; j < total; j += 64u) {
            const uint32_t it = items[j];
            stack.sp = 0;
            results[j] = firstBatch ? trace_item<STATS, LDS, false>(blob, bv, ac, rays, it & 0xFFu, it >> 8, stack, stats)      // first batch: the scan's verdict stands
                                    : trace_item<STATS, LDS, true>(blob, bv, ac, rays, it & 0xFFu, it >> 8, stack, stats);
        }
        __builtin_amdgcn_wave_barrier();
        PT_PROF_MARK(prof, 3);

        // ---- merge: every ray takes the results of its items of this batch, in the order they were listed
        {
            uint32_t off = 0;
            for (uint32_t pass = 0; pass < 2u; pass++) {
                const uint32_t ka = pass == 0u ? km : kq, kb = pass == 0u ? km2 : kq2, mine = pass == 0u ? nm : nq;
                for (uint32_t kk = ka; kk < kb; kk++) {
                    const unsigned long long bb = wave_ballot(mine > kk);
                    if (mine > kk) {
                        uint32_t x;
                        if (pass == 0u) { x = (uint32_t)__builtin_ctz(meshes); meshes &= meshes - 1u; }
                        else { x = (uint32_t)__builtin_ctz(quads); quads &= quads - 1u; }
                        merge_item(h, tmin, results[off + (uint32_t)__builtin_popcountll(bb & ltMask)], x);
                    }
                    off += (uint32_t)__builtin_popcountll(bb);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        PT_PROF_MARK(prof, 4);
        km = km2; kq = kq2; firstBatch = false;
    }
    stats->overflow += stack.overflow;
    if (h.inst != ~0u && !(h.t < tmax)) h.inst = ~0u;
    return h;
}

// ---------------------------------------------------------------------------------------------
// One lane, one ray, one stack over both levels: the plain form of TraceRay. ANYHIT = false: closest hit (primary rays of the
// G-buffer pass in scenes too large for the flat schedule, PT_DEBUG_TRAVERSAL_V1 cross-checks). ANYHIT = true:
// TraceRay<RAY_FLAG_FORCE_NON_OPAQUE | RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH> with the coloured-visibility IsOpaque
// (RaytracingHelpers.hlsli:7-55; the shape RTXDIAppBridge.hlsli:418-439 uses for shadow rays): every triangle inside
// (tmin, tmax) is a candidate exactly once (no leaf is visited twice), a blocking one ends the search; vis = product of the
// candidates' transmittances, the returned hit has inst == ~0u when nothing was committed.
// An instance transition parks the TLAS state as three stack entries: G, T (instances still to enter) and a marker.
// LOG (developer aid, pt_debug_trace_ray): every step of the walk is appended to log[] as (code, a, b, sp) words.
template <bool STATS, bool LDS, bool ANYHIT, typename STACK, bool LOG = false>
PT_DEV Hit trace_single(const BlobReader<LDS>& blob, const BlobView& bv, const AlphaContext& ac, v3 o, v3 d, float tmin, float tmax,
                        STACK& stack, TraceStats* stats, v3* vis, uint32_t* log = nullptr, uint32_t logCap = 0)
{
    uint32_t nlog = 0;
    #define PT_LOG(code, a, b) do { if (LOG && 4u * nlog + 4u <= logCap) { log[4 * nlog] = (code); log[4 * nlog + 1] = (a); log[4 * nlog + 2] = (b); log[4 * nlog + 3] = (uint32_t)stack.sp; nlog++; } } while (0)
    Hit h; h.t = tmax; h.u = 0.0f; h.v = 0.0f; h.inst = ~0u; h.geom = 0; h.prim = 0; h.slot = 0;
    if (ANYHIT) *vis = V3(1.0f, 1.0f, 1.0f);
    if (bv.instCount == 0 || !(tmin <= tmax)) return h;
    constexpr uint32_t kMarker = 0xFFFFFFFFu;
    BoxRay br = box_ray(o, d);
    RaySetup rs; rs.c1 = rs.c2 = false; rs.Sx = rs.Sy = rs.Sz = 0.0f;
    v3 ro = o;
    // curInst == ~0u: the walk is in the top level. (An integer on purpose: with a separate `bool bottom` flag hipcc 7.2 -O3 kept the
    // flag where a lane leaving its BLAS cleared it for its neighbours too -- seen on hardware with pt_debug_trace_ray.)
    uint32_t nodeBase16 = bv.nodeOff16, triBase16 = 0, curInst = ~0u;
    const bool oneInstance = bv.instCount == 1u;
    uint2 G = root_node_group(oneInstance), T = root_tri_group(oneInstance, 1u);
    stack.sp = 0;
    while (true) {
        if (T.y) {
            const uint32_t i = T.x + (uint32_t)__builtin_ctz(T.y);
            T.y &= T.y - 1u;
            if (curInst == ~0u) {                                           // a TLAS "triangle": enter the instance
                // the entry record (blob section enterOff16): worldToObject | node base, triangle base, count + mask, InstanceIndex | the
                // BLAS's root node -- one fetch, and the root is visited in the same iteration
                const uint32_t ia = bv.enterOff16 + i * kInst16;
                const f4v mk = blob.ld(ia + 3);
                const uint32_t cm = __float_as_uint(mk.z), ntri = cm & 0x00FFFFFFu, x = __float_as_uint(mk.w);
                if ((cm >> 24) && ntri != 0u) {
                    const f4v w0 = blob.ld(ia), w1 = blob.ld(ia + 1), w2 = blob.ld(ia + 2);
                    const f4v n0 = blob.ld(ia + 4), n1 = blob.ld(ia + 5), n2 = blob.ld(ia + 6), n3 = blob.ld(ia + 7), n4 = blob.ld(ia + 8);
                    ro = V3(sop3t(w0.x, o.x, w0.y, o.y, w0.z, o.z, w0.w), sop3t(w1.x, o.x, w1.y, o.y, w1.z, o.z, w1.w), sop3t(w2.x, o.x, w2.y, o.y, w2.z, o.z, w2.w));
                    const v3 rd = V3(sop3(w0.x, d.x, w0.y, d.y, w0.z, d.z), sop3(w1.x, d.x, w1.y, d.y, w1.z, d.z), sop3(w2.x, d.x, w2.y, d.y, w2.z, d.z));
                    rs = ray_setup(rd);
                    br = box_ray(ro, rd);
                    nodeBase16 = bv.nodeOff16 + __float_as_uint(mk.x) * kNode16;
                    triBase16 = bv.triOff16 + __float_as_uint(mk.y) * kTri16;
                    PT_LOG(1u, x, T.y);
                    stack.push(G); stack.push(T); stack.push(make_uint2(kMarker, 0u));
                    curInst = x;
                    if (blas_single_leaf(ntri)) { G = root_node_group(true); T = root_tri_group(true, ntri); }
                    else {
                        if (STATS) stats->nodes++;
                        PT_LOG(3u, 0u, 0x80000000u);
                        const uint32_t hits = wide_node_hits(n0, n1, n2, n3, n4, br, tmin, ANYHIT ? tmax : h.t);
                        G = make_uint2(__float_as_uint(n1.x), (hits & 0xFF000000u) | (__float_as_uint(n0.w) >> 24));
                        T = make_uint2(__float_as_uint(n1.y), hits & 0x00FFFFFFu);
                        PT_LOG(4u, G.y, T.y);
                    }
                }
            } else {
                const uint32_t ta = triBase16 + i * kTri16;
                const f4v pa = blob.ld(ta), pb = blob.ld(ta + 1), pc = blob.ld(ta + 2);
                // closest hit: a second triangle of the group is fetched with the first (one latency for both, as in the streaming walk)
                const bool two = !ANYHIT && T.y != 0u;
                uint32_t i2 = 0;
                f4v qa = pa, qb = pb, qc = pc;
                if (two) {
                    i2 = T.x + (uint32_t)__builtin_ctz(T.y);
                    T.y &= T.y - 1u;
                    const uint32_t tb = triBase16 + i2 * kTri16;
                    qa = blob.ld(tb); qb = blob.ld(tb + 1); qc = blob.ld(tb + 2);
                }
                if (STATS) stats->tris++;
                PT_LOG(2u, curInst, i);
                float t, u, v;
                if (tri_test(rs, ro, V3(pa.x, pa.y, pa.z), V3(pb.x, pb.y, pb.z), V3(pc.x, pc.y, pc.z), t, u, v)) {
                    const uint32_t geom = __float_as_uint(pa.w), prim = __float_as_uint(pb.w);
                    if (!ANYHIT) {
                        commit_candidate(ac, __float_as_uint(pc.w), h, tmin, t, u, v, curInst, geom, prim, i);
                    } else if (t > tmin && t < tmax) {
                        const PtObjectData* od = &ac.objects[ac.instances[curInst].instanceID + geom];
                        TexCoords tc;
                        get_texture_coordinates(od, ac.heap, prim, u, v, tc);
                        if (is_opaque_visibility(od, ac.heap, ac.srgbLut, tc, *vis)) {
                            h.t = t; h.u = u; h.v = v; h.inst = curInst; h.geom = geom; h.prim = prim; h.slot = i;
                            break;
                        }
                    }
                }
                if (two) {
                    if (STATS) stats->tris++;
                    PT_LOG(2u, curInst, i2);
                    if (tri_test(rs, ro, V3(qa.x, qa.y, qa.z), V3(qb.x, qb.y, qb.z), V3(qc.x, qc.y, qc.z), t, u, v))
                        commit_candidate(ac, __float_as_uint(qc.w), h, tmin, t, u, v, curInst, __float_as_uint(qa.w), __float_as_uint(qb.w), i2);
                }
            }
        } else if (G.y > 0x00FFFFFFu) {
            PT_LOG(3u, G.x, G.y);
            visit_node<STATS, LDS>(blob, nodeBase16, br, tmin, ANYHIT ? tmax : h.t, G, T, stack, stats);
            PT_LOG(4u, G.y, T.y);
        } else if (stack.sp > 0) {
            G = stack.pop();
            PT_LOG(5u, G.x, G.y);
            if (G.x == kMarker && G.y == 0u) {                              // leave the BLAS: back to the world-space ray
                if (stack.overflow) break;                                  // a refused push left the parked state incomplete (unreachable: the builders bound the depth)
                T = stack.pop(); G = stack.pop();
                PT_LOG(6u, G.y, T.y);
                br = box_ray(o, d); nodeBase16 = bv.nodeOff16; curInst = ~0u;
            }
        } else {
            break;
        }
    }
    PT_LOG(7u, h.inst, h.slot);
    #undef PT_LOG
    stats->overflow += stack.overflow;
    if (!ANYHIT && h.inst != ~0u && !(h.t < tmax)) h.inst = ~0u;
    return h;
}

} // namespace pt
