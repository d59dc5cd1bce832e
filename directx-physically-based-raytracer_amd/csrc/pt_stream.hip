// pt_stream.hip -- the two kernels of a round for scenes whose traversal copy does not fit LDS (gfx950): k_extend_stream, the streaming
// traversal, and k_shade, the shading half; the device functions they share with the fused round kernel are in pt_shade.hpp.
// (Measured once the two kernels had a translation unit to themselves: the iterative-ILP machine scheduler, which costs the fused round
// kernel 19 %, changes neither of them -- k_shade 33.6 against 33.9 us per launch on C3 -- so the whole library is built with the default one.)
#include "pt_shade.hpp"

namespace pt {

// counts: [0..nsq) traced, [nsq..2*nsq) fresh (nsq = 1 << sqShift sub-queues)
// k_shade<false> wants 132 VGPRs, one more than four waves per SIMD allow; held to 128 it spills nothing and the fourth wave is worth
// +1.8 % on C3 and +0.7 % on C5 (the kernel waits on its 268 B per ray, not on issue slots)
template <bool TEXTURED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_shade(SceneView sv, FrameView fv, const FrameConstants* __restrict__ fc, PtTextures tx,
                                               PathQueue qin, PathQueue qout, float2* aux, uint32_t segCap, const uint32_t* countIn, uint32_t* countOut,
                                               const uint4* __restrict__ primary, BlobView bv, const uint4* __restrict__ recA, const uint32_t* __restrict__ recB, uint32_t sqShift,
                                               uint32_t sqBase, uint32_t sqCount)
{
    BlobReader<false> blob; blob.p = bv.base;
    ShadeTables tables; tables.recA = recA; tables.recB = recB;            // the frame's normal records (null: vertices are fetched at the hit)
    __shared__ uint32_t lds[32];                                  // two sets of reservation words, taken in turn: a fast wave may enter the next tile's reservation while a slow one still reads this tile's
    uint32_t emits = 0;
    const PtCamera& cam = fc->cam; const PtSceneData& sd = fc->sd; const PtGraphicsSettings& gs = fc->gs;
    const uint32_t nsq = 1u << sqShift, bq = blockIdx.x / sqCount, sq = sqBase + (blockIdx.x - bq * sqCount), nbq = gridDim.x / sqCount;   // this launch serves sub-queues [sqBase, sqBase + sqCount): one chain of the frame
    const uint32_t nT = countIn[sq], nF = countIn[nsq + sq];
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
            shade_traced<TEXTURED>(sv, GeometryFromBlob<false>{ blob, bv, tables }, sd, gs, tx, aux, p, hr, rd.w, V3(rd.x, rd.y, rd.z), toTraced, toFresh, newO, newD);
        }
        emit_tile(qout, seg, segCap, &countOut[sq], &countOut[nsq + sq], lds + ((emits++ & 1u) << 4), toTraced, toFresh, p, newO, newD);
    }
    for (uint32_t tile = bq; tile * 256u < nF; tile += nbq) {                // fresh entries
        const uint32_t local = tile * 256u + threadIdx.x;
        bool toTraced = false, toFresh = false;
        PathRegs p; v3 newO = V3(0, 0, 0), newD = V3(0, 0, 1);
        if (local < nF) {
            p = load_path(qin, seg + (segCap - 1u - local));
            shade_fresh(fv, cam, gs, tx, aux, primary, p, toTraced, toFresh, newO, newD);
        }
        emit_tile(qout, seg, segCap, &countOut[sq], &countOut[nsq + sq], lds + ((emits++ & 1u) << 4), toTraced, toFresh, p, newO, newD);
    }
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
                                               const uint32_t* count, uint32_t* cursor, DeviceCounters* counters, uint32_t sqBase, uint32_t sqCount)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t bq = blockIdx.x / sqCount, sq = sqBase + (blockIdx.x - bq * sqCount);
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


hipError_t launch_shade(Context& c, const SceneView& sv, const FrameView& fv, const PtTextures& tx, const PathQueue& qin, const PathQueue& qout, float2* aux,
                        uint32_t segCap, const uint32_t* countIn, uint32_t* countOut, uint32_t grid, hipStream_t stream, uint32_t sqBase, uint32_t sqCount)
{
    const bool rec = normal_records_usable(c);
    const uint4* recA = rec ? c.shadeRecA : nullptr; const uint32_t* recB = rec ? c.shadeRecB : nullptr;
    if (c.heapHasTextures) k_shade<true><<<grid, 256, 0, stream>>>(sv, fv, c.frameConstants, tx, qin, qout, aux, segCap, countIn, countOut, c.primaryRecords, c.blob, recA, recB, c.sqShift, sqBase, sqCount);
    else k_shade<false><<<grid, 256, 0, stream>>>(sv, fv, c.frameConstants, tx, qin, qout, aux, segCap, countIn, countOut, c.primaryRecords, c.blob, recA, recB, c.sqShift, sqBase, sqCount);
    return hipGetLastError();
}

hipError_t launch_extend_stream(Context& c, const AlphaContext& ac, const PathQueue& q, uint32_t segCap, const uint32_t* count, uint32_t* cursor,
                                uint32_t grid, bool stats, bool writeT, hipStream_t stream, uint32_t sqBase, uint32_t sqCount)
{
    // Fewer, longer-lived waves than the other kernels: a wave only keeps its lanes busy if it refills them many times, and with
    // 8192 waves a 600 k-ray round gives each wave one batch of 64 (lane use then is mean / longest walk of the batch). Alone on
    // the GPU 1024 blocks are best (C3 1.42 -> 1.44 Grays/s); with other frames in flight on other streams, which fill the SIMD
    // slots a small grid leaves, 512 (C3 2.06 -> 2.22, C5 1.62 -> 1.89; 256: 2.02 / 1.80).
    // (a chain of the frame gets its share of that grid: whole blocks per sub-queue)
    const uint32_t nsq = 1u << c.sqShift;
    const uint32_t whole = std::max(nsq, std::min(grid, c.framesInFlight > 1 ? kStreamGridShared : kStreamGridAlone));
    const uint32_t sgrid = std::max(1u, whole / nsq) * sqCount;
    #define PT_XS(S, W) k_extend_stream<S, W><<<sgrid, 256, kStreamLdsStack, stream>>>(c.blob, ac, q, segCap, count, cursor, c.counters, sqBase, sqCount)
    if (stats) { if (writeT) PT_XS(true, true); else PT_XS(true, false); } else { if (writeT) PT_XS(false, true); else PT_XS(false, false); }
    #undef PT_XS
    return hipGetLastError();
}

} // namespace pt
