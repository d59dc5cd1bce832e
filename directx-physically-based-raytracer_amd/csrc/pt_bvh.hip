// pt_bvh.hip -- on-device construction of the compressed wide BVH (gfx950).
//
// Replaces what the reference delegates to the D3D12 driver through RTXMU:
//   Scene::CreateAccelerationStructures   Source/Scene.ixx:286-380        (PREFER_FAST_TRACE for static meshes, :329)
//   CreateGeometryDesc / BuildTopLevelAccelerationStructure   Source/RaytracingHelpers.ixx:28-105
//   CommandList::BuildAccelerationStructures / UpdateAccelerationStructures   Source/CommandList.ixx:217-241
//
// Pipeline (all kernels on the context stream):
//   triangle packets + boxes + scene bounds  ->  63-bit Morton code of the box centre (21 bits per axis), primitive index as
//   the sort payload  ->  rocPRIM radix sort of (key, index)  ->  leaves of 1..3 consecutive triangles, padded boxes  ->
//   Karras 2012 binary hierarchy over the leaves (ties between equal keys broken by position)  ->  bottom-up boxes (one
//   atomic arrival counter per internal node)  ->  COLLAPSE to 8-wide nodes: starting from a binary node, the child with the
//   largest surface area is opened until eight children stand (the SAH-greedy collapse of Ylitie et al. 2017), children are
//   dealt to the slots that make "slot xor ray octant" a front-to-back order, boxes are quantised to 8 bits outward  ->
//   triangle packets scattered into node order (a node's leaf triangles are contiguous).
// The TLAS runs the same builder over instance boxes; its "triangles" are entries of the instance order list.
// A bottom level built with PT_BUILD_FLAG_ALLOW_UPDATE keeps the binary topology, the Morton order and the slot assignment,
// so that an update (skinned mesh moved) is a true refit: packets rewritten in place, boxes bottom-up, nodes re-quantised --
// no allocation, no sort, no synchronisation.
#include "pt_internal.hpp"

#include <cstring>
#include <string.h>
#include <rocprim/rocprim.hpp>

namespace pt {

// ---------------------------------------------------------------------------------------------
// order-preserving float <-> uint for atomic min/max of bounds
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t f2ord(float f) { uint32_t u = __float_as_uint(f); return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u); }
__device__ __forceinline__ float ord2f(uint32_t u) { return __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu)); }

__global__ void k_init_bounds(uint32_t* bounds)           // [0..2] = min, [3..5] = max (ordered encoding)
{
    if (threadIdx.x < 3) bounds[threadIdx.x] = 0xFFFFFFFFu;
    else if (threadIdx.x < 6) bounds[threadIdx.x] = 0u;
}

__global__ void k_empty_bounds(float* rootBounds)        // an empty tree: lo = +inf, hi = -inf
{
    if (threadIdx.x < 6) rootBounds[threadIdx.x] = threadIdx.x < 3 ? INFINITY : -INFINITY;
}

// Scene bounds of a launch: every wave reduces its lanes, the waves of a workgroup meet in LDS, ONE lane issues the six atomics. The six
// words share a cache line and atomics on one line run one after the other (~11 ns each on MI355X: 10 002 instances x 6 atomics were
// 0.69 ms of a 0.75 ms top-level build, one set per wave), so what counts is how few are issued. Every thread of the block must call it.
__device__ __forceinline__ void block_bounds_atomics(float lo[3], float hi[3], uint32_t* bounds, float* lds /* 6 * waves */)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, waves = (blockDim.x + 63u) >> 6;
    #pragma unroll
    for (int a = 0; a < 3; a++)
        for (int off = 32; off > 0; off >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], off)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off)); }
    if (lane == 0u) for (int a = 0; a < 3; a++) { lds[wave * 6u + a] = lo[a]; lds[wave * 6u + 3u + a] = hi[a]; }
    __syncthreads();
    if (threadIdx.x < 3u) {
        const uint32_t a = threadIdx.x;
        float l = INFINITY, h = -INFINITY;
        for (uint32_t w = 0; w < waves; w++) { l = fminf(l, lds[w * 6u + a]); h = fmaxf(h, lds[w * 6u + 3u + a]); }
        if (l <= h) { atomicMin(&bounds[a], f2ord(l)); atomicMax(&bounds[3u + a], f2ord(h)); }
    }
}

__device__ __forceinline__ uint32_t load_index(const void* ib, uint32_t stride, uint32_t i)
{
    return stride == 2 ? (uint32_t)((const uint16_t*)ib)[i] : ((const uint32_t*)ib)[i];
}

// one thread per triangle of one geometry: packet, box; block-reduced scene bounds. slotOfPrim == nullptr (build): the
// packet goes to tris[triOffset + p] (primitive order, scattered later); else (refit) straight to its final slot.
__global__ void k_tri_setup(const uint8_t* __restrict__ vb, uint32_t vstride, const void* __restrict__ ib, uint32_t istride,
                            uint32_t nprims, uint32_t triOffset, uint32_t geomIndex, uint32_t flags,
                            TriPacket* __restrict__ tris, const uint32_t* __restrict__ slotOfPrim,
                            float4* __restrict__ boxLo, float4* __restrict__ boxHi, uint32_t* __restrict__ bounds, uint4* __restrict__ idxOut)
{
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    if (p < nprims) {
        float v[3][3]; uint32_t vi[3];
        for (int k = 0; k < 3; k++) {
            uint32_t idx = load_index(ib, istride, 3 * p + k);
            vi[k] = idx;
            const float* pv = (const float*)(vb + (size_t)vstride * idx);
            v[k][0] = pv[0]; v[k][1] = pv[1]; v[k][2] = pv[2];
        }
        TriPacket t;
        t.a = make_float4(v[0][0], v[0][1], v[0][2], __uint_as_float(geomIndex));
        t.b = make_float4(v[1][0], v[1][1], v[1][2], __uint_as_float(p));
        t.c = make_float4(v[2][0], v[2][1], v[2][2], __uint_as_float(flags));
        tris[slotOfPrim ? slotOfPrim[triOffset + p] : triOffset + p] = t;
        if (idxOut) idxOut[triOffset + p] = make_uint4(vi[0], vi[1], vi[2], 0u);       // build only: a refit moves vertices, not indices
        for (int a = 0; a < 3; a++) {
            lo[a] = fminf(fminf(v[0][a], v[1][a]), v[2][a]);
            hi[a] = fmaxf(fmaxf(v[0][a], v[1][a]), v[2][a]);
        }
        boxLo[triOffset + p] = make_float4(lo[0], lo[1], lo[2], 0.0f);
        boxHi[triOffset + p] = make_float4(hi[0], hi[1], hi[2], 0.0f);
    }
    if (!bounds) return;                                   // block-uniform
    __shared__ float sBounds[6 * 4];
    block_bounds_atomics(lo, hi, bounds, sBounds);
}

__device__ __forceinline__ uint64_t expand21(uint32_t v)      // 21 bits -> every third bit of 63
{
    uint64_t x = v & 0x1FFFFFu;
    x = (x | x << 32) & 0x1F00000000FFFFull;
    x = (x | x << 16) & 0x1F0000FF0000FFull;
    x = (x | x << 8) & 0x100F00F00F00F00Full;
    x = (x | x << 4) & 0x10C30C30C30C30C3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

constexpr bool kTlasCubicCells = true;            // cubic cells for the top level as well, now that the large instances are filed apart (C5 +1 %; before that: -2 %)
constexpr float kTlasLargeFraction = 0.25f;       // an instance spanning this share of the scene along some axis is filed next to the root
__global__ void k_morton(const float4* __restrict__ boxLo, const float4* __restrict__ boxHi, uint32_t n,
                         const uint32_t* __restrict__ bounds, uint64_t* __restrict__ keys, uint32_t* __restrict__ index, bool cubic, bool largeFirst)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float mn[3], ext[3];
    for (int a = 0; a < 3; a++) { mn[a] = ord2f(bounds[a]); ext[a] = ord2f(bounds[3 + a]) - mn[a]; }
    float4 lo = boxLo[i], hi = boxHi[i];
    float c[3] = { 0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z) };
    uint32_t q[3];
    // One scale for the three axes (the largest extent): cells are cubes, so the bits of a short axis start to differ only where a
    // split along it is as good as one along the others; per-axis scaling cuts a flat mesh into slabs first (C3: 17.0 -> 13.7 node
    // visits per ray, longest walk 158 -> 119, +12 %). (For a top level the per-axis scale was the better one as long as its few huge
    // boxes -- ground, light -- were filed among the small ones: the short axis separated them. They have their own key bit now.)
    const float ext0[3] = { ext[0], ext[1], ext[2] };
    if (cubic) {
        const float emax = fmaxf(ext[0], fmaxf(ext[1], ext[2]));
        ext[0] = ext[1] = ext[2] = emax;
    }
    for (int a = 0; a < 3; a++) {
        float f = ext[a] > 0.0f ? (c[a] - mn[a]) / ext[a] : 0.0f;
        f = f == f ? f : 0.0f;                                                  // an empty (inverted) box has no centre
        q[a] = (uint32_t)fminf(fmaxf(f * 2097152.0f, 0.0f), 2097151.0f);
    }
    uint64_t key = (expand21(q[0]) << 2) | (expand21(q[1]) << 1) | expand21(q[2]);
    if (largeFirst) {
        // Top level: an instance that spans a quarter of the scene along some axis (a ground plane, a sky dome) would blow up the box of
        // every node on its path if it were filed among its small neighbours; the highest key bit files such instances in a subtree of
        // their own, next to the root (the coordinates give up their lowest bit for it). C5: 12.3 -> 11.2 node visits per ray, +6 %; the threshold
        // is not sensitive (1/16 .. 1/2 measured). The same flag on the TRIANGLES of a bottom level loses 5-9 % on C3: there the large ones are many.
        bool large = false;
        for (int a = 0; a < 3; a++) large = large || (ext0[a] > 0.0f && (a == 0 ? hi.x - lo.x : a == 1 ? hi.y - lo.y : hi.z - lo.z) > kTlasLargeFraction * ext0[a]);
        key = (key >> 3) | (large ? 1ull << 62 : 0ull);
    }
    keys[i] = key;
    index[i] = i;
}

// conservative padding so that a hit reported by the watertight triangle test is never culled by
// fp32 rounding in the slab test
__device__ __forceinline__ void pad_box(float4& lo, float4& hi)
{
    float l[3] = { lo.x, lo.y, lo.z }, h[3] = { hi.x, hi.y, hi.z };
    for (int a = 0; a < 3; a++) {
        if (l[a] <= h[a]) {
            float e = 1e-5f * fmaxf(fabsf(l[a]), fabsf(h[a])) + 1e-6f * (h[a] - l[a]) + 1e-30f;
            l[a] -= e; h[a] += e;
        }
    }
    lo = make_float4(l[0], l[1], l[2], 0.0f); hi = make_float4(h[0], h[1], h[2], 0.0f);
}

// Leaves: leafSize consecutive entries of the sorted order (BLAS: triangles, TLAS: one instance); padded union box.
__global__ void k_leaves(const uint64_t* __restrict__ keysSorted, const uint32_t* __restrict__ indexSorted, const float4* __restrict__ boxLo,
                         const float4* __restrict__ boxHi, uint32_t nitems, uint32_t nleaves, uint32_t leafSize,
                         uint64_t* __restrict__ leafKeys, float4* __restrict__ leafLo, float4* __restrict__ leafHi)
{
    uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nleaves) return;
    uint32_t first = l * leafSize, count = min(leafSize, nitems - first);
    float4 lo = make_float4(INFINITY, INFINITY, INFINITY, 0.0f), hi = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.0f);
    for (uint32_t i = 0; i < count; i++) {
        const uint32_t s = indexSorted[first + i];
        const float4 a = boxLo[s], b = boxHi[s];
        lo.x = fminf(lo.x, a.x); lo.y = fminf(lo.y, a.y); lo.z = fminf(lo.z, a.z);
        hi.x = fmaxf(hi.x, b.x); hi.y = fmaxf(hi.y, b.y); hi.z = fmaxf(hi.z, b.z);
    }
    pad_box(lo, hi);
    if (leafKeys) leafKeys[l] = keysSorted[first];
    leafLo[l] = lo; leafHi[l] = hi;
}


// ---------------------------------------------------------------------------------------------
// World box of an instance. The 8 corners of the BLAS's root box pushed through ObjectToWorld bound it, but loosely for a rotated
// instance of anything rounder than a box; the union over the boxes two levels down (the children of the root node and of its
// internal children, as the traversal itself decodes them: conservative by construction) hugs the object. Both are bounds, so
// their intersection is one. A mesh of at most kExactBoxTriangles triangles is bounded by its transformed vertices instead.
// C5 (10 002 rotated, squashed icospheres): fewer rays enter an instance they then miss.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool wide_child_box(const WideNode& nd, int s, float lo[3], float hi[3])
{
    const uint32_t* ql[3] = { nd.qlox, nd.qloy, nd.qloz }; const uint32_t* qh[3] = { nd.qhix, nd.qhiy, nd.qhiz };
    for (int a = 0; a < 3; a++) {
        const uint32_t l = (ql[a][s >> 2] >> (8 * (s & 3))) & 0xFFu, h = (qh[a][s >> 2] >> (8 * (s & 3))) & 0xFFu;
        if (l > h) return false;                                             // an empty slot is stored as [255, 0]
        const float scale = __uint_as_float(((nd.expImask >> (8 * a)) & 0xFFu) << 23);
        lo[a] = nd.origin[a] + (float)l * scale; hi[a] = nd.origin[a] + (float)h * scale;
    }
    return true;
}
__device__ __forceinline__ void grow_by_transformed_box(const float* M, const float blo[3], const float bhi[3], float lo[3], float hi[3])
{
    for (int c = 0; c < 8; c++) {
        const float x = (c & 1) ? bhi[0] : blo[0], y = (c & 2) ? bhi[1] : blo[1], z = (c & 4) ? bhi[2] : blo[2];
        for (int a = 0; a < 3; a++) {
            const float w = M[4 * a] * x + M[4 * a + 1] * y + M[4 * a + 2] * z + M[4 * a + 3];
            lo[a] = fminf(lo[a], w); hi[a] = fmaxf(hi[a], w);
        }
    }
}
constexpr uint32_t kExactBoxTriangles = 1024;
// The 64 lanes of a wave work on ONE instance: the triangles of the exact path are dealt to the lanes and the result reduced; the
// node path is evaluated by every lane alike (a handful of boxes). All lanes return the same box.
__device__ void instance_world_box(const InstanceRecord& ir, const float* b, float lo[3], float hi[3])
{
    for (int a = 0; a < 3; a++) { lo[a] = INFINITY; hi[a] = -INFINITY; }
    if (!(b[0] <= b[3])) return;                                             // empty BLAS
    float M[12];                                                              // in registers: the loops below read it per vertex
    #pragma unroll
    for (int k = 0; k < 12; k++) M[k] = ir.objectToWorld[k];
    const float rlo[3] = { b[0], b[1], b[2] }, rhi[3] = { b[3], b[4], b[5] };
    grow_by_transformed_box(M, rlo, rhi, lo, hi);
    float tlo[3] = { INFINITY, INFINITY, INFINITY }, thi[3] = { -INFINITY, -INFINITY, -INFINITY };
    const TriPacket* tris = ir.tris; const uint32_t triCount = ir.triCount;
    if (tris && triCount <= kExactBoxTriangles) {                            // a small mesh: the vertices themselves (the tightest box there is)
        for (uint32_t t = threadIdx.x & 63u; t < triCount; t += 64u) {
            const TriPacket tp = tris[t];
            const float4 v[3] = { tp.a, tp.b, tp.c };
            #pragma unroll
            for (int k = 0; k < 3; k++)
                #pragma unroll
                for (int a = 0; a < 3; a++) {
                    const float w = M[4 * a] * v[k].x + M[4 * a + 1] * v[k].y + M[4 * a + 2] * v[k].z + M[4 * a + 3];
                    tlo[a] = fminf(tlo[a], w); thi[a] = fmaxf(thi[a], w);
                }
        }
        for (int a = 0; a < 3; a++)
            for (int off = 32; off > 0; off >>= 1) { tlo[a] = fminf(tlo[a], __shfl_xor(tlo[a], off)); thi[a] = fmaxf(thi[a], __shfl_xor(thi[a], off)); }
        for (int a = 0; a < 3; a++) { lo[a] = fmaxf(lo[a], tlo[a]); hi[a] = fminf(hi[a], thi[a]); }
        return;
    }
    if (!ir.nodes || blas_single_leaf(ir.triCount)) return;                  // a BLAS of one leaf has no node
    // two levels down: lane (s, t) = (lane / 8, lane % 8) takes grandchild t of child s -- or, for a leaf slot s, its lane t = 0 the child
    // box itself -- and the wave reduces. (Every lane walking all 64 boxes by itself took 70 us for the one large mesh of a scene.)
    {
        const WideNode* nodes = ir.nodes;
        const uint32_t lane = threadIdx.x & 63u, sl = lane >> 3, tl = lane & 7u;
        const WideNode root = nodes[0];
        const uint32_t imask = root.expImask >> 24;
        float clo[3], chi[3];
        if (wide_child_box(root, (int)sl, clo, chi)) {
            if (imask & (1u << sl)) {
                const WideNode child = nodes[root.childBase + __popc(imask & ((1u << sl) - 1u))];
                float glo[3], ghi[3];
                if (wide_child_box(child, (int)tl, glo, ghi)) {
                    for (int a = 0; a < 3; a++) { glo[a] = fmaxf(glo[a], clo[a]); ghi[a] = fminf(ghi[a], chi[a]); }    // inside its parent's box as well
                    if (glo[0] <= ghi[0] && glo[1] <= ghi[1] && glo[2] <= ghi[2]) grow_by_transformed_box(M, glo, ghi, tlo, thi);
                }
            } else if (tl == 0u) grow_by_transformed_box(M, clo, chi, tlo, thi);
        }
        for (int a = 0; a < 3; a++)
            for (int off = 32; off > 0; off >>= 1) { tlo[a] = fminf(tlo[a], __shfl_xor(tlo[a], off)); thi[a] = fmaxf(thi[a], __shfl_xor(thi[a], off)); }
    }
    for (int a = 0; a < 3; a++) { lo[a] = fmaxf(lo[a], tlo[a]); hi[a] = fminf(hi[a], thi[a]); }
    if (!(lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2])) for (int a = 0; a < 3; a++) { lo[a] = INFINITY; hi[a] = -INFINITY; }
}

// TLAS items: the world box of each instance. A wave works on one instance at a time and strides over the array; the scene bounds are
// accumulated per wave and leave the workgroup as six atomics.
__global__ __launch_bounds__(256) void k_instance_boxes(const InstanceRecord* __restrict__ inst, const float* const* __restrict__ blasBounds, uint32_t n,
                                 float4* __restrict__ boxLo, float4* __restrict__ boxHi, uint32_t* __restrict__ bounds)
{
    __shared__ float sBounds[6 * 4];
    float slo[3] = { INFINITY, INFINITY, INFINITY }, shi[3] = { -INFINITY, -INFINITY, -INFINITY };
    const uint32_t waveStride = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; i < n; i += waveStride) {     // wave-uniform
        float lo[3], hi[3];
        instance_world_box(inst[i], blasBounds[i], lo, hi);                   // blasBounds: lo.xyz hi.xyz of the BLAS root
        if ((threadIdx.x & 63u) == 0u) {
            boxLo[i] = make_float4(lo[0], lo[1], lo[2], 0.0f);
            boxHi[i] = make_float4(hi[0], hi[1], hi[2], 0.0f);
        }
        for (int a = 0; a < 3; a++) if (lo[a] <= hi[a]) { slo[a] = fminf(slo[a], lo[a]); shi[a] = fmaxf(shi[a], hi[a]); }
    }
    block_bounds_atomics(slo, shi, bounds, sBounds);
}

// ---------------------------------------------------------------------------------------------
// Karras, "Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees", HPG 2012
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int delta(const uint64_t* keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    const uint64_t x = keys[i] ^ keys[j];
    return x ? __clzll((long long)x) : 64 + __clz(i ^ j);      // equal keys: the positions break the tie (section 4 of the paper)
}

// internal node i in [0, n-1): children + parent links. child < 0 means leaf ~(leaf index).
__global__ void k_karras(const uint64_t* __restrict__ keys, int n, int2* __restrict__ children, int* __restrict__ parentInternal, int* __restrict__ parentLeaf)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2, div = 2; ; ) {
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
        div *= 2; t = (l + div - 1) / div;
    }
    int gamma = i + s * d + min(d, 0);
    int lo = min(i, j), hi = max(i, j);
    int left = (lo == gamma) ? ~gamma : gamma;
    int right = (hi == gamma + 1) ? ~(gamma + 1) : (gamma + 1);
    children[i] = make_int2(left, right);
    if (left < 0) parentLeaf[~left] = i; else parentInternal[left] = i;
    if (right < 0) parentLeaf[~right] = i; else parentInternal[right] = i;
    if (i == 0) parentInternal[0] = -1;
}

// Boxes of the binary nodes and, with dp != nullptr, the cost tables of the optimal collapse: one thread per leaf climbs
// towards the root; the second arrival at a node owns it. rootBounds: lo.xyz hi.xyz of the whole tree.
//
// Collapse cost model (Ylitie, Karras, Laine 2017, section 3.1). C(n, i) = cheapest SAH cost of the subtree of binary node n
// represented as a forest of at most i roots, each a leaf (<= maxLeafItems items) or a wide node:
//   C(n, 1) = min(C_leaf(n), C_internal(n)),  C_leaf(n) = A_n * P_n * c_item,  C_internal(n) = A_n * c_node + C_distribute(n, 8)
//   C(n, i) = min(C_distribute(n, i), C(n, i - 1)),   C_distribute(n, j) = min over 0 < k < j of C(left, k) + C(right, j - k)
// The decisions are kept (which k, leaf or node) and the collapse kernel follows them top-down.
struct alignas(16) DpNode { float cost[7]; uint8_t split[9]; uint8_t isLeaf; uint8_t _pad[2]; int2 children; };      // split[j], j = 2..8: k of the best distribution, 0 = "take C(n, j - 1)"
static_assert(sizeof(DpNode) == 48, "layout");
constexpr float kCostNode = 1.0f;
constexpr float kCostTriangle = 0.6f;         // a triangle test against a node visit. By instruction counts (~97 against ~225) 0.4; measured 0.3 -> 0.6: C3 +3 %, C5 +3 %
                                                     // (a triangle is also a step of the walk, and steps are what the streaming traversal pays for), flat from 0.6 to 0.9
constexpr float kCostInstance = 1.0f;        // entering an instance: look-up, ray transform, BLAS root

__device__ __forceinline__ float half_area(float4 lo, float4 hi)
{
    const float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
    const float a = dx * dy + dy * dz + dz * dx;
    return a == a && dx >= 0.0f ? a : 0.0f;               // empty boxes (inverted, infinite) never attract the collapse
}

// Hand-off between the threads of ONE launch (a child's box and cost table, written by the thread that finished the child, read by
// the thread that finishes the parent, on any CU of any XCD). The XCDs' L2s are not coherent with each other and a CU's L1 is never
// refreshed, so plain stores + __threadfence() -- an L2 write-back and an L1 invalidate per thread and level, ~3.5 us each and worse
// under load (MI355X_MICROARCH.md, inter-workgroup visibility) -- made this kernel the slowest of the build: 1.92 ms for 250 k
// triangles. Instead every handed-off byte is stored and loaded with sc1 (write-through / L1 bypass, 16 bytes per access), the
// storing lane waits for its stores (vmcnt(0)) before its agent-scope arrival atomic, and the lane whose atomic returns 1 (the second
// arrival) is the one that loads: the guide's measured "sc1 stores -> wait -> atomic add -> the adder that came last loads sc1" form.
typedef float refit_f4 __attribute__((ext_vector_type(4)));
// (s_nop 1: a store of more than 8 bytes reads its data registers after issue; the compiler keeps the required wait state for its own stores
// but cannot see into inline assembly, and did place a write of the data registers right behind this one.)
__device__ __forceinline__ void store16_sc1(void* p, refit_f4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ refit_f4 load16_sc1(const void* p)
{
    refit_f4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ float4 as_float4(refit_f4 v) { return make_float4(v.x, v.y, v.z, v.w); }

__global__ void k_refit(int nleaves, const float4* __restrict__ leafLo, const float4* __restrict__ leafHi,
                        const int2* __restrict__ children, const int* __restrict__ parentInternal, const int* __restrict__ parentLeaf,
                        float4* nodeLo, float4* nodeHi, uint32_t* arrival, float* rootBounds,
                        DpNode* dp, uint32_t nitems, uint32_t leafSize, uint32_t maxLeafItems, float costItem)
{
    int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nleaves) return;
    if (nleaves == 1) {
        const float4 lo = leafLo[0], hi = leafHi[0];
        rootBounds[0] = lo.x; rootBounds[1] = lo.y; rootBounds[2] = lo.z; rootBounds[3] = hi.x; rootBounds[4] = hi.y; rootBounds[5] = hi.z;
        return;
    }
    int cur = parentLeaf[l];
    while (cur >= 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // what this lane stored for the node below `cur` has left the CU
        if (atomicAdd(&arrival[cur], 1u) == 0u) return;      // first arrival: the sibling will finish the node
        arrival[cur] = 0u;                                  // ready for the next refit (a later launch)
        const int2 ch = children[cur];
        // leaf boxes come from an earlier launch (plain loads); an internal child's box was stored, sc1, by the thread that arrived first
        const float4 lo0 = ch.x < 0 ? leafLo[~ch.x] : as_float4(load16_sc1(&nodeLo[ch.x])), hi0 = ch.x < 0 ? leafHi[~ch.x] : as_float4(load16_sc1(&nodeHi[ch.x]));
        const float4 lo1 = ch.y < 0 ? leafLo[~ch.y] : as_float4(load16_sc1(&nodeLo[ch.y])), hi1 = ch.y < 0 ? leafHi[~ch.y] : as_float4(load16_sc1(&nodeHi[ch.y]));
        const float4 lo = make_float4(fminf(lo0.x, lo1.x), fminf(lo0.y, lo1.y), fminf(lo0.z, lo1.z), 0.0f);
        const float4 hi = make_float4(fmaxf(hi0.x, hi1.x), fmaxf(hi0.y, hi1.y), fmaxf(hi0.z, hi1.z), 0.0f);
        // items below a node travel up with its box (the w of the low corner): the tree may have any topology (a subtree's leaves need not
        // be a range of the sorted order)
        const uint32_t cnt0 = ch.x < 0 ? min(leafSize, nitems - (uint32_t)(~ch.x) * leafSize) : __float_as_uint(lo0.w);
        const uint32_t cnt1 = ch.y < 0 ? min(leafSize, nitems - (uint32_t)(~ch.y) * leafSize) : __float_as_uint(lo1.w);
        const uint32_t P = cnt0 + cnt1;
        store16_sc1(&nodeLo[cur], (refit_f4){ lo.x, lo.y, lo.z, __uint_as_float(P) });
        store16_sc1(&nodeHi[cur], (refit_f4){ hi.x, hi.y, hi.z, 0.0f });
        if (dp) {
            float c[2][7];
            #pragma unroll
            for (int side = 0; side < 2; side++) {
                const int r = side ? ch.y : ch.x;
                if (r < 0) {
                    const float v = half_area(side ? lo1 : lo0, side ? hi1 : hi0) * (float)(side ? cnt1 : cnt0) * costItem;
                    #pragma unroll
                    for (int i = 0; i < 7; i++) c[side][i] = v;
                } else {
                    const refit_f4 a = load16_sc1(&dp[r]), b2 = load16_sc1((const char*)&dp[r] + 16);       // cost[0..6]
                    c[side][0] = a.x; c[side][1] = a.y; c[side][2] = a.z; c[side][3] = a.w; c[side][4] = b2.x; c[side][5] = b2.y; c[side][6] = b2.z;
                }
            }
            DpNode d;
            float dist[9];
            #pragma unroll
            for (int j = 2; j <= 8; j++) {
                float best = INFINITY; int bk = 1;
                #pragma unroll
                for (int k = 1; k < j; k++) {
                    if (k > 7 || j - k > 7) continue;
                    const float v = c[0][k - 1] + c[1][j - k - 1];
                    if (v < best) { best = v; bk = k; }
                }
                dist[j] = best; d.split[j] = (uint8_t)bk;
            }
            const float A = half_area(lo, hi);
            const float cLeaf = P <= maxLeafItems ? A * (float)P * costItem : INFINITY;
            const float cInternal = A * kCostNode + dist[8];
            d.isLeaf = cLeaf <= cInternal; d.split[0] = d.split[1] = 0;
            d._pad[0] = d._pad[1] = 0; d.children = ch;        // the node's children ride along: the collapse reads ONE record per node it opens
            d.cost[0] = fminf(cLeaf, cInternal);
            #pragma unroll
            for (int i = 2; i <= 7; i++) {
                if (dist[i] <= d.cost[i - 2]) d.cost[i - 1] = dist[i];          // ties go to MORE roots: zero-area subtrees (all costs 0) still fill eight slots
                else { d.cost[i - 1] = d.cost[i - 2]; d.split[i] = 0; }
            }
            const refit_f4* dv = (const refit_f4*)&d;
            store16_sc1(&dp[cur], dv[0]); store16_sc1((char*)&dp[cur] + 16, dv[1]); store16_sc1((char*)&dp[cur] + 32, dv[2]);
        }
        if (cur == 0) {
            rootBounds[0] = lo.x; rootBounds[1] = lo.y; rootBounds[2] = lo.z; rootBounds[3] = hi.x; rootBounds[4] = hi.y; rootBounds[5] = hi.z;
        }
        cur = parentInternal[cur];
    }
}

// ---------------------------------------------------------------------------------------------
// collapse to compressed 8-wide nodes
// ---------------------------------------------------------------------------------------------
constexpr int kEmptyRef = 0x7FFFFFFF;
constexpr uint32_t kMaxWideLevels = 60;                  // a deeper tree cannot be traversed with kStackSize entries anyway
constexpr uint32_t kSingleGroupCollapseLeaves = 4096;    // up to here one workgroup collapses the tree in one launch
constexpr uint32_t kLevelLaunches = 32;                  // per-level launches of the large-tree collapse: 2 * depth + 4 <= kStackSize bounds a traversable tree at 30 levels

__device__ __forceinline__ void ref_box(int r, const float4* leafLo, const float4* leafHi, const float4* nodeLo, const float4* nodeHi, float4& lo, float4& hi)
{
    if (r < 0) { lo = leafLo[~r]; hi = leafHi[~r]; } else { lo = nodeLo[r]; hi = nodeHi[r]; }
}

// Origin, per-axis power-of-two scale and the 8-bit child boxes of a node (paper, section 3.3): e = ceil(log2(extent / 255)),
// q_lo = floor((lo - p) / 2^e), q_hi = ceil((hi - p) / 2^e), each checked against its decoded value so that the stored box
// contains the child box whatever the roundings were. Slots with refs[s] == kEmptyRef (or an empty child box) get lo > hi.
__device__ void quantise_node(WideNode& n, float4 nlo, float4 nhi, const float4* clo, const float4* chi, const int* refs)
{
    const float p[3] = { nlo.x, nlo.y, nlo.z }, ph[3] = { nhi.x, nhi.y, nhi.z };
    uint32_t q[2][3][8];
    uint32_t eb[3];
    const bool valid = p[0] <= ph[0] && p[1] <= ph[1] && p[2] <= ph[2] && isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2])
                       && isfinite(ph[0]) && isfinite(ph[1]) && isfinite(ph[2]);
    for (int a = 0; a < 3; a++) {
        const float ext = valid ? ph[a] - p[a] : 0.0f;
        int e;
        (void)frexpf(ext * (1.0f / 255.0f), &e);             // ext / 255 = m * 2^e with m in [0.5, 1): 2^e covers it
        if (!(ext > 0.0f)) e = -126;
        while (true) {
            if (e < -126) e = -126;
            const float scale = ldexpf(1.0f, e), inv = ldexpf(1.0f, -e);
            bool ok = true;
            for (int s = 0; s < 8; s++) {
                const float cl = a == 0 ? clo[s].x : (a == 1 ? clo[s].y : clo[s].z), ch = a == 0 ? chi[s].x : (a == 1 ? chi[s].y : chi[s].z);
                if (!valid || refs[s] == kEmptyRef || !(cl <= ch)) { q[0][a][s] = 255u; q[1][a][s] = 0u; continue; }
                float fl = floorf((cl - p[a]) * inv), fh = ceilf((ch - p[a]) * inv);
                fl = fminf(fmaxf(fl, 0.0f), 255.0f); fh = fmaxf(fh, 0.0f);
                while (fl > 0.0f && p[a] + fl * scale > cl) fl -= 1.0f;
                while (fh <= 255.0f && p[a] + fh * scale < ch) fh += 1.0f;
                if (fh > 255.0f) { ok = false; break; }
                q[0][a][s] = (uint32_t)fl; q[1][a][s] = (uint32_t)fh;
            }
            if (ok || e >= 127) break;
            e++;
        }
        eb[a] = (uint32_t)(e + 127);
    }
    n.origin[0] = valid ? p[0] : 0.0f; n.origin[1] = valid ? p[1] : 0.0f; n.origin[2] = valid ? p[2] : 0.0f;
    n.expImask = (n.expImask & 0xFF000000u) | eb[0] | (eb[1] << 8) | (eb[2] << 16);
    uint32_t* dst[2][3] = { { n.qlox, n.qloy, n.qloz }, { n.qhix, n.qhiy, n.qhiz } };
    for (int k = 0; k < 2; k++)
        for (int a = 0; a < 3; a++)
            for (int h = 0; h < 2; h++)
                dst[k][a][h] = q[k][a][4 * h] | (q[k][a][4 * h + 1] << 8) | (q[k][a][4 * h + 2] << 16) | (q[k][a][4 * h + 3] << 24);
}

struct CollapseArgs {
    const DpNode* dp;
    const int2* children; const float4* nodeLo; const float4* nodeHi; const float4* leafLo; const float4* leafHi;
    uint32_t nleaves, nitems, leafSize;
    WideNode* nodes; uint32_t nodeCapacity;
    int* binaryRootOf;                   // [nodeCapacity] the binary node a wide node was grown from (also the work queue)
    int* slotRefs;                       // [nodeCapacity * 8] binary ref held by each slot (refit re-quantises from these)
    uint32_t* leafDst;                   // [nleaves] first item of each leaf in node order
    WideHeader* header;
};

// Eight consecutive lanes share one wide node (a wave holds eight groups, all of them in the same code).
__device__ __forceinline__ uint32_t group_ballot(bool p) { return (uint32_t)(__builtin_amdgcn_ballot_w64(p) >> (threadIdx.x & 56u)) & 0xFFu; }
__device__ __forceinline__ uint32_t group_or(uint32_t v) { v |= __shfl_xor(v, 1, 8); v |= __shfl_xor(v, 2, 8); v |= __shfl_xor(v, 4, 8); return v; }
__device__ __forceinline__ uint32_t nth_set_bit(uint32_t m, uint32_t r)      // position of the r-th set bit of an 8-bit mask, 8 if there is none
{
    #pragma unroll
    for (uint32_t i = 0; i < 7; i++) if (i < r) m &= m - 1u;
    return m ? (uint32_t)__ffs(m) - 1u : 8u;
}

// One wide node per GROUP of eight lanes: w is its index, binaryRootOf[w] the binary node it grows from. The group follows the cost tables
// down to (at most) eight children, one per lane, deals them to octant-ordered slots, reserves the internal children (consecutive node
// indices) and the leaf items from the two counters, quantises, writes the node and queues the children (binaryRootOf of their indices).
// One thread did all of this until round 3, ~45 us of dependent loads and ~3000 serial instructions per node -- and a top-level rebuild walks
// its tree level by level, so a dynamic frame waited 4 x 45 us. Here the frontier of the cost-table walk is expanded by all its entries at
// once (<= 4 rounds of loads for a balanced choice instead of 15 loads in a row), every child's box, leaf walk and quantisation runs in its
// own lane, and the greedy slot assignment is eight lane-local scans + a three-step group minimum per round. Every lane of the wave must
// call it (inactive groups pass active = false): the group operations are wave operations.
// The result is the one the serial form produced: children are ranked in left-to-right order of the binary tree (the rank breaks cost ties,
// as the child index did), and the float expressions are the same.
__device__ void collapse_node(const CollapseArgs& A, uint32_t w, bool active, uint32_t* nodesCtr, uint32_t* itemsCtr, uint32_t* errorFlag)
{
    const uint32_t sub = threadIdx.x & 7u;
    const int root = active ? A.binaryRootOf[w] : 0;
    // 1. the children the cost tables chose: (binary node, roots it may use) entries, one per lane; an entry with more than one root to
    // spend splits in two as its table says, the right half moves to a free lane. key = the entry's path (bit 7 - d set: right at depth d),
    // so that ascending keys are the left-to-right order.
    int ref = root, budget = (active && sub == 0) ? 8 : 0;
    uint32_t key = 0, depth = 0;
    bool settled = false, leafChild = false;
    for (uint32_t pass = 0; pass < 9; pass++) {
        const bool open = budget > 0 && !settled;
        if (!__builtin_amdgcn_ballot_w64(open)) break;
        bool splits = false; int otherRef = 0, otherBudget = 0;
        if (open) {
            if (ref < 0) { settled = true; leafChild = true; }
            else {
                const uint4* rec = (const uint4*)&A.dp[ref];           // split[] sits in bytes 28..36, isLeaf in 37, the children in 40..47
                const uint4 r1 = rec[1], r2 = rec[2];
                auto split = [&](int j) -> int { return (int)((j < 4 ? r1.w >> (8 * j) : j < 8 ? r2.x >> (8 * (j - 4)) : r2.y) & 0xFFu); };
                if (pass) while (budget > 1 && split(budget) == 0) budget--;        // "take C(m, budget - 1)"; the root spends its eight as split[8] says
                if (budget == 1) { settled = true; leafChild = ((r2.y >> 8) & 0xFFu) != 0u; }
                else {
                    const int k = split(budget);
                    splits = true; otherRef = (int)r2.w; otherBudget = budget - k;
                    ref = (int)r2.z; budget = k;
                }
            }
        }
        const uint32_t splitMask = group_ballot(splits), freeMask = group_ballot(budget == 0);
        const uint32_t from = nth_set_bit(splitMask, __popc(freeMask & ((1u << sub) - 1u)));      // the r-th free lane takes from the r-th splitting one
        const uint32_t otherTag = (key | (1u << (7u - depth))) | ((depth + 1u) << 8);
        const int gotRef = __shfl(otherRef, (int)(from & 7u), 8), gotBudget = __shfl(otherBudget, (int)(from & 7u), 8);
        const uint32_t gotTag = __shfl(otherTag, (int)(from & 7u), 8);
        if (splits) depth++;
        if (budget == 0 && from < 8u) { ref = gotRef; budget = gotBudget; key = gotTag & 0xFFu; depth = gotTag >> 8; }
    }
    const bool occupied = budget > 0;
    if (occupied && !settled) *errorFlag = 1u;                               // cannot happen: eight roots are spent after seven splits
    uint32_t rank = 0;
    #pragma unroll
    for (int j = 0; j < 8; j++) { const uint32_t kj = __shfl(occupied ? key : 0x100u, j, 8); rank += (occupied && kj < key) ? 1u : 0u; }
    const uint32_t n = __popc(group_ballot(occupied));

    // 2. boxes
    float4 lo = make_float4(0, 0, 0, 0), hi = lo, nlo, nhi;
    if (occupied) ref_box(ref, A.leafLo, A.leafHi, A.nodeLo, A.nodeHi, lo, hi);
    ref_box(root, A.leafLo, A.leafHi, A.nodeLo, A.nodeHi, nlo, nhi);

    // 3. slots: child c goes where "slot xor octant" visits it in front-to-back order for rays of that octant (paper 3.2, greedy instead of
    // the auction: repeatedly the cheapest unassigned (child, slot) pair, the first such pair in (child, slot) order)
    float dX = 0.0f, dY = 0.0f, dZ = 0.0f;
    if (occupied) {
        const float cx = 0.5f * (nlo.x + nhi.x), cy = 0.5f * (nlo.y + nhi.y), cz = 0.5f * (nlo.z + nhi.z);
        dX = 0.5f * (lo.x + hi.x) - cx; dY = 0.5f * (lo.y + hi.y) - cy; dZ = 0.5f * (lo.z + hi.z) - cz;
        if (!(dX == dX)) dX = 0.0f;
        if (!(dY == dY)) dY = 0.0f;
        if (!(dZ == dZ)) dZ = 0.0f;
    }
    uint32_t freeSlot = 0xFFu, mySlot = 0; bool assigned = false;
    for (uint32_t round = 0; round < 8; round++) {
        if (!__builtin_amdgcn_ballot_w64(round < n)) break;
        float bcost = INFINITY; int bs = -1;
        if (occupied && !assigned) {
            #pragma unroll
            for (int s = 0; s < 8; s++) {
                if (!((freeSlot >> s) & 1u)) continue;
                const float cost = ((s & 4) ? -dX : dX) + ((s & 2) ? -dY : dY) + ((s & 1) ? -dZ : dZ);
                if (cost < bcost || bs < 0) { bcost = cost; bs = s; }
            }
        }
        uint32_t tag = bs >= 0 ? (rank << 8) | ((uint32_t)bs << 4) | sub : 0xFFFFFFFFu;
        #pragma unroll
        for (int x = 1; x < 8; x <<= 1) {
            const float oc = __shfl_xor(bcost, x, 8); const uint32_t ot = __shfl_xor(tag, x, 8);
            if (ot != 0xFFFFFFFFu && (tag == 0xFFFFFFFFu || oc < bcost || (oc == bcost && ot < tag))) { bcost = oc; tag = ot; }
        }
        if (tag != 0xFFFFFFFFu) {
            const uint32_t ws = (tag >> 4) & 0xFu;
            if ((tag & 0xFu) == sub) { assigned = true; mySlot = ws; }
            freeSlot &= ~(1u << ws);
        }
    }
    if (!occupied) mySlot = nth_set_bit(freeSlot, __popc(group_ballot(!occupied) & ((1u << sub) - 1u))) & 7u;     // the empty slots go to the idle lanes

    // 4. a leaf child is a leaf of the binary tree or a whole subtree of at most kMaxLeafTris items: its (at most three) binary leaves are found
    // by walking it -- they need not be neighbours in the sorted order
    const bool internalChild = occupied && !leafChild;
    int leafId[kMaxLeafTris] = { 0, 0, 0 }; uint32_t nl = 0, cnt = 0;
    if (occupied && leafChild) {
        auto add = [&](int m) {
            if (m >= 0) { *errorFlag = 1u; return; }                        // deeper than three leaves can be
            if (nl == 0) leafId[0] = ~m; else if (nl == 1) leafId[1] = ~m; else if (nl == 2) leafId[2] = ~m; else *errorFlag = 1u;
            nl++; cnt += min(A.leafSize, A.nitems - (uint32_t)(~m) * A.leafSize);
        };
        if (ref < 0) add(ref);
        else {
            const int2 c = A.children[ref];
            int2 cx = make_int2(0, 0), cy = make_int2(0, 0);
            if (c.x >= 0) cx = A.children[c.x];
            if (c.y >= 0) cy = A.children[c.y];
            if (c.x < 0) add(c.x); else { add(cx.x); add(cx.y); }
            if (c.y < 0) add(c.y); else { add(cy.x); add(cy.y); }
        }
        if (nl > kMaxLeafTris) nl = kMaxLeafTris;
        if (cnt > kMaxLeafTris) { *errorFlag = 1u; cnt = kMaxLeafTris; }
    }
    // what the group needs to know about every slot, in two words: internal (8 bits) | items (8 x 2 bits) | occupied (8 bits), and the lane that holds each slot
    const uint32_t bySlot = group_or((internalChild ? 1u << mySlot : 0u) | (cnt << (8u + 2u * mySlot)) | (occupied ? 1u << (24u + mySlot) : 0u));
    const uint32_t laneOfSlot = group_or(sub << (4u * mySlot));
    const uint32_t imask = bySlot & 0xFFu, itemFields = (bySlot >> 8) & 0xFFFFu;
    const uint32_t nInternal = __popc(imask), nItems = __popc(itemFields & 0x5555u) + 2u * __popc(itemFields & 0xAAAAu);
    const uint32_t ci = __popc(imask & ((1u << mySlot) - 1u));
    const uint32_t before = itemFields & ((1u << (2u * mySlot)) - 1u), ti = __popc(before & 0x5555u) + 2u * __popc(before & 0xAAAAu);
    uint32_t childBase = 0, itemBase = 0;
    if (active && sub == 0) {
        if (nInternal) childBase = atomicAdd(nodesCtr, nInternal);
        if (nItems) itemBase = atomicAdd(itemsCtr, nItems);
    }
    childBase = __shfl(childBase, 0, 8); itemBase = __shfl(itemBase, 0, 8);

    // 5. what each slot leaves behind
    uint32_t meta = 0;
    if (internalChild) {
        if (active) { if (childBase + ci < A.nodeCapacity) A.binaryRootOf[childBase + ci] = ref; else *errorFlag = 1u; }
        meta = 0x20u | (24u + mySlot);
    } else if (occupied) {
        if (ti + cnt > 24u) *errorFlag = 1u;
        uint32_t run = 0;
        #pragma unroll
        for (uint32_t k = 0; k < kMaxLeafTris; k++) {
            if (k >= nl) break;
            const uint32_t l = (uint32_t)leafId[k];
            if (active) A.leafDst[l] = itemBase + ti + run;
            run += min(A.leafSize, A.nitems - l * A.leafSize);
        }
        meta = ((((1u << cnt) - 1u) << 5) | ti) & 0xFFu;
    }
    if (active) A.slotRefs[(size_t)w * 8 + mySlot] = occupied ? ref : kEmptyRef;

    // 6. origin, per-axis power-of-two scale and this slot's 8-bit box (quantise_node's arithmetic, the eight slots side by side)
    const float p[3] = { nlo.x, nlo.y, nlo.z }, ph[3] = { nhi.x, nhi.y, nhi.z };
    const float cl3[3] = { lo.x, lo.y, lo.z }, ch3[3] = { hi.x, hi.y, hi.z };
    const bool valid = p[0] <= ph[0] && p[1] <= ph[1] && p[2] <= ph[2] && isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2])
                       && isfinite(ph[0]) && isfinite(ph[1]) && isfinite(ph[2]);
    uint32_t qlo[3], qhi[3], eb[3];
    #pragma unroll
    for (int a = 0; a < 3; a++) {
        const float ext = valid ? ph[a] - p[a] : 0.0f;
        int e;
        (void)frexpf(ext * (1.0f / 255.0f), &e);
        if (!(ext > 0.0f)) e = -126;
        const bool boxed = valid && occupied && cl3[a] <= ch3[a];
        bool done = false;
        uint32_t ql = 255u, qh = 0u;
        for (int tries = 0; tries < 256; tries++) {
            if (!__builtin_amdgcn_ballot_w64(!done)) break;
            if (!done) {
                if (e < -126) e = -126;
                const float scale = ldexpf(1.0f, e), inv = ldexpf(1.0f, -e);
                bool ok = true;
                if (boxed) {
                    float fl = floorf((cl3[a] - p[a]) * inv), fh = ceilf((ch3[a] - p[a]) * inv);
                    fl = fminf(fmaxf(fl, 0.0f), 255.0f); fh = fmaxf(fh, 0.0f);
                    while (fl > 0.0f && p[a] + fl * scale > cl3[a]) fl -= 1.0f;
                    while (fh <= 255.0f && p[a] + fh * scale < ch3[a]) fh += 1.0f;
                    ok = !(fh > 255.0f);
                    ql = (uint32_t)fl; qh = (uint32_t)fminf(fh, 255.0f);
                }
                const bool groupOk = group_ballot(!ok) == 0u;     // (lanes of a group are done together: the vote is among lanes in this branch)
                if (groupOk || e >= 127) done = true; else e++;
            }
        }
        qlo[a] = ql; qhi[a] = qh; eb[a] = (uint32_t)(e + 127);
    }
    // 7. the node: every lane collects the eight slots' bytes, lane 0 of the group stores
    const uint32_t mine0 = qlo[0] | (qlo[1] << 8) | (qlo[2] << 16) | (meta << 24), mine1 = qhi[0] | (qhi[1] << 8) | (qhi[2] << 16);
    uint32_t s0[8], s1[8];
    #pragma unroll
    for (int s = 0; s < 8; s++) {
        const int src = (int)((laneOfSlot >> (4 * s)) & 0xFu);
        s0[s] = __shfl(mine0, src, 8); s1[s] = __shfl(mine1, src, 8);
    }
    if (active && sub == 0) {
        auto bytes = [](const uint32_t* v, int first, int shift) -> uint32_t {
            return ((v[first] >> shift) & 0xFFu) | (((v[first + 1] >> shift) & 0xFFu) << 8) | (((v[first + 2] >> shift) & 0xFFu) << 16) | (((v[first + 3] >> shift) & 0xFFu) << 24);
        };
        WideNode node;
        node.origin[0] = valid ? p[0] : 0.0f; node.origin[1] = valid ? p[1] : 0.0f; node.origin[2] = valid ? p[2] : 0.0f;
        node.expImask = (imask << 24) | eb[0] | (eb[1] << 8) | (eb[2] << 16);
        node.childBase = childBase; node.triBase = itemBase;
        node.meta[0] = bytes(s0, 0, 24); node.meta[1] = bytes(s0, 4, 24);
        node.qlox[0] = bytes(s0, 0, 0);  node.qlox[1] = bytes(s0, 4, 0);
        node.qloy[0] = bytes(s0, 0, 8);  node.qloy[1] = bytes(s0, 4, 8);
        node.qloz[0] = bytes(s0, 0, 16); node.qloz[1] = bytes(s0, 4, 16);
        node.qhix[0] = bytes(s1, 0, 0);  node.qhix[1] = bytes(s1, 4, 0);
        node.qhiy[0] = bytes(s1, 0, 8);  node.qhiy[1] = bytes(s1, 4, 8);
        node.qhiz[0] = bytes(s1, 0, 16); node.qhiz[1] = bytes(s1, 4, 16);
        A.nodes[w] = node;
    }
}

// Small trees (top levels of a few thousand instances, small meshes): ONE workgroup walks the wide tree level by level (a node's
// children are allocated by its thread, the next level is the range allocated meanwhile): one launch, no inter-workgroup protocol, a
// loop bound every wave reaches.
__global__ __launch_bounds__(1024) void k_collapse(CollapseArgs A)
{
    __shared__ uint32_t sBegin, sEnd, sNodes, sItems, sError;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) { sBegin = 0; sEnd = 1; sNodes = 1; sItems = 0; sError = 0; A.binaryRootOf[0] = 0; }
    __syncthreads();
    if (A.nleaves <= 1) {                               // no internal node: traversal starts at the leaf itself, the node is a placeholder
        if (tid == 0) {
            WideNode n; memset(&n, 0, sizeof n);
            const float4 e = make_float4(0, 0, 0, 0);
            float4 cl[8], ch[8]; int refs[8];
            for (int s = 0; s < 8; s++) { cl[s] = e; ch[s] = e; refs[s] = kEmptyRef; A.slotRefs[s] = kEmptyRef; }
            quantise_node(n, e, e, cl, ch, refs);
            A.nodes[0] = n;
            if (A.nleaves == 1) A.leafDst[0] = 0;
            A.header->nodeCount = 1; A.header->itemCount = A.nitems; A.header->depth = 0; A.header->error = 0;
        }
        return;
    }
    uint32_t level = 0;
    for (; level < kMaxWideLevels; level++) {
        const uint32_t begin = sBegin, end = sEnd;
        if (begin >= end) break;
        for (uint32_t base = begin; base < end; base += blockDim.x / 8u) {              // uniform bounds: collapse_node is a wave-wide call
            const uint32_t w = base + tid / 8u;
            collapse_node(A, w, w < end, &sNodes, &sItems, &sError);
        }
        __syncthreads();
        if (tid == 0) { sBegin = end; sEnd = min(sNodes, A.nodeCapacity); }
        __syncthreads();
    }
    if (tid == 0) {
        A.header->nodeCount = min(sNodes, A.nodeCapacity); A.header->itemCount = sItems; A.header->depth = level;
        A.header->error = (sError || sBegin < sEnd) ? 1u : 0u;       // capacity exceeded (impossible: <= one wide node per binary node) or too deep
    }
}

// Large trees: one launch per LEVEL, every workgroup of the chip on it (the single workgroup above took 4.45 ms for the 31.6 k nodes of a
// 250 k-triangle mesh while 255 CUs idled). The level's node range sits in device memory; the last workgroup to finish a level
// publishes the next range (what the level allocated), so the launches need nothing from the host. The host enqueues kLevelLaunches
// of them -- as many levels as the traversal stack could walk anyway -- and a launch that finds an empty range returns at once.
struct CollapseState { uint32_t begin, end, nodes, items, error, ticket, depth, _pad; };

__global__ void k_collapse_begin(CollapseArgs A, CollapseState* st)
{
    st->begin = 0; st->end = 1; st->nodes = 1; st->items = 0; st->error = 0; st->ticket = 0; st->depth = 0;
    A.binaryRootOf[0] = 0;
}

__global__ __launch_bounds__(64) void k_collapse_level(CollapseArgs A, CollapseState* st)
{
    const uint32_t begin = st->begin, end = st->end;               // stable for the whole launch: only its last workgroup rewrites them
    if (begin >= end) return;
    for (uint32_t base = begin + blockIdx.x * (blockDim.x / 8u); base < end; base += gridDim.x * (blockDim.x / 8u)) {
        const uint32_t w = base + threadIdx.x / 8u;
        collapse_node(A, w, w < end, &st->nodes, &st->items, &st->error);
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(&st->ticket, 1u) == gridDim.x - 1u) {     // every workgroup has published its nodes
        __threadfence();
        st->ticket = 0;
        st->begin = end; st->end = min(atomicAdd(&st->nodes, 0u), A.nodeCapacity); st->depth += 1u;
    }
}

__global__ void k_collapse_end(CollapseArgs A, CollapseState* st)
{
    A.header->nodeCount = min(st->nodes, A.nodeCapacity); A.header->itemCount = st->items; A.header->depth = st->depth;
    A.header->error = (st->error || st->begin < st->end) ? 1u : 0u;             // too deep: levels left when the launches ran out
}

// after a refit: new boxes into the existing wide nodes (topology, slots, child and triangle references stay)
__global__ void k_requantise(uint32_t nodeCount, WideNode* nodes, const int* __restrict__ binaryRootOf, const int* __restrict__ slotRefs,
                             const float4* __restrict__ leafLo, const float4* __restrict__ leafHi, const float4* __restrict__ nodeLo, const float4* __restrict__ nodeHi)
{
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nodeCount) return;
    WideNode n = nodes[w];
    float4 nlo, nhi, clo[8], chi[8]; int refs[8];
    ref_box(binaryRootOf[w], leafLo, leafHi, nodeLo, nodeHi, nlo, nhi);
    for (int s = 0; s < 8; s++) {
        refs[s] = slotRefs[(size_t)w * 8 + s];
        clo[s] = make_float4(0, 0, 0, 0); chi[s] = clo[s];
        if (refs[s] != kEmptyRef) ref_box(refs[s], leafLo, leafHi, nodeLo, nodeHi, clo[s], chi[s]);
    }
    quantise_node(n, nlo, nhi, clo, chi, refs);
    nodes[w] = n;
}

// triangle packets from primitive order into node order; slotOfPrim remembers the way for refits
__global__ void k_scatter_tris(const TriPacket* __restrict__ src, const uint32_t* __restrict__ indexSorted, const uint32_t* __restrict__ leafDst,
                               uint32_t n, uint32_t leafSize, TriPacket* __restrict__ dst, uint32_t* __restrict__ slotOfPrim,
                               const uint4* __restrict__ srcIdx, uint4* __restrict__ dstIdx)
{
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;          // position in Morton order
    if (m >= n) return;
    const uint32_t leaf = m / leafSize, slot = leafDst[leaf] + (m - leaf * leafSize), prim = indexSorted[m];
    dst[slot] = src[prim];
    dstIdx[slot] = srcIdx[prim];
    if (slotOfPrim) slotOfPrim[prim] = slot;
}

// ---------------------------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------------------------
static inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

#define BVH_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { err = e_; goto fail; } } while (0)

void TreeBuffers::release()
{
    void* ptrs[] = { boxLo, boxHi, bounds, keys, keysSorted, index, indexSorted, sortTemp, leafKeys, leafLo, leafHi, children, parentInternal,
                     parentLeaf, nodeLo, nodeHi, arrival, binaryRootOf, slotRefs, leafDst, slotOfPrim, header, dp, collapseState };
    for (void* p : ptrs) if (p) hipFree(p);
    *this = TreeBuffers();
}

// (re)allocates the build buffers for nitems items in leaves of leafSize; grow-only, so a rebuild of the same size allocates nothing
static hipError_t ensure_tree_buffers(TreeBuffers& b, uint32_t nitems, uint32_t leafSize, bool withPrimSlots)
{
    const uint32_t nleaves = cdiv(nitems, leafSize), nint = nleaves > 1 ? nleaves - 1 : 1, nwide = wide_node_capacity(nleaves);
    if (b.header && nitems <= b.itemCapacity && nleaves <= b.leafCapacity && (!withPrimSlots || b.slotOfPrim)) return hipSuccess;
    b.release();
    hipError_t err = hipSuccess;
    const size_t ni = nitems ? nitems : 1, nl = nleaves ? nleaves : 1;
    size_t tmp = 0;
    BVH_CHECK(hipMalloc((void**)&b.boxLo, sizeof(float4) * ni));
    BVH_CHECK(hipMalloc((void**)&b.boxHi, sizeof(float4) * ni));
    BVH_CHECK(hipMalloc((void**)&b.bounds, sizeof(uint32_t) * 8));
    BVH_CHECK(hipMalloc((void**)&b.keys, sizeof(uint64_t) * ni));
    BVH_CHECK(hipMalloc((void**)&b.keysSorted, sizeof(uint64_t) * ni));
    BVH_CHECK(hipMalloc((void**)&b.index, sizeof(uint32_t) * ni));
    BVH_CHECK(hipMalloc((void**)&b.indexSorted, sizeof(uint32_t) * ni));
    BVH_CHECK(rocprim::radix_sort_pairs(nullptr, tmp, b.keys, b.keysSorted, b.index, b.indexSorted, ni, 0, 63, (hipStream_t)nullptr));
    b.sortTempBytes = tmp;
    BVH_CHECK(hipMalloc(&b.sortTemp, tmp ? tmp : 16));
    BVH_CHECK(hipMalloc((void**)&b.leafKeys, sizeof(uint64_t) * nl));
    BVH_CHECK(hipMalloc((void**)&b.leafLo, sizeof(float4) * nl));
    BVH_CHECK(hipMalloc((void**)&b.leafHi, sizeof(float4) * nl));
    BVH_CHECK(hipMalloc((void**)&b.children, sizeof(int2) * nint));
    BVH_CHECK(hipMalloc((void**)&b.parentInternal, sizeof(int) * nint));
    BVH_CHECK(hipMalloc((void**)&b.parentLeaf, sizeof(int) * nl));
    BVH_CHECK(hipMalloc((void**)&b.nodeLo, sizeof(float4) * nint));
    BVH_CHECK(hipMalloc((void**)&b.nodeHi, sizeof(float4) * nint));
    BVH_CHECK(hipMalloc((void**)&b.arrival, sizeof(uint32_t) * nint));
    BVH_CHECK(hipMemset(b.arrival, 0, sizeof(uint32_t) * nint));
    BVH_CHECK(hipMalloc((void**)&b.binaryRootOf, sizeof(int) * nwide));
    BVH_CHECK(hipMalloc((void**)&b.slotRefs, sizeof(int) * 8 * nwide));
    BVH_CHECK(hipMalloc((void**)&b.leafDst, sizeof(uint32_t) * nl));
    if (withPrimSlots) BVH_CHECK(hipMalloc((void**)&b.slotOfPrim, sizeof(uint32_t) * ni));
    BVH_CHECK(hipMalloc((void**)&b.header, sizeof(WideHeader)));
    BVH_CHECK(hipMalloc(&b.collapseState, 32));
    BVH_CHECK(hipMalloc(&b.dp, sizeof(DpNode) * nint));
    b.itemCapacity = (uint32_t)ni; b.leafCapacity = (uint32_t)nl;
    return hipSuccess;
fail:
    b.release();
    return err;
}

// items in b.boxLo/boxHi/bounds -> sorted -> binary tree -> wide nodes in `nodes` (capacity: wide_node_capacity(nleaves)); no sync
static hipError_t build_wide_tree(TreeBuffers& b, uint32_t nitems, uint32_t leafSize, uint32_t maxLeafItems, float costItem, bool cubicCells, bool largeFirst, WideNode* nodes, float* rootBounds, hipStream_t stream)
{
    hipError_t err = hipSuccess;
    const uint32_t nleaves = cdiv(nitems, leafSize);
    if (nitems) {
        k_morton<<<cdiv(nitems, 256), 256, 0, stream>>>(b.boxLo, b.boxHi, nitems, b.bounds, b.keys, b.index, cubicCells, largeFirst);
        size_t tmp = b.sortTempBytes;
        BVH_CHECK(rocprim::radix_sort_pairs(b.sortTemp, tmp, b.keys, b.keysSorted, b.index, b.indexSorted, nitems, 0, 63, stream));
        k_leaves<<<cdiv(nleaves, 256), 256, 0, stream>>>(b.keysSorted, b.indexSorted, b.boxLo, b.boxHi, nitems, nleaves, leafSize, b.leafKeys, b.leafLo, b.leafHi);
        if (nleaves > 1) k_karras<<<cdiv(nleaves - 1, 256), 256, 0, stream>>>(b.leafKeys, (int)nleaves, b.children, b.parentInternal, b.parentLeaf);
        k_refit<<<cdiv(nleaves, 256), 256, 0, stream>>>((int)nleaves, b.leafLo, b.leafHi, b.children, b.parentInternal, b.parentLeaf, b.nodeLo, b.nodeHi, b.arrival, rootBounds,
                                                     (DpNode*)b.dp, nitems, leafSize, maxLeafItems, costItem);
    } else {
        k_empty_bounds<<<1, 64, 0, stream>>>(rootBounds);
    }
    {
        CollapseArgs A;
        A.dp = (const DpNode*)b.dp;
        A.children = b.children; A.nodeLo = b.nodeLo; A.nodeHi = b.nodeHi; A.leafLo = b.leafLo; A.leafHi = b.leafHi;
        A.nleaves = nleaves; A.nitems = nitems; A.leafSize = leafSize; A.nodes = nodes; A.nodeCapacity = wide_node_capacity(nleaves);
        A.binaryRootOf = b.binaryRootOf; A.slotRefs = b.slotRefs; A.leafDst = b.leafDst; A.header = b.header;
        if (nleaves <= kSingleGroupCollapseLeaves) k_collapse<<<1, 1024, 0, stream>>>(A);
        else {
            CollapseState* st = (CollapseState*)b.collapseState;
            k_collapse_begin<<<1, 1, 0, stream>>>(A, st);
            for (uint32_t level = 0; level < kLevelLaunches; level++) k_collapse_level<<<2048, 64, 0, stream>>>(A, st);
            k_collapse_end<<<1, 1, 0, stream>>>(A, st);
        }
    }
    BVH_CHECK(hipGetLastError());
fail:
    return err;
}

hipError_t build_blas_device(const PtGeometryDesc* geoms, uint32_t ngeoms, bool allowUpdate, hipStream_t stream, Blas& out)
{
    hipError_t err = hipSuccess;
    uint32_t ntris = 0;
    for (uint32_t g = 0; g < ngeoms; g++) ntris += geoms[g].IndexCount / 3;
    const uint32_t leafSize = blas_leaf_tris(ntris);
    out.triCount = ntris;
    out.leafCount = cdiv(ntris, leafSize);
    const uint32_t capacity = wide_node_capacity(out.leafCount);
    TriPacket* unsorted = nullptr; uint4* unsortedIdx = nullptr;
    WideHeader hdr{};
    BVH_CHECK(ensure_tree_buffers(out.tree, ntris, leafSize, true));
    BVH_CHECK(hipMalloc((void**)&out.nodes, sizeof(WideNode) * capacity));
    BVH_CHECK(hipMalloc((void**)&out.tris, sizeof(TriPacket) * (ntris ? ntris : 1)));
    BVH_CHECK(hipMalloc((void**)&out.idx, sizeof(uint4) * (ntris ? ntris : 1)));
    BVH_CHECK(hipMalloc((void**)&out.rootBounds, sizeof(float) * 8));
    if (ntris) {
        BVH_CHECK(hipMalloc((void**)&unsorted, sizeof(TriPacket) * ntris));
        BVH_CHECK(hipMalloc((void**)&unsortedIdx, sizeof(uint4) * ntris));
        k_init_bounds<<<1, 64, 0, stream>>>(out.tree.bounds);
        uint32_t off = 0;
        for (uint32_t g = 0; g < ngeoms; g++) {
            uint32_t np = geoms[g].IndexCount / 3;
            if (np) k_tri_setup<<<cdiv(np, 256), 256, 0, stream>>>((const uint8_t*)geoms[g].VertexBuffer, geoms[g].VertexStride,
                                                                   geoms[g].IndexBuffer, geoms[g].IndexStride, np, off, g, geoms[g].Flags,
                                                                   unsorted, nullptr, out.tree.boxLo, out.tree.boxHi, out.tree.bounds, unsortedIdx);
            off += np;
        }
        BVH_CHECK(hipGetLastError());
    }
    BVH_CHECK(build_wide_tree(out.tree, ntris, leafSize, kMaxLeafTris, kCostTriangle, true, false, out.nodes, out.rootBounds, stream));
    if (ntris) k_scatter_tris<<<cdiv(ntris, 256), 256, 0, stream>>>(unsorted, out.tree.indexSorted, out.tree.leafDst, ntris, leafSize, out.tris, out.tree.slotOfPrim, unsortedIdx, out.idx);
    BVH_CHECK(hipMemcpyAsync(&hdr, out.tree.header, sizeof hdr, hipMemcpyDeviceToHost, stream));
    BVH_CHECK(hipStreamSynchronize(stream));     // build is a load-time operation (reference: CommandList::End after the BLAS build, Scene.ixx:184-188)
    out.nodeCount = hdr.nodeCount; out.depth = hdr.depth; out.buildError = hdr.error != 0;
    if (hdr.nodeCount < capacity && !hdr.error) {  // the collapse needed fewer nodes than the worst case (it always does): give the rest back
        WideNode* exact = nullptr;
        BVH_CHECK(hipMalloc((void**)&exact, sizeof(WideNode) * hdr.nodeCount));
        BVH_CHECK(hipMemcpy(exact, out.nodes, sizeof(WideNode) * hdr.nodeCount, hipMemcpyDeviceToDevice));
        hipFree(out.nodes); out.nodes = exact;
    }
    out.updatable = allowUpdate;                   // a static mesh never refits (Scene.ixx:329: no ALLOW_UPDATE): its build buffers are the caller's scratch
fail:
    if (unsorted) hipFree(unsorted);
    if (unsortedIdx) hipFree(unsortedIdx);
    return err;
}

// PERFORM_UPDATE: same topology, same Morton order, same slots -- packets and boxes only. Nothing is allocated and nothing waits.
hipError_t refit_blas_device(const PtGeometryDesc* geoms, uint32_t ngeoms, hipStream_t stream, Blas& b)
{
    if (!b.triCount) return hipSuccess;
    const uint32_t leafSize = blas_leaf_tris(b.triCount);
    uint32_t off = 0;
    for (uint32_t g = 0; g < ngeoms; g++) {
        uint32_t np = geoms[g].IndexCount / 3;
        if (np) k_tri_setup<<<cdiv(np, 256), 256, 0, stream>>>((const uint8_t*)geoms[g].VertexBuffer, geoms[g].VertexStride,
                                                               geoms[g].IndexBuffer, geoms[g].IndexStride, np, off, g, geoms[g].Flags,
                                                               b.tris, b.tree.slotOfPrim, b.tree.boxLo, b.tree.boxHi, nullptr, nullptr);
        off += np;
    }
    k_leaves<<<cdiv(b.leafCount, 256), 256, 0, stream>>>(nullptr, b.tree.indexSorted, b.tree.boxLo, b.tree.boxHi, b.triCount, b.leafCount, leafSize,
                                                         nullptr, b.tree.leafLo, b.tree.leafHi);
    k_refit<<<cdiv(b.leafCount, 256), 256, 0, stream>>>((int)b.leafCount, b.tree.leafLo, b.tree.leafHi, b.tree.children, b.tree.parentInternal,
                                                        b.tree.parentLeaf, b.tree.nodeLo, b.tree.nodeHi, b.tree.arrival, b.rootBounds, nullptr, b.triCount, leafSize, 0, 0.0f);
    if (b.leafCount > 1)
        k_requantise<<<cdiv(b.nodeCount, 256), 256, 0, stream>>>(b.nodeCount, b.nodes, b.tree.binaryRootOf, b.tree.slotRefs, b.tree.leafLo, b.tree.leafHi,
                                                                 b.tree.nodeLo, b.tree.nodeHi);
    return hipGetLastError();
}

// TLAS over n instances: nodes into out.nodes (the order of the leaf items is tree.indexSorted through tree.leafDst). build_tlas_prepare sizes the arrays (grow-only:
// a rebuild with no more instances than before allocates nothing); build_tlas_device only enqueues kernels -- no sync, no
// allocation; the header (node count, depth) stays on the device until the caller reads it.
hipError_t build_tlas_prepare(Tlas& out, uint32_t n)
{
    hipError_t err = hipSuccess;
    BVH_CHECK(ensure_tree_buffers(out.tree, n, 1, false));
    if (n > out.capacity || !out.nodes) {
        if (out.nodes) hipFree(out.nodes);
        if (out.rootBounds) hipFree(out.rootBounds);
        out.nodes = nullptr; out.rootBounds = nullptr; out.capacity = 0;
        const uint32_t cap = n ? n : 1;
        BVH_CHECK(hipMalloc((void**)&out.nodes, sizeof(WideNode) * wide_node_capacity(cap)));
        BVH_CHECK(hipMalloc((void**)&out.rootBounds, sizeof(float) * 8));
        out.capacity = cap;
    }
fail:
    return err;
}

hipError_t build_tlas_device(const InstanceRecord* dInstances, const float* const* dBlasBounds, uint32_t n, hipStream_t stream, Tlas& out)
{
    hipError_t err = hipSuccess;
    out.instanceCount = n;
    if (n) {                                                 // (launch_instance_records has reset out.tree.bounds)
        k_instance_boxes<<<std::min(cdiv(n, 4), 256u), 256, 0, stream>>>(dInstances, dBlasBounds, n, out.tree.boxLo, out.tree.boxHi, out.tree.bounds);
    }
    BVH_CHECK(build_wide_tree(out.tree, n, 1, 1, kCostInstance, kTlasCubicCells, true, out.nodes, out.rootBounds, stream));
    BVH_CHECK(hipGetLastError());
fail:
    return err;
}

// ---------------------------------------------------------------------------------------------
// instance records + compact traversal blob: [InstanceT | nodes (TLAS first) | triangle packets | order], see pt_trace2.hpp
// ---------------------------------------------------------------------------------------------
// worldToObject: inverse of the affine 3x4 evaluated in double, rounded once to float (DESIGN.md "Arithmetic spec"; DXR derives
// CommittedWorldToObject3x4 inside the driver). IEEE double operations in a fixed order, no contraction: the same bits on the
// host (pt_invert_3x4 for tests) and here.
__host__ __device__ void invert_3x4(const float m[12], float out[12])
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

// D3D12_RAYTRACING_INSTANCE_DESC (as uploaded by BuildTopLevelAccelerationStructure, RaytracingHelpers.ixx:60-63) -> instance
// records; the bottom-level table maps the id the host resolved to the arrays of that BLAS
__global__ void k_instance_records(const InstanceSource* __restrict__ src, const BlasEntry* __restrict__ table, uint32_t n,
                                   InstanceRecord* __restrict__ rec, const float** __restrict__ bounds, uint32_t* sceneBounds)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 6u) sceneBounds[i] = i < 3u ? 0xFFFFFFFFu : 0u;          // what k_init_bounds writes: k_instance_boxes, the next launch, accumulates into them
    if (i >= n) return;
    const InstanceSource s = src[i];
    const BlasEntry b = table[s.blasSlot];
    InstanceRecord r;
    for (int k = 0; k < 12; k++) r.objectToWorld[k] = s.transform[k];
    invert_3x4(s.transform, r.worldToObject);
    r.nodes = b.nodes; r.tris = b.tris; r.instanceID = s.instanceID; r.mask = s.mask; r.triCount = b.triCount; r.blasSlot = s.blasSlot;
    rec[i] = r;
    bounds[i] = b.rootBounds;
}

// What the traversal copy holds, in ONE launch (five until round 3; a dynamic frame pays ~5 us per launch): the first instBlocks workgroups
// take one instance each thread, in Morton order l -- instance i = indexSorted[l], TLAS leaf position p = leafDst[l] -- and write its record
// three times: API order (shading looks instances up by index), leaf order (the TLAS's "triangles"), and the entry record of the
// streaming traversal: what entering an instance needs in one fetch -- worldToObject (3 units), node base | triangle base | triangle
// count (24 bits) + mask | InstanceIndex (1 unit), and a copy of the BLAS's root node (5 units, read from the BLAS itself), so that the step that
// enters the instance also visits its root. The remaining workgroups run the copy jobs: BLAS nodes / packets / vertex indices and the
// TLAS's nodes into their sections (device to device, 16 B per lane, 64 workgroups per job).
__global__ __launch_bounds__(256) void k_blob_assemble(const InstanceRecord* __restrict__ inst, const float4* __restrict__ itemLo, const float4* __restrict__ itemHi,
                                                       const BlasEntry* __restrict__ table, uint32_t n, const uint32_t* __restrict__ indexSorted,
                                                       const uint32_t* __restrict__ leafDst, InstanceT* __restrict__ outInst, InstanceT* __restrict__ outLeaf,
                                                       uint4* __restrict__ outEnter, const BlobCopy* __restrict__ jobs, uint32_t njobs, uint32_t instBlocks)
{
    if (blockIdx.x >= instBlocks) {
        const uint32_t cb = blockIdx.x - instBlocks, lanesOfJob = 64u * 256u;
        for (uint32_t j = cb / 64u; j < njobs; j += (gridDim.x - instBlocks) / 64u) {
            const BlobCopy c = jobs[j];
            const uint4* s = (const uint4*)c.src; uint4* d = (uint4*)c.dst;
            for (uint64_t i = (uint64_t)(cb % 64u) * 256u + threadIdx.x; i < c.n16; i += lanesOfJob) d[i] = s[i];
        }
        return;
    }
    const uint32_t l = blockIdx.x * 256u + threadIdx.x;
    if (l >= n) return;
    const uint32_t i = indexSorted[l], p = leafDst[l];
    const InstanceRecord r = inst[i];
    InstanceT t;
    for (int k = 0; k < 12; k++) t.worldToObject[k] = r.worldToObject[k];
    float4 l4 = itemLo[i], h4 = itemHi[i];                                    // k_instance_boxes (the TLAS's items)
    pad_box(l4, h4);
    const BlasEntry e = table[r.blasSlot];
    t.boxLo[0] = l4.x; t.boxLo[1] = l4.y; t.boxLo[2] = l4.z; t.nodeBase = e.nodeBase;
    t.boxHi[0] = h4.x; t.boxHi[1] = h4.y; t.boxHi[2] = h4.z; t.triBase = e.triBase;
    t.mask = r.mask; t.triCount = r.triCount; t.instanceID = r.instanceID; t.instanceIndex = i;
    for (int k = 0; k < 12; k++) t.objectToWorld[k] = r.objectToWorld[k];
    outInst[i] = t;
    outLeaf[p] = t;
    const uint4* tv = (const uint4*)&t; const uint4* root = (const uint4*)r.nodes;
    uint4* en = outEnter + (size_t)p * kInst16;
    en[0] = tv[0]; en[1] = tv[1]; en[2] = tv[2];
    en[3] = make_uint4(t.nodeBase, t.triBase, (t.triCount < 0xFFFFFFu ? t.triCount : 0xFFFFFFu) | (t.mask << 24), t.instanceIndex);
    for (uint32_t k = 0; k < kNode16; k++) en[4 + k] = root[k];
}

hipError_t launch_instance_records(const InstanceSource* src, const BlasEntry* table, uint32_t n, InstanceRecord* rec, const float** bounds, uint32_t* sceneBounds, hipStream_t stream)
{
    if (n) k_instance_records<<<cdiv(n, 256), 256, 0, stream>>>(src, table, n, rec, bounds, sceneBounds);
    return hipGetLastError();
}

hipError_t launch_blob_assembly(const InstanceRecord* inst, const Tlas& tlas, const BlasEntry* table, uint32_t n, InstanceT* outInst,
                                InstanceT* outLeafInst, const BlobCopy* jobs, uint32_t njobs, f4v* outEnter, hipStream_t stream)
{
    const uint32_t instBlocks = cdiv(n, 256), copyBlocks = 64u * (njobs < 1024u ? njobs : 1024u);
    if (instBlocks + copyBlocks)
        k_blob_assemble<<<instBlocks + copyBlocks, 256, 0, stream>>>(inst, tlas.tree.boxLo, tlas.tree.boxHi, table, n, tlas.tree.indexSorted, tlas.tree.leafDst,
                                                                     outInst, outLeafInst, (uint4*)outEnter, jobs, njobs, instBlocks);
    return hipGetLastError();
}

} // namespace pt
