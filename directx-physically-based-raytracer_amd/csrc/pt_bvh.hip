// pt_bvh.hip -- on-device LBVH construction (gfx950).
//
// Replaces what the reference delegates to the D3D12 driver through RTXMU:
//   Scene::CreateAccelerationStructures   Source/Scene.ixx:286-380
//   CreateGeometryDesc / BuildTopLevelAccelerationStructure   Source/RaytracingHelpers.ixx:28-105
//   CommandList::BuildAccelerationStructures                  Source/CommandList.ixx:217-233
//
// Pipeline (all kernels on the context stream, no host round trip):
//   triangle packets + boxes + scene bounds  ->  30-bit Morton code of the box centre, made unique
//   by appending the primitive index  ->  rocPRIM radix sort of the 64-bit keys  ->  packets
//   gathered into Morton order, <= 4 consecutive triangles per leaf  ->  Karras 2012 hierarchy over
//   the leaves  ->  bottom-up refit (one atomic arrival counter per internal node) that emits the
//   final 64-byte two-box nodes.  The TLAS runs the same tree builder over instance boxes.
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

__device__ __forceinline__ uint32_t load_index(const void* ib, uint32_t stride, uint32_t i)
{
    return stride == 2 ? (uint32_t)((const uint16_t*)ib)[i] : ((const uint32_t*)ib)[i];
}

// one thread per triangle of one geometry: packet, box, centre; block-reduced scene bounds
__global__ void k_tri_setup(const uint8_t* __restrict__ vb, uint32_t vstride, const void* __restrict__ ib, uint32_t istride,
                            uint32_t nprims, uint32_t triOffset, uint32_t geomIndex, uint32_t flags,
                            TriPacket* __restrict__ tris, float4* __restrict__ boxLo, float4* __restrict__ boxHi,
                            uint32_t* __restrict__ bounds)
{
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    if (p < nprims) {
        float v[3][3];
        for (int k = 0; k < 3; k++) {
            uint32_t idx = load_index(ib, istride, 3 * p + k);
            const float* pv = (const float*)(vb + (size_t)vstride * idx);
            v[k][0] = pv[0]; v[k][1] = pv[1]; v[k][2] = pv[2];
        }
        TriPacket t;
        t.a = make_float4(v[0][0], v[0][1], v[0][2], __uint_as_float(geomIndex));
        t.b = make_float4(v[1][0], v[1][1], v[1][2], __uint_as_float(p));
        t.c = make_float4(v[2][0], v[2][1], v[2][2], __uint_as_float(flags));
        tris[triOffset + p] = t;
        for (int a = 0; a < 3; a++) {
            lo[a] = fminf(fminf(v[0][a], v[1][a]), v[2][a]);
            hi[a] = fmaxf(fmaxf(v[0][a], v[1][a]), v[2][a]);
        }
        boxLo[triOffset + p] = make_float4(lo[0], lo[1], lo[2], 0.0f);
        boxHi[triOffset + p] = make_float4(hi[0], hi[1], hi[2], 0.0f);
    }
    // wave64 reduction, then one atomic per wave and component
    for (int a = 0; a < 3; a++) {
        float l = lo[a], h = hi[a];
        for (int off = 32; off > 0; off >>= 1) { l = fminf(l, __shfl_xor(l, off)); h = fmaxf(h, __shfl_xor(h, off)); }
        if ((threadIdx.x & 63) == 0 && l <= h) { atomicMin(&bounds[a], f2ord(l)); atomicMax(&bounds[3 + a], f2ord(h)); }
    }
}

__device__ __forceinline__ uint32_t expand10(uint32_t v)
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__global__ void k_morton(const float4* __restrict__ boxLo, const float4* __restrict__ boxHi, uint32_t n,
                         const uint32_t* __restrict__ bounds, uint64_t* __restrict__ keys)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float mn[3], ext[3];
    for (int a = 0; a < 3; a++) { mn[a] = ord2f(bounds[a]); ext[a] = ord2f(bounds[3 + a]) - mn[a]; }
    float4 lo = boxLo[i], hi = boxHi[i];
    float c[3] = { 0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z) };
    uint32_t q[3];
    for (int a = 0; a < 3; a++) {
        float f = ext[a] > 0.0f ? (c[a] - mn[a]) / ext[a] : 0.0f;
        q[a] = (uint32_t)fminf(fmaxf(f * 1024.0f, 0.0f), 1023.0f);
    }
    uint32_t m = (expand10(q[0]) << 2) | (expand10(q[1]) << 1) | expand10(q[2]);
    keys[i] = ((uint64_t)m << 32) | (uint64_t)i;
}

__global__ void k_gather_tris(const TriPacket* __restrict__ src, const uint64_t* __restrict__ keys, uint32_t n, TriPacket* __restrict__ dst)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[(uint32_t)(keys[i] & 0xFFFFFFFFull)];
}

// BLAS leaves: <= blas_leaf_tris(ntris) consecutive Morton-ordered triangles; leaf box from the exact vertices
__global__ void k_blas_leaves(const TriPacket* __restrict__ tris, const uint64_t* __restrict__ triKeys, uint32_t ntris, uint32_t nleaves,
                              uint64_t* __restrict__ leafKeys, float4* __restrict__ leafLo, float4* __restrict__ leafHi, int* __restrict__ leafRef)
{
    uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nleaves) return;
    const uint32_t leafTris = blas_leaf_tris(ntris);
    uint32_t first = l * leafTris, count = min(leafTris, ntris - first);
    float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (uint32_t i = 0; i < count; i++) {
        TriPacket t = tris[first + i];
        const float vx[3] = { t.a.x, t.b.x, t.c.x }, vy[3] = { t.a.y, t.b.y, t.c.y }, vz[3] = { t.a.z, t.b.z, t.c.z };
        for (int k = 0; k < 3; k++) {
            lo[0] = fminf(lo[0], vx[k]); hi[0] = fmaxf(hi[0], vx[k]);
            lo[1] = fminf(lo[1], vy[k]); hi[1] = fmaxf(hi[1], vy[k]);
            lo[2] = fminf(lo[2], vz[k]); hi[2] = fmaxf(hi[2], vz[k]);
        }
    }
    leafKeys[l] = triKeys[first];
    leafLo[l] = make_float4(lo[0], lo[1], lo[2], 0.0f);
    leafHi[l] = make_float4(hi[0], hi[1], hi[2], 0.0f);
    leafRef[l] = ~(int)((first << 3) | (count - 1));
}

// TLAS items: world box of each instance = its BLAS root box pushed through ObjectToWorld (8 corners)
__global__ void k_instance_boxes(const InstanceRecord* __restrict__ inst, const float* const* __restrict__ blasBounds, uint32_t n,
                                 float4* __restrict__ boxLo, float4* __restrict__ boxHi, uint32_t* __restrict__ bounds)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    if (i < n) {
        const float* b = blasBounds[i];           // lo.xyz hi.xyz of the BLAS root
        const float* M = inst[i].objectToWorld;
        if (b[0] <= b[3]) {
            for (int c = 0; c < 8; c++) {
                float x = (c & 1) ? b[3] : b[0], y = (c & 2) ? b[4] : b[1], z = (c & 4) ? b[5] : b[2];
                for (int a = 0; a < 3; a++) {
                    float w = M[4 * a] * x + M[4 * a + 1] * y + M[4 * a + 2] * z + M[4 * a + 3];
                    lo[a] = fminf(lo[a], w); hi[a] = fmaxf(hi[a], w);
                }
            }
        }
        boxLo[i] = make_float4(lo[0], lo[1], lo[2], 0.0f);
        boxHi[i] = make_float4(hi[0], hi[1], hi[2], 0.0f);
    }
    for (int a = 0; a < 3; a++) {
        float l = lo[a], h = hi[a];
        for (int off = 32; off > 0; off >>= 1) { l = fminf(l, __shfl_xor(l, off)); h = fmaxf(h, __shfl_xor(h, off)); }
        if ((threadIdx.x & 63) == 0 && l <= h) { atomicMin(&bounds[a], f2ord(l)); atomicMax(&bounds[3 + a], f2ord(h)); }
    }
}

__global__ void k_tlas_leaves(const uint64_t* __restrict__ keys, const float4* __restrict__ boxLo, const float4* __restrict__ boxHi, uint32_t n,
                              float4* __restrict__ leafLo, float4* __restrict__ leafHi, int* __restrict__ leafRef)
{
    uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= n) return;
    uint32_t i = (uint32_t)(keys[l] & 0xFFFFFFFFull);
    leafLo[l] = boxLo[i]; leafHi[l] = boxHi[i];
    leafRef[l] = ~(int)i;
}

// ---------------------------------------------------------------------------------------------
// Karras, "Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees", HPG 2012
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int delta(const uint64_t* keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    return __clzll((long long)(keys[i] ^ keys[j]));      // keys are unique
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

__device__ __forceinline__ void write_node(BvhNode* node, float4 lo0, float4 hi0, int c0, float4 lo1, float4 hi1, int c1)
{
    BvhNode n;
    n.c0xy = make_float4(lo0.x, hi0.x, lo0.y, hi0.y);
    n.c1xy = make_float4(lo1.x, hi1.x, lo1.y, hi1.y);
    n.cz = make_float4(lo0.z, hi0.z, lo1.z, hi1.z);
    n.child = make_int4(c0, c1, 0, 0);
    *node = n;
}

// one thread per leaf climbs towards the root; the second arrival at a node owns it.
// nodeLo/nodeHi: box of each internal node (scratch). rootBounds: lo.xyz hi.xyz of the whole tree.
__global__ void k_refit(int nleaves, const float4* __restrict__ leafLo, const float4* __restrict__ leafHi, const int* __restrict__ leafRef,
                        const int2* __restrict__ children, const int* __restrict__ parentInternal, const int* __restrict__ parentLeaf,
                        float4* nodeLo, float4* nodeHi, uint32_t* arrival, BvhNode* nodes, float* rootBounds)
{
    int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nleaves) return;
    if (nleaves == 1) {
        float4 lo = leafLo[0], hi = leafHi[0];
        pad_box(lo, hi);
        float4 elo = make_float4(INFINITY, INFINITY, INFINITY, 0.0f), ehi = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.0f);
        write_node(&nodes[0], lo, hi, leafRef[0], elo, ehi, leafRef[0]);
        rootBounds[0] = lo.x; rootBounds[1] = lo.y; rootBounds[2] = lo.z; rootBounds[3] = hi.x; rootBounds[4] = hi.y; rootBounds[5] = hi.z;
        return;
    }
    int cur = parentLeaf[l];
    while (cur >= 0) {
        __threadfence();                                    // publish what this thread wrote below `cur`
        if (atomicAdd(&arrival[cur], 1u) == 0u) return;      // first arrival: the sibling will finish the node
        __threadfence();                                    // acquire the sibling subtree's boxes
        int2 ch = children[cur];
        float4 lo0, hi0, lo1, hi1; int c0, c1;
        if (ch.x < 0) { lo0 = leafLo[~ch.x]; hi0 = leafHi[~ch.x]; pad_box(lo0, hi0); c0 = leafRef[~ch.x]; }
        else { lo0 = nodeLo[ch.x]; hi0 = nodeHi[ch.x]; c0 = ch.x; }
        if (ch.y < 0) { lo1 = leafLo[~ch.y]; hi1 = leafHi[~ch.y]; pad_box(lo1, hi1); c1 = leafRef[~ch.y]; }
        else { lo1 = nodeLo[ch.y]; hi1 = nodeHi[ch.y]; c1 = ch.y; }
        write_node(&nodes[cur], lo0, hi0, c0, lo1, hi1, c1);
        float4 lo = make_float4(fminf(lo0.x, lo1.x), fminf(lo0.y, lo1.y), fminf(lo0.z, lo1.z), 0.0f);
        float4 hi = make_float4(fmaxf(hi0.x, hi1.x), fmaxf(hi0.y, hi1.y), fmaxf(hi0.z, hi1.z), 0.0f);
        nodeLo[cur] = lo; nodeHi[cur] = hi;
        if (cur == 0) {
            rootBounds[0] = lo.x; rootBounds[1] = lo.y; rootBounds[2] = lo.z; rootBounds[3] = hi.x; rootBounds[4] = hi.y; rootBounds[5] = hi.z;
        }
        cur = parentInternal[cur];
    }
}

__global__ void k_empty_tree(BvhNode* nodes, float* rootBounds)
{
    float4 elo = make_float4(INFINITY, INFINITY, INFINITY, 0.0f), ehi = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.0f);
    write_node(&nodes[0], elo, ehi, kEntryDone, elo, ehi, kEntryDone);
    rootBounds[0] = rootBounds[1] = rootBounds[2] = INFINITY; rootBounds[3] = rootBounds[4] = rootBounds[5] = -INFINITY;
}

// ---------------------------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------------------------
static inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

struct Scratch {                      // freed when the build's stream work has completed
    std::vector<void*> ptrs;
    hipError_t alloc(void** p, size_t bytes) { hipError_t e = hipMalloc(p, bytes ? bytes : 16); if (e == hipSuccess) ptrs.push_back(*p); return e; }
};

#define BVH_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { err = e_; goto fail; } } while (0)

static hipError_t sort_keys(uint64_t* keysIn, uint64_t* keysOut, uint32_t n, hipStream_t stream, Scratch& sc)
{
    size_t tmpBytes = 0;
    hipError_t e = rocprim::radix_sort_keys(nullptr, tmpBytes, keysIn, keysOut, n, 0, 62, stream);
    if (e != hipSuccess) return e;
    void* tmp = nullptr;
    e = sc.alloc(&tmp, tmpBytes);
    if (e != hipSuccess) return e;
    return rocprim::radix_sort_keys(tmp, tmpBytes, keysIn, keysOut, n, 0, 62, stream);
}

// builds the tree over nleaves prepared leaves (sorted unique keys, boxes, refs)
static hipError_t build_tree(uint32_t nleaves, const uint64_t* leafKeys, const float4* leafLo, const float4* leafHi, const int* leafRef,
                             BvhNode* nodes, float* rootBounds, hipStream_t stream, Scratch& sc)
{
    hipError_t err = hipSuccess;
    if (nleaves == 0) { k_empty_tree<<<1, 1, 0, stream>>>(nodes, rootBounds); return hipGetLastError(); }
    uint32_t nint = nleaves > 1 ? nleaves - 1 : 1;
    int2* children = nullptr; int* parentInternal = nullptr; int* parentLeaf = nullptr;
    float4* nodeLo = nullptr; float4* nodeHi = nullptr; uint32_t* arrival = nullptr;
    BVH_CHECK(sc.alloc((void**)&children, sizeof(int2) * nint));
    BVH_CHECK(sc.alloc((void**)&parentInternal, sizeof(int) * nint));
    BVH_CHECK(sc.alloc((void**)&parentLeaf, sizeof(int) * nleaves));
    BVH_CHECK(sc.alloc((void**)&nodeLo, sizeof(float4) * nint));
    BVH_CHECK(sc.alloc((void**)&nodeHi, sizeof(float4) * nint));
    BVH_CHECK(sc.alloc((void**)&arrival, sizeof(uint32_t) * nint));
    BVH_CHECK(hipMemsetAsync(arrival, 0, sizeof(uint32_t) * nint, stream));
    if (nleaves > 1) {
        k_karras<<<cdiv(nleaves - 1, 256), 256, 0, stream>>>(leafKeys, (int)nleaves, children, parentInternal, parentLeaf);
        BVH_CHECK(hipGetLastError());
    }
    k_refit<<<cdiv(nleaves, 256), 256, 0, stream>>>((int)nleaves, leafLo, leafHi, leafRef, children, parentInternal, parentLeaf,
                                                    nodeLo, nodeHi, arrival, nodes, rootBounds);
    BVH_CHECK(hipGetLastError());
fail:
    return err;
}

hipError_t build_blas_device(const PtGeometryDesc* geoms, uint32_t ngeoms, hipStream_t stream, Blas& out)
{
    hipError_t err = hipSuccess;
    Scratch sc;
    uint32_t ntris = 0;
    for (uint32_t g = 0; g < ngeoms; g++) ntris += geoms[g].IndexCount / 3;
    out.triCount = ntris;
    out.leafCount = cdiv(ntris, blas_leaf_tris(ntris));
    out.nodeCount = out.leafCount > 1 ? out.leafCount - 1 : 1;
    TriPacket* unsorted = nullptr; float4* boxLo = nullptr; float4* boxHi = nullptr; uint32_t* bounds = nullptr;
    uint64_t* keys = nullptr; uint64_t* keysSorted = nullptr;
    uint64_t* leafKeys = nullptr; float4* leafLo = nullptr; float4* leafHi = nullptr; int* leafRef = nullptr;
    BVH_CHECK(hipMalloc((void**)&out.nodes, sizeof(BvhNode) * out.nodeCount));
    BVH_CHECK(hipMalloc((void**)&out.tris, sizeof(TriPacket) * (ntris ? ntris : 1)));
    BVH_CHECK(hipMalloc((void**)&out.rootBounds, sizeof(float) * 8));
    if (ntris) {
        BVH_CHECK(sc.alloc((void**)&unsorted, sizeof(TriPacket) * ntris));
        BVH_CHECK(sc.alloc((void**)&boxLo, sizeof(float4) * ntris));
        BVH_CHECK(sc.alloc((void**)&boxHi, sizeof(float4) * ntris));
        BVH_CHECK(sc.alloc((void**)&bounds, sizeof(uint32_t) * 8));
        BVH_CHECK(sc.alloc((void**)&keys, sizeof(uint64_t) * ntris));
        BVH_CHECK(sc.alloc((void**)&keysSorted, sizeof(uint64_t) * ntris));
        BVH_CHECK(sc.alloc((void**)&leafKeys, sizeof(uint64_t) * out.leafCount));
        BVH_CHECK(sc.alloc((void**)&leafLo, sizeof(float4) * out.leafCount));
        BVH_CHECK(sc.alloc((void**)&leafHi, sizeof(float4) * out.leafCount));
        BVH_CHECK(sc.alloc((void**)&leafRef, sizeof(int) * out.leafCount));
        k_init_bounds<<<1, 64, 0, stream>>>(bounds);
        uint32_t off = 0;
        for (uint32_t g = 0; g < ngeoms; g++) {
            uint32_t np = geoms[g].IndexCount / 3;
            if (np) k_tri_setup<<<cdiv(np, 256), 256, 0, stream>>>((const uint8_t*)geoms[g].VertexBuffer, geoms[g].VertexStride,
                                                                   geoms[g].IndexBuffer, geoms[g].IndexStride, np, off, g, geoms[g].Flags,
                                                                   unsorted, boxLo, boxHi, bounds);
            off += np;
        }
        BVH_CHECK(hipGetLastError());
        k_morton<<<cdiv(ntris, 256), 256, 0, stream>>>(boxLo, boxHi, ntris, bounds, keys);
        BVH_CHECK(sort_keys(keys, keysSorted, ntris, stream, sc));
        k_gather_tris<<<cdiv(ntris, 256), 256, 0, stream>>>(unsorted, keysSorted, ntris, out.tris);
        k_blas_leaves<<<cdiv(out.leafCount, 256), 256, 0, stream>>>(out.tris, keysSorted, ntris, out.leafCount, leafKeys, leafLo, leafHi, leafRef);
        BVH_CHECK(hipGetLastError());
    }
    BVH_CHECK(build_tree(out.leafCount, leafKeys, leafLo, leafHi, leafRef, out.nodes, out.rootBounds, stream, sc));
    BVH_CHECK(hipStreamSynchronize(stream));     // build is a load-time operation (reference: CommandList::End after the BLAS build, Scene.ixx:184-188)
fail:
    for (void* p : sc.ptrs) hipFree(p);
    return err;
}

hipError_t build_tlas_device(const InstanceRecord* dInstances, const float* const* dBlasBounds, uint32_t n, hipStream_t stream, Tlas& out)
{
    hipError_t err = hipSuccess;
    Scratch sc;
    out.instanceCount = n;
    out.nodeCount = n > 1 ? n - 1 : 1;
    float4* boxLo = nullptr; float4* boxHi = nullptr; uint32_t* bounds = nullptr; uint64_t* keys = nullptr; uint64_t* keysSorted = nullptr;
    float4* leafLo = nullptr; float4* leafHi = nullptr; int* leafRef = nullptr; float* rootBounds = nullptr;
    BVH_CHECK(hipMalloc((void**)&out.nodes, sizeof(BvhNode) * out.nodeCount));
    BVH_CHECK(sc.alloc((void**)&rootBounds, sizeof(float) * 8));
    if (n) {
        BVH_CHECK(sc.alloc((void**)&boxLo, sizeof(float4) * n));
        BVH_CHECK(sc.alloc((void**)&boxHi, sizeof(float4) * n));
        BVH_CHECK(sc.alloc((void**)&bounds, sizeof(uint32_t) * 8));
        BVH_CHECK(sc.alloc((void**)&keys, sizeof(uint64_t) * n));
        BVH_CHECK(sc.alloc((void**)&keysSorted, sizeof(uint64_t) * n));
        BVH_CHECK(sc.alloc((void**)&leafLo, sizeof(float4) * n));
        BVH_CHECK(sc.alloc((void**)&leafHi, sizeof(float4) * n));
        BVH_CHECK(sc.alloc((void**)&leafRef, sizeof(int) * n));
        k_init_bounds<<<1, 64, 0, stream>>>(bounds);
        k_instance_boxes<<<cdiv(n, 256), 256, 0, stream>>>(dInstances, dBlasBounds, n, boxLo, boxHi, bounds);
        k_morton<<<cdiv(n, 256), 256, 0, stream>>>(boxLo, boxHi, n, bounds, keys);
        BVH_CHECK(hipGetLastError());
        BVH_CHECK(sort_keys(keys, keysSorted, n, stream, sc));
        k_tlas_leaves<<<cdiv(n, 256), 256, 0, stream>>>(keysSorted, boxLo, boxHi, n, leafLo, leafHi, leafRef);
        BVH_CHECK(hipGetLastError());
    }
    BVH_CHECK(build_tree(n, keysSorted, leafLo, leafHi, leafRef, out.nodes, rootBounds, stream, sc));
    BVH_CHECK(hipStreamSynchronize(stream));
fail:
    for (void* p : sc.ptrs) hipFree(p);
    return err;
}

// ---------------------------------------------------------------------------------------------
// compact traversal blob: [InstanceT | nodes (TLAS first) | triangle packets], see pt_trace2.hpp
// ---------------------------------------------------------------------------------------------
__global__ void k_blob_instances(const InstanceRecord* __restrict__ inst, const float* const* __restrict__ blasBounds,
                                 const uint2* __restrict__ bases, uint32_t n, InstanceT* __restrict__ out)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    InstanceT t;
    for (int k = 0; k < 12; k++) t.worldToObject[k] = inst[i].worldToObject[k];
    float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    const float* b = blasBounds[i];
    const float* M = inst[i].objectToWorld;
    if (b[0] <= b[3]) {
        for (int c = 0; c < 8; c++) {
            float x = (c & 1) ? b[3] : b[0], y = (c & 2) ? b[4] : b[1], z = (c & 4) ? b[5] : b[2];
            for (int a = 0; a < 3; a++) {
                float w = M[4 * a] * x + M[4 * a + 1] * y + M[4 * a + 2] * z + M[4 * a + 3];
                lo[a] = fminf(lo[a], w); hi[a] = fmaxf(hi[a], w);
            }
        }
    }
    float4 l4 = make_float4(lo[0], lo[1], lo[2], 0.0f), h4 = make_float4(hi[0], hi[1], hi[2], 0.0f);
    pad_box(l4, h4);
    t.boxLo[0] = l4.x; t.boxLo[1] = l4.y; t.boxLo[2] = l4.z; t.nodeBase = bases[i].x;
    t.boxHi[0] = h4.x; t.boxHi[1] = h4.y; t.boxHi[2] = h4.z; t.triBase = bases[i].y;
    t.mask = inst[i].mask; t.triCount = inst[i].triCount; t.instanceID = inst[i].instanceID; t._pad = 0;
    for (int k = 0; k < 12; k++) t.objectToWorld[k] = M[k];
    out[i] = t;
}

hipError_t build_blob_device(const Tlas& tlas, const float* const* dBlasBounds, const std::vector<BlobPiece>& pieces,
                             const std::vector<uint32_t>& pieceOfInstance, hipStream_t stream, void** outDev, BlobView* outView)
{
    hipError_t err = hipSuccess;
    const uint32_t n = tlas.instanceCount;
    uint32_t nodeCount = tlas.nodeCount, triCount = 0;
    for (const BlobPiece& p : pieces) { nodeCount += p.nodeCount; triCount += p.triCount; }
    const size_t instBytes = (size_t)n * sizeof(InstanceT), nodeBytes = (size_t)nodeCount * sizeof(BvhNode), triBytes = (size_t)triCount * sizeof(TriPacket);
    const size_t total = instBytes + nodeBytes + triBytes;
    uint8_t* blob = nullptr; uint2* dBases = nullptr;
    std::vector<uint2> bases(n ? n : 1);
    BVH_CHECK(hipMalloc((void**)&blob, total ? total : 16));
    BVH_CHECK(hipMalloc((void**)&dBases, sizeof(uint2) * (n ? n : 1)));
    for (uint32_t i = 0; i < n; i++) { const BlobPiece& p = pieces[pieceOfInstance[i]]; bases[i] = make_uint2(p.nodeBase, p.triBase); }
    if (n) BVH_CHECK(hipMemcpyAsync(dBases, bases.data(), sizeof(uint2) * n, hipMemcpyHostToDevice, stream));
    BVH_CHECK(hipMemcpyAsync(blob + instBytes, tlas.nodes, sizeof(BvhNode) * tlas.nodeCount, hipMemcpyDeviceToDevice, stream));
    for (const BlobPiece& p : pieces) {
        BVH_CHECK(hipMemcpyAsync(blob + instBytes + sizeof(BvhNode) * p.nodeBase, p.nodes, sizeof(BvhNode) * p.nodeCount, hipMemcpyDeviceToDevice, stream));
        if (p.triCount) BVH_CHECK(hipMemcpyAsync(blob + instBytes + nodeBytes + sizeof(TriPacket) * p.triBase, p.tris, sizeof(TriPacket) * p.triCount, hipMemcpyDeviceToDevice, stream));
    }
    if (n) k_blob_instances<<<cdiv(n, 256), 256, 0, stream>>>(tlas.instances, dBlasBounds, dBases, n, (InstanceT*)blob);
    BVH_CHECK(hipGetLastError());
    BVH_CHECK(hipStreamSynchronize(stream));
    outView->base = (const f4v*)blob;
    outView->instOff16 = 0; outView->nodeOff16 = (uint32_t)(instBytes / 16); outView->triOff16 = (uint32_t)((instBytes + nodeBytes) / 16);
    outView->instCount = n; outView->nodeCount = nodeCount; outView->triCount = triCount; outView->bytes = (uint32_t)total;
    *outDev = blob; blob = nullptr;
fail:
    if (blob) hipFree(blob);
    if (dBases) hipFree(dBases);
    return err;
}

} // namespace pt
