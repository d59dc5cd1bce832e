// pt_internal.hpp -- host-side context and the internal interfaces between the .hip files.
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/ptamd.h"
#include "pt_trace2.hpp"

namespace pt {

struct Blas {
    BvhNode* nodes = nullptr;
    TriPacket* tris = nullptr;
    float* rootBounds = nullptr;             // device: lo.xyz hi.xyz
    uint32_t triCount = 0, leafCount = 0, nodeCount = 0;
};

struct Tlas {
    BvhNode* nodes = nullptr;
    InstanceRecord* instances = nullptr;     // device, indexed by InstanceIndex
    uint32_t instanceCount = 0, nodeCount = 0;
    uint64_t triangleCount = 0;              // sum over instances
};


// everything a render kernel needs about the scene, passed by value as a kernel argument
struct SceneView {
    AccelView accel;
    const PtObjectData* objects;
    const PtInstanceData* instanceData;
    const HeapEntry* heap;
    const float* srgbLut;                    // 256-entry sRGB -> linear table (device)
    uint32_t objectCount, heapCount;
};

struct FrameView {                           // G-buffer geometry of this context's shard
    uint32_t width, height;                  // full frame
    uint32_t localRows;                      // rows held by this rank
    uint32_t rankIndex, rankCount, bandHeight;
};

// wavefront path state, structure of arrays, 16-byte records (DESIGN.md "Queues")
struct PathQueue {
    float4* s0;      // throughput.xyz | pixel (local index, bits)
    float4* s1;      // sampleRadiance.xyz | rng state (bits)
    float4* s2;      // radiance sum.xyz | sample << 16 | bounce << 1 | fresh (bits)
    float4* r0;      // ray origin.xyz | tmin
    float4* r1;      // ray direction.xyz | tmax
    uint4*  hit;     // instance | triangle slot in the BLAS | u bits | v bits      (instance ~0u = miss)
};

struct FrameConstants { PtCamera cam; PtSceneData sd; PtGraphicsSettings gs; };

struct DeviceCounters {
    unsigned long long primaryRays, secondaryRays, nodesVisited, trianglesTested;
    unsigned int mismatchCount, _pad;        // PT_DEBUG_BRUTE_FORCE: rays whose LBVH result differs from brute force
    float mismatchRay[16];                   // first such ray: o.xyz tmin d.xyz tmax | bvh inst slot t - | brute inst slot t -
};

struct Context {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string lastError;

    std::vector<HeapEntry> heapHost;
    HeapEntry* heapDev = nullptr; uint32_t heapDevCap = 0; bool heapDirty = true;
    float* srgbLutDev = nullptr;
    bool heapHasTextures = false;             // any Texture2D / TextureCube descriptor: selects the TEXTURED kernel variants

    std::map<uint64_t, Blas> blas; uint64_t nextBlasId = 1;
    Tlas tlas; bool haveTlas = false;

    PtCamera camera{}; PtSceneData sceneData{}; PtGraphicsSettings settings{};
    bool haveCamera = false, haveSceneData = false, haveSettings = false;
    const PtObjectData* objects = nullptr; uint32_t objectCount = 0;
    const PtInstanceData* instanceData = nullptr; uint32_t instanceDataCount = 0;
    PtSharding sharding{0, 1, 16, 0};

    void* blobDev = nullptr; BlobView blob{};        // compact traversal copy of TLAS + instances + every referenced BLAS

    PathQueue queue[2]{}; uint32_t queueCapacity = 0;
    float2* pixelAux = nullptr; uint32_t pixelAuxCapacity = 0;   // denoiser modes: first-bounce hit distance | isDiffuse per pixel
    FrameConstants* frameConstants = nullptr;
    hipGraphExec_t graphExec = nullptr; std::string graphKey; bool disableGraphs = false;
    uint32_t* queueCounts = nullptr; uint32_t queueCountsCap = 0;
    DeviceCounters* counters = nullptr;
    uint64_t lastIterations = 0;
    uint32_t debugFlags = 0;

    bool timing = false;
    std::vector<hipEvent_t> evExtend, evShade, evRound;     // begin/end pairs since timing was enabled
    uint32_t nExtend = 0, nShade = 0, nRound = 0;
};

// pt_bvh.hip
hipError_t build_blas_device(const PtGeometryDesc* geoms, uint32_t ngeoms, hipStream_t stream, Blas& out);
hipError_t build_tlas_device(const InstanceRecord* dInstances, const float* const* dBlasBounds, uint32_t n, hipStream_t stream, Tlas& out);
struct BlobPiece { const BvhNode* nodes; const TriPacket* tris; uint32_t nodeCount, triCount, nodeBase, triBase; };
hipError_t build_blob_device(const Tlas& tlas, const float* const* dBlasBounds, const std::vector<BlobPiece>& pieces,
                             const std::vector<uint32_t>& pieceOfInstance, hipStream_t stream, void** outDev, BlobView* outView);

// pt_skin.hip
hipError_t launch_skin(hipStream_t stream, const void* skeletal, const float* transforms, void* vertices, void* motion, uint32_t count);

// pt_kernels.hip
hipError_t launch_gbuffer(Context& c, const SceneView& sv, const FrameView& fv, uint32_t flags, const PtTextures& tx);
hipError_t launch_raytrace(Context& c, const SceneView& sv, const FrameView& fv, const PtTextures& tx);
hipError_t launch_visibility(Context& c, const SceneView& sv, const void* rays, uint32_t count, void* out);
hipError_t launch_bsdf_evaluate(hipStream_t stream, const float* q, uint32_t count, float* r);
hipError_t launch_deinterleave(hipStream_t stream, void* dst, const void* src, const uint64_t* rankOffsetsHost, uint32_t rankCount,
                               uint32_t bandHeight, uint32_t width, uint32_t height, uint32_t pixelBytes);

} // namespace pt
