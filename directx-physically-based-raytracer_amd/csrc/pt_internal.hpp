// pt_internal.hpp -- host-side context and the internal interfaces between the .hip files.
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/ptamd.h"
#include "pt_trace2.hpp"

namespace pt {

struct WideHeader { uint32_t nodeCount, itemCount, depth, error; };      // written by the collapse kernel

// Device buffers of one tree build. A bottom level built with PT_BUILD_FLAG_ALLOW_UPDATE and the top level keep them, so that a
// refit / rebuild allocates nothing; a static bottom level frees them after the build.
struct TreeBuffers {
    float4* boxLo = nullptr; float4* boxHi = nullptr; uint32_t* bounds = nullptr;
    uint64_t* keys = nullptr; uint64_t* keysSorted = nullptr; uint32_t* index = nullptr; uint32_t* indexSorted = nullptr;
    void* sortTemp = nullptr; size_t sortTempBytes = 0;
    uint64_t* leafKeys = nullptr; float4* leafLo = nullptr; float4* leafHi = nullptr;
    int2* children = nullptr; int* parentInternal = nullptr; int* parentLeaf = nullptr;
    float4* nodeLo = nullptr; float4* nodeHi = nullptr; uint32_t* arrival = nullptr;
    int* binaryRootOf = nullptr; int* slotRefs = nullptr; uint32_t* leafDst = nullptr; uint32_t* slotOfPrim = nullptr;
    WideHeader* header = nullptr; void* collapseState = nullptr;
    void* dp = nullptr;                             // collapse cost tables per binary node
    uint32_t itemCapacity = 0, leafCapacity = 0;
    void release();
};

// wide nodes a tree of nleaves leaves can need. A wide node absorbs at least one binary internal node (nleaves - 1 of them), and no more
// when the cost tables see no gain in opening a child (zero-area subtrees: degenerate triangles, instances of empty meshes): the worst
// case is one wide node per binary node. Typical trees use ~0.13 nodes per leaf; a static bottom level gives the rest back after its build.
inline uint32_t wide_node_capacity(uint32_t nleaves) { return nleaves + 1u; }

struct Blas {
    WideNode* nodes = nullptr;
    TriPacket* tris = nullptr;               // node order: the triangles of a node's leaf slots are contiguous
    uint4* idx = nullptr;                    // the three vertex indices of every triangle, same order (hit reconstruction: no index-buffer hop)
    float* rootBounds = nullptr;             // device: lo.xyz hi.xyz
    uint32_t triCount = 0, leafCount = 0, nodeCount = 0, depth = 0, geometryCount = 0;
    bool buildError = false, updatable = false;
    // a static bottom level lives in the context's traversal copy only, once a top-level build has seen it (pt_api.hip pt_build_top_level):
    // nodes / tris / idx above are null then, and these are the byte offsets of its three pieces in that copy
    bool inBlob = false; uint64_t blobNodeAt = 0, blobTriAt = 0, blobIdxAt = 0;
    TreeBuffers tree;                        // kept only when updatable
};

struct Tlas {
    WideNode* nodes = nullptr;               // capacity wide_node_capacity(capacity)
    float* rootBounds = nullptr;
    InstanceRecord* instances = nullptr;     // device, indexed by InstanceIndex
    const float** blasBounds = nullptr;      // device, per instance: root bounds of its BLAS
    uint32_t instanceCount = 0, capacity = 0;
    uint64_t triangleCount = 0;              // sum over instances
    TreeBuffers tree;
};

// host -> device inputs of a top-level build, one upload per build
struct InstanceSource { float transform[12]; uint32_t instanceID, mask, blasSlot, _pad; };
struct BlasEntry { const WideNode* nodes; const TriPacket* tris; const float* rootBounds; uint32_t triCount, nodeCount, nodeBase, triBase; const uint4* idx;
                   uint32_t objectBase, geometryCount; };   // InstanceID of the first instance that refers to the bottom level: ObjectData[objectBase + geometry] describes its meshes
struct BlobCopy { const void* src; void* dst; uint64_t n16; };

// What hit reconstruction needs of an object's geometry, resolved once per change of (ObjectData, heap) by the validation kernel: the
// object record -> descriptor table -> buffer chain of the reference (RaytracingHelpers.hlsli:82-85) is one fetch here.
struct alignas(16) ShadeGeom { const uint8_t* vb; uint32_t stride, nOff, tOff, uvOff[2], _pad; };   // offsets of normal / tangent / the two texture-coordinate sets inside a vertex; ~0u: attribute absent.
                                                                                                 // (a triangle's vertex indices come from the traversal copy, not from the object's index buffer)
static_assert(sizeof(ShadeGeom) == 32, "layout");

// everything a render kernel needs about the scene, passed by value as a kernel argument
struct SceneView {
    AccelView accel;
    const PtObjectData* objects;
    const ShadeGeom* shadeGeom;              // [objectCount]
    const HeapEntry* shadeTex;               // [objectCount * 7]: the objects' resolved texture slots (pt_texture.hpp TextureSlots)
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

// Queue geometry (pt_kernels.hip "wavefront path tracer"): the path queue is cut into independent sub-queues
// How many: a power of two chosen per frame by the form that renders it (Context::sqShift = its log2). The fused round kernel likes 32 (C2: 64 -0.4 %,
// 128 -0.9 %), the streaming form 128 (C5 +3 %, C3 +0.5 % over 32; 256 the same, 512 less): a sub-queue is then shared by 16 waves instead of 64,
// cursors are less contended and run dry at a finer grain. Per round the counters are: entries traced | fresh | cursor of the streaming form,
// one word per sub-queue each (3 << sqShift words).
constexpr uint32_t kSubQueueShiftFused = 5, kSubQueueShiftStream = 7, kSubQueuesMax = 1u << kSubQueueShiftStream;

struct FrameConstants { PtCamera cam; PtSceneData sd; PtGraphicsSettings gs; };
struct RoundArgs;

struct DeviceCounters {
    unsigned long long primaryRays, secondaryRays, nodesVisited, trianglesTested;
    unsigned int mismatchCount, stackOverflows;   // PT_DEBUG_BRUTE_FORCE: rays whose BVH result differs from brute force | refused stack pushes (must be 0)
    float mismatchRay[16];                   // first such ray: o.xyz tmin d.xyz tmax | bvh inst slot t - | brute inst slot t -
    unsigned int maxNodesPerRay, _pad2;      // PT_DEBUG_TRAVERSAL_STATS, streaming traversal: the longest walk
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
    TreeBuffers buildScratch;                         // build buffers of static bottom levels, reused from build to build (grow-only)
    Tlas tlas; bool haveTlas = false;
    Context* sceneOwner = nullptr;                    // pt_share_scene: tlas / blob below are views of that context's, never freed here
    int borrowers = 0;                                // contexts viewing THIS context's scene
    std::vector<Context*> viewers;                    // ... and who they are (pt_destroy of a viewed owner detaches them)
    uint32_t framesInFlight = 1;                      // pt_set_frames_in_flight: how many contexts render concurrently on this GPU (grid sizing)
    std::vector<uint64_t> tlasBlasIds;                // bottom levels the live TLAS refers to (pt_release_bottom_level checks)
    std::vector<uint8_t> tlasUploadHost; void* tlasUploadDev = nullptr; size_t tlasUploadCap = 0;   // InstanceSource | BlasEntry | BlobCopy
    struct UploadStage { void* host = nullptr; size_t capacity = 0; hipEvent_t event = nullptr; };  // pinned staging of that upload, two in turn
    UploadStage tlasStage[2]; uint32_t tlasStageNext = 0;
    WideHeader* tlasHeaderHost = nullptr; hipEvent_t tlasHeaderEvent = nullptr; bool tlasHeaderPending = false;   // lazy depth / error check
    uint32_t maxBlasDepth = 0, tlasInstanceCap = 0, tlasNodeReserve = 0;   // tlasNodeReserve: top-level nodes reserved at the head of the traversal copy by the last build
    size_t blobCapacity = 0;
    uint32_t tlasValidatedCount = ~0u, persistentGrid = 0;
    uint64_t tlasBindingHash = 0;                     // over (InstanceID, bottom-level id) of the instances, in order: what the shared-geometry check depends on
    uint64_t tlasObjectEnd = 0;                       // max over instances of InstanceID + geometry count: ObjectData must reach that far
    uint32_t sqShift = kSubQueueShiftFused;   // log2 of the number of sub-queues of the frame being enqueued (launch_raytrace)
    ShadeGeom* shadeGeomDev = nullptr; uint32_t shadeGeomCap = 0;
    HeapEntry* shadeTexDev = nullptr;                 // same capacity: 7 resolved texture slots per object
    // per-frame copy of the vertex normals, one record per triangle packet of the traversal copy (pt_shade.hpp ShadeTables)
    uint4* shadeRecA = nullptr; uint32_t* shadeRecB = nullptr; uint32_t shadeRecCap = 0;
    const BlasEntry* blasTableDev = nullptr; uint32_t blasTableCount = 0; uint32_t blasTableMaxTris = 0;    // the top-level build's table of bottom levels (device; a viewer: the owner's)
    const InstanceSource* instSourceDev = nullptr;
    bool normalsShared = false;              // every instance of a bottom level resolves to the same vertex buffer / stride / normal offset (checked with the objects)
    bool sharedVerdict = false;              // what the last shared-geometry check said (normalsShared is put aside while there is no top level)
    bool validated = false; uint32_t* validateDev = nullptr;   // descriptor / index validation of the scene inputs (pt_api.hip make_views)
    const void* validatedObjects = nullptr; uint32_t validatedObjectCount = 0;

    PtCamera camera{}; PtSceneData sceneData{}; PtGraphicsSettings settings{};
    bool haveCamera = false, haveSceneData = false, haveSettings = false;
    const PtObjectData* objects = nullptr; uint32_t objectCount = 0;
    const PtInstanceData* instanceData = nullptr; uint32_t instanceDataCount = 0;
    PtSharding sharding{0, 1, 16, 0};

    void* blobDev = nullptr; BlobView blob{};        // compact traversal copy of TLAS + instances + every referenced BLAS

    PathQueue queue[2]{}; uint32_t queueCapacity = 0;
    uint4* primaryRecords = nullptr; uint32_t primaryCapacity = 0;   // 48 B per local pixel: the primary surface as bounce 0 reads it (k_pt_init)
    float2* pixelAux = nullptr; uint32_t pixelAuxCapacity = 0;   // denoiser modes: first-bounce hit distance | isDiffuse per pixel
    FrameConstants* frameConstants = nullptr;
    hipGraphExec_t graphExec = nullptr; std::string graphKey; bool disableGraphs = false;
    // chains of a frame (pt_kernels.hip launch_raytrace): the rounds of a group of sub-queues need nothing from the other groups, so each group's
    // chain of launches is a linear graph replayed on a stream of its own, behind the frame's preamble and joined to the context's stream.
    static constexpr uint32_t kMaxChains = 4;
    uint32_t chains = 0;                              // 0: the library chooses (pt_set_round_chains)
    hipStream_t chainStream[kMaxChains - 1] = { nullptr, nullptr, nullptr }; hipEvent_t chainFork = nullptr, chainJoin[kMaxChains - 1] = { nullptr, nullptr, nullptr };
    hipGraphExec_t chainGraph[kMaxChains] = { nullptr, nullptr, nullptr, nullptr }; std::string chainGraphKey;
    uint32_t* queueCounts = nullptr; uint32_t queueCountsCap = 0;
    struct RoundArgs* roundArgs = nullptr; uint32_t roundArgsCap = 0; std::string roundArgsKey;   // per-round argument blocks of k_round (device)
    DeviceCounters* counters = nullptr;
    uint64_t lastIterations = 0;
    uint32_t debugFlags = 0;

    void* comm = nullptr; bool commOwned = false; uint32_t commRank = 0, commWorld = 1;   // ncclComm_t of pt_gather_bands (pt_comm.hip)

    bool timing = false;
    std::vector<hipEvent_t> evExtend, evShade, evRound;     // begin/end pairs since timing was enabled
    uint32_t nExtend = 0, nShade = 0, nRound = 0;
};

std::string& create_error();             // pt_api.hip: the message pt_last_error(NULL) returns (errors of the context-free entry points)

// pt_bvh.hip
hipError_t build_blas_device(const PtGeometryDesc* geoms, uint32_t ngeoms, bool allowUpdate, hipStream_t stream, Blas& out);
hipError_t refit_blas_device(const PtGeometryDesc* geoms, uint32_t ngeoms, hipStream_t stream, Blas& b);
hipError_t build_tlas_prepare(Tlas& out, uint32_t n);
hipError_t build_tlas_device(const InstanceRecord* dInstances, const float* const* dBlasBounds, uint32_t n, hipStream_t stream, Tlas& out);
hipError_t launch_instance_records(const InstanceSource* src, const BlasEntry* table, uint32_t n, InstanceRecord* rec, const float** bounds, uint32_t* sceneBounds, hipStream_t stream);
hipError_t launch_blob_assembly(const InstanceRecord* inst, const Tlas& tlas, const BlasEntry* table, uint32_t n, InstanceT* outInst,
                                InstanceT* outLeafInst, const BlobCopy* jobs, uint32_t njobs, f4v* outEnter, hipStream_t stream);
__host__ __device__ void invert_3x4(const float m[12], float out[12]);

// pt_skin.hip
hipError_t launch_skin(hipStream_t stream, const void* skeletal, const float* transforms, void* vertices, void* motion, uint32_t count);

// pt_stream.hip
hipError_t launch_shade(Context& c, const SceneView& sv, const FrameView& fv, const PtTextures& tx, const PathQueue& qin, const PathQueue& qout, float2* aux,
                        uint32_t segCap, const uint32_t* countIn, uint32_t* countOut, uint32_t grid, hipStream_t stream, uint32_t sqBase, uint32_t sqCount);
hipError_t launch_extend_stream(Context& c, const AlphaContext& ac, const PathQueue& q, uint32_t segCap, const uint32_t* count, uint32_t* cursor,
                                uint32_t grid, bool stats, bool writeT, hipStream_t stream, uint32_t sqBase, uint32_t sqCount);

// pt_kernels.hip
hipError_t launch_gbuffer(Context& c, const SceneView& sv, const FrameView& fv, uint32_t flags, const PtTextures& tx);
hipError_t launch_raytrace(Context& c, const SceneView& sv, const FrameView& fv, const PtTextures& tx);
hipError_t launch_visibility(Context& c, const SceneView& sv, const void* rays, uint32_t count, void* out);
hipError_t launch_bsdf_evaluate(hipStream_t stream, const float* q, uint32_t count, float* r);
hipError_t launch_debug_trace(Context& c, const SceneView& sv, const float* ray8, uint32_t* devLog, uint32_t logCap);
uint32_t round_objects_in_lds(const Context& c, uint32_t objectCount, bool haveShadeGeom);   // the fused round kernel's LDS tables (PtAccelStats)
uint32_t round_records_in_lds(const Context& c, uint32_t objectCount, bool haveShadeGeom);
inline bool normal_records_usable(const Context& c) { return c.normalsShared && c.blasTableDev && c.blasTableCount <= 65535u /* grid.y of k_capture_normals */ && c.shadeRecA && c.blob.triCount && c.blob.triCount <= c.shadeRecCap; }
hipError_t launch_check_shared_geometry(hipStream_t stream, const InstanceSource* src, const BlasEntry* table, uint32_t n, const ShadeGeom* shadeGeom, uint32_t* out);
hipError_t launch_validate_objects(hipStream_t stream, const PtObjectData* objects, uint32_t count, const HeapEntry* heap, uint32_t heapCount, uint32_t* out, ShadeGeom* shadeGeom, HeapEntry* shadeTex);
hipError_t launch_deinterleave(hipStream_t stream, void* dst, const void* src, const uint64_t* rankOffsetsHost, uint32_t rankCount,
                               uint32_t bandHeight, uint32_t width, uint32_t height, uint32_t pixelBytes);

} // namespace pt

struct PtContext { pt::Context c; };      // the opaque handle of include/ptamd.h
