// ptamd.hpp -- C++ host-side mirror of the reference's operators over the C ABI (include/ptamd.h).
//
// Same names, members and call order as the reference structs this path replaces, so that
// App::RenderScene (Source/App.cpp:1157-1329) keeps its shape when it is pointed at the MI355X path:
//     struct GBufferGeneration   Source/GBufferGeneration.ixx:27-122
//     struct Raytracing          Source/Raytracing.ixx:29-250      (DEFAULT permutation; SHARC overload absent)
//     BuildTopLevelAccelerationStructure / CreateGeometryDesc       Source/RaytracingHelpers.ixx:28-105
//     struct SkeletalMeshSkinning Source/SkeletalMeshSkinning.ixx:20-66
// Error behaviour: a failing status becomes the exception type the reference throws at the same place
// (std::invalid_argument for argument checks, std::system_error otherwise: Source/ErrorHelpers.ixx:16-32).
// Header-only, plain C++20 (std::span), no HIP headers needed by the including translation unit.
#pragma once
#include <cstdint>
#include <span>
#include <stdexcept>
#include <string>
#include <system_error>
#include <vector>

#include "../../include/ptamd.h"

namespace ptamd {

inline void ThrowIfFailed(PtContext* ctx, int status)
{
    if (status == PT_OK) return;
    std::string msg = pt_last_error(ctx);
    if (status == PT_ERROR_INVALID_ARGUMENT) throw std::invalid_argument(msg);
    throw std::system_error(status, std::generic_category(), msg);
}

// Stand-in for the reference's DeviceContext + CommandList pair: one HIP device + one stream.
struct CommandList {
    PtContext* Context = nullptr;

    explicit CommandList(int deviceOrdinal = 0, void* hipStream = nullptr)
    {
        int s = pt_create(deviceOrdinal, &Context);
        if (s != PT_OK) throw std::system_error(s, std::generic_category(), pt_last_error(nullptr));
        if (hipStream) ThrowIfFailed(Context, pt_set_stream(Context, hipStream));
    }
    CommandList(const CommandList&) = delete;
    CommandList& operator=(const CommandList&) = delete;
    ~CommandList() { pt_destroy(Context); }

    void End() { ThrowIfFailed(Context, pt_sync(Context)); }            // CommandList::End + Wait, Source/CommandList.ixx:86-119

    // Frames in flight: this command list renders the scene `owner` built (acceleration structures, descriptor table, object and
    // instance data), read-only, on its own stream with its own path queues. The reference has ONE Scene whatever the number of
    // frames in flight (Source/App.cpp:372-374). `owner` must outlive the sharing.
    void ShareScene(const CommandList& owner) { ThrowIfFailed(Context, pt_share_scene(Context, owner.Context)); }
};

// Multi-GPU (no reference counterpart: the reference presents from one device). One process per GPU; every rank renders its 16-row
// bands (pt_set_sharding) and the root assembles the frame with one grouped RCCL exchange over xGMI (pt_gather_bands).
struct BandSharding {
    static constexpr uint32_t BandHeight = 16;
    uint32_t Rank = 0, World = 1;

    // every rank calls this with the same 128-byte id (UniqueId() of one rank, distributed by the caller); returns when all have joined
    void Join(CommandList& commandList, uint32_t rank, uint32_t world, const void* uniqueId)
    {
        Rank = rank; World = world;
        const PtSharding s{ rank, world, BandHeight, 0 };
        ThrowIfFailed(commandList.Context, pt_set_sharding(commandList.Context, &s));
        if (world > 1 || uniqueId) ThrowIfFailed(commandList.Context, pt_comm_init(commandList.Context, uniqueId, rank, world));
    }
    static std::vector<uint8_t> UniqueId()
    {
        std::vector<uint8_t> id(PT_COMM_ID_BYTES);
        int s = pt_comm_get_unique_id(id.data());
        if (s != PT_OK) throw std::system_error(s, std::generic_category(), pt_last_error(nullptr));
        return id;
    }
    uint32_t LocalRows(uint32_t height) const
    {
        const PtSharding s{ Rank, World, BandHeight, 0 };
        uint32_t rows = 0;
        if (pt_local_rows(&s, height, &rows) != PT_OK) throw std::invalid_argument("invalid sharding");
        return rows;
    }
    // localTexture: this rank's rows of a texture; fullFrame (root only): H x W pixels. Enqueued on the command list's stream.
    void GatherBands(CommandList& commandList, const void* localTexture, void* fullFrame, uint32_t width, uint32_t height, uint32_t pixelBytes, uint32_t root = 0)
    {
        ThrowIfFailed(commandList.Context, pt_gather_bands(commandList.Context, localTexture, fullFrame, width, height, pixelBytes, root));
    }
};

// GPUBuffer* / Texture* slots of the reference become plain device pointers here.
struct GPUBuffer { const void* DevicePointer = nullptr; uint64_t Capacity = 0; uint32_t Stride = 0; };

namespace RaytracingHelpers {

// CreateGeometryDesc, Source/RaytracingHelpers.ixx:76-105 (same argument checks, same exception texts)
inline PtGeometryDesc CreateGeometryDesc(const GPUBuffer& vertices, const GPUBuffer& indices, uint32_t flags = 0)
{
    if (indices.Stride != sizeof(uint16_t) && indices.Stride != sizeof(uint32_t))
        throw std::invalid_argument("Triangle index format must be either uint16 or uint32");
    if (indices.Capacity % 3 != 0) throw std::invalid_argument("Triangle index count must be divisible by 3");
    PtGeometryDesc d{};
    d.VertexBuffer = vertices.DevicePointer; d.VertexCount = (uint32_t)vertices.Capacity; d.VertexStride = vertices.Stride;
    d.IndexBuffer = indices.DevicePointer; d.IndexCount = (uint32_t)indices.Capacity; d.IndexStride = indices.Stride;
    d.Flags = flags;
    return d;
}

struct TopLevelAccelerationStructure { bool Valid = false; };

// BuildTopLevelAccelerationStructure, Source/RaytracingHelpers.ixx:28-74
inline void BuildTopLevelAccelerationStructure(CommandList& commandList, uint32_t flags, std::span<const PtInstanceDesc> descs,
                                               bool /*resize*/, TopLevelAccelerationStructure& accelerationStructure)
{
    ThrowIfFailed(commandList.Context, pt_build_top_level(commandList.Context, descs.data(), (uint32_t)descs.size(), flags));
    accelerationStructure.Valid = true;
}

// CommandList::BuildAccelerationStructures for one bottom-level input, Source/CommandList.ixx:217-233
inline uint64_t BuildBottomLevelAccelerationStructure(CommandList& commandList, std::span<const PtGeometryDesc> geometryDescs, uint32_t flags)
{
    uint64_t id = 0;
    ThrowIfFailed(commandList.Context, pt_build_bottom_level(commandList.Context, geometryDescs.data(), (uint32_t)geometryDescs.size(), flags, &id));
    return id;
}

// CommandList::UpdateAccelerationStructures for one bottom-level input (PERFORM_UPDATE after skinning), Source/CommandList.ixx:235-241
inline void UpdateBottomLevelAccelerationStructure(CommandList& commandList, uint64_t id, std::span<const PtGeometryDesc> geometryDescs, uint32_t flags)
{
    ThrowIfFailed(commandList.Context, pt_update_bottom_level(commandList.Context, id, geometryDescs.data(), (uint32_t)geometryDescs.size(), flags));
}

// RTXMU RemoveAccelerationStructures in ~Scene / CollectGarbage, Source/Scene.ixx:108-123
inline void ReleaseBottomLevelAccelerationStructure(CommandList& commandList, uint64_t id)
{
    ThrowIfFailed(commandList.Context, pt_release_bottom_level(commandList.Context, id));
}

} // namespace RaytracingHelpers

// The shader-visible descriptor heap (DeviceContext::ResourceDescriptorHeap): slots that ObjectData / SceneData index.
struct DescriptorHeap {
    explicit DescriptorHeap(CommandList& commandList, uint32_t capacity) : m_context(commandList.Context)
    {
        ThrowIfFailed(m_context, pt_heap_resize(m_context, capacity));
    }
    // GPUBuffer::CreateSRV(Raw|Structured|Typed), Source/GPUBuffer.ixx: vertex / index / motion-vector buffers (App.cpp:1046-1050)
    void SetBuffer(uint32_t descriptor, const GPUBuffer& buffer)
    {
        ThrowIfFailed(m_context, pt_heap_set_buffer(m_context, descriptor, buffer.DevicePointer, buffer.Capacity * buffer.Stride, buffer.Stride));
    }
    // Texture::CreateSRV for a material map or the environment map (App.cpp:1021-1024,1052-1063); mip 0 texels, cube = 6 faces
    void SetTexture(uint32_t descriptor, const void* texels, uint32_t width, uint32_t height, PtFormat format, bool isCubeMap = false)
    {
        ThrowIfFailed(m_context, pt_heap_set_texture(m_context, descriptor, texels, width, height, (uint32_t)format, isCubeMap ? 1u : 0u));
    }
private:
    PtContext* m_context;
};

// struct SkeletalMeshSkinning, Source/SkeletalMeshSkinning.ixx:20-66 (Prepare has nothing to bind here)
struct SkeletalMeshSkinning {
    struct { const GPUBuffer* SkeletalVertices; const GPUBuffer* SkeletalTransforms; const GPUBuffer* Vertices; const GPUBuffer* MotionVectors; } GPUBuffers{};

    explicit SkeletalMeshSkinning(CommandList&) {}
    void Prepare(CommandList&) {}
    void Process(CommandList& commandList)                  // :43-65, vertexCount = Vertices->GetCapacity()
    {
        if (!GPUBuffers.SkeletalVertices || !GPUBuffers.SkeletalTransforms || !GPUBuffers.Vertices || !GPUBuffers.MotionVectors)
            throw std::invalid_argument("SkeletalMeshSkinning::GPUBuffers not set");
        ThrowIfFailed(commandList.Context, pt_skin_mesh(commandList.Context, GPUBuffers.SkeletalVertices->DevicePointer,
                                                        (const float*)GPUBuffers.SkeletalTransforms->DevicePointer,
                                                        const_cast<void*>(GPUBuffers.Vertices->DevicePointer),
                                                        const_cast<void*>(GPUBuffers.MotionVectors->DevicePointer),
                                                        (uint32_t)GPUBuffers.Vertices->Capacity));
    }
};

// The two calls the reference's direct-lighting bridge makes on the same scene data (Shaders/RTXDIAppBridge.hlsli:418-439,
// Shaders/BxDF.hlsli:247-285), batched over device arrays.
namespace DirectLighting {
inline void TraceVisibility(CommandList& commandList, const PtRayDesc* deviceRays, uint32_t count, float* deviceVisibility)
{
    ThrowIfFailed(commandList.Context, pt_trace_visibility(commandList.Context, deviceRays, count, deviceVisibility));
}
inline void EvaluateBSDF(CommandList& commandList, const PtBsdfQuery* deviceQueries, uint32_t count, PtBsdfResult* deviceResults)
{
    ThrowIfFailed(commandList.Context, pt_bsdf_evaluate(commandList.Context, deviceQueries, count, deviceResults));
}
} // namespace DirectLighting

struct GBufferGeneration {
    struct Flags {                                          // Source/GBufferGeneration.ixx:28-44
        enum {
            Position = 0x1, FlatNormal = 0x2, GeometricNormal = 0x4, LinearDepth = 0x8, NormalizedDepth = 0x10,
            MotionVector = 0x20, DiffuseAlbedo = 0x40, SpecularAlbedo = 0x80, Albedo = DiffuseAlbedo | SpecularAlbedo,
            NormalRoughness = 0x100, Radiance = 0x200,
            Geometry = Position | FlatNormal | GeometricNormal | LinearDepth | NormalizedDepth | MotionVector | NormalRoughness,
            Material = 0x400 | Albedo | NormalRoughness | Radiance
        };
    };
    struct Constants { uint32_t RenderSize[2]{}; uint32_t Flags{}; };

    // SceneData / Camera: host structs (the reference copies them into constant buffers each frame,
    // Source/App.cpp:540-561,1016-1026); InstanceData / ObjectData: device arrays.
    struct { const PtSceneData* SceneData; const PtCamera* Camera; const PtInstanceData* InstanceData; const PtObjectData* ObjectData;
             uint32_t InstanceCount, ObjectCount; } GPUBuffers{};
    PtTextures Textures{};                                  // same member order as the reference's Textures struct

    explicit GBufferGeneration(CommandList& commandList) : m_context(commandList.Context) {}

    void Render(CommandList& commandList, const RaytracingHelpers::TopLevelAccelerationStructure& topLevelAccelerationStructure, const Constants& constants)
    {
        if (!topLevelAccelerationStructure.Valid) throw std::invalid_argument("top-level acceleration structure has not been built");
        PtContext* c = commandList.Context;
        ThrowIfFailed(c, pt_set_scene_data(c, GPUBuffers.SceneData));
        ThrowIfFailed(c, pt_set_camera(c, GPUBuffers.Camera));
        ThrowIfFailed(c, pt_set_instance_data(c, GPUBuffers.InstanceData, GPUBuffers.InstanceCount));
        ThrowIfFailed(c, pt_set_object_data(c, GPUBuffers.ObjectData, GPUBuffers.ObjectCount));
        PtGBufferConstants k{ { constants.RenderSize[0], constants.RenderSize[1] }, constants.Flags };
        ThrowIfFailed(c, pt_gbuffer_render(c, &k, &Textures));
    }

private:
    PtContext* m_context;
};

struct Raytracing {
    struct GraphicsSettings {                               // Source/Raytracing.ixx:30-36
        uint32_t RenderSize[2]{};
        uint32_t FrameIndex{}, Bounces{}, SamplesPerPixel{};
        float ThroughputThreshold = 1e-3f;
        bool IsRussianRouletteEnabled{}, IsShaderExecutionReorderingEnabled{}, IsDIEnabled{};
        uint32_t Denoiser{};
    };

    struct { const PtSceneData* SceneData; const PtCamera* Camera; const PtObjectData* ObjectData; uint32_t ObjectCount; } GPUBuffers{};
    PtTextures Textures{};                                  // Position .. Radiance; Diffuse / Specular / SpecularHitDistance when GraphicsSettings.Denoiser != None

    explicit Raytracing(CommandList& commandList) : m_context(commandList.Context) {}

    void SetConstants(const GraphicsSettings& graphicsSettings) noexcept      // Source/Raytracing.ixx:92-104
    {
        m_graphicsSettings = {};
        m_graphicsSettings.RenderSize[0] = graphicsSettings.RenderSize[0];
        m_graphicsSettings.RenderSize[1] = graphicsSettings.RenderSize[1];
        m_graphicsSettings.FrameIndex = graphicsSettings.FrameIndex;
        m_graphicsSettings.Bounces = graphicsSettings.Bounces;
        m_graphicsSettings.SamplesPerPixel = graphicsSettings.SamplesPerPixel;
        m_graphicsSettings.ThroughputThreshold = graphicsSettings.ThroughputThreshold;
        m_graphicsSettings.IsRussianRouletteEnabled = graphicsSettings.IsRussianRouletteEnabled;
        m_graphicsSettings.IsShaderExecutionReorderingEnabled = graphicsSettings.IsShaderExecutionReorderingEnabled;
        m_graphicsSettings.IsDIEnabled = graphicsSettings.IsDIEnabled;
        m_graphicsSettings.Denoiser = graphicsSettings.Denoiser;
    }

    void Render(CommandList& commandList, const RaytracingHelpers::TopLevelAccelerationStructure& topLevelAccelerationStructure)   // :106-112
    {
        if (!topLevelAccelerationStructure.Valid) throw std::invalid_argument("top-level acceleration structure has not been built");
        PtContext* c = commandList.Context;
        ThrowIfFailed(c, pt_set_scene_data(c, GPUBuffers.SceneData));
        ThrowIfFailed(c, pt_set_camera(c, GPUBuffers.Camera));
        ThrowIfFailed(c, pt_set_object_data(c, GPUBuffers.ObjectData, GPUBuffers.ObjectCount));
        ThrowIfFailed(c, pt_raytrace_set_constants(c, &m_graphicsSettings));
        ThrowIfFailed(c, pt_raytrace_render(c, &Textures));
    }

private:
    PtContext* m_context;
    PtGraphicsSettings m_graphicsSettings{};
};

} // namespace ptamd
