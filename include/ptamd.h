/*
 * ptamd.h -- C ABI of the MI355X-native wavefront path tracer (libptamd.so).
 *
 * Drop-in boundary for ONE hot path of Hydr10n/DirectX-Physically-Based-Raytracer:
 *     GBufferGeneration::Render  (Source/GBufferGeneration.ixx:80-117 -> Shaders/GBufferGeneration.hlsl:116-232)
 *     Raytracing::Render         (Source/Raytracing.ixx:106-112      -> Shaders/Raytracing.hlsl:103-415, DEFAULT)
 *     acceleration-structure build (Source/Scene.ixx:286-380, Source/RaytracingHelpers.ixx:28-105,
 *                                   Source/CommandList.ixx:217-249)
 * Every entry point cites the reference interface it replaces. Plain pointers and sizes only:
 * no C++ types, no torch types. All `const void*` "device" arguments are HIP device pointers
 * owned by the caller; structs passed by pointer are HOST memory copied during the call.
 * Work is enqueued on the context's HIP stream (pt_set_stream) and is asynchronous unless
 * stated otherwise; pt_sync() blocks until it has completed.
 *
 * Error convention: every function returns PT_OK (0) or a negative PtStatus; the message is
 * available from pt_last_error(). Nothing throws across this boundary. (The reference throws
 * std::system_error / std::invalid_argument: Source/ErrorHelpers.ixx:16-32,
 * Source/RaytracingHelpers.ixx:83-88 -- the C++ mirror in directx-physically-based-raytracer_amd/host/
 * turns the status back into those exceptions.)
 */
#ifndef PTAMD_H
#define PTAMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTAMD_ABI_VERSION 4

typedef enum PtStatus {
    PT_OK = 0,
    PT_ERROR_INVALID_ARGUMENT = -1,   /* reference: Throw<std::invalid_argument> */
    PT_ERROR_HIP = -2,                /* reference: ThrowIfFailed(HRESULT) */
    PT_ERROR_OUT_OF_MEMORY = -3,
    PT_ERROR_NOT_READY = -4,          /* render before an acceleration structure exists */
    PT_ERROR_NO_DEVICE = -5,
    PT_ERROR_RCCL = -6                /* an RCCL call failed (message: ncclGetErrorString), or librccl could not be loaded */
} PtStatus;

/* ------------------------------------------------------------------------------------------
 * Reference data layouts, byte-exact. Field names are the reference's.
 * ------------------------------------------------------------------------------------------ */
typedef struct PtVertexDesc {             /* Source/Vertex.ixx:30-36, Shaders/Vertex.hlsli:5-12 */
    uint32_t Stride, _pad[3];
    struct { uint32_t Normal, Tangent, TextureCoordinates[2]; } AttributeOffsets;   /* ~0u = absent */
} PtVertexDesc;

typedef struct PtMeshDescriptors {        /* Source/CommonShaderData.ixx:28-30; indices into the descriptor heap */
    uint32_t Vertices, Indices, MotionVectors, _pad;
} PtMeshDescriptors;

typedef struct PtMaterial {               /* Source/Material.ixx:12-20, Shaders/Material.hlsli:8-22 */
    float BaseColor[4];
    float EmissiveStrength;
    float EmissiveColor[3];
    float Metallic, Roughness, IOR, Transmission;
    uint32_t AlphaMode;                   /* 0 Opaque, 1 Mask, 2 Blend */
    float AlphaCutoff;
    uint32_t _pad[2];
} PtMaterial;

typedef struct PtTextureMapInfo {         /* Source/Material.ixx:35-38 */
    uint32_t Descriptor, TextureCoordinateIndex, _pad[2];
} PtTextureMapInfo;

typedef struct PtObjectData {             /* Source/CommonShaderData.ixx:34-39 */
    PtVertexDesc VertexDesc;
    PtMeshDescriptors MeshDescriptors;
    PtMaterial Material;
    PtTextureMapInfo TextureMapInfoArray[7];
} PtObjectData;

typedef struct PtInstanceData {           /* Source/CommonShaderData.ixx:22-26 */
    uint32_t FirstGeometryIndex, _pad[3];
    float PreviousObjectToWorld[12];      /* XMFLOAT3X4: rows of the column-vector affine [R|t] */
    float ObjectToWorld[12];
} PtInstanceData;

typedef struct PtSceneData {              /* Source/CommonShaderData.ixx:15-20 */
    uint32_t IsStatic, IsEnvironmentLightTextureCubeMap;
    uint32_t EnvironmentLightTextureDescriptor, _pad;
    float EnvironmentLightColor[4];       /* a < 0 selects the procedural sky (Shaders/ShadingHelpers.hlsli:25-29) */
    float EnvironmentLightTransform[12];
} PtSceneData;

typedef struct PtCamera {                 /* Source/Camera.ixx:16-36, Shaders/Camera.hlsli:5-25 */
    uint32_t IsNormalizedDepthReversed;
    float PreviousPosition[3], Position[3], _pad0;
    float RightDirection[3], _pad1;
    float UpDirection[3], _pad2;
    float ForwardDirection[3];
    float ApertureRadius, NearDepth, FarDepth;
    float Jitter[2];
    float PreviousWorldToView[16], PreviousViewToProjection[16], PreviousWorldToProjection[16],
          PreviousProjectionToView[16], PreviousViewToWorld[16], WorldToProjection[16],
          ProjectionToView[16], ViewToWorld[16];
} PtCamera;

typedef struct PtGraphicsSettings {       /* Raytracing::GraphicsSettings, Source/Raytracing.ixx:30-36,151-166 */
    uint32_t RenderSize[2];
    uint32_t FrameIndex, Bounces, SamplesPerPixel;
    float ThroughputThreshold;            /* reference default 1e-3 */
    uint32_t IsRussianRouletteEnabled, IsShaderExecutionReorderingEnabled, IsDIEnabled;
    uint32_t Denoiser;                    /* PtDenoiser: selects which outputs Raytracing writes (Raytracing.hlsl:379-413) */
    uint32_t ExtFlags;                    /* reference: first padding word. PT_EXT_* build-side switches */
    uint32_t _pad;
    uint32_t SHARC[8];                    /* out of scope, ignored */
} PtGraphicsSettings;

/* enum class Denoiser, Source/Denoiser.ixx:8. The denoisers themselves (NRD, DLSS-RR) are out of scope; these values only
 * select the output packing the path tracer hands to them. */
enum PtDenoiser { PT_DENOISER_NONE = 0, PT_DENOISER_DLSS_RAY_RECONSTRUCTION = 1, PT_DENOISER_NRD_REBLUR = 2, PT_DENOISER_NRD_RELAX = 3 };

#define PT_EXT_LAMBERTIAN_ONLY 0x1u       /* BASELINE.json config C1: lobe weights {1,0,0}, DiffuseTerm = 1/pi */

typedef struct PtGBufferConstants {       /* GBufferGeneration::Constants, Source/GBufferGeneration.ixx:46-49 */
    uint32_t RenderSize[2];
    uint32_t Flags;                       /* PtGBufferFlags */
} PtGBufferConstants;

enum PtGBufferFlags {                     /* Source/GBufferGeneration.ixx:28-44 */
    PT_GB_Position = 0x1, PT_GB_FlatNormal = 0x2, PT_GB_GeometricNormal = 0x4, PT_GB_LinearDepth = 0x8,
    PT_GB_NormalizedDepth = 0x10, PT_GB_MotionVector = 0x20, PT_GB_DiffuseAlbedo = 0x40,
    PT_GB_SpecularAlbedo = 0x80, PT_GB_Albedo = 0xC0, PT_GB_NormalRoughness = 0x100, PT_GB_Radiance = 0x200,
    PT_GB_Geometry = 0x1 | 0x2 | 0x4 | 0x8 | 0x10 | 0x20 | 0x100,
    PT_GB_Material = 0x400 | 0xC0 | 0x100 | 0x200
};

/* The G-buffer "textures": linear row-major device arrays, one element per pixel of the LOCAL
 * framebuffer (see PtSharding), in the reference's DXGI formats (Source/App.cpp:438-455).
 * Same member order as GBufferGeneration::Textures (Source/GBufferGeneration.ixx:53-68);
 * Raytracing::Textures (Source/Raytracing.ixx:46-59) is the subset the path tracer reads plus
 * Radiance. NULL = not bound. DiffuseAlbedo / SpecularAlbedo (GBufferGeneration.hlsl:171-186) are written only when
 * their flag is set, which the reference does when a denoiser is selected (Source/App.cpp:1223). */
typedef struct PtTextures {
    void* Position;            /* R32G32B32A32_FLOAT  16 B : xyz world position, w = spawn offset; all +inf on miss */
    void* FlatNormal;          /* R16G16_SNORM         4 B : signed octahedral */
    void* GeometricNormal;     /* R16G16_SNORM         4 B */
    void* LinearDepth;         /* R32_FLOAT            4 B */
    void* NormalizedDepth;     /* R32_FLOAT            4 B */
    void* MotionVector;        /* R16G16B16A16_FLOAT   8 B */
    void* BaseColorMetalness;  /* R8G8B8A8_UNORM       4 B */
    void* DiffuseAlbedo;       /* R16G16B16A16_FLOAT   8 B  NRD_MaterialFactors diffuse (only with PT_GB_DiffuseAlbedo) */
    void* SpecularAlbedo;      /* R16G16B16A16_FLOAT   8 B  NRD_MaterialFactors specular (only with PT_GB_SpecularAlbedo) */
    void* NormalRoughness;     /* R16G16B16A16_SNORM   8 B */
    void* IOR;                 /* R16_FLOAT            2 B */
    void* Transmission;        /* R8_UNORM             1 B */
    void* Radiance;            /* R16G16B16A16_FLOAT   8 B : G-buffer emission/environment in, path-traced radiance out */
    void* RadianceF32;         /* build-side extra, optional: R32G32B32A32_FLOAT copy of the value stored to Radiance */
    /* denoiser-facing outputs of Raytracing::Textures (Source/Raytracing.ixx:55-58), written per Raytracing.hlsl:387-413 */
    void* Diffuse;             /* R16G16B16A16_FLOAT   8 B : NRD: indirect diffuse radiance, hit distance in a */
    void* Specular;            /* R16G16B16A16_FLOAT   8 B : NRD: indirect specular radiance, hit distance in a */
    void* SpecularHitDistance; /* R16_FLOAT            2 B : DLSS-RR: first-bounce hit distance of specular pixels */
} PtTextures;

/* ------------------------------------------------------------------------------------------
 * context
 * ------------------------------------------------------------------------------------------ */
typedef struct PtContext PtContext;

int  pt_abi_version(void);
/* One context per GPU (reference: one D3D12 device, Source/DeviceResources.cpp:101-170). */
int  pt_create(int device_ordinal, PtContext** out_ctx);
void pt_destroy(PtContext* ctx);
const char* pt_last_error(const PtContext* ctx);            /* ctx may be NULL: last error of pt_create */
/* hipStream_t to enqueue on (NULL = the default stream). Reference: CommandList recording order. */
int  pt_set_stream(PtContext* ctx, void* hip_stream);
int  pt_sync(PtContext* ctx);                               /* reference: CommandList::End/Wait, Source/CommandList.ixx:86-119 */
/* How many contexts render concurrently on this GPU (frames in flight on separate streams; reference: the swap chain's
 * back-buffer count, Source/DeviceResources.cpp). A performance hint only -- it sizes the persistent grids: a kernel that shares
 * the GPU with other frames' kernels launches fewer, longer-lived workgroups. Default 1. */
int  pt_set_frames_in_flight(PtContext* ctx, uint32_t frames);
/* Chains of a frame. Paths never leave their sub-queue of the path queue, so after the first bounce the rounds of a group of sub-queues depend
 * on nothing outside the group: the library runs `chains` such groups as independent chains of launches, each on a stream of its own (forked
 * from and joined to the context's stream, invisible to the caller), so that ONE frame's launches overlap the way frames in flight do -- for a
 * host that presents one frame at a time (the reference: Source/App.cpp:167). 0 = the library's choice: 3 for a scene beyond LDS rendered on a
 * caller-provided stream with pt_set_frames_in_flight(1), else 1 (measured: profiles/r04_ab/frame_chains.jsonl). On the default stream and while
 * kernel timing is enabled the chains run one after the other. The image does not depend on it. */
int  pt_set_round_chains(PtContext* ctx, uint32_t chains);            /* 0..4 */

/* ------------------------------------------------------------------------------------------
 * descriptor heap analogue. ObjectData.MeshDescriptors.{Vertices,Indices} index this table
 * (reference: ResourceDescriptorHeap[...], Shaders/RaytracingHelpers.hlsli:82-85).
 * stride: element size of a typed buffer (index buffer: 2 = R16_UINT or 4 = R32_UINT), 0 for raw (vertex) buffers,
 * 8 for the structured float16_t4 motion-vector buffer of a skinned mesh (MeshDescriptors.MotionVectors).
 * ------------------------------------------------------------------------------------------ */
int  pt_heap_resize(PtContext* ctx, uint32_t descriptor_count);
int  pt_heap_set_buffer(PtContext* ctx, uint32_t descriptor, const void* device_ptr, uint64_t bytes, uint32_t stride);
/* Texture2D / TextureCube SRV (reference: Texture::GetSRVDescriptor, indices stored in TextureMapInfo.Descriptor
 * at Source/App.cpp:1052-1063 and in SceneData.EnvironmentLightTextureDescriptor at :1021-1024). Mip 0 only: the
 * path samples with SampleLevel(sampler, uv, 0) (Shaders/ShadingHelpers.hlsli:58). Linear row-major texels;
 * a cube is 6 faces +X,-X,+Y,-Y,+Z,-Z back to back. */
typedef enum PtFormat {
    PT_FORMAT_R8G8B8A8_UNORM = 0,
    PT_FORMAT_R8G8B8A8_UNORM_SRGB = 1,     /* base colour / emissive textures (Source/GLTFHelpers.ixx:375-391) */
    PT_FORMAT_R32G32B32A32_FLOAT = 2       /* HDR environment maps */
} PtFormat;
int  pt_heap_set_texture(PtContext* ctx, uint32_t descriptor, const void* device_ptr, uint32_t width, uint32_t height,
                         uint32_t format, uint32_t is_cube);

/* ------------------------------------------------------------------------------------------
 * acceleration structures
 * ------------------------------------------------------------------------------------------ */
typedef struct PtGeometryDesc {           /* D3D12_RAYTRACING_GEOMETRY_DESC as filled by CreateGeometryDesc,
                                             Source/RaytracingHelpers.ixx:76-105 */
    const void* VertexBuffer;             /* device; position = 3 x f32 at offset 0 of each vertex */
    uint32_t VertexCount, VertexStride;
    const void* IndexBuffer;              /* device */
    uint32_t IndexCount, IndexStride;     /* stride 2 or 4; count divisible by 3 (else PT_ERROR_INVALID_ARGUMENT) */
    uint32_t Flags;                       /* PT_GEOMETRY_FLAG_OPAQUE */
    uint32_t _pad;
} PtGeometryDesc;
#define PT_GEOMETRY_FLAG_OPAQUE 0x1u      /* Source/Scene.ixx:320-324 */

#define PT_BUILD_FLAG_ALLOW_UPDATE      0x01u
#define PT_BUILD_FLAG_PREFER_FAST_TRACE 0x04u
#define PT_BUILD_FLAG_PREFER_FAST_BUILD 0x08u

/* Bottom level: one per MeshNode, one geometry per Mesh (Source/Scene.ixx:286-341 ->
 * CommandList::BuildAccelerationStructures, Source/CommandList.ixx:217-233). Returns an id
 * (reference: RTXMU accel-struct id). A compressed 8-wide BVH is built on the device, on the context stream (Morton order,
 * binary hierarchy, surface-area-greedy collapse, 8-bit child boxes); the call returns when it is built. With
 * PT_BUILD_FLAG_ALLOW_UPDATE (the reference sets it for skeletal meshes, Scene.ixx:329) the structure keeps what a refit needs. */
int  pt_build_bottom_level(PtContext* ctx, const PtGeometryDesc* geometries, uint32_t geometry_count,
                           uint32_t build_flags, uint64_t* out_blas_id);
/* D3D12_RAYTRACING_ACCELERATION_STRUCTURE_BUILD_FLAG_PERFORM_UPDATE for a skinned mesh node (Source/Scene.ixx:327-341,
 * CommandList::UpdateAccelerationStructures, Source/CommandList.ixx:235-241): same id, geometry re-read from the (moved)
 * vertex buffers. A structure built with PT_BUILD_FLAG_ALLOW_UPDATE and given the same triangle counts is REFITTED in place
 * (asynchronous: no allocation, no wait); any other is rebuilt under its id. Until pt_build_top_level has been called again
 * (Scene::CreateAccelerationStructures does so every dynamic frame) renders answer PT_ERROR_NOT_READY. */
int  pt_update_bottom_level(PtContext* ctx, uint64_t blas_id, const PtGeometryDesc* geometries, uint32_t geometry_count, uint32_t build_flags);
/* Frees a bottom level. If the live top level refers to it, that top level is dropped with it (renders answer
 * PT_ERROR_NOT_READY until the next pt_build_top_level). */
int  pt_release_bottom_level(PtContext* ctx, uint64_t blas_id);

/* SkeletalMeshSkinning::Process (Source/SkeletalMeshSkinning.ixx:38-57 -> Shaders/SkeletalMeshSkinning.hlsl:28-62).
 * skeletal_vertices: VertexPositionNormalTangentSkin[vertex_count] (48 B); skeletal_transforms: row-major float3x4 per
 * joint; vertices: the mesh's 32-byte vertex buffer (read for the old position, rewritten); motion_vectors: half4 per
 * vertex (xyz written: old - new object-space position). All device pointers; enqueued on the stream. */
int  pt_skin_mesh(PtContext* ctx, const void* skeletal_vertices, const float* skeletal_transforms, void* vertices,
                  void* motion_vectors, uint32_t vertex_count);       /* Scene::CollectGarbage, Source/Scene.ixx:382-387 */

typedef struct PtInstanceDesc {           /* D3D12_RAYTRACING_INSTANCE_DESC as filled at Source/Scene.ixx:365-377 */
    float Transform[12];
    uint32_t InstanceID;                  /* = InstanceData.FirstGeometryIndex; ObjectIndex = InstanceID + GeometryIndex */
    uint32_t InstanceMask;                /* low 8 bits; 0 hides the instance */
    uint64_t AccelerationStructure;       /* blas id */
} PtInstanceDesc;

/* Top level over all mesh-node instances (BuildTopLevelAccelerationStructure,
 * Source/RaytracingHelpers.ixx:28-74). descs is HOST memory, copied before the call returns. Rebuilds if one already exists.
 * Enqueued on the stream like the reference's build on its command list: instance records, the 8-wide tree over the instance
 * boxes and the traversal copy are made by kernels; a rebuild that needs no more room than the last one allocates nothing and
 * waits for nothing. A tree too deep for the traversal stack is reported by the next pt_sync / render call. */
int  pt_build_top_level(PtContext* ctx, const PtInstanceDesc* descs, uint32_t count, uint32_t build_flags);

typedef struct PtAccelStats {
    uint32_t InstanceCount, BottomLevelCount;
    uint64_t TriangleCount;               /* sum over instances */
    uint64_t NodeBytes, TriangleBytes;    /* device memory held by BVH nodes / triangle packets */
    uint32_t NodeSizeBytes, TriangleSizeBytes;   /* 80 (compressed 8-wide node) / 48 */
    uint32_t MaxBottomLevelDepth, TopLevelDepth; /* levels of 8-wide nodes */
    uint64_t BlobBytes;                   /* the traversal copy: instances + every referenced bottom level + top level (0 for a view) */
    uint32_t SharedScene;                 /* 1: this context views another context's scene (pt_share_scene) */
    uint32_t NormalRecords;               /* 1: hits take their vertex normals from the frame's normal records (every instance of a bottom level names the same
                                             vertex data; decided when the object data is resolved), 0: fetched through the hit's own object */
    uint64_t OwnedBottomLevelBytes;       /* nodes + packets + indices of bottom levels held OUTSIDE the traversal copy: updatable ones, and static ones no top level has
                                             seen yet. A static bottom level lives in the copy only (BlobBytes) once a top-level build has adopted it */
    uint32_t RoundObjectsInLds, RoundRecordsInLds;   /* what the fused round kernel stages in LDS behind the traversal copy: objects (resolved geometry + material) |
                                                        normal records; 0 = none: a hit's shading then fetches them from memory (measurement: which bytes reach HBM) */
} PtAccelStats;
int  pt_get_accel_stats(PtContext* ctx, PtAccelStats* out);           /* synchronises */

/* Frames in flight: several contexts (one HIP stream, one set of path queues each) rendering the SAME static scene need one copy
 * of it. After this call `ctx` renders from `source`'s acceleration structures, descriptor table and object / instance data, read-only
 * (reference: one Scene, App.cpp:372-374, whatever the number of frames the swap chain keeps in flight). `ctx` owns none of it:
 * `source` must outlive the sharing, and refuses to rebuild or release its structures while it is viewed. Synchronises both streams. */
int  pt_share_scene(PtContext* ctx, PtContext* source);

/* ------------------------------------------------------------------------------------------
 * per-frame inputs (reference: the GPUBuffers slots the caller fills before each Render,
 * Source/GBufferGeneration.ixx:51, Source/Raytracing.ixx:44; filled at Source/App.cpp:540-561,1016-1074)
 * ------------------------------------------------------------------------------------------ */
int  pt_set_camera(PtContext* ctx, const PtCamera* camera);                          /* host struct, copied */
int  pt_set_scene_data(PtContext* ctx, const PtSceneData* scene_data);               /* host struct, copied */
/* device array, referenced. Material and TextureMapInfoArray are read at every hit; VertexDesc and MeshDescriptors are resolved (and
 * checked against the descriptor heap) when the binding -- pointer or count --, the heap or the instances of the top level change, and
 * after pt_invalidate_object_data. Binding the same pointer and count again (a host that fills its GPUBuffers slots before every Render,
 * Source/App.cpp:540-561) is free and resolves nothing: a caller that REWRITES VertexDesc / MeshDescriptors of a bound array in place says
 * so with pt_invalidate_object_data, and the next render resolves and checks them again (one small kernel + one wait).
 * MeshDescriptors.Indices is checked but not dereferenced at a hit: a triangle's vertex indices are those of the index buffer given to
 * pt_build_bottom_level (kept next to its triangle packets), as DXR itself intersects the triangles of the build-time index buffer. */
int  pt_set_object_data(PtContext* ctx, const PtObjectData* device_objects, uint32_t count);
int  pt_invalidate_object_data(PtContext* ctx);
int  pt_set_instance_data(PtContext* ctx, const PtInstanceData* device_instances, uint32_t count); /* device array, referenced */

/* Multi-GPU framebuffer sharding (not a reference feature). The frame is cut into
 * horizontal bands of BandHeight rows; band b belongs to rank b % RankCount. A context renders
 * only its own bands; its textures hold those rows contiguously (local row = (b / RankCount) *
 * BandHeight + row-in-band). RNG seeds and camera rays use GLOBAL pixel coordinates, so the
 * gathered image is bit-identical to a single-GPU one. RankCount = 1 disables sharding. */
typedef struct PtSharding { uint32_t RankIndex, RankCount, BandHeight, _pad; } PtSharding;
int  pt_set_sharding(PtContext* ctx, const PtSharding* sharding);
int  pt_local_rows(const PtSharding* sharding, uint32_t frame_height, uint32_t* out_rows);
/* The gather (north_star: "image tiles shard naturally across the 8 GPUs of one node with a final RCCL gather over xGMI"; the
 * reference is single-GPU: Source/App.cpp:1157-1329 renders and presents on one device). One communicator per context, i.e. per
 * stream: frames in flight on separate streams never share one. RCCL (librccl.so.1) is loaded at the first of these calls.
 *   pt_comm_get_unique_id  ncclGetUniqueId: PT_COMM_ID_BYTES of host memory, made by one rank and handed to the others by the caller
 *                          (MPI, a file, torch.distributed's store ...)
 *   pt_comm_init           ncclCommInitRank on the context's device; returns when all `world` ranks have called it
 *   pt_comm_adopt          use an ncclComm_t the caller already owns (not destroyed by the library)
 *   pt_comm_destroy        also called by pt_destroy
 *   pt_gather_bands        enqueued on the stream: every non-root rank sends its bands (local_bands: its rows, contiguous, as
 *                          pt_set_sharding lays a texture out), the root receives each band straight into its rows of
 *                          dst_full[H][W] (grouped ncclSend / ncclRecv, one per band, no staging copy) and places its own bands
 *                          with one strided device copy. width * pixel_bytes must be a multiple of 16. Needs a communicator
 *                          whose rank / size equal the sharding's; with RankCount 1 no communicator is needed and nothing is sent.
 * Errors of RCCL come back as PT_ERROR_RCCL with ncclGetErrorString in pt_last_error.
 * pt_gather_plan is the host arithmetic behind it (no GPU, no RCCL): the messages rank sharding->RankIndex issues, in order. With
 * out == NULL only the count is returned. Tests check that the plans of all ranks pair up and tile the frame. */
typedef struct PtBandMessage {
    uint32_t Peer, IsSend, Band, _pad;    /* the other rank | 1 = ncclSend (non-root), 0 = ncclRecv (root) | global band index */
    uint64_t LocalOffset;                 /* byte offset in the sender's local texture */
    uint64_t FullOffset;                  /* byte offset in the root's full frame */
    uint64_t Bytes;
} PtBandMessage;
#define PT_COMM_ID_BYTES 128
int  pt_comm_get_unique_id(void* out_id_host);
int  pt_comm_init(PtContext* ctx, const void* id_host, uint32_t rank, uint32_t world);
int  pt_comm_adopt(PtContext* ctx, void* nccl_comm);
int  pt_comm_destroy(PtContext* ctx);
int  pt_gather_bands(PtContext* ctx, const void* local_bands, void* dst_full, uint32_t width, uint32_t height, uint32_t pixel_bytes, uint32_t root);
int  pt_gather_plan(const PtSharding* sharding, uint32_t height, uint64_t row_bytes, uint32_t root, PtBandMessage* out, uint32_t capacity, uint32_t* out_count);

/* dst_full[H][W] (pixel_bytes each) <- gathered[rank][local rows][W] (what a gather of every rank's
 * local buffer yields, rank r at byte offset rank_offsets[r]). Device pointers; enqueued on the stream. */
int  pt_deinterleave_bands(PtContext* ctx, void* dst_full, const void* gathered, const uint64_t* rank_offsets_host,
                           uint32_t rank_count, uint32_t band_height, uint32_t width, uint32_t height, uint32_t pixel_bytes);

/* ------------------------------------------------------------------------------------------
 * the two operators
 * ------------------------------------------------------------------------------------------ */
/* GBufferGeneration::Render(commandList, topLevelAccelerationStructure, constants),
 * Source/GBufferGeneration.ixx:80-117. Primary pinhole ray per pixel, closest hit, G-buffer stores. */
int  pt_gbuffer_render(PtContext* ctx, const PtGBufferConstants* constants, const PtTextures* textures);

/* Raytracing::SetConstants + Raytracing::Render (Source/Raytracing.ixx:92-112): spp x (Bounces+1)
 * path-tracing loop per pixel, wavefront-scheduled; reads the G-buffer textures written by
 * pt_gbuffer_render for bounce 0 and stores the per-pixel radiance.
 * Vertex normals are read from the objects' vertex buffers once per call, at its start (in stream order), not at every hit:
 * what a hit interpolates is what the buffers held when the call's work began. Tangents, texture coordinates and per-vertex
 * motion are read at the hit. */
int  pt_raytrace_set_constants(PtContext* ctx, const PtGraphicsSettings* settings);
int  pt_raytrace_render(PtContext* ctx, const PtTextures* textures);

/* ------------------------------------------------------------------------------------------
 * building blocks the reference's direct-lighting bridge calls on the same data
 * ------------------------------------------------------------------------------------------ */
typedef struct PtRayDesc { float Origin[3]; float TMin; float Direction[3]; float TMax; } PtRayDesc;   /* HLSL RayDesc, 32 B */
/* Shadow rays: TraceRay<RAY_FLAG_FORCE_NON_OPAQUE | RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH>(q, ray, RAY_FLAG_NONE, ~0u)
 * with the coloured-visibility IsOpaque (Shaders/RaytracingHelpers.hlsli:7-55, Shaders/ShadingHelpers.hlsli:117-159), as used by
 * GetFinalVisibility / GetConservativeVisibility (Shaders/RTXDIAppBridge.hlsli:426-439).
 * device_visibility: 4 floats per ray = visibility.rgb, then 1 if nothing was committed (unoccluded) else 0. */
int  pt_trace_visibility(PtContext* ctx, const PtRayDesc* device_rays, uint32_t count, float* device_visibility);

/* BSDFSample::Evaluate(surfaceVectors, L, V, lobeWeights, diffuse, specular) + EvaluatePDF(...), the all-lobe overloads
 * (Shaders/BxDF.hlsli:247-285) after Initialize + ComputeLobeWeights, batched. */
typedef struct PtBsdfQuery {
    float BaseColor[3], Metallic, Roughness, IOR, Transmission, IsFrontFace;     /* IsFrontFace: 0 or 1 */
    float GeometricNormal[3], ShadingNormal[3], V[3], L[3];
} PtBsdfQuery;                            /* 80 B */
typedef struct PtBsdfResult { float Diffuse[3], Specular[3], PDF, _pad; } PtBsdfResult;   /* 32 B */
int  pt_bsdf_evaluate(PtContext* ctx, const PtBsdfQuery* device_queries, uint32_t count, PtBsdfResult* device_results);

/* Measurement (no reference counterpart). Counters cover the work enqueued since the last reset;
 * reading them synchronises the stream. */
typedef struct PtCounters {
    uint64_t PrimaryRays;         /* rays cast by pt_gbuffer_render */
    uint64_t SecondaryRays;       /* rays cast by pt_raytrace_render (bounce rays actually traced) */
    uint64_t NodesVisited;        /* BVH node fetches   (only with PT_DEBUG_TRAVERSAL_STATS) */
    uint64_t TrianglesTested;     /* triangle tests     (only with PT_DEBUG_TRAVERSAL_STATS) */
    uint64_t WavefrontIterations; /* extend/shade rounds launched by the last pt_raytrace_render */
    uint64_t BvhMismatches;       /* PT_DEBUG_BRUTE_FORCE: rays whose BVH result differed from brute force (must be 0) */
    uint64_t StackOverflows;      /* traversal-stack pushes refused for lack of room (must be 0: the builders reject structures that are too deep) */
    uint64_t MaxNodesPerRay;      /* PT_DEBUG_TRAVERSAL_STATS, scenes beyond LDS: node visits of the longest ray */
} PtCounters;
int  pt_reset_counters(PtContext* ctx);
int  pt_get_counters(PtContext* ctx, PtCounters* out);
#define PT_DEBUG_TRAVERSAL_STATS 0x1u
#define PT_DEBUG_TRAVERSAL_V1   0x4u     /* bounce rays use the interleaved TLAS/BLAS traversal of k_gbuffer instead of the phase-aligned one */
#define PT_DEBUG_TRAVERSAL_PHASED 0x8u  /* bounce rays: the TLAS-walking phase-aligned schedule even when the scene is small enough for the flat one */
#define PT_DEBUG_UNFUSED_ROUNDS  0x10u    /* a round = k_shade + k_extend2 (two launches, hit records through HBM) instead of the fused k_round */
#define PT_DEBUG_LOCKSTEP        0x20u    /* scenes too large for LDS: the lock-step schedules (one tile of rays per wave) instead of the streaming form */
#define PT_DEBUG_GATHER_LOCAL_ONLY 0x40u  /* pt_gather_bands places the root's own bands and exchanges nothing: lets ONE GPU play every rank in turn (tests) */
#define PT_DEBUG_BRUTE_FORCE     0x2u     /* bounce rays test every triangle of every instance (validates the LBVH) */
#define PT_DEBUG_GATHER_SELF_EXCHANGE 0x80u /* pt_gather_bands with a world-size-1 communicator: the rank's own bands travel by ncclSend to itself + ncclRecv from
                                             itself (one pair per band, one group) into the full frame -- the grouped p2p path of a real gather on ONE GPU (tests, bench --rehearse-collective) */
int  pt_set_debug_flags(PtContext* ctx, uint32_t flags);
/* first mismatching ray under PT_DEBUG_BRUTE_FORCE: o.xyz tmin d.xyz tmax | bvh inst slot t - | brute inst slot t - */
int  pt_debug_read_mismatch(PtContext* ctx, float* out16);

/* Developer / test aid: the traversal copy of the scene (8-wide nodes of the top level and of every bottom level, triangle packets in node order,
 * their vertex indices, the instance records in API order and again in top-level leaf order, then -- after LeafInstanceOffset16 + 9 units per
 * instance -- the entry records of the streaming traversal: transform, bases and a copy of the bottom level's root node) copied to host memory,
 * with its section offsets in 16-byte units. tests/ decode it to check the structural invariants of the builder (containment, reference ranges, depth). Synchronises. */
typedef struct PtBlobLayout {
    uint32_t InstanceOffset16, NodeOffset16, TriangleOffset16, LeafInstanceOffset16;
    uint32_t InstanceCount, NodeCount, TriangleCount, Bytes;
} PtBlobLayout;
int  pt_debug_download_blob(PtContext* ctx, void* host_dst, uint64_t capacity_bytes, PtBlobLayout* out_layout);
/* Developer aid: one closest-hit ray through the one-lane two-level walk with a step log of (code, a, b, stack depth) words:
 * 1 enter instance a | 2 triangle b of instance a | 3 node group (a, b) about to be visited | 4 groups after the visit |
 * 5 popped (a, b) | 6 top-level state restored | 7 result (instance, triangle slot). Synchronises. */
int  pt_debug_trace_ray(PtContext* ctx, const PtRayDesc* host_ray, uint32_t* host_log, uint32_t log_words);

/* Per-kernel timing with HIP events recorded on the context stream around every extend / shade launch
 * issued after pt_enable_kernel_timing(ctx, 1); the getter synchronises and returns the sums since then. */
int  pt_enable_kernel_timing(PtContext* ctx, int enable);
int  pt_get_kernel_timing(PtContext* ctx, float* extend_ms, float* shade_ms, uint32_t* extend_launches, uint32_t* shade_launches);
/* the same for the fused round kernel (the default form of a round: trace + shade in one launch) */
int  pt_get_round_timing(PtContext* ctx, float* round_ms, uint32_t* round_launches);

#ifdef __cplusplus
}
#endif
#endif /* PTAMD_H */
