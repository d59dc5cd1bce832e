/*
 * pt_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's path-tracing hot path
 * (Shaders/GBufferGeneration.hlsl:main + Shaders/Raytracing.hlsl:RayGeneration,
 * DEFAULT permutation, Denoiser::None, DI off) used ONLY as the checker by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 * Nothing in the product (directx-physically-based-raytracer_amd/, include/) may
 * include, link or call this.
 *
 * PARITY STATUS: "parity unpinned" at the NVIDIA-RTX/MathLib boundary.
 * The reference has no tests/golden vectors and its
 * un-vendored math dependency (External/MathLib, ml.hlsli; version unpinned
 * in .gitmodules:1-3) is absent from /root/reference. The functions marked
 * [MathLib spec] below restate MathLib's published algorithms from the
 * literature they cite; everything else follows the reference file:line cited
 * next to it.
 *
 * Struct layouts follow SURVEY.md Appendix A (byte-exact with the reference's
 * Source/CommonShaderData.ixx:15-39, Material.ixx:10-38, Vertex.ixx:30-50,
 * Camera.ixx:16-36, Raytracing.ixx:151-166).
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- reference data layouts (names as in the reference) ------------------ */

typedef struct OrVertexDesc {            /* Source/Vertex.ixx:30-36 */
    uint32_t Stride, _pad[3];
    uint32_t Normal, Tangent, TextureCoordinates[2]; /* byte offsets, ~0u = absent */
} OrVertexDesc;                          /* 32 B */

typedef struct OrMeshDescriptors {       /* Source/CommonShaderData.ixx:28-30 */
    uint32_t Vertices, Indices, MotionVectors, _pad;
} OrMeshDescriptors;                     /* 16 B */

typedef struct OrMaterial {              /* Source/Material.ixx:12-20 */
    float BaseColor[4];
    float EmissiveStrength;
    float EmissiveColor[3];
    float Metallic, Roughness, IOR, Transmission;
    uint32_t AlphaMode;                  /* 0 Opaque, 1 Mask, 2 Blend */
    float AlphaCutoff;
    uint32_t _pad[2];
} OrMaterial;                            /* 64 B */

typedef struct OrTextureMapInfo {        /* Source/Material.ixx:35-38 */
    uint32_t Descriptor, TextureCoordinateIndex, _pad[2];
} OrTextureMapInfo;                      /* 16 B */

typedef struct OrObjectData {            /* Source/CommonShaderData.ixx:34-39 */
    OrVertexDesc VertexDesc;
    OrMeshDescriptors MeshDescriptors;
    OrMaterial Material;
    OrTextureMapInfo TextureMapInfoArray[7];
} OrObjectData;                          /* 224 B */

typedef struct OrInstanceData {          /* Source/CommonShaderData.ixx:22-26 */
    uint32_t FirstGeometryIndex, _pad[3];
    float PreviousObjectToWorld[12];     /* row-major 3x4 */
    float ObjectToWorld[12];
} OrInstanceData;                        /* 112 B */

typedef struct OrSceneData {             /* Source/CommonShaderData.ixx:15-20 */
    uint32_t IsStatic, IsEnvironmentLightTextureCubeMap;
    uint32_t EnvironmentLightTextureDescriptor, _pad;
    float EnvironmentLightColor[4];
    float EnvironmentLightTransform[12];
} OrSceneData;                           /* 80 B */

typedef struct OrCamera {                /* Source/Camera.ixx:16-36 */
    uint32_t IsNormalizedDepthReversed;
    float PreviousPosition[3], Position[3], _pad0;
    float RightDirection[3], _pad1;
    float UpDirection[3], _pad2;
    float ForwardDirection[3];
    float ApertureRadius, NearDepth, FarDepth;
    float Jitter[2];
    float PreviousWorldToView[16], PreviousViewToProjection[16],
          PreviousWorldToProjection[16], PreviousProjectionToView[16],
          PreviousViewToWorld[16], WorldToProjection[16], ProjectionToView[16],
          ViewToWorld[16];               /* XMFLOAT4X4, row-vector convention */
} OrCamera;                              /* 608 B */

typedef struct OrGraphicsSettings {      /* Source/Raytracing.ixx:151-166 */
    uint32_t RenderSize[2];
    uint32_t FrameIndex, Bounces, SamplesPerPixel;
    float ThroughputThreshold;
    uint32_t IsRussianRouletteEnabled, IsShaderExecutionReorderingEnabled, IsDIEnabled;
    uint32_t Denoiser;
    uint32_t ExtFlags;                   /* reference: padding word 0. build-side switches, see OR_EXT_* */
    uint32_t _pad;
    uint32_t SHARC[8];
} OrGraphicsSettings;                    /* 80 B */

#define OR_EXT_LAMBERTIAN_ONLY 0x1u      /* BASELINE.json config C1 switch */

typedef struct OrGBufferConstants {      /* Source/GBufferGeneration.ixx:46-49 */
    uint32_t RenderSize[2];
    uint32_t Flags;
} OrGBufferConstants;

/* Shaders/GBufferGeneration.hlsl:9-28 */
enum {
    OR_GB_Position = 0x1, OR_GB_FlatNormal = 0x2, OR_GB_GeometricNormal = 0x4,
    OR_GB_LinearDepth = 0x8, OR_GB_NormalizedDepth = 0x10, OR_GB_MotionVector = 0x20,
    OR_GB_DiffuseAlbedo = 0x40, OR_GB_SpecularAlbedo = 0x80, OR_GB_Albedo = 0xC0,
    OR_GB_NormalRoughness = 0x100, OR_GB_Radiance = 0x200,
    OR_GB_Geometry = 0x1 | 0x2 | 0x4 | 0x8 | 0x10 | 0x20 | 0x100,
    OR_GB_Material = 0x400 | 0xC0 | 0x100 | 0x200
};

/* G-buffer targets: linear row-major host arrays in the reference's texture
 * formats (Source/App.cpp:438-455). Any pointer may be NULL (not bound). */
typedef struct OrGBufferTextures {
    float*    Position;            /* RGBA32F   16 B/px */
    int16_t*  FlatNormal;          /* RG16_SNORM 4 B/px */
    int16_t*  GeometricNormal;     /* RG16_SNORM 4 B/px */
    float*    LinearDepth;         /* R32F */
    float*    NormalizedDepth;     /* R32F */
    uint16_t* MotionVector;        /* RGBA16F   8 B/px */
    uint8_t*  BaseColorMetalness;  /* RGBA8_UNORM */
    uint16_t* DiffuseAlbedo;       /* RGBA16F, only with OR_GB_DiffuseAlbedo */
    uint16_t* SpecularAlbedo;      /* RGBA16F, only with OR_GB_SpecularAlbedo */
    int16_t*  NormalRoughness;     /* RGBA16_SNORM 8 B/px */
    uint16_t* IOR;                 /* R16F */
    uint8_t*  Transmission;        /* R8_UNORM */
    uint16_t* Radiance;            /* RGBA16F   8 B/px */
    float*    RadianceF32;         /* build-side extra (not a reference texture): RGBA32F copy of what
                                      or_raytrace_render stores to Radiance, before fp16 rounding; may be NULL */
    uint16_t* Diffuse;             /* RGBA16F, NRD output (Raytracing.hlsl:400-413) */
    uint16_t* Specular;            /* RGBA16F */
    uint16_t* SpecularHitDistance; /* R16F, DLSS-RR output (:395-398) */
} OrGBufferTextures;

/* ---- acceleration-structure inputs (DXR-shaped) -------------------------- */

typedef struct OrGeometryDesc {          /* D3D12_RAYTRACING_GEOMETRY_DESC subset, Source/RaytracingHelpers.ixx:76-105 */
    const void* Vertices;  uint32_t VertexCount, VertexStride;   /* position = 3 x f32 at offset 0 */
    const void* Indices;   uint32_t IndexCount,  IndexStride;    /* 2 or 4 */
    uint32_t Flags;                                              /* 1 = OPAQUE */
    uint32_t _pad;
} OrGeometryDesc;

typedef struct OrBlasDesc { uint32_t FirstGeometry, GeometryCount; } OrBlasDesc;

typedef struct OrInstanceDesc {          /* D3D12_RAYTRACING_INSTANCE_DESC subset, Source/Scene.ixx:365-377 */
    float Transform[12];
    uint32_t InstanceID;
    uint32_t InstanceMask;
    uint32_t Blas;
    uint32_t _pad;
} OrInstanceDesc;

/* descriptor-heap analogue: ObjectData.MeshDescriptors.* / TextureMapInfo.Descriptor /
 * SceneData.EnvironmentLightTextureDescriptor index this table.
 * Kind 0 buffer : Stride = element size of a typed buffer (index buffers: 2 = R16_UINT, 4 = R32_UINT; raw: 0).
 * Kind 1 Texture2D, Kind 2 TextureCube (faces +X,-X,+Y,-Y,+Z,-Z contiguous): Bytes = width | height << 32,
 *        Stride = texel format (OR_FMT_*). Mip 0 only (the path samples with SampleLevel(..., 0)). */
typedef struct OrHeapEntry { const void* Ptr; uint64_t Bytes; uint32_t Stride; uint32_t Kind; } OrHeapEntry;
enum { OR_KIND_BUFFER = 0, OR_KIND_TEXTURE2D = 1, OR_KIND_TEXTURECUBE = 2 };
enum { OR_FMT_RGBA8_UNORM = 0, OR_FMT_RGBA8_UNORM_SRGB = 1, OR_FMT_RGBA32_FLOAT = 2 };

typedef struct OrScene OrScene;

/* accel_mode: 0 = brute force over every triangle of every instance (no BVH at all),
 *             1 = oracle's own top-down median-split BVH (independent of the product's LBVH). */
OrScene* or_scene_create(const OrGeometryDesc* geoms, uint32_t n_geoms,
                         const OrBlasDesc* blas, uint32_t n_blas,
                         const OrInstanceDesc* inst, uint32_t n_inst,
                         const OrObjectData* objects, uint32_t n_objects,
                         const OrInstanceData* inst_data,
                         const OrHeapEntry* heap, uint32_t n_heap,
                         int accel_mode);
void or_scene_destroy(OrScene*);

/* y0..y1 restrict the rows processed (global pixel coordinates are kept) so a bounded
 * sample of a big frame can be timed; pass 0, RenderSize[1] for the whole frame.
 * Returns the number of rays traced (primary rays for gbuffer, secondary for raytrace). */
uint64_t or_gbuffer_render(const OrScene*, const OrCamera*, const OrSceneData*,
                           const OrGBufferConstants*, const OrGBufferTextures*,
                           uint32_t y0, uint32_t y1, int n_threads);
uint64_t or_raytrace_render(const OrScene*, const OrCamera*, const OrSceneData*,
                            const OrGraphicsSettings*, const OrGBufferTextures*,
                            uint32_t y0, uint32_t y1, int n_threads);

/* ---- unit-level entry points for known-answer tests ---------------------- */
uint32_t or_rng_init(uint32_t px, uint32_t py, uint32_t frame);
float    or_rng_float(uint32_t* state);
void     or_sincos_2pi(float u, float* s, float* c);
uint16_t or_f32_to_f16(float f);
float    or_f16_to_f32(uint16_t h);
int16_t  or_f32_to_snorm16(float f);
float    or_snorm16_to_f32(int16_t v);
uint8_t  or_f32_to_unorm8(float f);
void     or_oct_encode(const float n[3], float out[2]);
void     or_oct_decode(const float p[2], float out[3]);
/* ray-triangle: returns 1 on hit and fills t,u,v (u weights v1, v weights v2) */
int      or_ray_triangle(const float o[3], const float d[3], float tmin, float tmax,
                         const float v0[3], const float v1[3], const float v2[3],
                         float* t, float* u, float* v);
/* BSDF: one call = ComputeLobeWeights + Sample + EvaluatePDF + Evaluate (single-lobe forms).
 * mat = {base.rgb, metallic, roughness, ior, transmission}; returns 1 if Sample() returned true. */
int      or_bsdf_sample(const float mat[7], int front_face, const float Ng[3], const float Ns[3],
                        const float V[3], const float rnd[4], uint32_t ext_flags,
                        float L[3], int* lobe, float* pdf, float f[3], float weights[3]);
void     or_env_term_rtg(const float f0[3], float NoV, float roughness, float out[3]);
void     or_safe_spawn(const float v[9], const float bary[2], const float o2w[12], const float w2o[12],
                       float objPos[3], float wldPos[3], float objN[3], float wldN[3], float* offset);
void     or_invert_3x4(const float m[12], float out[12]);
/* BSDFSample::Evaluate / EvaluatePDF, all-lobe overloads (BxDF.hlsli:247-285; consumer: RTXDIAppBridge.hlsli:249-263).
 * q: 20 floats per query = base.rgb metallic roughness ior transmission frontFace(0/1) Ng.xyz Ns.xyz V.xyz L.xyz
 * r:  8 floats per query = diffuse.rgb specular.rgb pdf 0 */
void     or_bsdf_evaluate(const float* q, uint32_t count, float* r);
/* TraceRay<FORCE_NON_OPAQUE | ACCEPT_FIRST_HIT_AND_END_SEARCH> with the coloured-visibility IsOpaque overload
 * (RaytracingHelpers.hlsli:7-55, ShadingHelpers.hlsli:117-159, RTXDIAppBridge.hlsli:418-439).
 * rays: 8 floats per ray = origin.xyz tmin dir.xyz tmax; out: 4 floats = visibility.rgb, 1 if nothing was committed else 0 */
void     or_trace_visibility(const OrScene* scene, const float* rays, uint32_t count, float* out);
/* SkeletalMeshSkinning.hlsl:28-62. skeletal: VertexPositionNormalTangentSkin (48 B: pos f32x3, normal i16x3, tangent i16x3,
 * joints u16x4, weights f32x4); transforms: row-major 3x4 per joint; vertices: 32-B vertex (in/out); motion: half4 per vertex */
void     or_skin_mesh(const void* skeletal, const float* transforms, void* vertices, uint16_t* motion, uint32_t count);
/* SampleLevel(sampler, uv, 0) of a heap texture: bilinear, wrap addressing, sRGB decode per texel */
void     or_texture_sample(const OrHeapEntry* tex, float u, float v, float out[4]);
void     or_cube_sample(const OrHeapEntry* tex, const float dir[3], float out[4]);

#ifdef __cplusplus
}
#endif
#endif
