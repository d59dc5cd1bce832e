"""Byte-exact numpy mirrors of the reference's GPU data layouts.

Reference: Source/CommonShaderData.ixx:15-39 (SceneData, InstanceData, MeshDescriptors, ObjectData),
Source/Material.ixx:10-38, Source/Vertex.ixx:30-50, Source/Camera.ixx:16-36,
Source/Raytracing.ixx:151-166 (GraphicsSettings), Source/GBufferGeneration.ixx:28-49.
The C side of the same layouts is include/ptamd.h (static_asserted there).
"""
import numpy as np

NONE = 0xFFFFFFFF  # "~0u": absent attribute / descriptor

VERTEX = np.dtype({  # VertexPositionNormalTangentTexture, 32 B
    "names": ["Position", "Normal", "Tangent", "TexCoord0", "TexCoord1"],
    "formats": [("<f4", 3), ("<i2", 3), ("<i2", 3), ("<f2", 2), ("<f2", 2)],
    "offsets": [0, 12, 18, 24, 28], "itemsize": 32})

SKELETAL_VERTEX = np.dtype({  # VertexPositionNormalTangentSkin, 48 B (Source/Vertex.ixx:52-57)
    "names": ["Position", "Normal", "Tangent", "Joints", "Weights"],
    "formats": [("<f4", 3), ("<i2", 3), ("<i2", 3), ("<u2", 4), ("<f4", 4)],
    "offsets": [0, 12, 18, 24, 32], "itemsize": 48})

VERTEX_DESC = np.dtype({
    "names": ["Stride", "Normal", "Tangent", "TexCoord"],
    "formats": ["<u4", "<u4", "<u4", ("<u4", 2)],
    "offsets": [0, 16, 20, 24], "itemsize": 32})

MESH_DESCRIPTORS = np.dtype({
    "names": ["Vertices", "Indices", "MotionVectors"],
    "formats": ["<u4", "<u4", "<u4"], "offsets": [0, 4, 8], "itemsize": 16})

MATERIAL = np.dtype({
    "names": ["BaseColor", "EmissiveStrength", "EmissiveColor", "Metallic", "Roughness", "IOR",
              "Transmission", "AlphaMode", "AlphaCutoff"],
    "formats": [("<f4", 4), "<f4", ("<f4", 3), "<f4", "<f4", "<f4", "<f4", "<u4", "<f4"],
    "offsets": [0, 16, 20, 32, 36, 40, 44, 48, 52], "itemsize": 64})

TEXTURE_MAP_INFO = np.dtype({
    "names": ["Descriptor", "TextureCoordinateIndex"], "formats": ["<u4", "<u4"],
    "offsets": [0, 4], "itemsize": 16})

OBJECT_DATA = np.dtype({
    "names": ["VertexDesc", "MeshDescriptors", "Material", "TextureMapInfoArray"],
    "formats": [VERTEX_DESC, MESH_DESCRIPTORS, MATERIAL, (TEXTURE_MAP_INFO, 7)],
    "offsets": [0, 32, 48, 112], "itemsize": 224})

INSTANCE_DATA = np.dtype({
    "names": ["FirstGeometryIndex", "PreviousObjectToWorld", "ObjectToWorld"],
    "formats": ["<u4", ("<f4", (3, 4)), ("<f4", (3, 4))],
    "offsets": [0, 16, 64], "itemsize": 112})

SCENE_DATA = np.dtype({
    "names": ["IsStatic", "IsEnvironmentLightTextureCubeMap", "EnvironmentLightTextureDescriptor",
              "EnvironmentLightColor", "EnvironmentLightTransform"],
    "formats": ["<u4", "<u4", "<u4", ("<f4", 4), ("<f4", (3, 4))],
    "offsets": [0, 4, 8, 16, 32], "itemsize": 80})

_M = ("<f4", (4, 4))
CAMERA = np.dtype({
    "names": ["IsNormalizedDepthReversed", "PreviousPosition", "Position", "RightDirection",
              "UpDirection", "ForwardDirection", "ApertureRadius", "NearDepth", "FarDepth", "Jitter",
              "PreviousWorldToView", "PreviousViewToProjection", "PreviousWorldToProjection",
              "PreviousProjectionToView", "PreviousViewToWorld", "WorldToProjection",
              "ProjectionToView", "ViewToWorld"],
    "formats": ["<u4", ("<f4", 3), ("<f4", 3), ("<f4", 3), ("<f4", 3), ("<f4", 3), "<f4", "<f4", "<f4",
                ("<f4", 2), _M, _M, _M, _M, _M, _M, _M, _M],
    "offsets": [0, 4, 16, 32, 48, 64, 76, 80, 84, 88, 96, 160, 224, 288, 352, 416, 480, 544],
    "itemsize": 608})

GRAPHICS_SETTINGS = np.dtype({
    "names": ["RenderSize", "FrameIndex", "Bounces", "SamplesPerPixel", "ThroughputThreshold",
              "IsRussianRouletteEnabled", "IsShaderExecutionReorderingEnabled", "IsDIEnabled", "Denoiser",
              "ExtFlags"],
    "formats": [("<u4", 2), "<u4", "<u4", "<u4", "<f4", "<u4", "<u4", "<u4", "<u4", "<u4"],
    "offsets": [0, 8, 12, 16, 20, 24, 28, 32, 36, 40], "itemsize": 80})

EXT_LAMBERTIAN_ONLY = 0x1   # build-side switch living in the reference's padding word (config C1)

GBUFFER_CONSTANTS = np.dtype({
    "names": ["RenderSize", "Flags"], "formats": [("<u4", 2), "<u4"], "offsets": [0, 8], "itemsize": 12})


class GBufferFlags:  # Shaders/GBufferGeneration.hlsl:9-28
    Position = 0x1
    FlatNormal = 0x2
    GeometricNormal = 0x4
    LinearDepth = 0x8
    NormalizedDepth = 0x10
    MotionVector = 0x20
    DiffuseAlbedo = 0x40
    SpecularAlbedo = 0x80
    Albedo = 0xC0
    NormalRoughness = 0x100
    Radiance = 0x200
    Geometry = 0x1 | 0x2 | 0x4 | 0x8 | 0x10 | 0x20 | 0x100
    Material = 0x400 | 0xC0 | 0x100 | 0x200
    # what App.cpp:1224 passes when the denoiser is None
    DefaultNoDenoiser = 0xFFFFFFFF & ~0xC0


# G-buffer texture formats (Source/App.cpp:438-455): name -> (numpy dtype, channels)
GBUFFER_FORMATS = {
    "Position": ("<f4", 4),            # RGBA32F
    "FlatNormal": ("<i2", 2),          # RG16_SNORM
    "GeometricNormal": ("<i2", 2),     # RG16_SNORM
    "LinearDepth": ("<f4", 1),         # R32F
    "NormalizedDepth": ("<f4", 1),     # R32F
    "MotionVector": ("<u2", 4),        # RGBA16F (raw half bits)
    "BaseColorMetalness": ("u1", 4),   # RGBA8_UNORM
    "DiffuseAlbedo": ("<u2", 4),       # RGBA16F, written when a denoiser is selected
    "SpecularAlbedo": ("<u2", 4),      # RGBA16F, written when a denoiser is selected
    "NormalRoughness": ("<i2", 4),     # RGBA16_SNORM
    "IOR": ("<u2", 1),                 # R16F
    "Transmission": ("u1", 1),         # R8_UNORM
    "Radiance": ("<u2", 4),            # RGBA16F
}
GBUFFER_ORDER = list(GBUFFER_FORMATS.keys())
# denoiser-facing outputs of the path tracer (Source/App.cpp:475-482)
DENOISER_FORMATS = {"Diffuse": ("<u2", 4), "Specular": ("<u2", 4), "SpecularHitDistance": ("<u2", 1)}
DENOISER_NONE, DENOISER_DLSS_RR, DENOISER_NRD_REBLUR, DENOISER_NRD_RELAX = 0, 1, 2, 3     # Source/Denoiser.ixx:8


def default_material():
    """Material() defaults, Source/Material.ixx:13-19."""
    m = np.zeros((), MATERIAL)
    m["BaseColor"] = (0, 0, 0, 1)
    m["EmissiveStrength"] = 1
    m["Roughness"] = 0.5
    m["IOR"] = 1.5
    m["AlphaCutoff"] = 0.5
    return m
