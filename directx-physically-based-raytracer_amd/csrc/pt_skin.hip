// pt_skin.hip -- GPU skinning for dynamic scenes.
//   k_skin  <- Shaders/SkeletalMeshSkinning.hlsl:28-62 (4-joint linear blend, inverse-transpose normals, per-vertex
//              object-space motion vectors consumed by CalculateMotionVector, Shaders/GBufferGeneration.hlsl:73-84)
// One thread per vertex, 256 threads per block like the reference ([numthreads(256,1,1)]). HBM-bound: 48 B read +
// 32 B read-modify-write + 8 B written per vertex.
#include "pt_internal.hpp"

namespace pt {

struct alignas(4) SkeletalVertex {     // VertexPositionNormalTangentSkin, Source/Vertex.ixx:52-57
    float Position[3];
    int16_t Normal[3], Tangent[3];
    uint16_t Joints[4];
    float Weights[4];
};
static_assert(sizeof(SkeletalVertex) == 48, "layout");

__device__ __forceinline__ int16_t pack_r16_snorm(float v) { return (int16_t)(clampf(v, -1.0f, 1.0f) * 32767.0f); }   // Packing.hlsli:3-6

__global__ __launch_bounds__(256) void k_skin(const SkeletalVertex* __restrict__ skeletal, const float* __restrict__ transforms,
                                              uint8_t* __restrict__ vertices, uint16_t* __restrict__ motion, uint32_t count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const SkeletalVertex sv = skeletal[i];
    const float w[4] = { sv.Weights[0], sv.Weights[1], sv.Weights[2], 1.0f - sv.Weights[0] - sv.Weights[1] - sv.Weights[2] };
    float M[12];
    #pragma unroll
    for (int k = 0; k < 12; k++) M[k] = 0.0f;
    #pragma unroll
    for (int j = 0; j < 4; j++) {
        const float* T = transforms + 12 * (size_t)sv.Joints[j];
        #pragma unroll
        for (int k = 0; k < 12; k++) M[k] = mad(w[j], T[k], M[k]);
    }
    const float* pos = sv.Position;
    const v3 p = V3(sop3t(M[0], pos[0], M[1], pos[1], M[2], pos[2], M[3]), sop3t(M[4], pos[0], M[5], pos[1], M[6], pos[2], M[7]), sop3t(M[8], pos[0], M[9], pos[1], M[10], pos[2], M[11]));
    float* dv = (float*)(vertices + 32 * (size_t)i);
    const v3 mv = V3(dv[0] - p.x, dv[1] - p.y, dv[2] - p.z);
    const v3 n = V3(unpack_r16_snorm(sv.Normal[0]), unpack_r16_snorm(sv.Normal[1]), unpack_r16_snorm(sv.Normal[2]));
    const v3 t = V3(unpack_r16_snorm(sv.Tangent[0]), unpack_r16_snorm(sv.Tangent[1]), unpack_r16_snorm(sv.Tangent[2]));
    const v3 r0 = V3(M[0], M[1], M[2]), r1 = V3(M[4], M[5], M[6]), r2 = V3(M[8], M[9], M[10]);
    const v3 v = cross(r0, r1);                                   // Math::InverseTranspose, Math.hlsli:23-27
    const float d = dot(v, r2);
    v3 i0 = cross(r1, r2), i1 = cross(r2, r0);
    i0 = V3(i0.x / d, i0.y / d, i0.z / d); i1 = V3(i1.x / d, i1.y / d, i1.z / d);
    const v3 i2 = V3(v.x / d, v.y / d, v.z / d);
    const v3 nn = normalize(V3(dot(i0, n), dot(i1, n), dot(i2, n)));
    const v3 tt = normalize(V3(dot(r0, t), dot(r1, t), dot(r2, t)));
    dv[0] = p.x; dv[1] = p.y; dv[2] = p.z;
    int16_t* q = (int16_t*)(vertices + 32 * (size_t)i + 12);
    q[0] = pack_r16_snorm(nn.x); q[1] = pack_r16_snorm(nn.y); q[2] = pack_r16_snorm(nn.z);
    q[3] = pack_r16_snorm(tt.x); q[4] = pack_r16_snorm(tt.y); q[5] = pack_r16_snorm(tt.z);
    motion[4 * (size_t)i + 0] = f32_to_f16(mv.x); motion[4 * (size_t)i + 1] = f32_to_f16(mv.y); motion[4 * (size_t)i + 2] = f32_to_f16(mv.z);
}

hipError_t launch_skin(hipStream_t stream, const void* skeletal, const float* transforms, void* vertices, void* motion, uint32_t count)
{
    if (!count) return hipSuccess;
    k_skin<<<(count + 255) / 256, 256, 0, stream>>>((const SkeletalVertex*)skeletal, transforms, (uint8_t*)vertices, (uint16_t*)motion, count);
    return hipGetLastError();
}

} // namespace pt
