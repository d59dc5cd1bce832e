// pt_api.hip -- the C ABI of include/ptamd.h: context, descriptor heap, acceleration-structure
// lifecycle, per-frame constants and the two render operators. No exception crosses this file.
#include "pt_internal.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>

using namespace pt;

struct PtContext { Context c; };

static thread_local std::string g_createError;

static int fail(Context* c, int status, const std::string& msg)
{
    if (c) c->lastError = msg; else g_createError = msg;
    return status;
}
static int fail_hip(Context* c, hipError_t e, const char* what)
{
    return fail(c, e == hipErrorOutOfMemory ? PT_ERROR_OUT_OF_MEMORY : PT_ERROR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define API_HIP(ctx, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail_hip(ctx, e_, #expr); } while (0)
#define API_ARG(ctx, cond, msg) do { if (!(cond)) return fail(ctx, PT_ERROR_INVALID_ARGUMENT, msg); } while (0)

// worldToObject: inverse of the affine 3x4 evaluated in double, rounded once to float (DESIGN.md
// "Arithmetic spec"; DXR derives CommittedWorldToObject3x4 inside the driver).
static void invert_3x4(const float m[12], float out[12])
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

extern "C" {

int pt_abi_version(void) { return PTAMD_ABI_VERSION; }

int pt_create(int device_ordinal, PtContext** out_ctx)
{
    if (!out_ctx) return fail(nullptr, PT_ERROR_INVALID_ARGUMENT, "out_ctx is NULL");
    *out_ctx = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(nullptr, PT_ERROR_NO_DEVICE, "no HIP device visible: the path tracer has no CPU fallback");
    if (device_ordinal < 0 || device_ordinal >= n) return fail(nullptr, PT_ERROR_INVALID_ARGUMENT, "device ordinal out of range");
    if ((e = hipSetDevice(device_ordinal)) != hipSuccess) return fail_hip(nullptr, e, "hipSetDevice");
    PtContext* p = new (std::nothrow) PtContext();
    if (!p) return fail(nullptr, PT_ERROR_OUT_OF_MEMORY, "host allocation failed");
    p->c.device = device_ordinal;
    if ((e = hipMalloc((void**)&p->c.counters, sizeof(DeviceCounters))) != hipSuccess) { delete p; return fail_hip(nullptr, e, "hipMalloc(counters)"); }
    hipMemset(p->c.counters, 0, sizeof(DeviceCounters));
    *out_ctx = p;
    return PT_OK;
}

static void free_blas(Blas& b) { if (b.nodes) hipFree(b.nodes); if (b.tris) hipFree(b.tris); if (b.rootBounds) hipFree(b.rootBounds); b = Blas(); }
static void free_tlas(Tlas& t) { if (t.nodes) hipFree(t.nodes); if (t.instances) hipFree(t.instances); t = Tlas(); }

void pt_destroy(PtContext* ctx)
{
    if (!ctx) return;
    Context& c = ctx->c;
    hipSetDevice(c.device);
    hipStreamSynchronize(c.stream);
    for (auto& kv : c.blas) free_blas(kv.second);
    free_tlas(c.tlas);
    if (c.heapDev) hipFree(c.heapDev);
    if (c.srgbLutDev) hipFree(c.srgbLutDev);
    if (c.blobDev) hipFree(c.blobDev);
    for (int k = 0; k < 2; k++) {
        PathQueue& q = c.queue[k];
        void* ptrs[6] = { q.s0, q.s1, q.s2, q.r0, q.r1, q.hit };
        for (void* p : ptrs) if (p) hipFree(p);
    }
    if (c.graphExec) hipGraphExecDestroy(c.graphExec);
    if (c.frameConstants) hipFree(c.frameConstants);
    if (c.pixelAux) hipFree(c.pixelAux);
    if (c.queueCounts) hipFree(c.queueCounts);
    if (c.counters) hipFree(c.counters);
    for (auto e : c.evExtend) hipEventDestroy(e);
    for (auto e : c.evRound) hipEventDestroy(e);
    for (auto e : c.evShade) hipEventDestroy(e);
    delete ctx;
}

const char* pt_last_error(const PtContext* ctx) { return ctx ? ctx->c.lastError.c_str() : g_createError.c_str(); }

int pt_set_stream(PtContext* ctx, void* hip_stream)
{
    if (!ctx) return PT_ERROR_INVALID_ARGUMENT;
    ctx->c.stream = (hipStream_t)hip_stream;
    return PT_OK;
}

int pt_sync(PtContext* ctx)
{
    if (!ctx) return PT_ERROR_INVALID_ARGUMENT;
    API_HIP(&ctx->c, hipSetDevice(ctx->c.device));
    API_HIP(&ctx->c, hipStreamSynchronize(ctx->c.stream));
    return PT_OK;
}

// ---- descriptor heap ------------------------------------------------------------------------
int pt_heap_resize(PtContext* ctx, uint32_t descriptor_count)
{
    if (!ctx) return PT_ERROR_INVALID_ARGUMENT;
    ctx->c.heapHost.resize(descriptor_count, HeapEntry{ nullptr, 0, 0, 0 });
    ctx->c.heapDirty = true;
    return PT_OK;
}

int pt_heap_set_buffer(PtContext* ctx, uint32_t descriptor, const void* device_ptr, uint64_t bytes, uint32_t stride)
{
    if (!ctx) return PT_ERROR_INVALID_ARGUMENT;
    Context& c = ctx->c;
    API_ARG(&c, descriptor < c.heapHost.size(), "descriptor index beyond pt_heap_resize");
    API_ARG(&c, stride == 0 || stride == 2 || stride == 4 || stride == 8, "buffer stride must be 0 (raw), 2 / 4 (typed index buffer) or 8 (structured half4 motion vectors)");
    c.heapHost[descriptor] = HeapEntry{ device_ptr, bytes, stride, kKindBuffer };
    c.heapDirty = true;
    return PT_OK;
}

int pt_heap_set_texture(PtContext* ctx, uint32_t descriptor, const void* device_ptr, uint32_t width, uint32_t height, uint32_t format, uint32_t is_cube)
{
    if (!ctx) return PT_ERROR_INVALID_ARGUMENT;
    Context& c = ctx->c;
    API_ARG(&c, descriptor < c.heapHost.size(), "descriptor index beyond pt_heap_resize");
    API_ARG(&c, device_ptr && width && height, "texture pointer / size is null");
    API_ARG(&c, format <= PT_FORMAT_R32G32B32A32_FLOAT, "unsupported texture format");
    API_ARG(&c, !is_cube || width == height, "cube faces must be square");
    c.heapHost[descriptor] = HeapEntry{ device_ptr, (uint64_t)width | ((uint64_t)height << 32), format, is_cube ? kKindTextureCube : kKindTexture2D };
    c.heapDirty = true;
    return PT_OK;
}

static int upload_heap(Context& c)
{
    if (!c.srgbLutDev) {                                  // sRGB -> linear table, evaluated in double (arithmetic spec)
        float lut[256];
        for (int i = 0; i < 256; i++) { double v = i / 255.0; lut[i] = (float)(v <= 0.04045 ? v / 12.92 : pow((v + 0.055) / 1.055, 2.4)); }
        API_HIP(&c, hipMalloc((void**)&c.srgbLutDev, sizeof lut));
        API_HIP(&c, hipMemcpy(c.srgbLutDev, lut, sizeof lut, hipMemcpyHostToDevice));
    }
    if (!c.heapDirty) return PT_OK;
    uint32_t n = (uint32_t)c.heapHost.size();
    c.heapHasTextures = false;
    for (const HeapEntry& e : c.heapHost) if (e.kind != kKindBuffer) c.heapHasTextures = true;
    if (n > c.heapDevCap) {
        if (c.heapDev) hipFree(c.heapDev);
        c.heapDev = nullptr; c.heapDevCap = 0;
        API_HIP(&c, hipMalloc((void**)&c.heapDev, sizeof(HeapEntry) * n));
        c.heapDevCap = n;
    }
    if (n) API_HIP(&c, hipMemcpyAsync(c.heapDev, c.heapHost.data(), sizeof(HeapEntry) * n, hipMemcpyHostToDevice, c.stream));
    API_HIP(&c, hipStreamSynchronize(c.stream));     // heapHost may change right after
    c.heapDirty = false;
    return PT_OK;
}

// ---- acceleration structures ----------------------------------------------------------------
int pt_build_bottom_level(PtContext* ctx, const PtGeometryDesc* geometries, uint32_t geometry_count, uint32_t build_flags, uint64_t* out_blas_id)
{
    (void)build_flags;
    if (!ctx) return PT_ERROR_INVALID_ARGUMENT;
    Context& c = ctx->c;
    API_ARG(&c, out_blas_id, "out_blas_id is NULL");
    API_ARG(&c, geometries || geometry_count == 0, "geometries is NULL");
    for (uint32_t g = 0; g < geometry_count; g++) {
        // same argument checks as CreateGeometryDesc, Source/RaytracingHelpers.ixx:82-89
        API_ARG(&c, geometries[g].IndexStride == 2 || geometries[g].IndexStride == 4, "Triangle index format must be either uint16 or uint32");
        API_ARG(&c, geometries[g].IndexCount % 3 == 0, "Triangle index count must be divisible by 3");
        API_ARG(&c, geometries[g].IndexCount == 0 || (geometries[g].VertexBuffer && geometries[g].IndexBuffer), "geometry buffer is NULL");
        API_ARG(&c, geometries[g].VertexStride >= 12, "vertex stride must cover a float3 position");
    }
    API_HIP(&c, hipSetDevice(c.device));
    Blas b;
    hipError_t e = build_blas_device(geometries, geometry_count, c.stream, b);
    if (e != hipSuccess) { free_blas(b); return fail_hip(&c, e, "bottom-level build"); }
    uint64_t id = c.nextBlasId++;
    c.blas[id] = b;
    *out_blas_id = id;
    return PT_OK;
}

int pt_update_bottom_level(PtContext* ctx, uint64_t blas_id, const PtGeometryDesc* geometries, uint32_t geometry_count, uint32_t build_flags)
{
    (void)build_flags;
    if (!ctx) return PT_ERROR_INVALID_ARGUMENT;
    Context& c = ctx->c;
    auto it = c.blas.find(blas_id);
    API_ARG(&c, it != c.blas.end(), "unknown bottom-level id");
    API_ARG(&c, geometries || geometry_count == 0, "geometries is NULL");
    for (uint32_t g = 0; g < geometry_count; g++) {
        API_ARG(&c, geometries[g].IndexStride == 2 || geometries[g].IndexStride == 4, "Triangle index format must be either uint16 or uint32");
        API_ARG(&c, geometries[g].IndexCount % 3 == 0, "Triangle index count must be divisible by 3");
    }
    API_HIP(&c, hipSetDevice(c.device));
    // PERFORM_UPDATE (Source/Scene.ixx:327-341, CommandList::UpdateAccelerationStructures): the skinned vertices moved.
    // The LBVH is rebuilt (Morton order may change) under the same id; a device build of a skinned mesh is cheaper than
    // keeping a stale topology. The TLAS must be rebuilt afterwards (pt_build_top_level), as the reference does.
    API_HIP(&c, hipStreamSynchronize(c.stream));
    Blas b;
    hipError_t e = build_blas_device(geometries, geometry_count, c.stream, b);
    if (e != hipSuccess) { free_blas(b); return fail_hip(&c, e, "bottom-level update"); }
    free_blas(it->second);
    it->second = b;
    c.haveTlas = false;                                  // instance records point at the freed BLAS
    return PT_OK;
}

int pt_skin_mesh(PtContext* ctx, const void* skeletal_vertices, const float* skeletal_transforms, void* vertices, void* motion_vectors, uint32_t vertex_count)
{
    if (!ctx) return PT_ERROR_INVALID_ARGUMENT;
    Context& c = ctx->c;
    API_ARG(&c, vertex_count == 0 || (skeletal_vertices && skeletal_transforms && vertices && motion_vectors), "a skinning buffer is NULL");
    API_HIP(&c, hipSetDevice(c.device));
    API_HIP(&c, launch_skin(c.stream, skeletal_vertices, skeletal_transforms, vertices, motion_vectors, vertex_count));
    return PT_OK;
}

int pt_release_bottom_level(PtContext* ctx, uint64_t blas_id)
{
    if (!ctx) return PT_ERROR_INVALID_ARGUMENT;
    Context& c = ctx->c;
    auto it = c.blas.find(blas_id);
    API_ARG(&c, it != c.blas.end(), "unknown bottom-level id");
    hipStreamSynchronize(c.stream);
    free_blas(it->second);
    c.blas.erase(it);
    return PT_OK;
}

int pt_build_top_level(PtContext* ctx, const PtInstanceDesc* descs, uint32_t count, uint32_t build_flags)
{
    (void)build_flags;
    if (!ctx) return PT_ERROR_INVALID_ARGUMENT;
    Context& c = ctx->c;
    API_ARG(&c, descs || count == 0, "descs is NULL");
    API_HIP(&c, hipSetDevice(c.device));
    std::vector<InstanceRecord> rec(count ? count : 1);
    std::vector<const float*> bounds(count ? count : 1);
    std::vector<BlobPiece> pieces; std::vector<uint32_t> pieceOfInstance(count ? count : 1);
    std::map<uint64_t, uint32_t> pieceOfBlas;
    uint32_t blobNodes = count > 1 ? count - 1 : 1, blobTris = 0;       // TLAS nodes come first in the blob
    uint64_t tris = 0;
    for (uint32_t i = 0; i < count; i++) {
        auto it = c.blas.find(descs[i].AccelerationStructure);
        API_ARG(&c, it != c.blas.end(), "instance refers to an unknown bottom-level id");
        InstanceRecord& r = rec[i];
        memcpy(r.objectToWorld, descs[i].Transform, sizeof(float) * 12);
        invert_3x4(descs[i].Transform, r.worldToObject);
        r.nodes = it->second.nodes; r.tris = it->second.tris;
        r.instanceID = descs[i].InstanceID & 0xFFFFFFu;
        r.mask = descs[i].InstanceMask & 0xFFu;
        r.triCount = it->second.triCount; r._pad = 0;
        bounds[i] = it->second.rootBounds;
        tris += it->second.triCount;
        auto pb = pieceOfBlas.find(descs[i].AccelerationStructure);
        if (pb == pieceOfBlas.end()) {
            const Blas& b = it->second;
            pieces.push_back(BlobPiece{ b.nodes, b.tris, b.nodeCount, b.triCount, blobNodes, blobTris });
            blobNodes += b.nodeCount; blobTris += b.triCount;
            pb = pieceOfBlas.emplace(descs[i].AccelerationStructure, (uint32_t)pieces.size() - 1).first;
        }
        pieceOfInstance[i] = pb->second;
    }
    API_HIP(&c, hipStreamSynchronize(c.stream));      // nothing may still be traversing the old TLAS
    free_tlas(c.tlas); c.haveTlas = false;
    Tlas t;
    const float** dBounds = nullptr;
    hipError_t e = hipMalloc((void**)&t.instances, sizeof(InstanceRecord) * (count ? count : 1));
    if (e == hipSuccess) e = hipMalloc((void**)&dBounds, sizeof(float*) * (count ? count : 1));
    if (e == hipSuccess && count) e = hipMemcpy(t.instances, rec.data(), sizeof(InstanceRecord) * count, hipMemcpyHostToDevice);
    if (e == hipSuccess && count) e = hipMemcpy(dBounds, bounds.data(), sizeof(float*) * count, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = build_tlas_device(t.instances, dBounds, count, c.stream, t);
    if (c.blobDev) { hipFree(c.blobDev); c.blobDev = nullptr; c.blob = BlobView{}; }
    if (e == hipSuccess) e = build_blob_device(t, dBounds, pieces, pieceOfInstance, c.stream, &c.blobDev, &c.blob);
    if (dBounds) hipFree(dBounds);
    if (e != hipSuccess) { free_tlas(t); return fail_hip(&c, e, "top-level build"); }
    t.triangleCount = tris;
    c.tlas = t; c.haveTlas = true;
    return PT_OK;
}

int pt_get_accel_stats(PtContext* ctx, PtAccelStats* out)
{
    if (!ctx || !out) return PT_ERROR_INVALID_ARGUMENT;
    Context& c = ctx->c;
    memset(out, 0, sizeof *out);
    out->InstanceCount = c.tlas.instanceCount;
    out->BottomLevelCount = (uint32_t)c.blas.size();
    out->TriangleCount = c.tlas.triangleCount;
    out->NodeSizeBytes = sizeof(BvhNode); out->TriangleSizeBytes = sizeof(TriPacket);
    uint64_t nb = (uint64_t)c.tlas.nodeCount * sizeof(BvhNode), tb = 0;
    for (auto& kv : c.blas) { nb += (uint64_t)kv.second.nodeCount * sizeof(BvhNode); tb += (uint64_t)kv.second.triCount * sizeof(TriPacket); }
    out->NodeBytes = nb; out->TriangleBytes = tb;
    return PT_OK;
}

// ---- per-frame inputs -----------------------------------------------------------------------
int pt_set_camera(PtContext* ctx, const PtCamera* camera)
{
    if (!ctx) return PT_ERROR_INVALID_ARGUMENT;
    API_ARG(&ctx->c, camera, "camera is NULL");
    ctx->c.camera = *camera; ctx->c.haveCamera = true;
    return PT_OK;
}
int pt_set_scene_data(PtContext* ctx, const PtSceneData* scene_data)
{
    if (!ctx) return PT_ERROR_INVALID_ARGUMENT;
    API_ARG(&ctx->c, scene_data, "scene_data is NULL");
    ctx->c.sceneData = *scene_data; ctx->c.haveSceneData = true;
    return PT_OK;
}
int pt_set_object_data(PtContext* ctx, const PtObjectData* device_objects, uint32_t count)
{
    if (!ctx) return PT_ERROR_INVALID_ARGUMENT;
    API_ARG(&ctx->c, device_objects || count == 0, "device_objects is NULL");
    ctx->c.objects = device_objects; ctx->c.objectCount = count;
    return PT_OK;
}
int pt_set_instance_data(PtContext* ctx, const PtInstanceData* device_instances, uint32_t count)
{
    if (!ctx) return PT_ERROR_INVALID_ARGUMENT;
    ctx->c.instanceData = device_instances; ctx->c.instanceDataCount = count;
    return PT_OK;
}

int pt_set_sharding(PtContext* ctx, const PtSharding* s)
{
    if (!ctx) return PT_ERROR_INVALID_ARGUMENT;
    API_ARG(&ctx->c, s && s->RankCount >= 1 && s->RankIndex < s->RankCount && s->BandHeight >= 1, "invalid sharding");
    ctx->c.sharding = *s;
    return PT_OK;
}

int pt_local_rows(const PtSharding* s, uint32_t frame_height, uint32_t* out_rows)
{
    if (!s || !out_rows || s->RankCount == 0 || s->BandHeight == 0 || s->RankIndex >= s->RankCount) return PT_ERROR_INVALID_ARGUMENT;
    uint32_t rows = 0;
    for (uint32_t b = s->RankIndex; (uint64_t)b * s->BandHeight < frame_height; b += s->RankCount) {
        uint32_t y0 = b * s->BandHeight;
        rows += (frame_height - y0 < s->BandHeight) ? frame_height - y0 : s->BandHeight;
    }
    *out_rows = rows;
    return PT_OK;
}

int pt_deinterleave_bands(PtContext* ctx, void* dst_full, const void* gathered, const uint64_t* rank_offsets_host, uint32_t rank_count,
                          uint32_t band_height, uint32_t width, uint32_t height, uint32_t pixel_bytes)
{
    if (!ctx) return PT_ERROR_INVALID_ARGUMENT;
    Context& c = ctx->c;
    API_ARG(&c, dst_full && gathered && rank_offsets_host && rank_count >= 1 && band_height >= 1, "invalid de-interleave arguments");
    API_ARG(&c, pixel_bytes % 4 == 0, "pixel size must be a multiple of 4 bytes");
    API_HIP(&c, hipSetDevice(c.device));
    API_ARG(&c, rank_count <= 64, "at most 64 ranks");
    hipError_t e = launch_deinterleave(c.stream, dst_full, gathered, rank_offsets_host, rank_count, band_height, width, height, pixel_bytes);
    if (e != hipSuccess) return fail_hip(&c, e, "de-interleave");
    return PT_OK;
}

// ---- operators ------------------------------------------------------------------------------
static int make_views(Context& c, uint32_t width, uint32_t height, SceneView& sv, FrameView& fv, bool needFrameInputs = true)
{
    if (!c.haveTlas) return fail(&c, PT_ERROR_NOT_READY, "no top-level acceleration structure: call pt_build_top_level first");
    if (needFrameInputs && (!c.haveCamera || !c.haveSceneData)) return fail(&c, PT_ERROR_NOT_READY, "camera / scene data not set");
    if (!c.objects && c.tlas.instanceCount) return fail(&c, PT_ERROR_NOT_READY, "object data not set");
    int s = upload_heap(c);
    if (s != PT_OK) return s;
    sv.accel.tlasNodes = c.tlas.nodes; sv.accel.instances = c.tlas.instances; sv.accel.instanceCount = c.tlas.instanceCount;
    sv.objects = c.objects; sv.objectCount = c.objectCount;
    sv.instanceData = c.instanceData;
    sv.heap = c.heapDev; sv.heapCount = (uint32_t)c.heapHost.size();
    sv.srgbLut = c.srgbLutDev;
    fv.width = width; fv.height = height;
    fv.rankIndex = c.sharding.RankIndex; fv.rankCount = c.sharding.RankCount; fv.bandHeight = c.sharding.BandHeight;
    pt_local_rows(&c.sharding, height, &fv.localRows);
    return PT_OK;
}

int pt_gbuffer_render(PtContext* ctx, const PtGBufferConstants* constants, const PtTextures* textures)
{
    if (!ctx) return PT_ERROR_INVALID_ARGUMENT;
    Context& c = ctx->c;
    API_ARG(&c, constants && textures, "constants / textures is NULL");
    API_HIP(&c, hipSetDevice(c.device));
    SceneView sv; FrameView fv; memset(&sv, 0, sizeof sv); memset(&fv, 0, sizeof fv);
    int s = make_views(c, constants->RenderSize[0], constants->RenderSize[1], sv, fv);
    if (s != PT_OK) return s;
    API_HIP(&c, launch_gbuffer(c, sv, fv, constants->Flags, *textures));
    return PT_OK;
}

int pt_raytrace_set_constants(PtContext* ctx, const PtGraphicsSettings* settings)
{
    if (!ctx) return PT_ERROR_INVALID_ARGUMENT;
    API_ARG(&ctx->c, settings, "settings is NULL");
    ctx->c.settings = *settings; ctx->c.haveSettings = true;
    return PT_OK;
}

int pt_raytrace_render(PtContext* ctx, const PtTextures* tx)
{
    if (!ctx) return PT_ERROR_INVALID_ARGUMENT;
    Context& c = ctx->c;
    API_ARG(&c, tx, "textures is NULL");
    if (!c.haveSettings) return fail(&c, PT_ERROR_NOT_READY, "call pt_raytrace_set_constants first");
    API_ARG(&c, c.settings.Denoiser <= PT_DENOISER_NRD_RELAX, "unknown Denoiser value");
    API_ARG(&c, !c.settings.IsDIEnabled, "IsDIEnabled needs the RTXDI passes, which are out of scope");
    API_ARG(&c, !(c.settings.Denoiser >= PT_DENOISER_NRD_REBLUR) || (tx->Diffuse && tx->Specular), "NRD modes write Textures.Diffuse / Textures.Specular: not bound");
    API_ARG(&c, c.settings.SamplesPerPixel < 65536 && c.settings.Bounces < 32768, "SamplesPerPixel / Bounces out of range");
    API_ARG(&c, tx->Position && tx->FlatNormal && tx->GeometricNormal && tx->BaseColorMetalness && tx->NormalRoughness && tx->IOR
                 && tx->Transmission && tx->Radiance, "a G-buffer texture the path tracer reads is not bound (Raytracing::Textures)");
    API_HIP(&c, hipSetDevice(c.device));
    SceneView sv; FrameView fv; memset(&sv, 0, sizeof sv); memset(&fv, 0, sizeof fv);
    int s = make_views(c, c.settings.RenderSize[0], c.settings.RenderSize[1], sv, fv);
    if (s != PT_OK) return s;
    if (c.settings.Bounces == 0) return PT_OK;          // reference: the pass is not dispatched, Source/App.cpp:1277-1279
    API_HIP(&c, launch_raytrace(c, sv, fv, *tx));
    return PT_OK;
}

int pt_trace_visibility(PtContext* ctx, const PtRayDesc* device_rays, uint32_t count, float* device_visibility)
{
    if (!ctx) return PT_ERROR_INVALID_ARGUMENT;
    Context& c = ctx->c;
    API_ARG(&c, count == 0 || (device_rays && device_visibility), "ray / visibility buffer is NULL");
    API_HIP(&c, hipSetDevice(c.device));
    SceneView sv; FrameView fv; memset(&sv, 0, sizeof sv); memset(&fv, 0, sizeof fv);
    int s = make_views(c, 1, 1, sv, fv, false);                    // only the scene half of the views is needed
    if (s != PT_OK) return s;
    API_HIP(&c, launch_visibility(c, sv, device_rays, count, device_visibility));
    return PT_OK;
}

int pt_bsdf_evaluate(PtContext* ctx, const PtBsdfQuery* device_queries, uint32_t count, PtBsdfResult* device_results)
{
    if (!ctx) return PT_ERROR_INVALID_ARGUMENT;
    Context& c = ctx->c;
    API_ARG(&c, count == 0 || (device_queries && device_results), "query / result buffer is NULL");
    API_HIP(&c, hipSetDevice(c.device));
    API_HIP(&c, launch_bsdf_evaluate(c.stream, (const float*)device_queries, count, (float*)device_results));
    return PT_OK;
}

// ---- measurement ----------------------------------------------------------------------------
int pt_reset_counters(PtContext* ctx)
{
    if (!ctx) return PT_ERROR_INVALID_ARGUMENT;
    API_HIP(&ctx->c, hipSetDevice(ctx->c.device));
    API_HIP(&ctx->c, hipMemsetAsync(ctx->c.counters, 0, sizeof(DeviceCounters), ctx->c.stream));
    return PT_OK;
}

int pt_get_counters(PtContext* ctx, PtCounters* out)
{
    if (!ctx || !out) return PT_ERROR_INVALID_ARGUMENT;
    Context& c = ctx->c;
    API_HIP(&c, hipSetDevice(c.device));
    DeviceCounters d;
    API_HIP(&c, hipMemcpyAsync(&d, c.counters, sizeof d, hipMemcpyDeviceToHost, c.stream));
    API_HIP(&c, hipStreamSynchronize(c.stream));
    memset(out, 0, sizeof *out);
    out->PrimaryRays = d.primaryRays; out->SecondaryRays = d.secondaryRays;
    out->NodesVisited = d.nodesVisited; out->TrianglesTested = d.trianglesTested;
    out->WavefrontIterations = c.lastIterations;
    out->BvhMismatches = d.mismatchCount;
    return PT_OK;
}

int pt_debug_read_mismatch(PtContext* ctx, float* out16)
{
    if (!ctx || !out16) return PT_ERROR_INVALID_ARGUMENT;
    Context& c = ctx->c;
    API_HIP(&c, hipSetDevice(c.device));
    DeviceCounters d;
    API_HIP(&c, hipMemcpyAsync(&d, c.counters, sizeof d, hipMemcpyDeviceToHost, c.stream));
    API_HIP(&c, hipStreamSynchronize(c.stream));
    memcpy(out16, d.mismatchRay, sizeof(float) * 16);
    return PT_OK;
}

int pt_set_debug_flags(PtContext* ctx, uint32_t flags)
{
    if (!ctx) return PT_ERROR_INVALID_ARGUMENT;
    ctx->c.debugFlags = flags;
    return PT_OK;
}

int pt_enable_kernel_timing(PtContext* ctx, int enable)
{
    if (!ctx) return PT_ERROR_INVALID_ARGUMENT;
    ctx->c.timing = enable != 0;
    ctx->c.nExtend = ctx->c.nShade = ctx->c.nRound = 0;          // accumulation restarts
    return PT_OK;
}

int pt_get_kernel_timing(PtContext* ctx, float* extend_ms, float* shade_ms, uint32_t* extend_launches, uint32_t* shade_launches)
{
    if (!ctx) return PT_ERROR_INVALID_ARGUMENT;
    Context& c = ctx->c;
    API_HIP(&c, hipSetDevice(c.device));
    API_HIP(&c, hipStreamSynchronize(c.stream));
    float te = 0.0f, ts = 0.0f;
    if (c.timing) {
        for (uint32_t k = 0; k < c.nExtend && 2 * k + 1 < c.evExtend.size(); k++) { float ms = 0; hipEventElapsedTime(&ms, c.evExtend[2 * k], c.evExtend[2 * k + 1]); te += ms; }
        for (uint32_t k = 0; k < c.nShade && 2 * k + 1 < c.evShade.size(); k++) { float ms = 0; hipEventElapsedTime(&ms, c.evShade[2 * k], c.evShade[2 * k + 1]); ts += ms; }
    }
    if (extend_ms) *extend_ms = te;
    if (shade_ms) *shade_ms = ts;
    if (extend_launches) *extend_launches = c.nExtend;
    if (shade_launches) *shade_launches = c.nShade;
    return PT_OK;
}

int pt_get_round_timing(PtContext* ctx, float* round_ms, uint32_t* round_launches)
{
    if (!ctx) return PT_ERROR_INVALID_ARGUMENT;
    Context& c = ctx->c;
    API_HIP(&c, hipSetDevice(c.device));
    API_HIP(&c, hipStreamSynchronize(c.stream));
    float t = 0.0f;
    if (c.timing)
        for (uint32_t k = 0; k < c.nRound && 2 * k + 1 < c.evRound.size(); k++) { float ms = 0; hipEventElapsedTime(&ms, c.evRound[2 * k], c.evRound[2 * k + 1]); t += ms; }
    if (round_ms) *round_ms = t;
    if (round_launches) *round_launches = c.nRound;
    return PT_OK;
}

} // extern "C"
