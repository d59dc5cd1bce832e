// pt_demo.cpp -- C++ host of the path: builds the BASELINE Cornell box, drives libptamd.so through the
// reference-shaped operators of ptamd.hpp (GBufferGeneration / Raytracing / RaytracingHelpers), times it and
// optionally dumps the radiance for the parity test (tests/test_host_cpp.py compares it with the oracle).
//
//   pt_demo [--width W] [--height H] [--spp S] [--bounces B] [--frames N] [--out file.bin] [--ranks R]
//
// --ranks R: one process per GPU. The parent (which never touches a GPU) starts R children `--rank r --world R --id-file F`; rank 0
// makes the RCCL unique id and leaves it in F, the others pick it up; every rank renders its 16-row bands (BandSharding) and rank 0
// assembles the frame with pt_gather_bands (grouped ncclSend / ncclRecv over xGMI) -- the timed loop includes the gather. R may be 1
// (the N > 1 code path on a one-GPU box); R > the number of visible GPUs is refused (RCCL wants one device per rank).
//
// The scene construction mirrors scenes.py:cornell_box(variant="ggx") value for value (same double-precision
// expressions), so both hosts feed the library identical bytes. Build: make -C ../csrc ../pt_demo (hipcc, host code).
#include <hip/hip_runtime.h>
#include <signal.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "ptamd.hpp"

using namespace ptamd;

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

struct Vertex {                    // VertexPositionNormalTangentTexture, Source/Vertex.ixx:38-50
    float Position[3];
    int16_t Normal[3], Tangent[3];
    uint16_t TexCoord[2][2];
};
static_assert(sizeof(Vertex) == 32, "layout");

static int16_t snorm16(double v)   // MathLib float2_to_snorm_16_16 (host side, Source/Vertex.ixx:17-22)
{
    v = std::fmin(std::fmax(v, -1.0), 1.0) * 32767.0;
    return (int16_t)(v >= 0 ? std::floor(v + 0.5) : std::ceil(v - 0.5));
}

struct HostMesh { std::vector<Vertex> vertices; std::vector<uint16_t> indices; PtMaterial material; };

static PtMaterial make_material(float r, float g, float b, float er = 0, float eg = 0, float eb = 0, float strength = 1, float metallic = 0,
                                float roughness = 0.5f, float ior = 1.5f, float transmission = 0)
{
    PtMaterial m{};                // Material() defaults, Source/Material.ixx:13-19
    m.BaseColor[0] = r; m.BaseColor[1] = g; m.BaseColor[2] = b; m.BaseColor[3] = 1;
    m.EmissiveStrength = strength; m.EmissiveColor[0] = er; m.EmissiveColor[1] = eg; m.EmissiveColor[2] = eb;
    m.Metallic = metallic; m.Roughness = roughness; m.IOR = ior; m.Transmission = transmission;
    m.AlphaMode = 0; m.AlphaCutoff = 0.5f;
    return m;
}

static HostMesh quad(const double p[4][3], const double n[3], const PtMaterial& mat)
{
    HostMesh m; m.material = mat;
    for (int i = 0; i < 4; i++) {
        Vertex v{};
        for (int k = 0; k < 3; k++) { v.Position[k] = (float)p[i][k]; v.Normal[k] = snorm16(n[k]); }
        m.vertices.push_back(v);
    }
    m.indices = { 0, 1, 2, 0, 2, 3 };
    return m;
}

static HostMesh box(const PtMaterial& mat)   // unit cube, 24 vertices with face normals (scenes.py:box_mesh)
{
    HostMesh m; m.material = mat;
    for (int axis = 0; axis < 3; axis++)
        for (int si = 0; si < 2; si++) {
            const double sgn = si ? 1.0 : -1.0;
            const int a = (axis + 1) % 3, b = (axis + 2) % 3;
            const int sa[4] = { -1, 1, 1, -1 }, sb[4] = { -1, -1, 1, 1 };
            double corners[4][3];
            for (int c = 0; c < 4; c++) { corners[c][axis] = 0.5 * sgn; corners[c][a] = 0.5 * sa[c]; corners[c][b] = 0.5 * sb[c]; }
            const uint16_t base = (uint16_t)m.vertices.size();
            for (int c = 0; c < 4; c++) {
                const double* p = corners[sgn < 0 ? 3 - c : c];
                Vertex v{};
                for (int k = 0; k < 3; k++) { v.Position[k] = (float)p[k]; v.Normal[k] = snorm16(k == axis ? sgn : 0.0); }
                m.vertices.push_back(v);
            }
            const uint16_t q[6] = { 0, 1, 2, 0, 2, 3 };
            for (uint16_t i : q) m.indices.push_back((uint16_t)(base + i));
        }
    return m;
}

static void trs(float out[12], double tx, double ty, double tz, double yawDeg, double sx, double sy, double sz)
{   // scenes.py:trs with pitch = 0: T * R_y(yaw) * S, column-vector 3x4
    const double a = yawDeg * (M_PI / 180.0), cy = std::cos(a), sn = std::sin(a);
    const double ry[3][3] = { { cy, 0, sn }, { 0, 1, 0 }, { -sn, 0, cy } };
    const double s[3] = { sx, sy, sz }, t[3] = { tx, ty, tz };
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) out[4 * i + j] = (float)(ry[i][j] * s[j]); out[4 * i + 3] = (float)t[i]; }
}

template <typename T> static T* upload(const std::vector<T>& v)
{
    T* d = nullptr;
    HIP_OK(hipMalloc((void**)&d, v.size() * sizeof(T) + 16));
    HIP_OK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

// --ranks R: start one child per rank and wait for them (no GPU call has been made in this process, and none will be).
// The unique id travels through a file in a directory of our own (mkdtemp: mode 0700, unpredictable name). The first rank that fails
// ends the run: the others would wait for it inside ncclCommInitRank or a grouped receive for ever, so they are killed.
static int launch_ranks(int argc, char** argv, uint32_t ranks)
{
    char dirTemplate[] = "/tmp/pt_demo_XXXXXX";
    if (!mkdtemp(dirTemplate)) { perror("mkdtemp"); return 2; }
    const std::string dir = dirTemplate, idFile = dir + "/id";
    std::vector<pid_t> pids;
    int rc = 0;
    for (uint32_t r = 0; r < ranks && rc == 0; r++) {
        const pid_t pid = fork();
        if (pid < 0) { perror("fork"); rc = 2; break; }
        if (pid == 0) {
            std::vector<std::string> args(argv, argv + argc);
            for (const char* extra : { "--rank", "", "--world", "", "--id-file", "" }) args.push_back(extra);
            args[args.size() - 5] = std::to_string(r); args[args.size() - 3] = std::to_string(ranks); args[args.size() - 1] = idFile;
            std::vector<char*> cargs;
            for (auto& a : args) cargs.push_back(a.data());
            cargs.push_back(nullptr);
            execv("/proc/self/exe", cargs.data());
            perror("execv"); _exit(127);
        }
        pids.push_back(pid);
    }
    size_t left = pids.size();
    while (left) {
        int st = 0;
        const pid_t p = waitpid(-1, &st, 0);
        if (p < 0) { if (errno == EINTR) continue; break; }
        auto it = std::find(pids.begin(), pids.end(), p);
        if (it == pids.end()) continue;
        *it = -1; left--;
        const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 2;
        if (code != 0 && rc == 0) {
            rc = code;
            for (pid_t q : pids) if (q > 0) kill(q, SIGTERM);                     // exactly the children started above
        }
    }
    unlink((idFile + ".tmp").c_str()); unlink(idFile.c_str()); rmdir(dir.c_str());
    return rc;
}

static std::vector<uint8_t> exchange_unique_id(uint32_t rank, const std::string& idFile)
{
    std::vector<uint8_t> id(PT_COMM_ID_BYTES);
    if (rank == 0) {
        id = BandSharding::UniqueId();
        const std::string tmp = idFile + ".tmp";
        FILE* fp = fopen(tmp.c_str(), "wb");
        if (!fp || fwrite(id.data(), 1, id.size(), fp) != id.size()) throw std::runtime_error("cannot write " + tmp);
        fclose(fp);
        if (rename(tmp.c_str(), idFile.c_str()) != 0) throw std::runtime_error("cannot publish " + idFile);
        return id;
    }
    for (int tries = 0; tries < 1200; tries++) {                    // rank 0 publishes with an atomic rename: a file that exists is complete
        if (FILE* fp = fopen(idFile.c_str(), "rb")) {
            const size_t n = fread(id.data(), 1, id.size(), fp);
            fclose(fp);
            if (n == id.size()) return id;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
    }
    throw std::runtime_error("rank 0 never published the RCCL unique id");
}

int main(int argc, char** argv)
{
    uint32_t W = 1920, H = 1080, spp = 4, bounces = 8, frames = 10, ranks = 0, rank = 0, world = 1;
    std::string out, idFile;
    for (int i = 1; i + 1 < argc; i += 2) {
        std::string k = argv[i];
        if (k == "--width") W = atoi(argv[i + 1]); else if (k == "--height") H = atoi(argv[i + 1]);
        else if (k == "--spp") spp = atoi(argv[i + 1]); else if (k == "--bounces") bounces = atoi(argv[i + 1]);
        else if (k == "--frames") frames = atoi(argv[i + 1]); else if (k == "--out") out = argv[i + 1];
        else if (k == "--ranks") ranks = atoi(argv[i + 1]); else if (k == "--rank") rank = atoi(argv[i + 1]);
        else if (k == "--world") world = atoi(argv[i + 1]); else if (k == "--id-file") idFile = argv[i + 1];
    }
    const bool sharded = !idFile.empty();
    if (ranks && !sharded) return launch_ranks(argc, argv, ranks);
    try {
        int deviceCount = 0;
        HIP_OK(hipGetDeviceCount(&deviceCount));
        if ((int)world > deviceCount) { fprintf(stderr, "pt_demo: %u ranks need %u GPUs, %d visible\n", world, world, deviceCount); return 4; }
        HIP_OK(hipSetDevice((int)rank));
        hipStream_t stream; HIP_OK(hipStreamCreate(&stream));
        CommandList commandList((int)rank, stream);
        BandSharding sharding;
        if (sharded) sharding.Join(commandList, rank, world, exchange_unique_id(rank, idFile).data());
        const uint32_t localRows = sharding.LocalRows(H);

        // ---- scene (scenes.py:cornell_box, variant "ggx")
        const PtMaterial white = make_material(0.73f, 0.73f, 0.73f), red = make_material(0.65f, 0.05f, 0.05f), green = make_material(0.12f, 0.45f, 0.15f);
        const double floorP[4][3] = { { -1, -1, -1 }, { -1, -1, 1 }, { 1, -1, 1 }, { 1, -1, -1 } }, up[3] = { 0, 1, 0 };
        const double ceilP[4][3] = { { -1, 1, -1 }, { 1, 1, -1 }, { 1, 1, 1 }, { -1, 1, 1 } }, down[3] = { 0, -1, 0 };
        const double backP[4][3] = { { -1, -1, 1 }, { -1, 1, 1 }, { 1, 1, 1 }, { 1, -1, 1 } }, toCam[3] = { 0, 0, -1 };
        const double leftP[4][3] = { { -1, -1, -1 }, { -1, 1, -1 }, { -1, 1, 1 }, { -1, -1, 1 } }, px[3] = { 1, 0, 0 };
        const double rightP[4][3] = { { 1, -1, -1 }, { 1, -1, 1 }, { 1, 1, 1 }, { 1, 1, -1 } }, nx[3] = { -1, 0, 0 };
        const double lightP[4][3] = { { -0.25, 0, -0.25 }, { 0.25, 0, -0.25 }, { 0.25, 0, 0.25 }, { -0.25, 0, 0.25 } };
        std::vector<HostMesh> meshes = {
            quad(floorP, up, white), quad(ceilP, down, white), quad(backP, toCam, white), quad(leftP, px, red), quad(rightP, nx, green),
            quad(lightP, down, make_material(0.78f, 0.78f, 0.78f, 1, 1, 1, 15.0f)),
            box(make_material(0.95f, 0.93f, 0.88f, 0, 0, 0, 1, 1.0f, 0.05f)), box(make_material(0.73f, 0.73f, 0.73f, 0, 0, 0, 1, 0, 0.2f)) };
        float xf[8][12];
        for (int i = 0; i < 5; i++) trs(xf[i], 0, 0, 0, 0, 1, 1, 1);
        trs(xf[5], 0, 0.998, 0.1, 0, 1, 1, 1);
        trs(xf[6], -0.35, -0.4, 0.35, -18.0, 0.6, 1.2, 0.6);
        trs(xf[7], 0.35, -0.7, -0.25, 15.0, 0.6, 0.6, 0.6);

        // ---- buffers, descriptor heap, ObjectData / InstanceData (App::UpdateScene, Source/App.cpp:1028-1074)
        const uint32_t n = (uint32_t)meshes.size();
        ThrowIfFailed(commandList.Context, pt_heap_resize(commandList.Context, 2 * n));
        std::vector<PtObjectData> objectData(n); std::vector<PtInstanceData> instanceData(n);
        std::vector<uint64_t> blas(n); std::vector<PtInstanceDesc> instanceDescs(n);
        for (uint32_t i = 0; i < n; i++) {
            Vertex* dv = upload(meshes[i].vertices); uint16_t* di = upload(meshes[i].indices);
            ThrowIfFailed(commandList.Context, pt_heap_set_buffer(commandList.Context, 2 * i, dv, meshes[i].vertices.size() * sizeof(Vertex), 0));
            ThrowIfFailed(commandList.Context, pt_heap_set_buffer(commandList.Context, 2 * i + 1, di, meshes[i].indices.size() * 2, 2));
            PtObjectData& od = objectData[i]; memset(&od, 0, sizeof od);
            od.VertexDesc.Stride = sizeof(Vertex);
            od.VertexDesc.AttributeOffsets.Normal = 12; od.VertexDesc.AttributeOffsets.Tangent = ~0u;
            od.VertexDesc.AttributeOffsets.TextureCoordinates[0] = od.VertexDesc.AttributeOffsets.TextureCoordinates[1] = ~0u;
            od.MeshDescriptors.Vertices = 2 * i; od.MeshDescriptors.Indices = 2 * i + 1; od.MeshDescriptors.MotionVectors = ~0u;
            od.Material = meshes[i].material;
            for (auto& t : od.TextureMapInfoArray) t.Descriptor = ~0u;
            PtInstanceData& id = instanceData[i]; memset(&id, 0, sizeof id);
            id.FirstGeometryIndex = i;
            memcpy(id.ObjectToWorld, xf[i], 48); memcpy(id.PreviousObjectToWorld, xf[i], 48);
            // Scene::CreateAccelerationStructures: one BLAS per mesh node, one geometry per mesh
            const PtGeometryDesc g = RaytracingHelpers::CreateGeometryDesc({ dv, meshes[i].vertices.size(), sizeof(Vertex) }, { di, meshes[i].indices.size(), 2 },
                                                                           PT_GEOMETRY_FLAG_OPAQUE);
            blas[i] = RaytracingHelpers::BuildBottomLevelAccelerationStructure(commandList, std::span(&g, 1), PT_BUILD_FLAG_PREFER_FAST_TRACE);
            PtInstanceDesc& d = instanceDescs[i]; memset(&d, 0, sizeof d);
            memcpy(d.Transform, xf[i], 48); d.InstanceID = i; d.InstanceMask = ~0u; d.AccelerationStructure = blas[i];
        }
        RaytracingHelpers::TopLevelAccelerationStructure tlas;
        RaytracingHelpers::BuildTopLevelAccelerationStructure(commandList, PT_BUILD_FLAG_PREFER_FAST_TRACE, instanceDescs, false, tlas);
        PtObjectData* dObjects = upload(objectData); PtInstanceData* dInstances = upload(instanceData);

        // ---- camera (scenes.py:make_camera((0,0,-1.95), hfov 90, near 0.01, far inf)) and scene data
        PtCamera cam{}; const double aspect = (double)W / (double)H;
        const double rightLen = std::tan((90.0 * (M_PI / 180.0)) / 2), upLen = rightLen / aspect, nearD = 0.01;
        cam.IsNormalizedDepthReversed = 1;
        cam.Position[2] = cam.PreviousPosition[2] = -1.95f;
        cam.RightDirection[0] = (float)rightLen; cam.UpDirection[1] = (float)upLen; cam.ForwardDirection[2] = 1;
        cam.NearDepth = (float)nearD; cam.FarDepth = std::numeric_limits<float>::infinity();
        double w2v[4][4] = { { 1, 0, 0, 0 }, { 0, 1, 0, 0 }, { 0, 0, 1, 0 }, { 0, 0, 1.95, 1 } }, v2p[4][4] = {}, w2p[4][4] = {};
        v2p[0][0] = 1 / rightLen; v2p[1][1] = aspect / rightLen; v2p[2][3] = 1; v2p[3][2] = nearD;
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) for (int k = 0; k < 4; k++) w2p[i][j] += w2v[i][k] * v2p[k][j];
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
            cam.WorldToProjection[4 * i + j] = cam.PreviousWorldToProjection[4 * i + j] = (float)w2p[i][j];
            cam.PreviousWorldToView[4 * i + j] = (float)w2v[i][j]; cam.PreviousViewToProjection[4 * i + j] = (float)v2p[i][j];
        }
        PtSceneData sd{}; sd.IsStatic = 1; sd.EnvironmentLightTextureDescriptor = ~0u; sd.EnvironmentLightColor[3] = 1;
        sd.EnvironmentLightTransform[0] = sd.EnvironmentLightTransform[5] = sd.EnvironmentLightTransform[10] = 1;

        // ---- textures (Source/App.cpp:438-455 formats)
        const size_t px_ = (size_t)W * std::max(localRows, 1u), fullPx = (size_t)W * H;     // a rank's textures hold its own rows
        PtTextures tx{}; float* radianceF32 = nullptr;
        auto alloc = [&](size_t bytes) { void* p = nullptr; HIP_OK(hipMalloc(&p, bytes)); HIP_OK(hipMemset(p, 0, bytes)); return p; };
        tx.Position = alloc(px_ * 16); tx.FlatNormal = alloc(px_ * 4); tx.GeometricNormal = alloc(px_ * 4); tx.LinearDepth = alloc(px_ * 4);
        tx.NormalizedDepth = alloc(px_ * 4); tx.MotionVector = alloc(px_ * 8); tx.BaseColorMetalness = alloc(px_ * 4); tx.NormalRoughness = alloc(px_ * 8);
        tx.IOR = alloc(px_ * 2); tx.Transmission = alloc(px_); tx.Radiance = alloc(px_ * 8);
        tx.RadianceF32 = radianceF32 = (float*)alloc(px_ * 16);
        void* fullRadiance = sharded && rank == 0 ? alloc(fullPx * 8) : nullptr;      // the assembled frame (R16G16B16A16_FLOAT), root only
        float* fullRadianceF32 = sharded && rank == 0 && !out.empty() ? (float*)alloc(fullPx * 16) : nullptr;

        // ---- App::RenderScene
        GBufferGeneration gbuffer(commandList);
        gbuffer.GPUBuffers = { &sd, &cam, dInstances, dObjects, n, n };
        gbuffer.Textures = tx;
        Raytracing raytracing(commandList);
        raytracing.GPUBuffers = { &sd, &cam, dObjects, n };
        raytracing.Textures = tx;
        PtCounters counters{};
        double ms = 0;
        auto renderFrame = [&](uint32_t frameIndex) {
            gbuffer.Render(commandList, tlas, { { W, H }, ~0u & ~(uint32_t)GBufferGeneration::Flags::Albedo });     // App.cpp:1224
            Raytracing::GraphicsSettings gs; gs.RenderSize[0] = W; gs.RenderSize[1] = H;
            gs.FrameIndex = frameIndex; gs.Bounces = bounces; gs.SamplesPerPixel = spp; gs.IsRussianRouletteEnabled = true;
            raytracing.SetConstants(gs);
            raytracing.Render(commandList, tlas);
            if (sharded) sharding.GatherBands(commandList, tx.Radiance, fullRadiance, W, H, 8);
        };
        renderFrame(12345);                                         // warm-up
        commandList.End();
        // timed run: frames 1..N back to back
        ThrowIfFailed(commandList.Context, pt_reset_counters(commandList.Context));
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t f = 0; f < frames; f++) renderFrame(frames - 1 - f);   // the last frame has FrameIndex 0 (dumped below)
        commandList.End();
        ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        ThrowIfFailed(commandList.Context, pt_get_counters(commandList.Context, &counters));
        const double rays = (double)(counters.PrimaryRays + counters.SecondaryRays);
        if (!sharded)
            printf("{\"host\": \"c++\", \"width\": %u, \"height\": %u, \"spp\": %u, \"bounces\": %u, \"frames\": %u, \"rays\": %.0f, \"ms_per_frame\": %.4f, \"mrays_per_s\": %.1f}\n",
                   W, H, spp, bounces, frames, rays, ms / frames, rays / ms / 1e3);
        else       // rays are this rank's; rank 0's time includes every peer's bands arriving
            printf("{\"host\": \"c++\", \"rank\": %u, \"world\": %u, \"local_rows\": %u, \"width\": %u, \"height\": %u, \"spp\": %u, \"bounces\": %u, \"frames\": %u, \"rays_this_rank\": %.0f, \"ms_per_frame\": %.4f}\n",
                   rank, world, localRows, W, H, spp, bounces, frames, rays, ms / frames);
        if (sharded && fullRadianceF32) {                          // parity dump of a sharded run: the fp32 copy, assembled the same way
            sharding.GatherBands(commandList, radianceF32, fullRadianceF32, W, H, 16);
            commandList.End();
        } else if (sharded && !out.empty() && rank != 0) {
            sharding.GatherBands(commandList, radianceF32, nullptr, W, H, 16);
            commandList.End();
        }
        if (!out.empty() && rank == 0) {                            // last frame rendered has FrameIndex 0
            std::vector<float> host(fullPx * 4);
            HIP_OK(hipMemcpy(host.data(), fullRadianceF32 ? fullRadianceF32 : radianceF32, fullPx * 16, hipMemcpyDeviceToHost));
            FILE* fp = fopen(out.c_str(), "wb");
            if (!fp) { fprintf(stderr, "cannot open %s\n", out.c_str()); return 3; }
            fwrite(host.data(), 16, fullPx, fp); fclose(fp);
        }
    } catch (const std::exception& e) {
        fprintf(stderr, "pt_demo: %s\n", e.what());
        return 1;
    }
    return 0;
}
