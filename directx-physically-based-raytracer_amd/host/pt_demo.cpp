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
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "ptamd.hpp"
#include "pt_ingest.hpp"

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

// What the host hands the library, whichever way it came about (built in code below, or loaded by pt_ingest.hpp): mesh nodes (one bottom level
// each, one geometry per mesh), instances of them, a camera and the environment.
struct HostGeometry { std::vector<uint8_t> vertices; uint32_t vertexCount = 0; std::vector<uint8_t> indices; uint32_t indexCount = 0, indexStride = 2;
                      bool hasNormals = true, hasTangents = false, hasUV[2] = { false, false }; PtMaterial material{};
                      std::shared_ptr<ingest::Texture> textures[7]; uint32_t texCoord[7] = { 0, 0, 0, 0, 0, 0, 0 }; };   // slots in the order of Material.ixx:22-33
struct HostNode { std::vector<HostGeometry> meshes; };
struct HostObject { uint32_t node = 0; float transform[12]; bool visible = true; };
struct HostScene {
    std::vector<HostNode> nodes; std::vector<HostObject> objects;
    double cameraPosition[3] = { 0, 0, 0 }, cameraForward[3] = { 0, 0, 1 }, cameraUp[3] = { 0, 1, 0 };
    float envColor[4] = { 0, 0, 0, 1 }, envTransform[12] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0 };
};

static HostGeometry to_geometry(const HostMesh& m)
{
    HostGeometry g;
    g.vertexCount = (uint32_t)m.vertices.size(); g.vertices.resize(m.vertices.size() * sizeof(Vertex)); memcpy(g.vertices.data(), m.vertices.data(), g.vertices.size());
    g.indexCount = (uint32_t)m.indices.size(); g.indices.resize(m.indices.size() * 2); memcpy(g.indices.data(), m.indices.data(), g.indices.size());
    g.material = m.material;
    return g;
}

// scenes.py:cornell_box, variant "ggx": one mesh node per object, camera at the front opening looking +Z
static HostScene cornell_scene()
{
    HostScene sc;
    const PtMaterial white = make_material(0.73f, 0.73f, 0.73f), red = make_material(0.65f, 0.05f, 0.05f), green = make_material(0.12f, 0.45f, 0.15f);
    const double floorP[4][3] = { { -1, -1, -1 }, { -1, -1, 1 }, { 1, -1, 1 }, { 1, -1, -1 } }, up[3] = { 0, 1, 0 };
    const double ceilP[4][3] = { { -1, 1, -1 }, { 1, 1, -1 }, { 1, 1, 1 }, { -1, 1, 1 } }, down[3] = { 0, -1, 0 };
    const double backP[4][3] = { { -1, -1, 1 }, { -1, 1, 1 }, { 1, 1, 1 }, { 1, -1, 1 } }, toCam[3] = { 0, 0, -1 };
    const double leftP[4][3] = { { -1, -1, -1 }, { -1, 1, -1 }, { -1, 1, 1 }, { -1, -1, 1 } }, px[3] = { 1, 0, 0 };
    const double rightP[4][3] = { { 1, -1, -1 }, { 1, -1, 1 }, { 1, 1, 1 }, { 1, 1, -1 } }, nx[3] = { -1, 0, 0 };
    const double lightP[4][3] = { { -0.25, 0, -0.25 }, { 0.25, 0, -0.25 }, { 0.25, 0, 0.25 }, { -0.25, 0, 0.25 } };
    const std::vector<HostMesh> meshes = {
        quad(floorP, up, white), quad(ceilP, down, white), quad(backP, toCam, white), quad(leftP, px, red), quad(rightP, nx, green),
        quad(lightP, down, make_material(0.78f, 0.78f, 0.78f, 1, 1, 1, 15.0f)),
        box(make_material(0.95f, 0.93f, 0.88f, 0, 0, 0, 1, 1.0f, 0.05f)), box(make_material(0.73f, 0.73f, 0.73f, 0, 0, 0, 1, 0, 0.2f)) };
    float xf[8][12];
    for (int i = 0; i < 5; i++) trs(xf[i], 0, 0, 0, 0, 1, 1, 1);
    trs(xf[5], 0, 0.998, 0.1, 0, 1, 1, 1);
    trs(xf[6], -0.35, -0.4, 0.35, -18.0, 0.6, 1.2, 0.6);
    trs(xf[7], 0.35, -0.7, -0.25, 15.0, 0.6, 0.6, 0.6);
    for (size_t i = 0; i < meshes.size(); i++) {
        HostNode n; n.meshes.push_back(to_geometry(meshes[i])); sc.nodes.push_back(std::move(n));
        HostObject o; o.node = (uint32_t)i; memcpy(o.transform, xf[i], 48); sc.objects.push_back(o);
    }
    sc.cameraPosition[2] = -1.95;
    return sc;
}

// a scene descriptor in the reference's schema (Source/MyScene.ixx:33-90) with its glTF models, through host/pt_ingest.hpp.
// Camera: App::ResetCamera = CameraController::SetPosition / SetRotation (Source/Camera.ixx:84-97), default lens.
static HostScene ingested_scene(const std::string& path)
{
    const ingest::Scene in = ingest::load_scene(path);
    HostScene sc;
    for (auto& node : in.Nodes) {
        HostNode n;
        for (const ingest::MeshData& md : node->Meshes) {
            HostGeometry g;
            g.vertexCount = (uint32_t)md.Vertices.size(); g.vertices.resize(md.Vertices.size() * sizeof(ingest::Vertex)); memcpy(g.vertices.data(), md.Vertices.data(), g.vertices.size());
            g.indices = md.Indices; g.indexCount = md.IndexCount; g.indexStride = md.IndexStride;
            g.hasNormals = md.HasNormals; g.hasTangents = md.HasTangents; g.hasUV[0] = md.HasUV[0]; g.hasUV[1] = md.HasUV[1];
            g.material = md.HasMaterial ? md.Material : ingest::default_material();            // App.cpp:1044: Material() when a mesh names none
            for (int k = 0; k < 7; k++) { g.textures[k] = md.Textures[k]; g.texCoord[k] = md.TextureCoordinateIndex[k]; }
            for (const std::string& slot : md.SkippedTextures) fprintf(stderr, "pt_demo: %s texture of a material not loaded (this host decodes 8-bit PNG only)\n", slot.c_str());
            n.meshes.push_back(std::move(g));
        }
        sc.nodes.push_back(std::move(n));
    }
    for (const ingest::RenderObject& ro : in.Objects) { HostObject o; o.node = ro.Node; memcpy(o.transform, ro.Transform, 48); o.visible = ro.IsVisible; sc.objects.push_back(o); }
    const ingest::M4& r = in.CameraRotation;                         // forward = (0,0,1,0) R, right = (1,0,0,0) R, up = forward x right (ingest.py camera_from_desc)
    const double fwd[3] = { r.m[2][0], r.m[2][1], r.m[2][2] }, right[3] = { r.m[0][0], r.m[0][1], r.m[0][2] };
    for (int k = 0; k < 3; k++) { sc.cameraPosition[k] = in.CameraPosition[k]; sc.cameraForward[k] = fwd[k]; }
    sc.cameraUp[0] = fwd[1] * right[2] - fwd[2] * right[1]; sc.cameraUp[1] = fwd[2] * right[0] - fwd[0] * right[2]; sc.cameraUp[2] = fwd[0] * right[1] - fwd[1] * right[0];
    memcpy(sc.envColor, in.EnvironmentLightColor, 16); memcpy(sc.envTransform, in.EnvironmentLightTransform, 48);
    return sc;
}

// scenes.py:make_camera (hfov 90, near 0.01, far inf): XMMatrixLookToLH + SetupByHalfFovxInf, row-vector convention
static PtCamera make_camera(const HostScene& sc, double aspect)
{
    auto norm = [](double v[3]) { const double l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); v[0] /= l; v[1] /= l; v[2] /= l; };
    auto cross = [](const double a[3], const double b[3], double o[3]) { o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; };
    auto dot = [](const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
    double f[3] = { sc.cameraForward[0], sc.cameraForward[1], sc.cameraForward[2] }, r[3], u[3];
    norm(f); cross(sc.cameraUp, f, r); norm(r); cross(f, r, u);
    const double* pos = sc.cameraPosition;
    const double rightLen = std::tan((90.0 * (M_PI / 180.0)) / 2), upLen = rightLen / aspect, nearD = 0.01;
    PtCamera cam{};
    cam.IsNormalizedDepthReversed = 1;
    for (int k = 0; k < 3; k++) {
        cam.Position[k] = cam.PreviousPosition[k] = (float)pos[k];
        cam.RightDirection[k] = (float)(r[k] * rightLen); cam.UpDirection[k] = (float)(u[k] * upLen); cam.ForwardDirection[k] = (float)f[k];
    }
    cam.NearDepth = (float)nearD; cam.FarDepth = std::numeric_limits<float>::infinity();
    double w2v[4][4] = {}, v2p[4][4] = {}, w2p[4][4] = {};
    for (int k = 0; k < 3; k++) { w2v[k][0] = r[k]; w2v[k][1] = u[k]; w2v[k][2] = f[k]; }
    w2v[3][0] = -dot(pos, r); w2v[3][1] = -dot(pos, u); w2v[3][2] = -dot(pos, f); w2v[3][3] = 1;
    v2p[0][0] = 1 / rightLen; v2p[1][1] = aspect / rightLen; v2p[2][3] = 1; v2p[3][2] = nearD;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) for (int k = 0; k < 4; k++) w2p[i][j] += w2v[i][k] * v2p[k][j];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
        cam.WorldToProjection[4 * i + j] = cam.PreviousWorldToProjection[4 * i + j] = (float)w2p[i][j];
        cam.PreviousWorldToView[4 * i + j] = (float)w2v[i][j]; cam.PreviousViewToProjection[4 * i + j] = (float)v2p[i][j];
    }
    return cam;
}

// --dump-scene: what would be handed to the library, byte for byte, without touching a GPU (tests/test_host_cpp.py compares it with the harness's ingest)
static int dump_scene(const HostScene& sc, const std::string& path)
{
    FILE* fp = fopen(path.c_str(), "wb");
    if (!fp) { fprintf(stderr, "cannot open %s\n", path.c_str()); return 3; }
    printf("{\"nodes\": [");
    for (size_t n = 0; n < sc.nodes.size(); n++) {
        printf("%s[", n ? ", " : "");
        for (size_t m = 0; m < sc.nodes[n].meshes.size(); m++) {
            const HostGeometry& g = sc.nodes[n].meshes[m];
            printf("%s{\"vertices\": %u, \"indices\": %u, \"index_stride\": %u, \"has_normals\": %d, \"has_tangents\": %d, \"has_uv\": [%d, %d], \"textures\": [", m ? ", " : "",
                   g.vertexCount, g.indexCount, g.indexStride, g.hasNormals, g.hasTangents, g.hasUV[0], g.hasUV[1]);
            fwrite(g.vertices.data(), 1, g.vertices.size(), fp); fwrite(g.indices.data(), 1, g.indices.size(), fp); fwrite(&g.material, sizeof(PtMaterial), 1, fp);
            bool firstTex = true;
            for (int k = 0; k < 7; k++) if (g.textures[k]) {              // slot, size, sRGB, coordinate set; the texels follow the material in the file
                printf("%s[%d, %u, %u, %d, %u]", firstTex ? "" : ", ", k, g.textures[k]->Texels.Width, g.textures[k]->Texels.Height, g.textures[k]->SRGB ? 1 : 0, g.texCoord[k]);
                fwrite(g.textures[k]->Texels.RGBA.data(), 1, g.textures[k]->Texels.RGBA.size(), fp); firstTex = false;
            }
            printf("]}");
        }
        printf("]");
    }
    printf("], \"objects\": [");
    for (size_t i = 0; i < sc.objects.size(); i++) { printf("%s{\"node\": %u, \"visible\": %d}", i ? ", " : "", sc.objects[i].node, sc.objects[i].visible ? 1 : 0); fwrite(sc.objects[i].transform, 4, 12, fp); }
    printf("]}\n");
    fwrite(sc.envColor, 4, 4, fp); fwrite(sc.envTransform, 4, 12, fp);
    const PtCamera cam = make_camera(sc, 16.0 / 9.0);
    fwrite(&cam, sizeof cam, 1, fp);
    fclose(fp);
    return 0;
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
    std::string out, idFile, scenePath, dumpPath;
    for (int i = 1; i + 1 < argc; i += 2) {
        std::string k = argv[i];
        if (k == "--width") W = atoi(argv[i + 1]); else if (k == "--height") H = atoi(argv[i + 1]);
        else if (k == "--spp") spp = atoi(argv[i + 1]); else if (k == "--bounces") bounces = atoi(argv[i + 1]);
        else if (k == "--frames") frames = atoi(argv[i + 1]); else if (k == "--out") out = argv[i + 1];
        else if (k == "--ranks") ranks = atoi(argv[i + 1]); else if (k == "--rank") rank = atoi(argv[i + 1]);
        else if (k == "--world") world = atoi(argv[i + 1]); else if (k == "--id-file") idFile = argv[i + 1];
        else if (k == "--scene") scenePath = argv[i + 1]; else if (k == "--dump-scene") dumpPath = argv[i + 1];
    }
    const bool sharded = !idFile.empty();
    if (!dumpPath.empty()) {                                        // no GPU call on this path
        try { return dump_scene(scenePath.empty() ? cornell_scene() : ingested_scene(scenePath), dumpPath); }
        catch (const std::exception& e) { fprintf(stderr, "pt_demo: %s\n", e.what()); return 1; }
    }
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

        // ---- scene: the Cornell box built in code (scenes.py:cornell_box, variant "ggx"), or --scene <descriptor.json> through pt_ingest.hpp
        const HostScene scene = scenePath.empty() ? cornell_scene() : ingested_scene(scenePath);

        // ---- buffers, descriptor heap, bottom levels (Scene::CreateAccelerationStructures: one per mesh node, one geometry per mesh)
        uint32_t heapSize = 0;
        std::map<const ingest::Texture*, uint32_t> textureDescriptor;        // one descriptor per texture, however many meshes share it
        for (const HostNode& nd : scene.nodes) {
            heapSize += 2 * (uint32_t)nd.meshes.size();
            for (const HostGeometry& hg : nd.meshes) for (auto& t : hg.textures) if (t && !textureDescriptor.count(t.get())) textureDescriptor[t.get()] = 0;
        }
        const uint32_t firstTextureDescriptor = heapSize;
        heapSize += (uint32_t)textureDescriptor.size();
        ThrowIfFailed(commandList.Context, pt_heap_resize(commandList.Context, heapSize));
        {
            uint32_t next = firstTextureDescriptor;
            for (auto& kv : textureDescriptor) {                             // texel arrays as pt_heap_set_texture takes them (App.cpp:1052-1063 fills the descriptor indices)
                kv.second = next++;
                const ingest::Image& im = kv.first->Texels;
                uint8_t* dt = upload(im.RGBA);
                ThrowIfFailed(commandList.Context, pt_heap_set_texture(commandList.Context, kv.second, dt, im.Width, im.Height,
                                                                       kv.first->SRGB ? PT_FORMAT_R8G8B8A8_UNORM_SRGB : PT_FORMAT_R8G8B8A8_UNORM, 0));
            }
        }
        struct GeometryOnDevice { uint32_t heapVertices, heapIndices; };
        std::vector<std::vector<GeometryOnDevice>> onDevice(scene.nodes.size());
        std::vector<uint64_t> blas(scene.nodes.size());
        uint32_t heapNext = 0;
        for (size_t ni = 0; ni < scene.nodes.size(); ni++) {
            std::vector<PtGeometryDesc> geoms;
            for (const HostGeometry& hg : scene.nodes[ni].meshes) {
                uint8_t* dv = upload(hg.vertices); uint8_t* di = upload(hg.indices);
                ThrowIfFailed(commandList.Context, pt_heap_set_buffer(commandList.Context, heapNext, dv, hg.vertices.size(), 0));
                ThrowIfFailed(commandList.Context, pt_heap_set_buffer(commandList.Context, heapNext + 1, di, hg.indices.size(), hg.indexStride));
                onDevice[ni].push_back({ heapNext, heapNext + 1 }); heapNext += 2;
                // Scene.ixx:320-324: FLAG_OPAQUE unless the material is alpha-tested / blended
                geoms.push_back(RaytracingHelpers::CreateGeometryDesc({ dv, hg.vertexCount, sizeof(Vertex) }, { di, hg.indexCount, hg.indexStride },
                                                                      hg.material.AlphaMode == 0 ? PT_GEOMETRY_FLAG_OPAQUE : 0u));
            }
            blas[ni] = RaytracingHelpers::BuildBottomLevelAccelerationStructure(commandList, geoms, PT_BUILD_FLAG_PREFER_FAST_TRACE);
        }
        // ---- ObjectData / InstanceData (App::UpdateScene, Source/App.cpp:1028-1074): one object record per (instance, geometry), InstanceID = the
        // instance's first object (Scene.ixx:371)
        std::vector<PtObjectData> objectData; std::vector<PtInstanceData> instanceData(scene.objects.size());
        std::vector<PtInstanceDesc> instanceDescs(scene.objects.size());
        for (size_t i = 0; i < scene.objects.size(); i++) {
            const HostObject& ho = scene.objects[i];
            const uint32_t first = (uint32_t)objectData.size();
            for (size_t g = 0; g < scene.nodes[ho.node].meshes.size(); g++) {
                const HostGeometry& hg = scene.nodes[ho.node].meshes[g];
                PtObjectData od; memset(&od, 0, sizeof od);
                od.VertexDesc.Stride = sizeof(Vertex);
                od.VertexDesc.AttributeOffsets.Normal = hg.hasNormals ? 12u : ~0u; od.VertexDesc.AttributeOffsets.Tangent = hg.hasTangents ? 18u : ~0u;
                od.VertexDesc.AttributeOffsets.TextureCoordinates[0] = hg.hasUV[0] ? 24u : ~0u; od.VertexDesc.AttributeOffsets.TextureCoordinates[1] = hg.hasUV[1] ? 28u : ~0u;
                od.MeshDescriptors.Vertices = onDevice[ho.node][g].heapVertices; od.MeshDescriptors.Indices = onDevice[ho.node][g].heapIndices; od.MeshDescriptors.MotionVectors = ~0u;
                od.Material = hg.material;
                for (int k = 0; k < 7; k++) {
                    od.TextureMapInfoArray[k].Descriptor = hg.textures[k] ? textureDescriptor[hg.textures[k].get()] : ~0u;
                    od.TextureMapInfoArray[k].TextureCoordinateIndex = hg.texCoord[k];
                }
                objectData.push_back(od);
            }
            PtInstanceData& id = instanceData[i]; memset(&id, 0, sizeof id);
            id.FirstGeometryIndex = first;
            memcpy(id.ObjectToWorld, ho.transform, 48); memcpy(id.PreviousObjectToWorld, ho.transform, 48);
            PtInstanceDesc& d = instanceDescs[i]; memset(&d, 0, sizeof d);
            memcpy(d.Transform, ho.transform, 48); d.InstanceID = first; d.InstanceMask = ho.visible ? ~0u : 0u; d.AccelerationStructure = blas[ho.node];
        }
        const uint32_t n = (uint32_t)objectData.size(), nInstances = (uint32_t)scene.objects.size();
        RaytracingHelpers::TopLevelAccelerationStructure tlas;
        RaytracingHelpers::BuildTopLevelAccelerationStructure(commandList, PT_BUILD_FLAG_PREFER_FAST_TRACE, instanceDescs, false, tlas);
        PtObjectData* dObjects = upload(objectData); PtInstanceData* dInstances = upload(instanceData);

        // ---- camera (scenes.py:make_camera: hfov 90, near 0.01, far inf) and scene data
        PtCamera cam = make_camera(scene, (double)W / (double)H);
        PtSceneData sd{}; sd.IsStatic = 1; sd.EnvironmentLightTextureDescriptor = ~0u;
        memcpy(sd.EnvironmentLightColor, scene.envColor, 16); memcpy(sd.EnvironmentLightTransform, scene.envTransform, 48);

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
        gbuffer.GPUBuffers = { &sd, &cam, dInstances, dObjects, nInstances, n };
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
