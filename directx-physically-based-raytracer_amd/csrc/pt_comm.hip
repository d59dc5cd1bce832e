// pt_comm.hip -- the multi-GPU half of the C ABI: row bands of every rank into the root's full frame, over RCCL (xGMI).
//
// The reference renders on one GPU; north_star asks for "image tiles shard naturally across the 8 GPUs of one node with a final RCCL
// gather over xGMI". Sharding is pt_set_sharding (16-row bands, band b -> rank b % N); this file is the gather:
//   pt_comm_get_unique_id / pt_comm_init / pt_comm_adopt / pt_comm_destroy    one communicator per context (= per stream)
//   pt_gather_bands    grouped ncclSend / ncclRecv, one message per band, received STRAIGHT into the strided rows of the root's full
//                      frame: no staging buffer, no de-interleave pass; the root's own bands are one strided device copy. xGMI is
//                      point-to-point, so the root's seven links each carry only their peer's bands (2 MB per peer at 1080p, N = 8).
// RCCL is bound at the first pt_comm_* call (dlopen of librccl.so.1): a single-GPU host never loads it, and inside a process that
// already holds RCCL (PyTorch) the same copy is used.
#include "pt_internal.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

using namespace pt;

namespace {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

Rccl g_rccl;
std::once_flag g_rcclOnce;

void load_rccl()
{
    const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    for (const char* n : names) if ((g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!g_rccl.handle) { g_rccl.error = std::string("librccl.so.1 not found: ") + dlerror(); return; }
    #define PT_SYM(field, sym) do { g_rccl.field = (decltype(g_rccl.field))dlsym(g_rccl.handle, #sym); \
                                    if (!g_rccl.field) { g_rccl.error = "librccl lacks " #sym; return; } } while (0)
    PT_SYM(GetUniqueId, ncclGetUniqueId); PT_SYM(CommInitRank, ncclCommInitRank); PT_SYM(CommDestroy, ncclCommDestroy);
    PT_SYM(CommCount, ncclCommCount); PT_SYM(CommUserRank, ncclCommUserRank); PT_SYM(GroupStart, ncclGroupStart);
    PT_SYM(GroupEnd, ncclGroupEnd); PT_SYM(Send, ncclSend); PT_SYM(Recv, ncclRecv); PT_SYM(GetErrorString, ncclGetErrorString);
    #undef PT_SYM
}

int fail(Context* c, int status, const std::string& msg) { if (c) c->lastError = msg; else create_error() = msg; return status; }

const Rccl* rccl(Context* c, int& status)
{
    std::call_once(g_rcclOnce, load_rccl);
    if (!g_rccl.error.empty()) { status = fail(c, PT_ERROR_RCCL, g_rccl.error); return nullptr; }
    status = PT_OK;
    return &g_rccl;
}

#define API_NCCL(ctx, R, expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) return fail(ctx, PT_ERROR_RCCL, std::string(#expr) + ": " + (R)->GetErrorString(r_)); } while (0)

// the root's own bands: local rows [j * bandRows ...) -> rows of band j * rankCount + rank of the full frame. 16 bytes per lane.
__global__ __launch_bounds__(256) void k_place_own_bands(uint4* __restrict__ dst, const uint4* __restrict__ src, uint32_t rank, uint32_t rankCount,
                                                         uint32_t bandHeight, uint32_t row16, uint32_t localRows)
{
    const uint64_t total = (uint64_t)localRows * row16;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t ly = (uint32_t)(t / row16), x = (uint32_t)(t - (uint64_t)ly * row16);
        const uint32_t band = ly / bandHeight, y = (band * rankCount + rank) * bandHeight + (ly - band * bandHeight);
        dst[(uint64_t)y * row16 + x] = src[t];
    }
}

} // namespace

extern "C" {

int pt_comm_get_unique_id(void* out_id)
{
    if (!out_id) return fail(nullptr, PT_ERROR_INVALID_ARGUMENT, "out_id is NULL");
    int s; const Rccl* R = rccl(nullptr, s);
    if (!R) return s;
    static_assert(sizeof(ncclUniqueId) == PT_COMM_ID_BYTES, "PT_COMM_ID_BYTES");
    ncclUniqueId id;
    API_NCCL(nullptr, R, R->GetUniqueId(&id));
    memcpy(out_id, &id, sizeof id);
    return PT_OK;
}

int pt_comm_destroy(PtContext* ctx)
{
    if (!ctx) return PT_ERROR_INVALID_ARGUMENT;
    Context& c = ctx->c;
    if (c.comm && c.commOwned) {
        int s; const Rccl* R = rccl(&c, s);
        if (!R) return s;
        hipSetDevice(c.device);
        hipStreamSynchronize(c.stream);
        API_NCCL(&c, R, R->CommDestroy((ncclComm_t)c.comm));
    }
    c.comm = nullptr; c.commOwned = false; c.commRank = 0; c.commWorld = 1;
    return PT_OK;
}

int pt_comm_init(PtContext* ctx, const void* id, uint32_t rank, uint32_t world)
{
    if (!ctx) return PT_ERROR_INVALID_ARGUMENT;
    Context& c = ctx->c;
    if (!id || world == 0 || rank >= world) return fail(&c, PT_ERROR_INVALID_ARGUMENT, "pt_comm_init: id is NULL or rank >= world");
    int s = pt_comm_destroy(ctx);
    if (s != PT_OK) return s;
    const Rccl* R = rccl(&c, s);
    if (!R) return s;
    if (hipSetDevice(c.device) != hipSuccess) return fail(&c, PT_ERROR_HIP, "hipSetDevice");
    ncclUniqueId uid; memcpy(&uid, id, sizeof uid);
    ncclComm_t comm = nullptr;
    API_NCCL(&c, R, R->CommInitRank(&comm, (int)world, uid, (int)rank));       // returns when every rank has joined
    c.comm = comm; c.commOwned = true; c.commRank = rank; c.commWorld = world;
    return PT_OK;
}

int pt_comm_adopt(PtContext* ctx, void* nccl_comm)
{
    if (!ctx) return PT_ERROR_INVALID_ARGUMENT;
    Context& c = ctx->c;
    if (!nccl_comm) return fail(&c, PT_ERROR_INVALID_ARGUMENT, "pt_comm_adopt: communicator is NULL");
    int s = pt_comm_destroy(ctx);
    if (s != PT_OK) return s;
    const Rccl* R = rccl(&c, s);
    if (!R) return s;
    int n = 0, r = 0;
    API_NCCL(&c, R, R->CommCount((ncclComm_t)nccl_comm, &n));
    API_NCCL(&c, R, R->CommUserRank((ncclComm_t)nccl_comm, &r));
    c.comm = nccl_comm; c.commOwned = false; c.commRank = (uint32_t)r; c.commWorld = (uint32_t)n;
    return PT_OK;
}

// The messages of one gather, as rank `sharding->RankIndex` sees them, in issue order: band b travels from its owner b % N to `root`
// as ONE message of (rows of b) x row_bytes, read at LocalOffset of the owner's local texture and written at FullOffset of the root's
// full frame. A non-root rank gets its sends, the root its receives (ascending band order on both sides, so the k-th send of a peer
// meets the k-th receive from it) -- and nobody gets the root's own bands: those never leave the device. Host arithmetic only.
int pt_gather_plan(const PtSharding* sh, uint32_t height, uint64_t row_bytes, uint32_t root, PtBandMessage* out, uint32_t capacity, uint32_t* out_count)
{
    if (!sh || !out_count || sh->RankCount == 0 || sh->BandHeight == 0 || sh->RankIndex >= sh->RankCount || root >= sh->RankCount) return PT_ERROR_INVALID_ARGUMENT;
    const uint32_t N = sh->RankCount, me = sh->RankIndex, BH = sh->BandHeight, bands = (height + BH - 1) / BH;
    uint32_t n = 0;
    for (uint32_t b = 0; b < bands; b++) {
        const uint32_t owner = b % N;
        if (owner == root || (me != root && owner != me)) continue;
        const uint32_t y0 = b * BH, rows = height - y0 < BH ? height - y0 : BH;
        if (out && n < capacity) {
            PtBandMessage& m = out[n];
            m.Peer = me == root ? owner : root; m.IsSend = me == root ? 0u : 1u; m.Band = b; m._pad = 0;
            m.LocalOffset = (uint64_t)(b / N) * BH * row_bytes;        // every band before the last one of a rank is full
            m.FullOffset = (uint64_t)y0 * row_bytes;
            m.Bytes = (uint64_t)rows * row_bytes;
        }
        n++;
    }
    *out_count = n;
    return (out && n > capacity) ? PT_ERROR_INVALID_ARGUMENT : PT_OK;
}

int pt_gather_bands(PtContext* ctx, const void* local_bands, void* dst_full, uint32_t width, uint32_t height, uint32_t pixel_bytes, uint32_t root)
{
    if (!ctx) return PT_ERROR_INVALID_ARGUMENT;
    Context& c = ctx->c;
    const PtSharding& sh = c.sharding;
    const uint32_t N = sh.RankCount, me = sh.RankIndex, BH = sh.BandHeight;
    if (root >= N) return fail(&c, PT_ERROR_INVALID_ARGUMENT, "pt_gather_bands: root is not a rank of the sharding");
    if (!local_bands || (me == root && !dst_full)) return fail(&c, PT_ERROR_INVALID_ARGUMENT, "pt_gather_bands: a buffer is NULL");
    const uint64_t rowBytes = (uint64_t)width * pixel_bytes;
    if (rowBytes == 0 || rowBytes % 16 != 0) return fail(&c, PT_ERROR_INVALID_ARGUMENT, "pt_gather_bands: a row must be a multiple of 16 bytes");
    // PT_DEBUG_GATHER_SELF_EXCHANGE (rehearsal on ONE GPU, world-size-1 communicator): the rank's own bands -- whoever the sharding says it
    // is -- take the path a foreign band takes: ncclSend to itself and ncclRecv from itself, one pair per band inside one group, the
    // receive aimed straight at the band's rows of the full frame. The plan arithmetic, the grouped p2p calls, ncclGroupEnd and RCCL's
    // kernel on the context's stream all really run; only the peer is the sender itself.
    const bool selfExchange = (c.debugFlags & PT_DEBUG_GATHER_SELF_EXCHANGE) != 0;
    const bool exchange = !selfExchange && N > 1 && !(c.debugFlags & PT_DEBUG_GATHER_LOCAL_ONLY);
    if (selfExchange && (!c.comm || c.commWorld != 1))
        return fail(&c, PT_ERROR_NOT_READY, "pt_gather_bands: PT_DEBUG_GATHER_SELF_EXCHANGE needs a communicator of world size 1");
    if (exchange && (!c.comm || c.commWorld != N || c.commRank != me))
        return fail(&c, PT_ERROR_NOT_READY, "pt_gather_bands: no communicator matching the sharding (pt_comm_init / pt_comm_adopt with the rank and rank count of pt_set_sharding)");
    if (hipSetDevice(c.device) != hipSuccess) return fail(&c, PT_ERROR_HIP, "hipSetDevice");
    uint32_t localRows = 0;
    pt_local_rows(&sh, height, &localRows);

    if (exchange || selfExchange) {
        int s; const Rccl* R = rccl(&c, s);
        if (!R) return s;
        std::vector<PtBandMessage> plan;
        if (selfExchange) {
            if (!dst_full) return fail(&c, PT_ERROR_INVALID_ARGUMENT, "pt_gather_bands: a buffer is NULL");
            // the messages a root would RECEIVE from rank `me`: pt_gather_plan seen from a root that is not `me`; when N == 1 (or every
            // other rank would be `me` too) the bands are listed directly
            const uint32_t bands = (height + BH - 1) / BH;
            for (uint32_t b = me; b < bands; b += N) {
                const uint32_t y0 = b * BH, rows = height - y0 < BH ? height - y0 : BH;
                PtBandMessage m{}; m.Peer = 0; m.IsSend = 1u; m.Band = b;
                m.LocalOffset = (uint64_t)(b / N) * BH * rowBytes; m.FullOffset = (uint64_t)y0 * rowBytes; m.Bytes = (uint64_t)rows * rowBytes;
                plan.push_back(m);
            }
        } else {
            uint32_t count = 0;
            pt_gather_plan(&sh, height, rowBytes, root, nullptr, 0, &count);
            plan.resize(count);
            pt_gather_plan(&sh, height, rowBytes, root, plan.data(), count, &count);
        }
        ncclComm_t comm = (ncclComm_t)c.comm;
        API_NCCL(&c, R, R->GroupStart());
        ncclResult_t r = ncclSuccess;
        for (const PtBandMessage& m : plan) {
            if (m.IsSend) r = R->Send((const uint8_t*)local_bands + m.LocalOffset, (size_t)m.Bytes, ncclUint8, (int)m.Peer, comm, c.stream);
            if (r == ncclSuccess && (!m.IsSend || selfExchange)) r = R->Recv((uint8_t*)dst_full + m.FullOffset, (size_t)m.Bytes, ncclUint8, (int)m.Peer, comm, c.stream);
            if (r != ncclSuccess) break;
        }
        const ncclResult_t e = R->GroupEnd();
        if (r != ncclSuccess) return fail(&c, PT_ERROR_RCCL, std::string("ncclSend / ncclRecv: ") + R->GetErrorString(r));
        if (e != ncclSuccess) return fail(&c, PT_ERROR_RCCL, std::string("ncclGroupEnd: ") + R->GetErrorString(e));
    }
    if (!selfExchange && me == root && localRows && local_bands != dst_full) {
        const uint32_t row16 = (uint32_t)(rowBytes / 16);
        const uint64_t total = (uint64_t)localRows * row16;
        const uint32_t grid = (uint32_t)std::min<uint64_t>((total + 255) / 256, 2048);
        k_place_own_bands<<<grid, 256, 0, c.stream>>>((uint4*)dst_full, (const uint4*)local_bands, me, N, BH, row16, localRows);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(&c, PT_ERROR_HIP, std::string("k_place_own_bands: ") + hipGetErrorString(e));
    }
    return PT_OK;
}

} // extern "C"
