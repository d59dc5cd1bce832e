// Developer aid: throughput of dependent chains of tiny kernels replayed as hipGraphs on S streams.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned* c, unsigned* out, int spin, unsigned busy)
{
    extern __shared__ char lds[];
    if (c[blockIdx.x & 31] == 12345u) out[0] = 1;      // never true; makes the load live
    if (spin && blockIdx.x < busy) { long long t0 = wall_clock64(); while (wall_clock64() - t0 < spin) {} }
}
int main(int argc, char** argv)
{
    int S = argc > 1 ? atoi(argv[1]) : 1, grid = argc > 2 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 78, spin = argc > 4 ? atoi(argv[4]) : 0;
    int prio = argc > 7 ? atoi(argv[7]) : 0; int ldsBytes = argc > 6 ? atoi(argv[6]) : 30000, reps = 50; unsigned busy = argc > 5 ? atoi(argv[5]) : 1u << 30;
    unsigned* d; hipMalloc(&d, 4096); hipMemset(d, 0, 4096);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, ldsBytes);
    std::vector<hipStream_t> st(S); std::vector<hipGraphExec_t> ge(S);
    for (int s = 0; s < S; s++) {
        if (prio) { int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi); int pr = prio == 1 ? hi + (s % (lo - hi + 1)) : (prio == 2 ? hi : (prio == 3 ? lo : (prio == 4 ? (s < 3 ? 0 : hi) : (s % 2 ? hi : 0)))); hipStreamCreateWithPriority(&st[s], hipStreamNonBlocking, pr); if (s == 0) printf("priority range %d..%d\n", lo, hi); }
        else hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking);
        hipGraph_t g; hipStreamBeginCapture(st[s], hipStreamCaptureModeRelaxed);
        for (int i = 0; i < K; i++) k<<<grid, 256, ldsBytes, st[s]>>>(d, d + 64, spin, busy);
        hipStreamEndCapture(st[s], &g); hipGraphInstantiate(&ge[s], g, nullptr, nullptr, 0); hipGraphDestroy(g);
    }
    for (int s = 0; s < S; s++) hipGraphLaunch(ge[s], st[s]);
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; r++) for (int s = 0; s < S; s++) hipGraphLaunch(ge[s], st[s]);
    hipDeviceSynchronize();
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("busy %u lds %d streams %d grid %d chain %d spin %d: %.2f us per kernel (throughput), %.1f us per chain-replay per stream\n", busy, ldsBytes, S, grid, K, spin, us / (reps * S * K), us / reps);
    return 0;
}
