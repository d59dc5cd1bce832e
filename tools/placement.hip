// Developer aid: where does the dispatcher put the workgroups of a partially filled grid? (HW_ID per block)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <map>
#include <vector>
__global__ void k(uint32_t* out, int spin)
{
    extern __shared__ char lds[];
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = xcc; lds[0] = 1; }
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
}
int main(int argc, char** argv)
{
    int grid = argc > 1 ? atoi(argv[1]) : 384, ldsBytes = argc > 2 ? atoi(argv[2]) : 30000;
    uint32_t* d; hipMalloc(&d, grid * 8);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, ldsBytes);
    k<<<grid, 256, ldsBytes>>>(d, 2000);
    hipDeviceSynchronize();
    std::vector<uint32_t> h(grid * 2); hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost);
    std::map<uint32_t, int> perCu; std::map<uint32_t, int> perXcc;
    for (int b = 0; b < grid; b++) {
        uint32_t hw = h[2 * b], xcc = h[2 * b + 1] & 0xF;
        uint32_t cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        perCu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++; perXcc[xcc]++;
        if (b < 16) printf("block %d: xcc %u se %u sh %u cu %u simd %u\n", b, xcc, se, sh, cu, (hw >> 4) & 3);
    }
    std::map<int, int> hist; for (auto& kv : perCu) hist[kv.second]++;
    printf("grid %d lds %d: distinct CUs %zu; blocks-per-CU histogram:", grid, ldsBytes, perCu.size());
    for (auto& kv : hist) printf(" %dx%d", kv.second, kv.first);
    printf("\nper XCC:"); for (auto& kv : perXcc) printf(" %u:%d", kv.first, kv.second); printf("\n");
    return 0;
}
