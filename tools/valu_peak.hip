// valu_peak.hip -- what a gfx950 chip issues per second in wave64 vector instructions (developer tool, not part of the product).
//
//   hipcc -O2 --offload-arch=gfx950 tools/valu_peak.hip -o build/valu_peak && build/valu_peak > profiles/r03_valu_peak.json
//
// bench.py's roofline.valu_issue prices the render kernels against the "fma_indep" figure at 4 and 8 waves per SIMD that this
// program measures (the round-2 figure of 614 G/s assumed 4 cycles per wave64 instruction, which is what ONE wave
// alone on a SIMD sustains; a SIMD-32 retires a wave64 instruction in 2 cycles once two or more waves feed it).
//
// Every block is 256 threads = one wave per SIMD of its CU; a dynamic-LDS request of 160 KB / k makes exactly k blocks resident per
// CU, so a launch of CUs x k blocks runs k waves on every SIMD of the chip, all at once. Streams:
//   fma_indep    8 independent v_fma_f32 accumulators per lane (issue-bound)
//   fma_dep      1 accumulator: every instruction waits for the previous one (dependent-issue latency)
//   pk_fma_indep 8 independent v_pk_fma_f32 (two fp32 FMAs per lane and instruction)
//   mixed        v_fma_f32 interleaved with v_cndmask / v_max / v_cvt_f32_ubyte in the proportion of the node test
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kUnroll = 32;        // instruction groups per loop iteration

template <int MODE>
__global__ __launch_bounds__(256) void k_stream(float* out, int iters, float x, float y)
{
    extern __shared__ unsigned char smem[];
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = { a0, a1 }, p1 = { a2, a3 }, p2 = { a4, a5 }, p3 = { a6, a7 }, p4 = { a1, a0 }, p5 = { a3, a2 }, p6 = { a5, a4 }, p7 = { a7, a6 };
    f2 xx = { x, x }, yy = { y, y };
    unsigned u = threadIdx.x * 0x01010101u;
    for (int i = 0; i < iters; i++) {
        #pragma unroll
        for (int k = 0; k < kUnroll; k++) {
            if (MODE == 0) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
            } else if (MODE == 1) {
                asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2"
                             : "+v"(a0) : "v"(x), "v"(y));
            } else if (MODE == 2) {
                asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                             "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(xx), "v"(yy));
            } else {
                asm volatile("v_cvt_f32_ubyte0 %0, %8\n v_fma_f32 %1, %0, %9, %1\n v_cvt_f32_ubyte1 %2, %8\n v_fma_f32 %3, %2, %9, %3\n"
                             "v_max_f32 %4, %1, %3\n v_min_f32 %5, %1, %3\n v_cmp_le_f32 vcc, %4, %5\n v_cndmask_b32 %6, %6, %7, vcc"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(u), "v"(x) : "vcc");
            }
        }
    }
    if (MODE == 2) { a0 = p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y; }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)smem[0];
}

template <int MODE>
static double run(int cus, int wavesPerSimd, int iters, float* out)
{
    const size_t lds = (size_t)(160 * 1024) / wavesPerSimd - (wavesPerSimd > 1 ? 512 : 0);   // exactly wavesPerSimd blocks fit one CU
    CHECK(hipFuncSetAttribute((const void*)k_stream<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = cus * wavesPerSimd;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k_stream<MODE><<<grid, 256, lds>>>(out, 16, 1.0f, 0.0f);       // warm-up
    CHECK(hipDeviceSynchronize());
    double best = 1e30;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0));
        k_stream<MODE><<<grid, 256, lds>>>(out, iters, 1.0f, 0.0f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double insts = (double)grid * 4.0 * (double)iters * kUnroll * 8.0;      // wave-instructions
    return insts / (best * 1e-3);
}

int main()
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    float* out; CHECK(hipMalloc(&out, sizeof(float) * 256 * cus * 8));
    const int iters = 2048;
    const char* names[4] = { "fma_indep", "fma_dep", "pk_fma_indep", "mixed" };
    printf("{\"device\": \"%s\", \"arch\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"unit\": \"G wave64 instructions/s, whole chip\",\n", p.name, p.gcnArchName, cus, p.clockRate / 1000);
    printf(" \"nominal\": {\"per_simd_2_cycles\": %.1f, \"per_simd_4_cycles\": %.1f},\n", cus * 4.0 * p.clockRate * 1e3 / 2.0 / 1e9, cus * 4.0 * p.clockRate * 1e3 / 4.0 / 1e9);
    for (int m = 0; m < 4; m++) {
        printf(" \"%s\": {", names[m]);
        const int ws[4] = { 1, 2, 4, 8 };
        for (int k = 0; k < 4; k++) {
            double r = 0;
            if (m == 0) r = run<0>(cus, ws[k], iters, out);
            if (m == 1) r = run<1>(cus, ws[k], iters, out);
            if (m == 2) r = run<2>(cus, ws[k], iters, out);
            if (m == 3) r = run<3>(cus, ws[k], iters, out);
            printf("\"waves_per_simd_%d\": %.1f%s", ws[k], r / 1e9, k < 3 ? ", " : "");
        }
        printf("}%s\n", m < 3 ? "," : "");
    }
    printf("}\n");
    CHECK(hipFree(out));
    return 0;
}
