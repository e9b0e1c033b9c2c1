// Micro-benchmark: issue rate of v_mfma_f32_32x32x16_bf16 / v_mfma_f32_16x16x32_bf16 for dependent chains of 1..3
// accumulators, one or two waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int CHAIN>   // NACC accumulators, CHAIN consecutive MFMAs on the same accumulator before moving on
__global__ void k32(float* out, unsigned long long* cyc, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f - threadIdx.x * 0.002f); }
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n) acc[n] = f32x16{0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n)
#pragma unroll
            for (int c = 0; c < CHAIN; ++c) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) s += acc[n][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 8 * 4096);
    unsigned long long h[4096];
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define RUN(NACC, CHAIN, THREADS)                                                                       \
    hipLaunchKernelGGL((k32<NACC, CHAIN>), dim3(256), dim3(THREADS), 0, 0, out, cyc, iters);          \
    hipEventRecord(e0); hipLaunchKernelGGL((k32<NACC, CHAIN>), dim3(256), dim3(THREADS), 0, 0, out, cyc, iters); hipEventRecord(e1); \
    hipDeviceSynchronize(); hipMemcpy(h, cyc, 8 * 256, hipMemcpyDeviceToHost);                         \
    { float ms; hipEventElapsedTime(&ms, e0, e1);                                                       \
      const double nm = double(iters) * NACC * CHAIN;                                                   \
      printf("32x32x16: %d acc x chain %d, %d waves/CU: %.1f cycles per MFMA per wave; %.3f ms -> %.0f TFLOP/s, clock %.2f GHz\n", NACC, CHAIN, THREADS / 64, \
             double(h[7]) / nm, ms, 256.0 * (THREADS / 64) * nm * 32768.0 / (ms * 1e-3) / 1e12, double(h[7]) / (ms * 1e-3) / 1e9); }
    RUN(1, 1, 256) RUN(1, 3, 256) RUN(2, 3, 256) RUN(3, 3, 256) RUN(2, 1, 256)
    RUN(1, 3, 512) RUN(2, 3, 512) RUN(2, 1, 512)
    return 0;
}
