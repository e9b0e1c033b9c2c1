// Micro-benchmark: VALU issue rate on gfx950 by waves per SIMD.  Each wave runs a long stream of INDEPENDENT
// v_fma_f32 (16 accumulators) / v_pk_fma_f32; reports SIMD cycles per wave-instruction = cycles / (instructions
// per wave x waves per SIMD).  Settles whether one SIMD retires a wave64 f32 VALU op every 4 cycles (SIMD-16
// cadence) or every 2 (SIMD-32) once >= 2 waves are resident -- i.e. what "VALU-bound" means for the featuriser.
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>   // 0: v_fma_f32, 1: v_pk_fma_f32, 2: v_fma_f32 in a dependent chain per 4 accumulators
__global__ void k(float* out, unsigned long long* cyc, int iters, float s) {
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
    const float m = 1.0f + s * 1e-9f, c = s * 1e-7f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        } else if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    f2 v = {a[i], a[i + 1]}, mm = {m, m}, cc = {c, c};
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(mm), "v"(cc));
                    a[i] = v[0]; a[i + 1] = v[1];
                }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float t = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 8 * 4096);
    unsigned long long h[4096];
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"v_fma_f32 independent", "v_pk_fma_f32 independent", "v_fma_f32 4 chains (dependent every 4th)"};
#define RUN(MODE, THREADS)                                                                                  \
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(THREADS), 0, 0, out, cyc, iters, 1.0f);                   \
    hipEventRecord(e0); hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(THREADS), 0, 0, out, cyc, iters, 1.0f); hipEventRecord(e1); \
    hipDeviceSynchronize(); hipMemcpy(h, cyc, 8 * 256, hipMemcpyDeviceToHost);                             \
    { float ms; hipEventElapsedTime(&ms, e0, e1);                                                           \
      const double ninst = double(iters) * (MODE == 1 ? 32 : 64), wps = THREADS / 256.0;                    \
      printf("%-44s %4.1f waves/SIMD: %.2f cycles per wave-instruction per wave, %.2f SIMD cycles per instruction; %.3f ms, clock %.2f GHz\n", \
             names[MODE], wps, double(h[7]) / ninst, double(h[7]) / (ninst * (wps < 1 ? 1 : wps)), ms, double(h[7]) / (ms * 1e-3) / 1e9); }
    RUN(0, 256) RUN(0, 512) RUN(0, 768) RUN(0, 1024)
    RUN(1, 256) RUN(1, 512) RUN(1, 1024)
    RUN(2, 256) RUN(2, 512) RUN(2, 768) RUN(2, 1024)
    return 0;
}
