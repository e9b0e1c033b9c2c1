// How many 256-thread workgroups share a CU as a function of their dynamic LDS size on gfx950: what the occupancy API says, and
// what the hardware does (a kernel whose workgroups spin for a fixed number of cycles: 12 workgroups per CU take 12 / occupancy
// rounds).  hipcc --offload-arch=gfx950 -O2 tools/micro/lds_occupancy.hip -o build_ab/lds_occupancy && build_ab/lds_occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256, 3) void k(float* p, long long cycles) {
    extern __shared__ float s[];
    s[threadIdx.x] = p[threadIdx.x];
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
    if (p[0] == 12345.f) p[threadIdx.x] = s[255 - threadIdx.x];
}
int main() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    float* d;
    (void)hipMalloc(&d, 4096);
    (void)hipMemset(d, 0, 4096);
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    int prev = -1;
    float prev_ms = 0;
    for (int lds = 36864; lds <= 163840; lds += 256) {
        int n = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 256, lds);
        hipLaunchKernelGGL(k, dim3(cus * 12), dim3(256), lds, 0, d, 20000LL);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(cus * 12), dim3(256), lds, 0, d, 20000LL);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (n != prev || ms > prev_ms * 1.15f || ms < prev_ms * 0.87f)
            printf("dynamic LDS %6d B: API %d workgroups per CU, %d x 12 spinning workgroups take %.3f ms\n", lds, n, cus, ms);
        prev = n;
        prev_ms = ms;
    }
    return 0;
}
