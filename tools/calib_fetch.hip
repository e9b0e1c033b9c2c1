// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the featurise kernel
// uses (MI355X_MICROARCH.md "HBM": FETCH_SIZE reads 1/2 of a 16-B/lane stream; other widths uncalibrated).
// Streams a buffer larger than the 256 MiB Infinity Cache with 4-, 8- and 16-byte-per-lane loads and
// 4- / 8-byte-per-lane stores; compare the counters with the known byte counts printed here.
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T>
__global__ void rd(const T* __restrict__ p, size_t n, float* sink) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        T v = p[i];
        acc += reinterpret_cast<const float*>(&v)[0];
    }
    if (acc == 1234.5f) *sink = acc;
}
template <typename T>
__global__ void wr(T* __restrict__ p, size_t n) {
    T v; for (unsigned k = 0; k < sizeof(T) / 4; ++k) reinterpret_cast<float*>(&v)[k] = 1.0f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
int main() {
    const size_t bytes = size_t(1) << 30;   // 1 GiB
    void* buf; float* sink;
    hipMalloc(&buf, bytes); hipMalloc(&sink, 4); hipMemset(buf, 0, bytes);
    hipDeviceSynchronize();
    rd<float><<<2048, 256>>>((const float*)buf, bytes / 4, sink);
    rd<float2><<<2048, 256>>>((const float2*)buf, bytes / 8, sink);
    rd<float4><<<2048, 256>>>((const float4*)buf, bytes / 16, sink);
    wr<float><<<2048, 256>>>((float*)buf, bytes / 4);
    wr<float2><<<2048, 256>>>((float2*)buf, bytes / 8);
    hipDeviceSynchronize();
    printf("each kernel moves %zu bytes = %zu KiB\n", bytes, bytes / 1024);
    return 0;
}
