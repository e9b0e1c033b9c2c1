// Shared host/device helpers for libcough_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/cough_amd.h"

namespace cough {

void set_error(const char* fmt, ...);

#define COUGH_HIP_CHECK(expr)                                                              \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            ::cough::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return COUGH_EHIP;                                                             \
        }                                                                                  \
    } while (0)

#define COUGH_REQUIRE(cond, code, ...)       \
    do {                                     \
        if (!(cond)) {                       \
            ::cough::set_error(__VA_ARGS__); \
            return (code);                   \
        }                                    \
    } while (0)

// ---- wave64 / block reductions -------------------------------------------------------------
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Intra-wave LDS hand-off: LDS executes a wave's accesses in order, so only the compiler
// has to be kept from reordering across this point.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

}  // namespace cough
