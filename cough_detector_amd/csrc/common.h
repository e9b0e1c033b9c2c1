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
// DPP reductions (no LDS crossbar round trips): quad swaps, half-row / row mirrors give every lane its
// 16-lane row total; two xor-shuffles then combine the four rows.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));    // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp_mov<0x4E>(v));    // quad_perm [2,3,0,1]
    v = fmaxf(v, dpp_mov<0x141>(v));   // row_half_mirror
    v = fmaxf(v, dpp_mov<0x140>(v));   // row_mirror: every lane holds its row's max
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);            // every lane holds its row's sum
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// A float32 product the compiler cannot contract into an FMA with a neighbouring add / subtract: hipcc's __fmul_rn / __fadd_rn
// are the plain operators, and under HIP's default -ffp-contract=fast `__fsub_rn(x, __fmul_rn(c, y))` becomes ONE v_fma_f32
// (exact product), 1 ulp away from torch's separately rounded `x - c * y` in a third of the elements.  Wherever the
// reference's arithmetic is two IEEE operations (pre-emphasis, the clip generator's parameter arithmetic) use this.
__device__ __forceinline__ float mul_rn(float a, float b) {
    float r;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Intra-wave LDS hand-off: LDS executes a wave's accesses in order, so only the compiler
// has to be kept from reordering across this point.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

}  // namespace cough
