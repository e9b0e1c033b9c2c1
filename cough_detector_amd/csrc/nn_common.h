// Element types and small helpers shared by the classifier translation units (resnet.hip, cnn.hip).
#pragma once
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

namespace cough {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
typedef uint16_t bf16_t;   // storage type of bf16 activations / weights

__device__ __forceinline__ bf16_t f2bf(float f) {
    __hip_bfloat16 b = __float2bfloat16(f);
    return *reinterpret_cast<bf16_t*>(&b);
}
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(uint32_t(v) << 16); }
inline bf16_t f2bf_host(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return bf16_t((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return bf16_t(u >> 16);
}

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return f2bf(v); }
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return bf2f(v); }

// LDS image of NHWC bf16 pixels: the 16-byte channel chunks of a pixel are XOR-swizzled by the pixel index so that a
// ds_read_b128 of one chunk of 16 consecutive pixels touches 16 different 16-byte slots of the 256-byte bank row.
template <int C>
__device__ __forceinline__ int swz_off(int P, int j) {   // bf16 offset of 16-byte chunk j of pixel P
    constexpr int CH = C / 8, PPR = 16 / CH;             // chunks per pixel, pixels per 256-byte bank row
    return P * C + 8 * (j ^ int((unsigned(P) / PPR) & (CH - 1)));   // P >= 0: unsigned division is one shift
}

// f32 pair -> packed bf16 pair (round to nearest even, v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {a, b};
    const bf16x2_t p = __builtin_convertvector(v, bf16x2_t);
    return __builtin_bit_cast(uint32_t, p);
}
// x = hi + lo: hi = bf16(x), lo = bf16(x - hi) for 4 consecutive channels
__device__ __forceinline__ void split4(float a, float b, float c, float d, uint2& hi, uint2& lo) {
    hi.x = pk_bf16(a, b);
    hi.y = pk_bf16(c, d);
    lo.x = pk_bf16(a - __uint_as_float(hi.x << 16), b - __uint_as_float(hi.x & 0xffff0000u));
    lo.y = pk_bf16(c - __uint_as_float(hi.y << 16), d - __uint_as_float(hi.y & 0xffff0000u));
}

}  // namespace
}  // namespace cough
