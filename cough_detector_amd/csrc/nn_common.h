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

// NaN rule of the reference's classifiers.  torch's ReLU and max-pool PROPAGATE NaN (relu(NaN) = NaN, max(NaN, x) = NaN), every
// convolution mixes all input channels and every input pixel lies under some window of every (strided) conv, and the global
// average pool sees every position: ONE NaN pixel in a clip's image makes every logit of that clip NaN -- which is what the
// reference's own default constructor produces (6 contrast bands -> NaN rows, /root/reference/src/preprocessing.py:272-300), so
// its engine never fires on such a configuration.  ReLU and pooling here are v_max_f32, which returns the non-NaN operand;
// instead of paying a compare + select at every ReLU, the rule is applied where it ends up: this kernel runs after the head, one
// workgroup per clip, takes the verdict of the stem's staging loop (flag) or scans the image itself (flag == nullptr), and
// overwrites the clip's logits / probabilities with NaN (argmax of two NaNs: index 0, as torch.argmax).
__global__ __launch_bounds__(256) void nan_rule_kernel(const float* __restrict__ x, long long per_clip, const int* __restrict__ flag,
                                                       float* __restrict__ logits, float* __restrict__ probs,
                                                       int* __restrict__ preds) {
    const long long clip = blockIdx.x;
    int bad = 0;
    if (flag) {
        bad = flag[clip];
    } else {
        const float* p = x + clip * per_clip;
        for (long long i = threadIdx.x; i < per_clip; i += 256) bad |= (p[i] != p[i]) ? 1 : 0;
        bad = __syncthreads_or(bad);
    }
    if (bad && threadIdx.x == 0) {
        const float nan = __builtin_nanf("");
        logits[clip * 2] = nan;
        logits[clip * 2 + 1] = nan;
        if (probs) {
            probs[clip * 2] = nan;
            probs[clip * 2 + 1] = nan;
        }
        if (preds) preds[clip] = 0;
    }
}

}  // namespace
}  // namespace cough
