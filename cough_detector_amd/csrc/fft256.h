// Register-resident FFT building blocks shared by the featuriser (featurize.hip) and the stand-alone STFT
// (spectrogram.hip): a 256-point complex FFT is radix-16 (registers) x radix-16 (registers) around one LDS
// transpose; 16 lanes hold one frame, 16 points each.
#pragma once
#include <hip/hip_runtime.h>

namespace cough {
namespace {

constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, RH = 0.70710678118654752f;

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// a * W16^M, W16 = exp(-2*pi*i/16)
template <int M>
__device__ __forceinline__ float2 mul_w16(float2 a) {
    if constexpr (M == 0) return a;
    else if constexpr (M == 4) return make_float2(a.y, -a.x);
    else if constexpr (M == 2) return make_float2(RH * (a.x + a.y), RH * (a.y - a.x));
    else if constexpr (M == 6) return make_float2(RH * (a.y - a.x), -RH * (a.x + a.y));
    else {
        constexpr float c = (M == 1) ? C1 : (M == 3) ? S1 : -C1;   // M == 9: (-C1, -S1)
        constexpr float s = (M == 1) ? S1 : (M == 3) ? C1 : -S1;
        return make_float2(a.x * c + a.y * s, a.y * c - a.x * s);
    }
}

__device__ __forceinline__ void radix4(float2& a0, float2& a1, float2& a2, float2& a3) {
    const float2 s0 = make_float2(a0.x + a2.x, a0.y + a2.y), s1 = make_float2(a0.x - a2.x, a0.y - a2.y);
    const float2 s2 = make_float2(a1.x + a3.x, a1.y + a3.y), s3 = make_float2(a1.x - a3.x, a1.y - a3.y);
    a0 = make_float2(s0.x + s2.x, s0.y + s2.y);
    a2 = make_float2(s0.x - s2.x, s0.y - s2.y);
    a1 = make_float2(s1.x + s3.y, s1.y - s3.x);
    a3 = make_float2(s1.x - s3.y, s1.y + s3.x);
}

// In-register forward 16-point DFT, natural order in and out (radix-4 x radix-4).
__device__ __forceinline__ void dft16(float2 (&x)[16]) {
    float2 t[16];
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) {
        float2 a0 = x[n2], a1 = x[4 + n2], a2 = x[8 + n2], a3 = x[12 + n2];
        radix4(a0, a1, a2, a3);
        t[4 * n2 + 0] = a0; t[4 * n2 + 1] = a1; t[4 * n2 + 2] = a2; t[4 * n2 + 3] = a3;
    }
    t[5] = mul_w16<1>(t[5]);   t[6] = mul_w16<2>(t[6]);   t[7] = mul_w16<3>(t[7]);
    t[9] = mul_w16<2>(t[9]);   t[10] = mul_w16<4>(t[10]); t[11] = mul_w16<6>(t[11]);
    t[13] = mul_w16<3>(t[13]); t[14] = mul_w16<6>(t[14]); t[15] = mul_w16<9>(t[15]);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
        float2 b0 = t[k1], b1 = t[4 + k1], b2 = t[8 + k1], b3 = t[12 + k1];
        radix4(b0, b1, b2, b3);
        x[k1] = b0; x[k1 + 4] = b1; x[k1 + 8] = b2; x[k1 + 12] = b3;
    }
}

// W32^k2 = exp(-2*pi*i*k2/32), k2 = 0..7 (compile-time constants of the real-input split)
__device__ constexpr float W32C[8] = {1.0f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f,
                                      0.70710678118654752f, 0.55557023301960218f, 0.38268343236508977f,
                                      0.19509032201612825f};
__device__ constexpr float W32S[8] = {0.0f, -0.19509032201612825f, -0.38268343236508977f, -0.55557023301960218f,
                                      -0.70710678118654752f, -0.83146961230254524f, -0.92387953251128674f,
                                      -0.98078528040323043f};

}  // namespace
}  // namespace cough
