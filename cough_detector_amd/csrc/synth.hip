// On-device synthetic clip source for the benchmark stream (BASELINE.json configs[3]: "1M-clip synthetic stream",
// "generated on-device from seed").  The value distributions follow the reference's own
// synthetic generators (/root/reference/setup_coughvid.py:381-441: cough-like burst, silence, white noise, hum,
// clicks, speech-like sine stacks; /root/reference/prepare_data.py:136-163), 1 s @ 16 kHz, mixture by seed % 6 --
// the recipe of cough_detector_amd/synth.py:make_clip -- with numpy's sequential PCG64 stream replaced by a
// counter-based hash RNG, so that every sample is a pure function of (seed, sample index) and any rank can produce its
// own shard.  cough_detector_amd/synth.py:make_clip_counter is the host mirror (same float32 operation sequence);
// tests compare the two sample by sample.
#include "common.h"

namespace cough {
namespace {

constexpr int SN = 16000;
constexpr float TWO_PI = 6.28318530717958647692f;

__device__ __forceinline__ uint32_t hash32(uint32_t x) {   // "lowbias32"
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t key(uint32_t seed, uint32_t stream, uint32_t idx) {
    return hash32(hash32(seed * 0x9E3779B9u + stream) + idx);
}
__device__ __forceinline__ float u01(uint32_t k) { return float(k >> 8) * 5.9604644775390625e-8f; }          // [0, 1)
__device__ __forceinline__ float u01o(uint32_t k) { return float((k >> 8) + 1u) * 5.9604644775390625e-8f; }  // (0, 1]
__device__ __forceinline__ float normal(uint32_t seed, uint32_t stream, int i) {   // Box-Muller
    const float u1 = u01o(key(seed, stream, 2u * i)), u2 = u01(key(seed, stream, 2u * i + 1u));
    return sqrtf(-2.0f * logf(u1)) * cosf(TWO_PI * u2);
}
__device__ __forceinline__ float param(uint32_t seed, int p) { return u01(key(seed, 0xF00Du, uint32_t(p))); }
// sin(2*pi*f*t) at sample i for a frequency of fq/16 Hz: the phase f*i/16000 = fq*i/256000 is reduced in exact
// integer arithmetic (fq <= 16000, i < 16000: the product fits 32 bits), so it carries no float32 rounding of a
// large argument and the host mirror reproduces it bit for bit
__device__ __forceinline__ float tone(int fq, int i) {
    const uint32_t r = (uint32_t(fq) * uint32_t(i)) % 256000u;
    return sinf(TWO_PI * mul_rn(float(r), 3.90625e-6f));
}
__device__ __forceinline__ int freq16(float lo16, float span16, float x) { return int(__fadd_rn(lo16, mul_rn(span16, x))); }
// parameter arithmetic is pinned to separate IEEE multiplies / adds (no FMA contraction): the host mirror must
// reproduce every floor() below exactly
__device__ __forceinline__ float affine(float a, float b, float x) { return __fadd_rn(a, mul_rn(b, x)); }

__global__ __launch_bounds__(256) void synth_clips_kernel(float* __restrict__ out, long long stride, long long first_seed,
                                                          long long seed_stride) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const long long seed64 = first_seed + (long long)blockIdx.x * seed_stride;
    const uint32_t seed = uint32_t(seed64);
    const int kind = int(seed64 % 6);
    float* o = out + (long long)blockIdx.x * stride;
    if (kind == 0) {          // cough-like burst: 20 ms linear attack + exp(-5u) decay over a noise floor
        const float dur = affine(0.3f, 0.5f, param(seed, 0));
        const int n_burst = int(mul_rn(dur, 16000.0f));
        const int start = int(mul_rn(mul_rn(param(seed, 1), __fsub_rn(1.0f, dur)), 16000.0f));
        const int n_att = 320;
        const int f1 = freq16(1280.0f, 1120.0f, param(seed, 2)), f2 = freq16(3200.0f, 3200.0f, param(seed, 3));   // 80-150, 200-400 Hz
        const float inv_att = 1.0f / float(n_att - 1), inv_dec = 5.0f / float(n_burst - n_att - 1);
        float mx = 0.f;
        for (int i = tid; i < SN; i += 256) {
            const int j = i - start;
            float v = 0.f;
            if (j >= 0 && j < n_burst) {
                const float env = j < n_att ? float(j) * inv_att : expf(-(float(j - n_att) * inv_dec));
                v = env * (0.7f * normal(seed, 1, i) + 0.2f * tone(f1, i) + 0.1f * tone(f2, i));
            }
            o[i] = v;
            mx = fmaxf(mx, fabsf(v));
        }
        mx = wave_max(mx);
        if ((tid & 63) == 0) red[tid >> 6] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        const float g = 0.8f / (mx + 1e-8f);
        for (int i = tid; i < SN; i += 256) o[i] = o[i] * g + 0.01f * normal(seed, 2, i);   // own elements: no barrier needed
    } else if (kind == 1) {   // near silence
        for (int i = tid; i < SN; i += 256) o[i] = 0.005f * normal(seed, 1, i);
    } else if (kind == 2) {   // white noise
        const float sigma = affine(0.02f, 0.08f, param(seed, 0));
        for (int i = tid; i < SN; i += 256) o[i] = sigma * normal(seed, 1, i);
    } else if (kind == 3) {   // mains-like hum
        const int sel = int(mul_rn(param(seed, 0), 4.0f));
        const int f = sel == 0 ? 800 : sel == 1 ? 960 : sel == 2 ? 1600 : 1920;   // 50 / 60 / 100 / 120 Hz
        for (int i = tid; i < SN; i += 256) o[i] = 0.1f * tone(f, i) + 0.02f * normal(seed, 1, i);
    } else if (kind == 4) {   // clicks on a floor: 1-4 plateaus of 50 samples, later ones overwrite earlier ones
        const int cnt = 1 + int(mul_rn(param(seed, 0), 4.0f));
        int pos[4];
        float val[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            pos[c] = int(mul_rn(param(seed, 1 + 2 * c), float(SN - 100)));
            val[c] = affine(-0.3f, 0.6f, param(seed, 2 + 2 * c));
        }
        for (int i = tid; i < SN; i += 256) {
            float v = 0.01f * normal(seed, 1, i);
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < cnt && i >= pos[c] && i < pos[c] + 50) v = val[c];
            o[i] = v;
        }
    } else {                  // speech-like stack of 2-4 sines
        const int cnt = 2 + int(mul_rn(param(seed, 0), 3.0f));
        int f[4];
        float a[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f[c] = freq16(1600.0f, 14400.0f, param(seed, 1 + 2 * c));   // 100-1000 Hz
            a[c] = affine(0.05f, 0.1f, param(seed, 2 + 2 * c));
        }
        for (int i = tid; i < SN; i += 256) {
            float v = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < cnt) v += a[c] * tone(f[c], i);
            o[i] = v + 0.02f * normal(seed, 1, i);
        }
    }
}

}  // namespace
}  // namespace cough

extern "C" int cough_synth_clips(float* d_out, long long stride, int n_clips, long long first_seed, long long seed_stride,
                                 void* stream) {
    using namespace cough;
    COUGH_REQUIRE(d_out, COUGH_EINVAL, "cough_synth_clips: NULL argument");
    COUGH_REQUIRE(n_clips >= 0 && stride >= SN && first_seed >= 0 && seed_stride >= 1, COUGH_EINVAL,
                  "cough_synth_clips: need n_clips >= 0, stride >= 16000, first_seed >= 0, seed_stride >= 1");
    if (n_clips == 0) return COUGH_OK;
    hipLaunchKernelGGL(synth_clips_kernel, dim3(n_clips), dim3(256), 0, static_cast<hipStream_t>(stream), d_out, stride,
                       first_seed, seed_stride);
    COUGH_HIP_CHECK(hipGetLastError());
    return COUGH_OK;
}
