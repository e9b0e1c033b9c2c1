// K1 -- fused featuriser for gfx950: waveform -> log-mel(64) + MFCC(13) + delta(13) [+ delta-delta].
//
// Replaces AudioPreprocessor.normalize / extract_features
// (/root/reference/src/preprocessing.py:199-212, :432-489) and the torchaudio transforms it calls
// (T.MelSpectrogram :94-106, T.AmplitudeToDB :109-112, T.MFCC :116-127).  The reference runs the
// STFT->mel->dB chain twice per clip (:398 and :425); here one pass feeds both branches.
//
// One workgroup (8 waves) per clip; the whole clip lives in LDS.  All three per-clip reductions
// (peak for normalize, dB max for top_db, MFCC mean/std) are block reductions, so batches keep the
// reference's per-clip semantics.
//
//   P0  stage: coalesced float4 global loads -> (optional peak-normalise) -> LDS, reflect pad 256|N|256
//   P1  per frame (16 lanes each, 4 frames per wave pass): window * samples, packed as 256 complex
//       points; 256-pt FFT as radix-16 x radix-16 with ONE LDS transpose; real-input split for
//       bins 0..127 (lanes i and 16-i trade their upper halves by shuffle); |X|^2 -> LDS;
//       sparse mel (<=8 taps per band, taps and start bin held in registers); 10*log10 -> LDS
//   P2  block max -> top_db floor -> mel rows out; 13x64 DCT; mean / unbiased std; z-score, deltas out
//
// Only bins 4..127 feed the shipped 100 Hz-4 kHz filterbank, so the upper half
// of the spectrum is never formed.
#include <cmath>
#include <cstring>
#include <vector>

#include "common.h"

namespace cough {
namespace {

constexpr int NS = 16000, NFFT = 512, HOP = 160, WIN = 400, NFRAMES = 101, NMEL = 64, NMFCC = 13;
constexpr int PADL = NFFT / 2, NPAD = NS + 2 * PADL;
constexpr int NBIN = 128;   // spectrum bins formed (0..127)
constexpr int MAXW = 8;     // max non-zero taps of one mel band
constexpr int THREADS = 512, WAVES = THREADS / 64;
constexpr int FPW = 4;      // frames per wave pass
constexpr int NGROUP = (NFRAMES + FPW - 1) / FPW;
constexpr int XROW = 17;    // float2 per transpose row (16 + 1 pad: conflict-free ds_read_b64)
constexpr int XFRAME = 16 * XROW;
constexpr int NMF = NMFCC * NFRAMES;  // 1313

struct FeatTables {
    float win[NFFT];          // periodic Hann(400) zero-padded 56|400|56
    float2 tw256[16][16];     // W256^(j*k1), [j][k1]
    float2 tw512[NBIN];       // W512^k
    int mel_start[NMEL];      // first bin of band m
    float mel_w[NMEL][MAXW];  // taps of band m from mel_start
    float dct_t[NMFCC][NMEL]; // DCT-II ortho, [coeff][mel]
};

constexpr size_t LDS_PAD = size_t(NPAD) * 4;
constexpr size_t LDS_XCH = size_t(WAVES) * FPW * XFRAME * 8;
constexpr size_t LDS_MEL = size_t(NMEL) * NFRAMES * 4;
constexpr size_t LDS_RED = 64 * 4;
constexpr size_t LDS_TOTAL = LDS_PAD + LDS_XCH + LDS_MEL + LDS_RED;
static_assert(LDS_TOTAL <= 160 * 1024, "LDS budget");
static_assert(LDS_XCH >= size_t(3) * NMF * 4, "MFCC / delta buffers alias the transpose scratch");

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

__device__ __forceinline__ float block_max(float v, float* red, int tid) {
    v = wave_max(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float r = red[0];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) r = fmaxf(r, red[w]);
    return r;
}
__device__ __forceinline__ float block_sum(float v, float* red, int tid) {
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float r = red[0];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) r += red[w];
    return r;
}

__global__ __launch_bounds__(THREADS) void featurize_kernel(
    const float* __restrict__ wav, long long wav_stride, float* __restrict__ out, int nfeat,
    const FeatTables* __restrict__ tb, int normalize, int pre_emph, float pre_coef, int delta_delta) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* pad = reinterpret_cast<float*>(smem);
    float2* xch = reinterpret_cast<float2*>(smem + LDS_PAD);
    float* melbuf = reinterpret_cast<float*>(smem + LDS_PAD + LDS_XCH);
    float* red = reinterpret_cast<float*>(smem + LDS_PAD + LDS_XCH + LDS_MEL);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long clip = blockIdx.x;
    const float* x = wav + clip * wav_stride;
    float* o = out + clip * (long long)nfeat * NFRAMES;

    // ---------------- P0: stage the clip ----------------
    {
        float4 v[8];
        float amax = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int idx = r * THREADS + tid;
            v[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < NS / 4) v[r] = reinterpret_cast<const float4*>(x)[idx];
            amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[r].x), fabsf(v[r].y)), fmaxf(fabsf(v[r].z), fabsf(v[r].w))));
        }
        if (normalize) {   // waveform / waveform.abs().max() if max > 0 (preprocessing.py:209-212)
            const float m = block_max(amax, red, tid);
            if (m > 0.f) {
#pragma unroll
                for (int r = 0; r < 8; ++r) { v[r].x /= m; v[r].y /= m; v[r].z /= m; v[r].w /= m; }
            }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int idx = r * THREADS + tid;
            if (idx < NS / 4) reinterpret_cast<float4*>(pad + PADL)[idx] = v[r];
        }
    }
    __syncthreads();
    if (pre_emph) {   // y[n] = x[n] - coef*x[n-1], y[0] = x[0] (preprocessing.py:235-238); no FMA contraction
        float y[(NS + THREADS - 1) / THREADS];
#pragma unroll
        for (int r = 0; r < (NS + THREADS - 1) / THREADS; ++r) {
            const int n = r * THREADS + tid;
            y[r] = 0.f;
            if (n < NS) y[r] = (n == 0) ? pad[PADL] : __fsub_rn(pad[PADL + n], __fmul_rn(pre_coef, pad[PADL + n - 1]));
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < (NS + THREADS - 1) / THREADS; ++r) {
            const int n = r * THREADS + tid;
            if (n < NS) pad[PADL + n] = y[r];
        }
        __syncthreads();
    }
    // reflect padding (torch.stft center=True, pad_mode="reflect"): 512 pad samples, one per thread
    if (tid < PADL) pad[tid] = pad[2 * PADL - tid];
    else pad[NS + tid] = pad[NS + 2 * PADL - 2 - tid];   // p = NS+PADL+q <- PADL + (NS-2-q), q = tid-PADL
    __syncthreads();

    // ---------------- P1: STFT power -> mel -> dB ----------------
    const int j = lane & 15, fsub = lane >> 4;
    float w_re[16], w_im[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
        w_re[n1] = tb->win[32 * n1 + 2 * j];
        w_im[n1] = tb->win[32 * n1 + 2 * j + 1];
    }
    float2 tw_a[16];
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) tw_a[k1] = tb->tw256[j][k1];
    float2 tw_r[8];
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) tw_r[k2] = tb->tw512[j + 16 * k2];
    float mw[MAXW];
#pragma unroll
    for (int q = 0; q < MAXW; ++q) mw[q] = tb->mel_w[lane][q];
    const int mstart = tb->mel_start[lane];

    float2* myx = xch + (wave * FPW + fsub) * XFRAME;
    float run_max = -INFINITY;

    for (int g = wave; g < NGROUP; g += WAVES) {
        const int t = FPW * g + fsub;
        const float* fp = pad + (t < NFRAMES ? t : NFRAMES - 1) * HOP;   // idle sub-frames redo the last frame
        float2 a[16];
        a[0] = make_float2(0.f, 0.f);    // window is zero on samples [0,56) and [456,512)
        a[15] = make_float2(0.f, 0.f);
#pragma unroll
        for (int n1 = 1; n1 < 15; ++n1) {
            const float2 s = *reinterpret_cast<const float2*>(fp + 32 * n1 + 2 * j);
            a[n1] = make_float2(s.x * w_re[n1], s.y * w_im[n1]);
        }
        dft16(a);
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) myx[k1 * XROW + j] = (k1 == 0) ? a[0] : cmul(a[k1], tw_a[k1]);
        wave_lds_fence();
        float2 z[16];
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) z[n2] = myx[j * XROW + n2];
        dft16(z);   // z[k2] = Z[j + 16*k2]

        // partner Z[256-k] lives in lane (16-j)&15, register 15-k2 (j>=1) or 16-k2 (j==0)
        const int src = (lane & 48) | ((16 - j) & 15);
        float2 rv[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            rv[r].x = __shfl(z[8 + r].x, src, 64);
            rv[r].y = __shfl(z[8 + r].y, src, 64);
        }
        wave_lds_fence();
        float* pf = reinterpret_cast<float*>(myx);
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) {
            const float2 zk = z[k2];
            const float2 zp0 = (k2 == 0) ? z[0] : rv[8 - k2];   // j == 0
            const float2 zp = (j == 0) ? zp0 : rv[7 - k2];
            const float ex = 0.5f * (zk.x + zp.x), ey = 0.5f * (zk.y - zp.y);
            const float ox = 0.5f * (zk.y + zp.y), oy = -0.5f * (zk.x - zp.x);
            const float xr = ex + tw_r[k2].x * ox - tw_r[k2].y * oy;
            const float xi = ey + tw_r[k2].x * oy + tw_r[k2].y * ox;
            pf[j + 16 * k2] = xr * xr + xi * xi;
        }
        wave_lds_fence();
        // sparse mel: lane = band, the wave's 4 frames
#pragma unroll
        for (int f = 0; f < FPW; ++f) {
            const float* p = reinterpret_cast<const float*>(xch + (wave * FPW + f) * XFRAME) + mstart;
            float acc = 0.f;
#pragma unroll
            for (int q = 0; q < MAXW; ++q) acc += mw[q] * p[q];
            const int tf = FPW * g + f;
            if (tf < NFRAMES) {
                const float db = 10.0f * log10f(fmaxf(acc, 1e-10f));   // AmplitudeToDB('power'), amin 1e-10
                melbuf[lane * NFRAMES + tf] = db;
                run_max = fmaxf(run_max, db);
            }
        }
        wave_lds_fence();
    }

    // ---------------- P2: top_db floor, mel rows, DCT, z-score, deltas ----------------
    const float floor_db = block_max(run_max, red, tid) - 80.0f;   // per-clip max
    for (int idx = tid; idx < NMEL * NFRAMES; idx += THREADS) {
        const float d = fmaxf(melbuf[idx], floor_db);
        melbuf[idx] = d;
        o[idx] = fminf(fmaxf((d + 80.0f) / 80.0f, 0.f), 1.f);      // preprocessing.py:409-410
    }
    __syncthreads();
    float* mf = reinterpret_cast<float*>(xch);   // [13][101] MFCC, then z-scored in place
    float* dl = mf + NMF;                        // delta (needed in LDS only for delta-delta)
    float lsum = 0.f;
    for (int item = tid; item < NMF; item += THREADS) {
        const int c = item / NFRAMES, t = item - c * NFRAMES;
        float acc = 0.f;
#pragma unroll 16
        for (int m = 0; m < NMEL; ++m) acc += tb->dct_t[c][m] * melbuf[m * NFRAMES + t];
        mf[item] = acc;
        lsum += acc;
    }
    const float mean = block_sum(lsum, red, tid) / float(NMF);
    float lsq = 0.f;
    for (int item = tid; item < NMF; item += THREADS) {
        const float d = mf[item] - mean;
        lsq += d * d;
    }
    const float sd = sqrtf(block_sum(lsq, red, tid) / float(NMF - 1));   // torch.std: unbiased
    const float denom = sd + 1e-8f;                                        // preprocessing.py:428
    for (int item = tid; item < NMF; item += THREADS) mf[item] = (mf[item] - mean) / denom;
    __syncthreads();
    float* o_mfcc = o + NMEL * NFRAMES;
    float* o_delta = o_mfcc + NMF;
    for (int item = tid; item < NMF; item += THREADS) {
        const int c = item / NFRAMES, t = item - c * NFRAMES;
        const float* row = mf + c * NFRAMES;
        const float d = (row[t < NFRAMES - 1 ? t + 1 : t] - row[t > 0 ? t - 1 : 0]) / 2.0f;   // :353-355
        o_mfcc[item] = row[t];
        o_delta[item] = d;
        if (delta_delta) dl[item] = d;
    }
    if (delta_delta) {
        __syncthreads();
        float* o_dd = o_delta + NMF;
        for (int item = tid; item < NMF; item += THREADS) {
            const int c = item / NFRAMES, t = item - c * NFRAMES;
            const float* row = dl + c * NFRAMES;
            o_dd[item] = (row[t < NFRAMES - 1 ? t + 1 : t] - row[t > 0 ? t - 1 : 0]) / 2.0f;
        }
    }
}

}  // namespace
}  // namespace cough

// ------------------------------------------------------------------------------------------ C-ABI
struct cough_featurizer {
    cough_feat_config cfg;
    cough::FeatTables* d_tables;
    int nfeat;
};

extern "C" int cough_featurizer_create(cough_featurizer** out, const cough_feat_config* cfg,
                                       const float* window, const float* mel_fb, const float* dct) {
    using namespace cough;
    COUGH_REQUIRE(out && cfg && window && mel_fb && dct, COUGH_EINVAL, "cough_featurizer_create: NULL argument");
    COUGH_REQUIRE(cfg->sample_rate == 16000 && cfg->n_fft == NFFT && cfg->hop_length == HOP &&
                      cfg->win_length == WIN && cfg->n_mels == NMEL && cfg->n_mfcc == NMFCC &&
                      cfg->segment_samples == NS,
                  COUGH_EUNSUPPORTED,
                  "featuriser geometry not implemented on the HIP path: need sample_rate=16000 n_fft=512 "
                  "hop_length=160 win_length=400 n_mels=64 n_mfcc=13 segment=16000 samples");
    std::vector<FeatTables> host(1);
    FeatTables& t = host[0];
    std::memset(&t, 0, sizeof(t));
    const int left = (NFFT - WIN) / 2;
    for (int n = 0; n < WIN; ++n) t.win[left + n] = window[n];
    const double PI = 3.14159265358979323846;
    for (int jj = 0; jj < 16; ++jj)
        for (int k1 = 0; k1 < 16; ++k1) {
            const double a = -2.0 * PI * double(jj * k1) / 256.0;
            t.tw256[jj][k1] = make_float2(float(std::cos(a)), float(std::sin(a)));
        }
    for (int k = 0; k < NBIN; ++k) {
        const double a = -2.0 * PI * double(k) / 512.0;
        t.tw512[k] = make_float2(float(std::cos(a)), float(std::sin(a)));
    }
    const int nfreq = NFFT / 2 + 1;
    for (int m = 0; m < NMEL; ++m) {
        int first = -1, last = -1;
        for (int k = 0; k < nfreq; ++k)
            if (mel_fb[k * NMEL + m] != 0.f) { if (first < 0) first = k; last = k; }
        if (first < 0) { first = 0; last = 0; }   // empty band: all-zero taps
        COUGH_REQUIRE(last < NBIN && last - first < MAXW, COUGH_EUNSUPPORTED,
                      "mel band %d spans bins %d..%d: the HIP path needs bands within bins 0..127 and <= 8 taps "
                      "(f_max <= sample_rate/4)", m, first, last);
        if (first > NBIN - MAXW) first = NBIN - MAXW;   // keep start+8 inside the power buffer
        t.mel_start[m] = first;
        for (int q = 0; q < MAXW; ++q) t.mel_w[m][q] = mel_fb[(first + q) * NMEL + m];
    }
    for (int c = 0; c < NMFCC; ++c)
        for (int m = 0; m < NMEL; ++m) t.dct_t[c][m] = dct[m * NMFCC + c];

    cough_featurizer* f = new cough_featurizer();
    f->cfg = *cfg;
    f->nfeat = NMEL + 2 * NMFCC + (cfg->use_delta_delta ? NMFCC : 0);
    f->d_tables = nullptr;
    hipError_t e = hipMalloc(&f->d_tables, sizeof(FeatTables));
    if (e == hipSuccess) e = hipMemcpy(f->d_tables, &t, sizeof(FeatTables), hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(featurize_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, int(LDS_TOTAL));
    if (e != hipSuccess) {
        set_error("cough_featurizer_create: %s", hipGetErrorString(e));
        if (f->d_tables) (void)hipFree(f->d_tables);
        delete f;
        return COUGH_EHIP;
    }
    *out = f;
    return COUGH_OK;
}

extern "C" void cough_featurizer_destroy(cough_featurizer* f) {
    if (!f) return;
    if (f->d_tables) (void)hipFree(f->d_tables);
    delete f;
}

extern "C" int cough_featurizer_num_features(const cough_featurizer* f) { return f ? f->nfeat : -1; }
extern "C" int cough_featurizer_num_frames(const cough_featurizer* f) { return f ? cough::NFRAMES : -1; }

extern "C" int cough_featurize(const cough_featurizer* f, const float* d_wav, long long wav_stride,
                               float* d_feat, int n_clips, int flags, void* stream) {
    using namespace cough;
    COUGH_REQUIRE(f && d_wav && d_feat, COUGH_EINVAL, "cough_featurize: NULL argument");
    COUGH_REQUIRE(n_clips >= 0, COUGH_EINVAL, "cough_featurize: n_clips < 0");
    COUGH_REQUIRE(wav_stride >= NS && (wav_stride & 3) == 0 && (reinterpret_cast<size_t>(d_wav) & 15) == 0,
                  COUGH_EINVAL, "cough_featurize: d_wav must be 16-byte aligned with a row stride >= 16000, multiple of 4");
    if (n_clips == 0) return COUGH_OK;
    hipLaunchKernelGGL(featurize_kernel, dim3(n_clips), dim3(THREADS), LDS_TOTAL, static_cast<hipStream_t>(stream),
                       d_wav, wav_stride, d_feat, f->nfeat, f->d_tables, (flags & COUGH_FEAT_NORMALIZE) ? 1 : 0,
                       f->cfg.use_pre_emphasis, f->cfg.pre_emphasis_coef, f->cfg.use_delta_delta);
    COUGH_HIP_CHECK(hipGetLastError());
    return COUGH_OK;
}
