// Generic-geometry featuriser for gfx950: every AudioPreprocessor constructor geometry at n_fft = 512 that the tuned
// one-kernel path (featurize.hip: 16 kHz, hop 160, win 400, 64 mel bands of <= 8 taps below bin 128, 13 MFCC, 1 s) does
// not cover -- any sample_rate / hop_length / win_length <= 512 / n_mels <= 256 / n_mfcc <= n_mels / f_min / f_max
// (dense filterbanks, bins up to 256) / segment_duration (any frame count), with every flag of the constructor.
//
// Replaces, for those geometries, AudioPreprocessor.__init__ / extract_features / normalize
// (/root/reference/src/preprocessing.py:32-144, :199-212, :432-489), RealtimePreprocessor(window_duration=...) (:559-580)
// and the engine's re-construction of the preprocessor from a checkpoint's config (/root/reference/src/inference.py:89-108).
//
// Not the throughput path: a chain of small kernels over a sub-batch whose intermediates (the power spectrogram and the
// mel powers) stay in L2 / the Infinity Cache, kernel boundaries instead of LDS capacity limits -- so the frame count is
// unbounded -- and the precise libm forms (log10f, powf, true divisions) of the reference's arithmetic:
//   gen_peak       per-clip max |x|                                         (normalize(), :199-212; only when asked)
//   gen_stft       16 frames per workgroup, the featuriser's 256-point complex FFT (fft256.h) with all 257 bins formed;
//                  samples are normalised (x * (1 / peak)) and pre-emphasised (:214-240) as they are loaded; reflect padding of
//                  torch.stft(center=True); <MEL>: the mel projection with a CSR filterbank (bands of any width) runs on
//                  the tile of powers in LDS                                -> melpow [clip][n_mels][T]
//                  (without MEL: the spectrogram [clip][257][T] itself -- cough_spectrogram, the contrast rows)
//   gen_dbstat     per-clip max dB (AmplitudeToDB's top_db floor is per clip), PCEN min / max
//   gen_rows       mel rows (log-mel :405-410 or PCEN :305-340, :400-404) and the raw MFCC rows (DCT of the floored dB)
//   gen_zscore     (x - mean) / (std + 1e-8) over a block of rows, unbiased std (:428, :300)
//   gen_delta      compute_deltas (:342-356) once or twice (:464, :471-474)
//   gen_contrast   spectral-contrast + centroid rows (:242-303) out of a power and a Hann(512) magnitude spectrogram
#include <cmath>
#include <cstring>
#include <vector>

#include "common.h"
#include "contrast_rank.h"
#include "fft256.h"
#include "internal.h"

namespace cough {
namespace {

constexpr int G_NFFT = 512, G_NFREQ = 257, G_PADL = 256;
constexpr int G_XROW = 17, G_XFRAME = 16 * G_XROW;
constexpr int G_FPB = 16;        // frames per gen_stft workgroup: 4 waves x 4 frames
constexpr int G_TT = 64;         // frames per workgroup of the per-frame kernels (lane = frame)
constexpr int G_MAX_MELS = 256;   // gen_rows_kernel's dB tile is n_mels x 64 frames of dynamic LDS: 64 KB at 256 bands
constexpr int G_CT_BINS = 128;   // spectral-contrast bands up to this many bins fill the 32 KB tile with 64 frames (wider: fewer frames)
constexpr size_t G_SUB_BYTES = size_t(192) << 20;   // intermediates of one sub-batch: inside the 256 MiB Infinity Cache

__device__ __forceinline__ float g_block_max(float v, float* red, int tid) {
    v = wave_max(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ __forceinline__ float g_block_sum(float v, float* red, int tid) {
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// ------------------------------------------------------------------------------------------------ peak
__global__ __launch_bounds__(256) void gen_peak_kernel(const float* __restrict__ wav, long long stride, int N,
                                                       float* __restrict__ peaks) {
    __shared__ float red[4];
    const float* x = wav + blockIdx.x * stride;
    float m = 0.f;
    int isnan = 0;   // torch's max propagates NaN (`waveform.abs().max()`, :209): the clip is then left unscaled (NaN > 0 is False)
    for (int i = threadIdx.x; i < N; i += 256) {
        const float v = x[i];
        m = fmaxf(m, fabsf(v));
        isnan |= v != v;
    }
    m = g_block_max(m, red, threadIdx.x);
    isnan = __syncthreads_or(isnan);
    if (threadIdx.x == 0) peaks[blockIdx.x] = isnan ? __builtin_nanf("") : m;
}

// x / peak of normalize() (:209-212) as x * (1 / peak) -- within 1 ulp, a third of the STFT kernels' VALU cheaper -- for every
// peak whose reciprocal is a normal number; a denormal peak (1 / peak overflows) and an infinite one (x / inf = 0, inf / inf =
// NaN) take the true division.  A NaN peak (a NaN sample) and a zero one leave the clip unscaled.
struct PeakScale {
    float m, inv;
    bool div;
    __device__ __forceinline__ explicit PeakScale(float peak) {
        const bool norm = peak > 0.f;            // False for NaN
        div = norm && !(peak >= 1.2e-38f && peak <= 3.0e38f);
        m = peak;
        inv = norm && !div ? 1.0f / peak : 1.0f;
    }
    __device__ __forceinline__ float operator()(float x) const { return div ? x / m : x * inv; }
};

// ------------------------------------------------------------------------------------------------ STFT
// One workgroup = 16 consecutive frames of one clip; 16 lanes per frame, as in featurize.hip / spectrogram.hip.
// MEL: the mel projection (CSR filterbank) runs here, out of the tile of powers in LDS, and `out` receives the mel powers
// [clip][n_mels][T] -- the 257-row spectrogram (103 KB per 101 frames) is then neither written nor re-read.
struct GenMel {
    int n_mels;
    const int *lo, *hi, *off;
    const float* w;
    int n_taps;   // length of w
};
// TAIL (n_fft = 512 only; `out` is then the feature image and the spectrogram never leaves the workgroup): 1 = the raw
// spectral-contrast rows straight out of the tile of powers (thread = (frame, band): contrast_rank.h's selection networks on LDS
// columns; bands of <= 64 bins, which the reference's 1 .. 16 geometric bands of 257 bins are); 2 = the spectral-centroid row out of the tile of magnitudes.
struct GenTail {
    ContrastCfg cfg;
    const float* freqs;   // [257] bin frequencies in Hz
    float nyquist;
    int nfeat, row0;      // rows of the feature image, first contrast row
};
// G_CHUNK consecutive 16-frame tiles per workgroup: the next tile's samples are requested (registers) before the current tile's
// FFTs run, and the window taps stay in registers across the tiles.  PE: pre-emphasis (a template parameter: its left-neighbour
// samples cost 32 more registers).
constexpr int G_CHUNK = 4;
template <bool MAG, bool MEL, int TAIL = 0, bool PE = false>
__global__ __launch_bounds__(256, TAIL == 1 ? 3 : 1) void gen_stft_kernel(const float* __restrict__ wav, long long stride, int N, int hop, int T,
                                                       const float* __restrict__ win, const float2* __restrict__ tw256,
                                                       const float2* __restrict__ tw512, const float* __restrict__ peaks,
                                                       float coef, float* __restrict__ out, GenMel mel,
                                                       long long n_rows /* clips x T */, GenTail tail) {
    static_assert(TAIL == 0 || (!MEL && MAG == (TAIL == 2)), "contrast rows from powers, the centroid from magnitudes");
    __shared__ float xs[4 * 4 * G_XFRAME];
    __shared__ float otile[G_NFREQ * (G_FPB + 1)];
    __shared__ float2 twl[16 * G_XROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, fsub = lane >> 4;
    // the 16 frames of a tile are 16 consecutive (clip, frame) pairs of the launch: T is not padded to whole tiles
    const int f16 = wave * 4 + fsub;
    const long long n_tiles = (n_rows + G_FPB - 1) / G_FPB, tile0 = (long long)blockIdx.x * G_CHUNK;
    const int ntile = int(n_tiles - tile0 < G_CHUNK ? n_tiles - tile0 : G_CHUNK);
    twl[(tid >> 4) * G_XROW + (tid & 15)] = tw256[tid];
    __syncthreads();
    const float2* tw_row = twl + j * G_XROW;
    const float2 tw_j = tw512[j];
    [[maybe_unused]] float2 wreg[TAIL == 1 ? 1 : 16];   // TAIL == 1: the selection networks need the registers, taps come from L1
    if constexpr (TAIL != 1) {
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) wreg[n1] = *reinterpret_cast<const float2*>(win + 32 * n1 + 2 * j);
    }
    // raw samples of this lane's frame of one tile (and, PE, their left neighbours in clip order; 0 in front of the clip: y[0] = x[0])
    float2 raw[16];
    [[maybe_unused]] float2 rawl[PE ? 16 : 1];
    float pk = 0.f;
    auto load_tile = [&](long long tile) {
        const long long r0 = tile * G_FPB, row = r0 + f16 < n_rows ? r0 + f16 : n_rows - 1;
        const long long clip = row / T;
        const int t = int(row - clip * T);
        const float* x = wav + clip * stride;
        pk = peaks ? peaks[clip] : 0.f;
        const int s0 = hop * t - G_PADL + 2 * j;
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            int i0 = s0 + 32 * n1, i1 = i0 + 1;   // reflect padding (N >= 257: one reflection suffices)
            i0 = i0 < 0 ? -i0 : (i0 >= N ? 2 * (N - 1) - i0 : i0);
            i1 = i1 < 0 ? -i1 : (i1 >= N ? 2 * (N - 1) - i1 : i1);
            raw[n1] = make_float2(x[i0], x[i1]);
            if constexpr (PE) rawl[n1] = make_float2(i0 > 0 ? x[i0 - 1] : 0.f, i1 > 0 ? x[i1 - 1] : 0.f);
        }
    };
    load_tile(tile0);
#pragma unroll 1
    for (int c = 0; c < ntile; ++c) {
    const long long row0 = (tile0 + c) * G_FPB;
    float2 a[16], z[16];
    {
        const PeakScale scale(pk);   // "if max_val > 0: waveform / max_val" (:209-212)
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            float v0 = scale(raw[n1].x), v1 = scale(raw[n1].y);
            if constexpr (PE) {   // y[n] = x[n] - coef x[n-1], y[0] = x[0] (:235-238), on the normalised signal
                v0 = __fsub_rn(v0, mul_rn(coef, scale(rawl[n1].x)));
                v1 = __fsub_rn(v1, mul_rn(coef, scale(rawl[n1].y)));
            }
            const float2 wt = TAIL == 1 ? *reinterpret_cast<const float2*>(win + 32 * n1 + 2 * j) : wreg[TAIL == 1 ? 0 : n1];
            a[n1] = make_float2(v0 * wt.x, v1 * wt.y);
        }
    }
    if (c + 1 < ntile) load_tile(tile0 + c + 1);   // lands while this tile's FFTs run
    dft16(a);
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) a[k1] = cmul(a[k1], tw_row[k1]);
    float* myx = xs + (wave * 4 + fsub) * G_XFRAME;
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) myx[k1 * G_XROW + j] = a[k1].x;
    wave_lds_fence();
#pragma unroll
    for (int n2 = 0; n2 < 16; ++n2) z[n2].x = myx[j * G_XROW + n2];
    wave_lds_fence();
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) myx[k1 * G_XROW + j] = a[k1].y;
    wave_lds_fence();
#pragma unroll
    for (int n2 = 0; n2 < 16; ++n2) z[n2].y = myx[j * G_XROW + n2];
    dft16(z);   // z[k2] = Z[j + 16 k2]
    float2 rv[8];   // z[8 + r] of lane (16 - j) & 15 of the same frame: row_mirror, then rotate right by one
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        rv[r].x = dpp_mov<0x121>(dpp_mov<0x140>(z[8 + r].x));
        rv[r].y = dpp_mov<0x121>(dpp_mov<0x140>(z[8 + r].y));
    }
    auto put = [&](int bin, float pwr4) {   // pwr4 = |2X|^2
        otile[bin * (G_FPB + 1) + f16] = MAG ? 0.5f * sqrtf(pwr4) : 0.25f * pwr4;
    };
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) {
        const float2 zk = z[k2];
        const float2 zp0 = (k2 == 0) ? z[0] : rv[8 - k2];   // j == 0
        const float2 zp = (j == 0) ? zp0 : rv[7 - k2];
        // 2E = Zk + conj Zp, 2O = -i (Zk - conj Zp); 2X[k] = 2E + W^k 2O, 2X[256 - k] = conj(2E - W^k 2O)
        const float ex = zk.x + zp.x, ey = zk.y - zp.y;
        const float ox = zk.y + zp.y, oy = zp.x - zk.x;
        const float qx = W32C[k2] * ox - W32S[k2] * oy, qy = W32C[k2] * oy + W32S[k2] * ox;
        const float px = tw_j.x * qx - tw_j.y * qy, py = tw_j.x * qy + tw_j.y * qx;
        const float ar = ex + px, ai = ey + py, br = ex - px, bi = ey - py;
        const int k = j + 16 * k2;
        put(k, ar * ar + ai * ai);
        put(G_NFFT / 2 - k, br * br + bi * bi);
    }
    if (j == 0) put(128, 4.0f * (z[8].x * z[8].x + z[8].y * z[8].y));   // X[128] = conj Z[128]
    __syncthreads();
    if constexpr (TAIL == 1) {
        // raw contrast rows (:272-293): thread = (frame f of the tile, band i), a selection network per thread (contrast_rank.h).
        // (Ranking a band's bins with 16 threads per frame -- every thread busy -- issues more instructions in total, 4.0 k against
        // 3.1 k wave-instructions per tile, and was slower: the idle SIMDs here serve the CU's other workgroups.)
        const int f = tid & 15, i = tid >> 4;
        const long long orow = row0 + f;
        if (i < tail.cfg.n_bands && orow < n_rows) {
            const long long oclip = orow / T;
            int low = tail.cfg.edges[i], high = tail.cfg.edges[i + 1];
            if (high <= low) high = low + 1;
            if (high > G_NFREQ) high = G_NFREQ;
            float peaks_mean, valleys, chk = 0.f;
            contrast_select<13>(otile + low * (G_FPB + 1) + f, G_FPB + 1, high - low, 1.0f, peaks_mean, valleys, chk);   // nb <= 64
            out[(oclip * tail.nfeat + tail.row0 + i) * T + (orow - oclip * T)] = (log1pf(peaks_mean) - log1pf(valleys)) + chk;
        }
    } else if constexpr (TAIL == 2) {   // torchaudio.functional.spectral_centroid / (sample_rate / 2), :295-298
        const int f = tid & 15, grp = tid >> 4;
        float num = 0.f, den = 0.f;
        for (int k = grp; k < G_NFREQ; k += 16) {
            const float m = otile[k * (G_FPB + 1) + f];
            num = fmaf(tail.freqs[k], m, num);
            den += m;
        }
        float2* part = reinterpret_cast<float2*>(xs);   // [16 bin groups][16 frames]: the transpose scratch is free
        part[grp * 16 + f] = make_float2(num, den);
        __syncthreads();
        const long long orow = row0 + tid;
        if (tid < 16 && orow < n_rows) {
            float sn = 0.f, sd = 0.f;
            for (int q = 0; q < 16; ++q) {
                sn += part[q * 16 + tid].x;
                sd += part[q * 16 + tid].y;
            }
            const long long oclip = orow / T;
            out[(oclip * tail.nfeat + tail.row0 + tail.cfg.n_bands) * T + (orow - oclip * T)] = (sn / sd) / tail.nyquist;
        }
    } else if constexpr (MEL) {
        // thread = (frame f of the tile, band m mod 16): bins of a band in ascending order.  The CSR filterbank moves into the
        // (now free) transpose scratch first when it fits -- a triangular bank has ~2 taps per bin --, so a tap is an LDS read
        // instead of a dependent global load
        constexpr int W_CAP = 4 * 4 * G_XFRAME - 3 * G_MAX_MELS;
        float* wl = xs;
        int* meta = reinterpret_cast<int*>(xs + W_CAP);   // lo | hi | off
        const bool staged = mel.n_taps <= W_CAP;
        if (staged)
            for (int i = tid; i < mel.n_taps; i += 256) wl[i] = mel.w[i];
        for (int i = tid; i < mel.n_mels; i += 256) {
            meta[i] = mel.lo[i];
            meta[G_MAX_MELS + i] = mel.hi[i];
            meta[2 * G_MAX_MELS + i] = mel.off[i];
        }
        __syncthreads();
        const int f = tid & 15;                         // the output row of this thread: its own (clip, frame)
        const long long orow = row0 + f;
        const bool live = orow < n_rows;
        const long long oclip = live ? orow / T : 0;
        float* o = out + oclip * (long long)mel.n_mels * T + (orow - oclip * T);
        auto project = [&](const float* wbase) {
            for (int m = tid >> 4; m < mel.n_mels; m += 16) {
                const int l = meta[m], hb = meta[G_MAX_MELS + m];
                const float* wm = wbase + meta[2 * G_MAX_MELS + m] - l;
                const float* col = otile + f;
                float acc = 0.f;
#pragma unroll 4
                for (int k = l; k < hb; ++k) acc = fmaf(wm[k], col[k * (G_FPB + 1)], acc);
                if (live) o[(long long)m * T] = acc;
            }
        };
        if (staged) project(wl);
        else project(mel.w);
    } else {
        const int f = tid & 15;                         // idx % 16 == tid % 16: one output row per thread
        const long long orow = row0 + f;
        const bool live = orow < n_rows;
        const long long oclip = live ? orow / T : 0;
        float* o = out + oclip * (long long)G_NFREQ * T + (orow - oclip * T);
        for (int k = tid >> 4; k < G_NFREQ; k += 16)
            if (live) o[(long long)k * T] = otile[k * (G_FPB + 1) + f];
    }
    __syncthreads();   // the tile of powers and the transpose scratch are written again by the next tile
    }
}

// ------------------------------------------------------------------------------------------------ STFT, any power-of-two n_fft
// n_fft in {64 .. 2048} other than 512: one WAVE per frame, the n_fft / 2-point complex FFT of the packed real frame as a
// radix-4 Stockham autosort (ping-pong between two LDS buffers, twiddles from a table computed in double), then the
// real-input split with W_n_fft^k.  4 frames per workgroup.  Slower than the register radix-16 x radix-16 of the 512-point
// kernels (the shipped n_fft), and fully general in n_fft.
template <bool MAG, bool MEL>
__global__ __launch_bounds__(256) void gen_stft_pow2_kernel(const float* __restrict__ wav, long long stride, int N, int hop, int T,
                                                            int nfft, const float* __restrict__ win /* [nfft] */,
                                                            const float2* __restrict__ twn /* [nfft/2 + 1]: W_nfft^k */,
                                                            const float* __restrict__ peaks, int pre_emph, float coef,
                                                            float* __restrict__ out, GenMel mel) {
    extern __shared__ __attribute__((aligned(16))) char smem_p[];
    const int NC = nfft / 2, nfreq = NC + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float2* buf0 = reinterpret_cast<float2*>(smem_p) + size_t(wave) * 2 * NC;
    float2* buf1 = buf0 + NC;
    float* spec = reinterpret_cast<float*>(reinterpret_cast<float2*>(smem_p) + size_t(4) * 2 * NC) + size_t(wave) * (nfreq + 1);
    float2* tws = reinterpret_cast<float2*>(reinterpret_cast<float*>(reinterpret_cast<float2*>(smem_p) + size_t(4) * 2 * NC) +
                                            size_t(4) * (nfreq + 1));   // [NC + 1] W_nfft^k, shared by the 4 waves
    for (int k = tid; k <= NC; k += 256) tws[k] = twn[k];
    __syncthreads();
    const long long clip = blockIdx.y;
    const int t_raw = blockIdx.x * 4 + wave, t = t_raw < T ? t_raw : T - 1;
    const float* x = wav + clip * stride;
    const PeakScale scale(peaks ? peaks[clip] : 0.f);
    auto sample = [&](int i) -> float { return scale(x[i]); };
    auto value = [&](int i) -> float {
        i = i < 0 ? -i : (i >= N ? 2 * (N - 1) - i : i);          // reflect padding (N > n_fft / 2: one reflection suffices)
        float v = sample(i);
        if (pre_emph && i > 0) v = __fsub_rn(v, mul_rn(coef, sample(i - 1)));
        return v;
    };
    const int s0 = hop * t - NC;
    for (int n = lane; n < NC; n += 64)
        buf0[n] = make_float2(value(s0 + 2 * n) * win[2 * n], value(s0 + 2 * n + 1) * win[2 * n + 1]);
    wave_lds_fence();
    // Stockham autosort, decimation in frequency, radix 4 (one radix-2 stage first when log2(NC) is odd): a stage works on
    // sub-transforms of length n = NC / s interleaved at stride s; butterfly b -> (p = b / s, q = b % s), twiddles W_n^p =
    // W_nfft^(2 p s) out of the LDS copy of the table (W_nfft^(k + NC) = -W_nfft^k covers the third twiddle's range)
    float2* src = buf0;
    float2* dst = buf1;
    int ls = 0;   // log2(s)
    if (__builtin_ctz(NC) & 1) {
        const int l = NC >> 1;
        for (int b = lane; b < l; b += 64) {
            const float2 c0 = src[b], c1 = src[b + l];
            const float2 d = make_float2(c0.x - c1.x, c0.y - c1.y);
            dst[2 * b] = make_float2(c0.x + c1.x, c0.y + c1.y);
            dst[2 * b + 1] = cmul(d, tws[2 * b]);
        }
        wave_lds_fence();
        float2* tmp = src; src = dst; dst = tmp;
        ls = 1;
    }
    for (int n1 = NC >> (ls + 2); n1 >= 1; n1 >>= 2, ls += 2) {
        const int sm = (1 << ls) - 1;
        for (int b = lane; b < (NC >> 2); b += 64) {
            const int pp = b >> ls, q = b & sm;
            const float2* xin = src + q + (pp << ls);
            const float2 a = xin[0], bb = xin[n1 << ls], c = xin[(2 * n1) << ls], d = xin[(3 * n1) << ls];
            const int i1 = (2 * pp) << ls;                          // < NC / 2
            int i3 = 3 * i1;                                        // < 3 NC / 2
            const bool neg = i3 >= NC;
            i3 -= neg ? NC : 0;
            const float2 w1 = tws[i1], w2 = tws[2 * i1];
            float2 w3 = tws[i3];
            if (neg) w3 = make_float2(-w3.x, -w3.y);
            const float2 apc = make_float2(a.x + c.x, a.y + c.y), amc = make_float2(a.x - c.x, a.y - c.y);
            const float2 bpd = make_float2(bb.x + d.x, bb.y + d.y), bmd = make_float2(bb.x - d.x, bb.y - d.y);
            float2* yo = dst + q + ((4 * pp) << ls);
            yo[0] = make_float2(apc.x + bpd.x, apc.y + bpd.y);
            yo[1 << ls] = cmul(make_float2(amc.x + bmd.y, amc.y - bmd.x), w1);        // (a - c) - i (b - d)
            yo[2 << ls] = cmul(make_float2(apc.x - bpd.x, apc.y - bpd.y), w2);
            yo[3 << ls] = cmul(make_float2(amc.x - bmd.y, amc.y + bmd.x), w3);        // (a - c) + i (b - d)
        }
        wave_lds_fence();
        float2* tmp = src; src = dst; dst = tmp;
    }
    // real-input split: X[k] = E + W_nfft^k O, E = (Zk + conj Zp) / 2, O = -i (Zk - conj Zp) / 2, Zp = Z[(NC - k) mod NC]
    for (int k = lane; k <= NC; k += 64) {
        const float2 zk = src[k == NC ? 0 : k], zq = src[k == 0 || k == NC ? 0 : NC - k];
        const float ex = 0.5f * (zk.x + zq.x), ey = 0.5f * (zk.y - zq.y);
        const float ox = 0.5f * (zk.y + zq.y), oy = 0.5f * (zq.x - zk.x);
        const float2 w = tws[k];
        const float xr = ex + w.x * ox - w.y * oy, xi = ey + w.x * oy + w.y * ox;
        const float pw = xr * xr + xi * xi;
        spec[k] = MAG ? sqrtf(pw) : pw;
    }
    wave_lds_fence();
    if (t_raw >= T) return;
    if constexpr (MEL) {
        float* o = out + clip * (long long)mel.n_mels * T;
        for (int mb = lane; mb < mel.n_mels; mb += 64) {
            const int l = mel.lo[mb], hb = mel.hi[mb];
            const float* wm = mel.w + mel.off[mb];
            float acc = 0.f;
            for (int k = l; k < hb; ++k) acc = fmaf(wm[k - l], spec[k], acc);
            o[(long long)mb * T + t] = acc;
        }
    } else {
        float* o = out + clip * (long long)nfreq * T;
        for (int k = lane; k <= NC; k += 64) o[(long long)k * T + t] = spec[k];
    }
}

// ------------------------------------------------------------------------------------------------ STFT, ANY n_fft (DFT as a matrix product)
// An n_fft that is not a power of two (torchaudio's own default is 400): X[k] = sum_n x[n] w[n] e^{-2 pi i k n / n_fft} evaluated
// directly -- on the matrix cores in EXACT f32 (v_mfma_f32_32x32x2_f32: an fmaf chain per output, bit for bit what a VALU loop
// would give, at the VALU's peak rate but with every operand reused 32 times out of registers).  The real input halves the sum:
//   Re X[k] =  sum_{n = 0}^{n_fft / 2} s[n] cos(2 pi k n / n_fft),  s[n] = x[n] + x[n_fft - n]  (s[0] = x[0]; s[n_fft / 2] = x[n_fft / 2])
//   Im X[k] = -sum d[n] sin(2 pi k n / n_fft),                      d[n] = x[n] - x[n_fft - n]  (0 at the self-paired samples)
// i.e. two products [32 frames x n] . [n x bins] against constant tables (cos and -sin of (k n mod n_fft), computed in double on
// the host, zero-padded to whole tiles).  A workgroup = 32 frames of one clip; s / d are staged 64 values of n at a time
// ([n][32 frames]: the A fragment is one conflict-free ds_read_b32), the table rows stream from L2 (the B fragment is one
// coalesced 128-byte row per half-wave); a wave owns two 32-bin tiles per pass (64 accumulator registers), 8 tiles per pass.
// The spectrum tile [32][bins + 1] collects the passes for the mel projection / the frame-major store.  4 M FMA per 1 s clip at
// n_fft = 400: the generality fallback, not a throughput path.
using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int G_DFT_M = 32, G_DFT_KC = 64;
struct GenDft {
    // cos(2 pi (k n mod n_fft) / n_fft) and -sin(...) in FRAGMENT order: [bin tile][group of 8 n][lane = (n & 1) * 32 + bin % 32]
    // as float4 over n = 8 group + 2 q + (n & 1), q = 0 .. 3 -- one 16-byte load per lane feeds four k-steps, a wave's load is
    // one contiguous KB; n padded to whole chunks, bins to whole tiles, with zeros
    const float4* cos_t;
    const float4* sin_t;
    int pitch;    // bins rounded up to 32
    int groups;   // groups of 8 n per tile
};
inline size_t dft_lds_bytes(int pitch) { return size_t(2) * G_DFT_KC * G_DFT_M * 4 + size_t(G_DFT_M) * (pitch + 1) * 4; }

template <bool MAG, bool MEL>
__global__ __launch_bounds__(256) void gen_stft_dft_kernel(const float* __restrict__ wav, long long stride, int N, int hop, int T,
                                                           int nfft, const float* __restrict__ win /* [nfft] */, GenDft tab,
                                                           const float* __restrict__ peaks, int pre_emph, float coef,
                                                           float* __restrict__ out, GenMel mel, long long n_rows /* clips x T */) {
    extern __shared__ __attribute__((aligned(16))) char smem_d[];
    const int nfreq = nfft / 2 + 1, pad = nfft / 2, nh = nfft / 2;   // n = 0 .. nh
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pitch = tab.pitch, n_tiles = pitch >> 5, SP = pitch + 1;
    float* sp = reinterpret_cast<float*>(smem_d);       // [KC][32] sums of the current chunk
    float* df = sp + G_DFT_KC * G_DFT_M;                // [KC][32] differences
    float* spec = df + G_DFT_KC * G_DFT_M;              // [32][pitch + 1]
    // the 32 rows of a workgroup are 32 consecutive (clip, frame) pairs of the launch: no padding of T to whole tiles, a tile may
    // straddle two clips.  A thread stages the same row f = tid % 32 throughout.
    const long long row = (long long)blockIdx.x * G_DFT_M + (tid & 31);
    const bool row_live = row < n_rows;
    const long long clip = row_live ? row / T : (n_rows - 1) / T;
    const int t = row_live ? int(row - clip * T) : T - 1;
    const float* x = wav + clip * stride;
    const PeakScale scale(peaks ? peaks[clip] : 0.f);
    auto sample = [&](int i) -> float { return scale(x[i]); };
    auto value = [&](int i) -> float {
        i = i < 0 ? -i : (i >= N ? 2 * (N - 1) - i : i);
        float v = sample(i);
        if (pre_emph && i > 0) v = __fsub_rn(v, mul_rn(coef, sample(i - 1)));
        return v;
    };
    const int s0f = hop * t - pad;
    const int col = lane & 31, half = lane >> 5;
    for (int pass0 = 0; pass0 < n_tiles; pass0 += 8) {
        const int tile_a = pass0 + wave, tile_b = pass0 + 4 + wave;   // wave-uniform
        const bool has_a = tile_a < n_tiles, has_b = tile_b < n_tiles;
        f32x16 ra = {}, ia = {}, rb = {}, ib = {};
        float4 nca = {}, nsa = {}, ncb = {}, nsb = {};
        auto load_b = [&](int g) {
            const size_t ia4 = (size_t(tile_a) * tab.groups + g) * 64 + lane;
            nca = tab.cos_t[ia4];
            nsa = tab.sin_t[ia4];
            if (has_b) {
                const size_t ib4 = (size_t(tile_b) * tab.groups + g) * 64 + lane;
                ncb = tab.cos_t[ib4];
                nsb = tab.sin_t[ib4];
            }
        };
        if (has_a) load_b(0);
        for (int n0 = 0; n0 <= nh; n0 += G_DFT_KC) {
            __syncthreads();   // the previous chunk has been consumed
            for (int i = tid; i < G_DFT_KC * G_DFT_M; i += 256) {
                const int n = n0 + (i >> 5);   // i % 32 == tid % 32: this thread's row
                float a = 0.f, b = 0.f;
                bool paired = false;
                if (n <= nh) {
                    const int n2 = nfft - n;
                    a = value(s0f + n) * win[n];
                    paired = n > 0 && n2 != n;
                    if (paired) b = value(s0f + n2) * win[n2];
                }
                sp[i] = a + b;
                df[i] = paired ? a - b : 0.f;
            }
            __syncthreads();
            if (has_a) {
                const int gn = tab.groups - (n0 >> 3) < G_DFT_KC / 8 ? tab.groups - (n0 >> 3) : G_DFT_KC / 8;   // groups of 8 n left
                for (int gi = 0; gi < gn; ++gi) {
                    const int g = (n0 >> 3) + gi;
                    const float4 ca = nca, sa = nsa, cb = ncb, sb = nsb;
                    if (g + 1 < tab.groups) load_b(g + 1);   // the next group's fragments fly while this one is multiplied
                    const float* as = sp + (8 * gi + half) * G_DFT_M + col;
                    const float* ad = df + (8 * gi + half) * G_DFT_M + col;
                    const float s0 = as[0], s1 = as[2 * G_DFT_M], s2 = as[4 * G_DFT_M], s3 = as[6 * G_DFT_M];
                    const float d0 = ad[0], d1 = ad[2 * G_DFT_M], d2 = ad[4 * G_DFT_M], d3 = ad[6 * G_DFT_M];
                    ra = __builtin_amdgcn_mfma_f32_32x32x2f32(s0, ca.x, ra, 0, 0, 0);
                    ia = __builtin_amdgcn_mfma_f32_32x32x2f32(d0, sa.x, ia, 0, 0, 0);
                    if (has_b) {
                        rb = __builtin_amdgcn_mfma_f32_32x32x2f32(s0, cb.x, rb, 0, 0, 0);
                        ib = __builtin_amdgcn_mfma_f32_32x32x2f32(d0, sb.x, ib, 0, 0, 0);
                    }
                    ra = __builtin_amdgcn_mfma_f32_32x32x2f32(s1, ca.y, ra, 0, 0, 0);
                    ia = __builtin_amdgcn_mfma_f32_32x32x2f32(d1, sa.y, ia, 0, 0, 0);
                    if (has_b) {
                        rb = __builtin_amdgcn_mfma_f32_32x32x2f32(s1, cb.y, rb, 0, 0, 0);
                        ib = __builtin_amdgcn_mfma_f32_32x32x2f32(d1, sb.y, ib, 0, 0, 0);
                    }
                    ra = __builtin_amdgcn_mfma_f32_32x32x2f32(s2, ca.z, ra, 0, 0, 0);
                    ia = __builtin_amdgcn_mfma_f32_32x32x2f32(d2, sa.z, ia, 0, 0, 0);
                    if (has_b) {
                        rb = __builtin_amdgcn_mfma_f32_32x32x2f32(s2, cb.z, rb, 0, 0, 0);
                        ib = __builtin_amdgcn_mfma_f32_32x32x2f32(d2, sb.z, ib, 0, 0, 0);
                    }
                    ra = __builtin_amdgcn_mfma_f32_32x32x2f32(s3, ca.w, ra, 0, 0, 0);
                    ia = __builtin_amdgcn_mfma_f32_32x32x2f32(d3, sa.w, ia, 0, 0, 0);
                    if (has_b) {
                        rb = __builtin_amdgcn_mfma_f32_32x32x2f32(s3, cb.w, rb, 0, 0, 0);
                        ib = __builtin_amdgcn_mfma_f32_32x32x2f32(d3, sb.w, ib, 0, 0, 0);
                    }
                }
            }
        }
        // C layout of 32x32: register i of lane l = row 8 (i / 4) + 4 (l / 32) + i % 4 (frame), column l % 32 (bin of the tile)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int f = 8 * (i >> 2) + 4 * half + (i & 3);
            if (has_a) {
                const float p = ra[i] * ra[i] + ia[i] * ia[i];
                spec[f * SP + tile_a * 32 + col] = MAG ? sqrtf(p) : p;
            }
            if (has_b) {
                const float p = rb[i] * rb[i] + ib[i] * ib[i];
                spec[f * SP + tile_b * 32 + col] = MAG ? sqrtf(p) : p;
            }
        }
    }
    __syncthreads();
    // frame-major stores: thread = (row f = tid % 32, bin / band i / 32); a row's clip and frame are this thread's own
    if constexpr (MEL) {
        float* o = out + clip * (long long)mel.n_mels * T + t;
        const float* srow = spec + (tid & 31) * SP;
        for (int mb = tid >> 5; mb < mel.n_mels; mb += 8) {
            const int l = mel.lo[mb], hb = mel.hi[mb];
            const float* wm = mel.w + mel.off[mb];
            float acc = 0.f;
            for (int k = l; k < hb; ++k) acc = fmaf(wm[k - l], srow[k], acc);
            if (row_live) o[(long long)mb * T] = acc;
        }
    } else {
        float* o = out + clip * (long long)nfreq * T + t;
        const float* srow = spec + (tid & 31) * SP;
        for (int k = tid >> 5; k < nfreq; k += 8)
            if (row_live) o[(long long)k * T] = srow[k];
    }
}

// AmplitudeToDB('power'): 10 log10(max(x, amin = 1e-10)) (ref = 1: no offset)
__device__ __forceinline__ float g_db(float p) { return 10.0f * log10f(fmaxf(p, 1e-10f)); }
// PCEN value of apply_pcen (:305-340): smooth = 10-frame moving average (avg_pool2d kernel 10, padding 5, zeros counted,
// output trimmed to T: frames t - 5 .. t + 4), (mel / (1e-6 + smooth)^0.98 + 2)^0.5 - 2^0.5
__device__ __forceinline__ float g_pcen(const float* __restrict__ row, int t, int T) {
    float sm = 0.f;
#pragma unroll
    for (int q = -5; q < 5; ++q) {
        const int u = t + q;
        sm += (u >= 0 && u < T) ? row[u] : 0.f;
    }
    return sqrtf(row[t] / powf(1e-6f + sm / 10.0f, 0.98f) + 2.0f) - 1.41421356237309515f;
}

// per clip: stat[clip] = {max dB, PCEN min, PCEN max, poison}.  poison = 0, or NaN for a clip with a non-finite mel power -- a NaN /
// Inf sample under a frame, an f32 overflow of the power: the reference's dense `spec @ fb` turns an Inf bin into NaN bands
// (0 * inf), AmplitudeToDB's per-clip amax is then NaN and its floor makes EVERY cell of the clip NaN (:405-410), PCEN's min /
// max likewise (:402-404), the MFCC z-score and the deltas follow (:428); gen_rows_kernel adds the poison to what it writes
__global__ __launch_bounds__(256) void gen_dbstat_kernel(const float* __restrict__ melpow, int T, int n_mels, int pcen,
                                                         float* __restrict__ stat) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const long long clip = blockIdx.x;
    const float* mp = melpow + clip * (long long)n_mels * T;
    const long long n = (long long)n_mels * T;
    float mx = -INFINITY, pmin = INFINITY, pmax = -INFINITY;
    int bad = 0;
    for (long long i = tid; i < n; i += 256) {
        bad |= !(mp[i] < INFINITY);
        mx = fmaxf(mx, g_db(mp[i]));
        if (pcen) {
            const int m = int(i / T), t = int(i - (long long)m * T);
            const float v = g_pcen(mp + (long long)m * T, t, T);
            pmin = fminf(pmin, v);
            pmax = fmaxf(pmax, v);
        }
    }
    mx = g_block_max(mx, red, tid);
    pmin = -g_block_max(-pmin, red, tid);
    pmax = g_block_max(pmax, red, tid);
    bad = __syncthreads_or(bad);
    if (tid == 0) {
        stat[clip * 4] = mx; stat[clip * 4 + 1] = pmin; stat[clip * 4 + 2] = pmax;
        stat[clip * 4 + 3] = bad ? __builtin_nanf("") : 0.f;
    }
}

// mel rows + raw MFCC rows of 64 frames of one clip
__global__ __launch_bounds__(256) void gen_rows_kernel(const float* __restrict__ melpow, int T, int n_mels, int n_mfcc,
                                                       int pcen, const float* __restrict__ stat,
                                                       const float* __restrict__ dct_t /* [n_mfcc][n_mels] */,
                                                       float* __restrict__ feat, int nfeat) {
    extern __shared__ float dbt[];   // [n_mels][G_TT] floored dB of this tile
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long clip = blockIdx.y;
    const int t = blockIdx.x * G_TT + lane;
    const float* mp = melpow + clip * (long long)n_mels * T;
    float* o = feat + clip * (long long)nfeat * T;
    const float floor_db = stat[clip * 4] - 80.0f;   // top_db = 80 below the clip's maximum
    const float pmin = stat[clip * 4 + 1], prange = stat[clip * 4 + 2] - pmin + 1e-8f;
    const float poison = stat[clip * 4 + 3];   // 0, or NaN: a clip with a non-finite mel power is NaN in every cell
    for (int m = wave; m < n_mels; m += 4) {
        float d = 0.f;
        if (t < T) {
            d = fmaxf(g_db(mp[(long long)m * T + t]), floor_db) + poison;
            float v;
            if (pcen) v = (g_pcen(mp + (long long)m * T, t, T) - pmin) / prange;          // :402-404
            else v = fminf(fmaxf((d + 80.0f) / 80.0f, 0.f), 1.f);                         // :409-410
            o[(long long)m * T + t] = v + poison;
        }
        dbt[m * G_TT + lane] = d;
    }
    __syncthreads();
    for (int c = wave; c < n_mfcc; c += 4) {   // T.MFCC: DCT-II (ortho) of the floored dB, coefficients wave-uniform
        const float* dr = dct_t + c * n_mels;
        float acc = 0.f;
        for (int m = 0; m < n_mels; ++m) acc = fmaf(dr[m], dbt[m * G_TT + lane], acc);
        if (t < T) o[(long long)(n_mels + c) * T + t] = acc;
    }
}

// in place over rows [row0, row0 + nr) of every clip: (x - mean) / (std + 1e-8), std unbiased
__global__ __launch_bounds__(256) void gen_zscore_kernel(float* __restrict__ feat, int nfeat, int T, int row0, int nr) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    float* p = feat + (blockIdx.x * (long long)nfeat + row0) * T;
    const long long n = (long long)nr * T;
    float s = 0.f;
    for (long long i = tid; i < n; i += 256) s += p[i];
    const float mean = g_block_sum(s, red, tid) / float(n);
    float q = 0.f;
    for (long long i = tid; i < n; i += 256) {
        const float d = p[i] - mean;
        q += d * d;
    }
    const float sd = sqrtf(g_block_sum(q, red, tid) / float(n - 1));
    const float den = sd + 1e-8f;
    for (long long i = tid; i < n; i += 256) p[i] = (p[i] - mean) / den;
}

// compute_deltas (:342-356): replicate-pad by one, (x[t + 1] - x[t - 1]) / 2; twice for delta-delta
__global__ __launch_bounds__(256) void gen_delta_kernel(float* __restrict__ feat, int nfeat, int T, int n_mels, int n_mfcc,
                                                        int delta_delta) {
    const long long clip = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_mfcc * T) return;
    const int c = idx / T, t = idx - c * T;
    float* base = feat + clip * (long long)nfeat * T;
    const float* z = base + (long long)(n_mels + c) * T;
    auto d1 = [&](int u) -> float {
        const int a = u + 1 < T ? u + 1 : T - 1, b = u - 1 > 0 ? u - 1 : 0;
        return (z[a] - z[b]) * 0.5f;
    };
    base[(long long)(n_mels + n_mfcc + c) * T + t] = d1(t);
    if (delta_delta) {
        const int a = t + 1 < T ? t + 1 : T - 1, b = t - 1 > 0 ? t - 1 : 0;
        base[(long long)(n_mels + 2 * n_mfcc + c) * T + t] = (d1(a) - d1(b)) * 0.5f;
    }
}

// One raw spectral-contrast row (blockIdx.z < n_bands) or the centroid row (blockIdx.z == n_bands) of 64 frames of one clip; the
// joint z-score follows in gen_zscore_kernel.  Thread = frame.  A band of up to 128 bins (every band of n_fft <= 1024) streams
// once past the frame's selection network (contrast_rank.h: select_sums, straight from L2, consecutive frames at consecutive
// addresses, no LDS).  The wider bands of n_fft = 2048 go to LDS (dynamic: 32 KB, requested only when such a band exists) and
// each frame ranks its bins (rank = number of smaller values, ties by index -- the position torch.sort would give) and sums the
// values of rank >= top_idx and < bot_idx: the reference's sorted-slice means without sorting; an empty top slice divides 0 by 0
// as the mean of an empty tensor does (:272-293).
__global__ __launch_bounds__(64) void gen_contrast_kernel(const float* __restrict__ P, const float* __restrict__ M, int T, int nfreq,
                                                          ContrastCfg cfg, const float* __restrict__ freqs, float nyquist,
                                                          float* __restrict__ feat, int nfeat, int row0) {
    extern __shared__ __attribute__((aligned(16))) float4 band4[];   // [G_CT_BINS / 4 * G_TT] quad-planar, contrast_rank.h
    const int lane = threadIdx.x, i = blockIdx.z;
    const long long clip = blockIdx.y;
    const int t = blockIdx.x * G_TT + lane;
    const float* Pc = P + clip * (long long)nfreq * T;
    float* o = feat + (clip * (long long)nfeat + row0) * T;
    if (i == cfg.n_bands) {   // torchaudio.functional.spectral_centroid / (sample_rate / 2), :295-298
        if (t < T) {
            const float* Mc = M + clip * (long long)nfreq * T;
            float num = 0.f, den = 0.f;
            for (int k = 0; k < nfreq; ++k) {
                const float m = Mc[(long long)k * T + t];
                num += freqs[k] * m;
                den += m;
            }
            o[(long long)cfg.n_bands * T + t] = (num / den) / nyquist;
        }
        return;
    }
    int low = cfg.edges[i], high = cfg.edges[i + 1];   // :272-278
    if (high <= low) high = low + 1;
    if (high > nfreq) high = nfreq;
    const int nb = high - low;
    if (nb <= G_CT_BINS) {   // workgroup-uniform
        if (t < T) {
            float peaks, valleys, chk = 0.f;
            contrast_select(Pc + (long long)low * T + t, T, nb, 1.0f, peaks, valleys, chk);
            o[(long long)i * T + t] = (log1pf(peaks) - log1pf(valleys)) + chk;
        }
        return;
    }
    const int nq = (nb + 3) >> 2;
    int top_idx = (int)((double)nb * 0.8), bot_idx = (int)((double)nb * 0.2);   // python int(n_bins * 0.8)
    if (top_idx < 1) top_idx = 1;
    if (bot_idx < 1) bot_idx = 1;
    // the band's rows of FT frames at a time in the 32 KB tile: 32 / 16 / 8 frames for bands of up to 256 / 512 / 1024 bins
    const int FT = nb <= 2 * G_CT_BINS ? G_TT / 2 : nb <= 4 * G_CT_BINS ? G_TT / 4 : G_TT / 8;
    for (int sub = 0; sub < G_TT / FT; ++sub) {
        const int ts = blockIdx.x * G_TT + sub * FT + lane;
        if (blockIdx.x * G_TT + sub * FT >= T) break;   // wave-uniform: nothing left of this tile
        __syncthreads();
        if (lane < FT) {
            const float nan = __builtin_nanf("");
            for (int q = 0; q < nq; ++q) {
                const float* src = Pc + (long long)(low + 4 * q) * T + ts;
                float4 u;
                u.x = ts < T ? src[0] : 0.f;
                u.y = 4 * q + 1 < nb ? (ts < T ? src[T] : 0.f) : nan;
                u.z = 4 * q + 2 < nb ? (ts < T ? src[2LL * T] : 0.f) : nan;
                u.w = 4 * q + 3 < nb ? (ts < T ? src[3LL * T] : 0.f) : nan;
                band4[q * FT + lane] = u;
            }
        }
        __syncthreads();
        if (lane < FT) {
            float top, bot;
            contrast_band_sums(band4, FT, lane, nb, top_idx, bot_idx, top, bot);
            const float peaks = top / float(nb - top_idx);   // 0 / 0 = NaN when the top slice is empty
            const float valleys = bot / float(bot_idx);
            if (ts < T) o[(long long)i * T + ts] = log1pf(peaks) - log1pf(valleys);
        }
    }
}

// ------------------------------------------------------------------------------------------------ stand-alone helpers
// AudioPreprocessor.apply_pre_emphasis / compute_deltas / apply_pcen called on their own (:214-240, :342-356, :305-340)
__global__ __launch_bounds__(256) void pre_emphasis_kernel(const float* __restrict__ in, long long in_stride, float* __restrict__ out,
                                                           long long out_stride, int n, float coef) {
    const float* x = in + blockIdx.y * in_stride;
    float* y = out + blockIdx.y * out_stride;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = i > 0 ? __fsub_rn(x[i], mul_rn(coef, x[i - 1])) : x[i];   // first sample kept
}
__global__ __launch_bounds__(256) void deltas_kernel(const float* __restrict__ in, float* __restrict__ out, long long total, int T) {
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= total) return;
    const int t = int(i % T);
    const float* row = in + (i - t);
    const int a = t + 1 < T ? t + 1 : T - 1, b = t - 1 > 0 ? t - 1 : 0;   // replicate padding
    out[i] = (row[a] - row[b]) / 2.0f;
}
__global__ __launch_bounds__(256) void pcen_kernel(const float* __restrict__ in, float* __restrict__ out, long long total, int T,
                                                   float alpha, float delta, float r, float eps, float delta_pow_r) {
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= total) return;
    const int t = int(i % T);
    const float* row = in + (i - t);
    float sm = 0.f;
#pragma unroll
    for (int q = -5; q < 5; ++q) {   // avg_pool2d(kernel (1, 10), padding (0, 5), zeros counted), trimmed to T
        const int u = t + q;
        sm += (u >= 0 && u < T) ? row[u] : 0.f;
    }
    const float base = row[t] / powf(eps + sm / 10.0f, alpha) + delta;
    out[i] = (r == 0.5f ? sqrtf(base) : powf(base, r)) - delta_pow_r;
}

size_t align256g(size_t v) { return (v + 255) & ~size_t(255); }
size_t pow2_lds_bytes(int nfft) { return size_t(4) * 2 * (nfft / 2) * 8 + size_t(4) * (nfft / 2 + 2) * 4 + size_t(nfft / 2 + 2) * 8; }

}  // namespace

struct GenFeat {
    int N, hop, T, n_mels, n_mfcc, sample_rate;   // N / T: the constructor's segment and its frame count -- the tables hold for ANY
                                                  // waveform length, which is a launch parameter (gen_frames)
    int nfft, nfreq;       // n_fft (16 .. 2048) and n_fft / 2 + 1
    char* d_blob;
    const float* win;      // [n_fft] caller's window centred in the frame
    const float* win_full; // [n_fft] periodic Hann(n_fft)
    const float2* tw256;   // [16][16]            (the 512-point kernels)
    const float2* tw512;   // [128]
    const float2* twn;     // [n_fft / 2 + 1] W_n_fft^k (gen_stft_pow2_kernel)
    int n_taps;            // length of mel_w
    GenDft dft;            // cos / -sin matrices of gen_stft_dft_kernel (n_fft not a power of two; else null)
    const int *mel_lo, *mel_hi, *mel_off;
    const float* mel_w;    // CSR taps
    const float* dct_t;    // [n_mfcc][n_mels]
    const float* freqs;    // [257] torch.linspace(0, sample_rate // 2, 257)
};

namespace {
// one STFT launch of the chain: the register radix-16 x radix-16 kernel at n_fft = 512, the Stockham kernel for the other
// powers of two, the direct DFT for everything else
template <bool MAG, bool MEL>
void gen_launch_stft(const GenFeat* g, int N, int T, const float* w, long long wav_stride, int nc, const float* win, const float* peaks,
                     int pre_emph, float coef, float* out, const GenMel& mel, hipStream_t stream) {
    if (g->nfft == G_NFFT) {
        const long long n_rows = (long long)nc * T;
        const dim3 gs((unsigned)((n_rows + G_FPB * G_CHUNK - 1) / (G_FPB * G_CHUNK)));
        if (MEL && pre_emph)
            hipLaunchKernelGGL((gen_stft_kernel<MAG, MEL, 0, MEL>), gs, dim3(256), 0, stream, w, wav_stride, N, g->hop, T, win, g->tw256,
                               g->tw512, peaks, coef, out, mel, n_rows, GenTail{});
        else
            hipLaunchKernelGGL((gen_stft_kernel<MAG, MEL>), gs, dim3(256), 0, stream, w, wav_stride, N, g->hop, T, win, g->tw256,
                               g->tw512, peaks, coef, out, mel, n_rows, GenTail{});
    } else if (g->nfft >= 64 && (g->nfft & (g->nfft - 1)) == 0) {
        hipLaunchKernelGGL((gen_stft_pow2_kernel<MAG, MEL>), dim3((T + 3) / 4, nc), dim3(256), pow2_lds_bytes(g->nfft), stream, w,
                           wav_stride, N, g->hop, T, g->nfft, win, g->twn, peaks, pre_emph, coef, out, mel);
    } else {
        const long long n_rows = (long long)nc * T;
        hipLaunchKernelGGL((gen_stft_dft_kernel<MAG, MEL>), dim3((unsigned)((n_rows + G_DFT_M - 1) / G_DFT_M)), dim3(256),
                           dft_lds_bytes(g->dft.pitch), stream, w, wav_stride, N, g->hop, T, g->nfft, win, g->dft, peaks, pre_emph,
                           coef, out, mel, n_rows);
    }
}
}  // namespace

int gen_feat_create(GenFeat** out, const cough_feat_config* cfg, const float* window, const float* mel_fb, const float* dct) {
    const int nfft = cfg->n_fft, nfreq = nfft / 2 + 1;
    COUGH_REQUIRE(nfft >= 16 && nfft <= 2048, COUGH_EUNSUPPORTED,
                  "n_fft = %d: the HIP path implements 16 .. 2048 (512 on the register FFT kernels, the other powers of two "
                  "from 64 on a radix-4 Stockham kernel, anything else by direct DFT)", nfft);
    COUGH_REQUIRE(cfg->win_length >= 1 && cfg->win_length <= nfft, COUGH_EUNSUPPORTED,
                  "win_length = %d: need 1 <= win_length <= n_fft", cfg->win_length);
    COUGH_REQUIRE(cfg->hop_length >= 1, COUGH_EINVAL, "hop_length = %d", cfg->hop_length);
    COUGH_REQUIRE(cfg->segment_samples > nfft / 2, COUGH_EUNSUPPORTED,
                  "segment of %d samples: reflect padding (torch.stft center=True) needs more than n_fft / 2 = %d",
                  cfg->segment_samples, nfft / 2);
    COUGH_REQUIRE(cfg->n_mels >= 1 && cfg->n_mels <= G_MAX_MELS, COUGH_EUNSUPPORTED,
                  "n_mels = %d: the HIP path takes 1..%d", cfg->n_mels, G_MAX_MELS);
    COUGH_REQUIRE(!cfg->use_mfcc || (cfg->n_mfcc >= 1 && cfg->n_mfcc <= cfg->n_mels), COUGH_EINVAL,
                  "n_mfcc = %d: cannot select more MFCC coefficients than mel bins (%d)", cfg->n_mfcc, cfg->n_mels);
    COUGH_REQUIRE(cfg->sample_rate >= 2, COUGH_EINVAL, "sample_rate = %d", cfg->sample_rate);
    const int n_mels = cfg->n_mels, n_mfcc = cfg->use_mfcc ? cfg->n_mfcc : 0;   // T.MFCC exists only with use_mfcc (:116-127)
    const double PI = 3.14159265358979323846;
    std::vector<float> win(nfft, 0.f), hann(nfft), freqs(nfreq), dct_t(size_t(n_mfcc) * n_mels), taps;
    std::vector<float2> tw256(256), tw512(128), twn(nfreq);
    std::vector<int> lo(n_mels), hi(n_mels), off(n_mels);
    const int left = (nfft - cfg->win_length) / 2;   // torch.stft centres a short window in the frame
    for (int n = 0; n < cfg->win_length; ++n) win[left + n] = window[n];
    for (int n = 0; n < nfft; ++n) hann[n] = float(0.5 - 0.5 * std::cos(2.0 * PI * double(n) / double(nfft)));
    for (int k = 0; k < nfft; ++k) {
        const double a = -2.0 * PI * double(k) / double(nfft);
        if (k < nfreq) twn[k] = make_float2(float(std::cos(a)), float(std::sin(a)));
    }
    // gen_stft_dft_kernel's matrices (n_fft not a power of two): [n rounded up to the chunk][bins rounded up to 32], zero padded
    const bool use_dft = !(nfft >= 64 && (nfft & (nfft - 1)) == 0);
    const int dft_pitch = use_dft ? (nfreq + 31) / 32 * 32 : 0;
    const int dft_rows = use_dft ? (nfft / 2 + 1 + 7) / 8 * 8 : 0;   // n padded to whole groups of 8 (zero rows)
    std::vector<float> dft_cos(size_t(dft_rows) * dft_pitch, 0.f), dft_sin(size_t(dft_rows) * dft_pitch, 0.f);
    if (use_dft) {
        std::vector<double> cd(nfft), sd(nfft);
        for (int j = 0; j < nfft; ++j) {
            cd[j] = std::cos(2.0 * PI * double(j) / double(nfft));
            sd[j] = -std::sin(2.0 * PI * double(j) / double(nfft));
        }
        const int groups = dft_rows / 8;
        for (int n = 0; n <= nfft / 2; ++n)
            for (int k = 0; k < nfreq; ++k) {
                const int j = int((long long)k * n % nfft);
                const size_t at = ((size_t(k >> 5) * groups + (n >> 3)) * 64 + (n & 1) * 32 + (k & 31)) * 4 + ((n & 7) >> 1);
                dft_cos[at] = float(cd[j]);
                dft_sin[at] = float(sd[j]);
            }
    }
    for (int jj = 0; jj < 16; ++jj)
        for (int k1 = 0; k1 < 16; ++k1) {
            const double a = -2.0 * PI * double(jj * k1) / 256.0;
            tw256[jj * 16 + k1] = make_float2(float(std::cos(a)), float(std::sin(a)));
        }
    for (int k = 0; k < 128; ++k) {
        const double a = -2.0 * PI * double(k) / 512.0;
        tw512[k] = make_float2(float(std::cos(a)), float(std::sin(a)));
    }
    for (int m = 0; m < n_mels; ++m) {   // CSR: the band's first .. last non-zero bin (zeros inside the range kept)
        int first = -1, last = -1;
        for (int k = 0; k < nfreq; ++k)
            if (mel_fb[k * n_mels + m] != 0.f) { if (first < 0) first = k; last = k; }
        if (first < 0) { first = 0; last = -1; }   // empty band: no taps, mel power 0
        lo[m] = first;
        hi[m] = last + 1;
        off[m] = int(taps.size());
        for (int k = first; k <= last; ++k) taps.push_back(mel_fb[k * n_mels + m]);
    }
    if (taps.empty()) taps.push_back(0.f);
    for (int c = 0; c < n_mfcc; ++c)
        for (int m = 0; m < n_mels; ++m) dct_t[size_t(c) * n_mels + m] = dct[m * n_mfcc + c];
    if (dct_t.empty()) dct_t.push_back(0.f);
    {   // torch.linspace(0, sample_rate // 2, 257) in float32: start + step * i below the midpoint, end - step * (n - 1 - i) above
        const float end = float(cfg->sample_rate / 2);
        const volatile float step = end / float(nfreq - 1);
        for (int i = 0; i < nfreq; ++i) {
            volatile float prod = i < nfreq / 2 ? step * float(i) : step * float(nfreq - i - 1);
            freqs[i] = i < nfreq / 2 ? prod : end - prod;
        }
    }
    // one device blob
    size_t o_win = 0, o_hann = o_win + align256g(nfft * 4), o_tw256 = o_hann + align256g(nfft * 4),
           o_tw512 = o_tw256 + align256g(256 * 8), o_twn = o_tw512 + align256g(128 * 8), o_dc = o_twn + align256g(nfreq * 8), o_ds = o_dc + align256g(dft_cos.size() * 4),
           o_lo = o_ds + align256g(dft_sin.size() * 4),
           o_hi = o_lo + align256g(n_mels * 4),
           o_off = o_hi + align256g(n_mels * 4), o_taps = o_off + align256g(n_mels * 4),
           o_dct = o_taps + align256g(taps.size() * 4), o_freqs = o_dct + align256g(dct_t.size() * 4),
           total = o_freqs + align256g(nfreq * 4);
    std::vector<char> host(total, 0);
    std::memcpy(host.data() + o_win, win.data(), nfft * 4);
    std::memcpy(host.data() + o_hann, hann.data(), nfft * 4);
    std::memcpy(host.data() + o_tw256, tw256.data(), 256 * 8);
    std::memcpy(host.data() + o_tw512, tw512.data(), 128 * 8);
    std::memcpy(host.data() + o_twn, twn.data(), nfreq * 8);
    if (use_dft) {
        std::memcpy(host.data() + o_dc, dft_cos.data(), dft_cos.size() * 4);
        std::memcpy(host.data() + o_ds, dft_sin.data(), dft_sin.size() * 4);
    }
    std::memcpy(host.data() + o_lo, lo.data(), n_mels * 4);
    std::memcpy(host.data() + o_hi, hi.data(), n_mels * 4);
    std::memcpy(host.data() + o_off, off.data(), n_mels * 4);
    std::memcpy(host.data() + o_taps, taps.data(), taps.size() * 4);
    std::memcpy(host.data() + o_dct, dct_t.data(), dct_t.size() * 4);
    std::memcpy(host.data() + o_freqs, freqs.data(), nfreq * 4);
    GenFeat* g = new GenFeat();
    g->N = cfg->segment_samples;
    g->hop = cfg->hop_length;
    // frames of torch.stft(center=True): 1 + (N + 2 (n_fft / 2) - n_fft) / hop = N / hop + 1 (get_expected_time_frames(),
    // :532-534) for an even n_fft, one sample less of signal for an odd one
    g->T = (cfg->segment_samples - (nfft & 1)) / cfg->hop_length + 1;
    g->n_mels = n_mels;
    g->n_mfcc = n_mfcc;
    g->sample_rate = cfg->sample_rate;
    g->nfft = nfft;
    g->nfreq = nfreq;
    g->d_blob = nullptr;
    hipError_t e = hipMalloc(&g->d_blob, total);
    if (e == hipSuccess) e = hipMemcpy(g->d_blob, host.data(), total, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        set_error("cough_featurizer_create (generic geometry): %s", hipGetErrorString(e));
        if (g->d_blob) (void)hipFree(g->d_blob);
        delete g;
        return COUGH_EHIP;
    }
    const char* b = g->d_blob;
    g->win = reinterpret_cast<const float*>(b + o_win);
    g->win_full = reinterpret_cast<const float*>(b + o_hann);
    g->tw256 = reinterpret_cast<const float2*>(b + o_tw256);
    g->tw512 = reinterpret_cast<const float2*>(b + o_tw512);
    g->twn = reinterpret_cast<const float2*>(b + o_twn);
    g->dft.cos_t = use_dft ? reinterpret_cast<const float4*>(b + o_dc) : nullptr;
    g->dft.sin_t = use_dft ? reinterpret_cast<const float4*>(b + o_ds) : nullptr;
    g->dft.pitch = dft_pitch;
    g->dft.groups = dft_rows / 8;
    g->mel_lo = reinterpret_cast<const int*>(b + o_lo);
    g->mel_hi = reinterpret_cast<const int*>(b + o_hi);
    g->mel_off = reinterpret_cast<const int*>(b + o_off);
    g->mel_w = reinterpret_cast<const float*>(b + o_taps);
    g->n_taps = int(taps.size());
    g->dct_t = reinterpret_cast<const float*>(b + o_dct);
    g->freqs = reinterpret_cast<const float*>(b + o_freqs);
    if (nfft != G_NFFT) {   // more than 64 KB of dynamic LDS: the Stockham kernel at n_fft = 2048, the DFT kernel above ~ 750
        const void* fns[] = {reinterpret_cast<const void*>(gen_stft_pow2_kernel<false, false>),
                             reinterpret_cast<const void*>(gen_stft_pow2_kernel<false, true>),
                             reinterpret_cast<const void*>(gen_stft_pow2_kernel<true, false>),
                             reinterpret_cast<const void*>(gen_stft_dft_kernel<false, false>),
                             reinterpret_cast<const void*>(gen_stft_dft_kernel<false, true>),
                             reinterpret_cast<const void*>(gen_stft_dft_kernel<true, false>)};
        const size_t need = pow2_lds_bytes(2048) > dft_lds_bytes(1024) ? pow2_lds_bytes(2048) : dft_lds_bytes(1024);
        for (const void* fn : fns)
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, int(need)) != hipSuccess) {
                set_error("cough_featurizer_create (generic geometry): hipFuncSetAttribute failed");
                gen_feat_destroy(g);
                return COUGH_EHIP;
            }
    }
    *out = g;
    return COUGH_OK;
}

void gen_feat_destroy(GenFeat* g) {
    if (!g) return;
    if (g->d_blob) (void)hipFree(g->d_blob);
    delete g;
}

int gen_segment_samples(const GenFeat* g) { return g->N; }
// frames of torch.stft(center=True) for a waveform of n_samples (0: the constructor's segment): n / hop + 1 for an even n_fft
// (get_expected_time_frames(), :532-534), one sample less of signal for an odd one
int gen_frames(const GenFeat* g, int n_samples) {
    return n_samples <= 0 ? g->T : (n_samples - (g->nfft & 1)) / g->hop + 1;
}

namespace {
// n_fft = 512 with bands of <= 64 bins (every geometric band layout of 257 bins): the contrast and centroid rows come straight out of
// the STFT workgroups' tiles (gen_stft_kernel TAIL) -- no spectrogram in the workspace
bool contrast_in_stft(const GenFeat* g, int n_bands, const int* edges) {
    bool ok = g->nfft == G_NFFT && n_bands >= 1 && n_bands <= 16;
    for (int i = 0; ok && i < n_bands; ++i) ok = edges[i + 1] - edges[i] <= 64;   // (1 .. 16 geometric bands of 257 bins: <= 64 bins)
    return ok;
}
struct GenCarve {
    int sub;                 // clips per sub-batch
    size_t o_peaks, o_stat, o_P, o_M, o_mel, total;
};
GenCarve gen_carve(const GenFeat* g, bool contrast, int T, int n_clips) {
    GenCarve c;
    const size_t spec = size_t(g->nfreq) * T * 4, mel = size_t(g->n_mels) * T * 4;
    const size_t per = spec * (contrast ? 2 : 0) + mel;   // the spectrograms exist only for the contrast rows
    size_t sub = G_SUB_BYTES / per;
    if (sub < 1) sub = 1;
    if (sub > 32768) sub = 32768;   // grid.y
    if (sub > size_t(n_clips)) sub = size_t(n_clips);
    c.sub = int(sub);
    c.o_peaks = 0;   // one peak per clip of the CALL (the one-launch kernel leaves them for the contrast rows of a run-time geometry)
    c.o_stat = c.o_peaks + align256g(size_t(n_clips) * 4);
    c.o_P = c.o_stat + align256g(sub * 16);
    c.o_M = c.o_P + (contrast ? align256g(sub * spec) : 0);
    c.o_mel = c.o_M + (contrast ? align256g(sub * spec) : 0);
    c.total = c.o_mel + align256g(sub * mel);
    return c;
}
}  // namespace

size_t gen_workspace_bytes(const GenFeat* g, const cough_feat_config& cfg, int n_samples, int n_clips) {
    const bool spectrograms = cfg.use_spectral_contrast && !contrast_in_stft(g, cfg.n_contrast_bands, cfg.contrast_edges);
    return n_clips > 0 ? gen_carve(g, spectrograms, gen_frames(g, n_samples), n_clips).total : 0;
}

int gen_spectrogram(const GenFeat* g, const float* d_wav, long long wav_stride, int n_samples, float* d_spec, int n_clips, int flags,
                    hipStream_t stream) {
    const int N = n_samples > 0 ? n_samples : g->N, T = gen_frames(g, N);
    COUGH_REQUIRE(N > g->nfft / 2, COUGH_EINVAL, "cough_spectrogram: %d samples: reflect padding needs more than n_fft / 2 = %d", N,
                  g->nfft / 2);
    COUGH_REQUIRE(wav_stride >= N, COUGH_EINVAL, "cough_spectrogram: row stride %lld < %d samples", wav_stride, N);
    const bool full = flags & COUGH_SPEC_FULL_WINDOW, mag = flags & COUGH_SPEC_MAGNITUDE;
    const float* win = full ? g->win_full : g->win;
    const GenMel none{0, nullptr, nullptr, nullptr, nullptr, 0};
    if (mag) gen_launch_stft<true, false>(g, N, T, d_wav, wav_stride, n_clips, win, nullptr, 0, 0.f, d_spec, none, stream);
    else gen_launch_stft<false, false>(g, N, T, d_wav, wav_stride, n_clips, win, nullptr, 0, 0.f, d_spec, none, stream);
    COUGH_HIP_CHECK(hipGetLastError());
    return COUGH_OK;
}

int gen_featurize(const GenFeat* g, const cough_feat_config& cfg, const ContrastCfg& contrast, const float* d_wav,
                  long long wav_stride, int n_samples, float* d_feat, int nfeat, int nbase, int n_clips, int normalize,
                  void* d_workspace, size_t workspace_bytes, hipStream_t stream, bool contrast_rows_only) {
    const int N = n_samples > 0 ? n_samples : g->N, T = gen_frames(g, N);
    COUGH_REQUIRE(N > g->nfft / 2, COUGH_EINVAL, "cough_featurize: %d samples: the reflect padding of torch.stft(center=True) needs "
                  "more than n_fft / 2 = %d", N, g->nfft / 2);
    COUGH_REQUIRE(wav_stride >= N, COUGH_EINVAL, "cough_featurize: row stride %lld < %d samples", wav_stride, N);
    const bool want_contrast = contrast.n_bands > 0;
    const bool narrow = want_contrast && contrast_in_stft(g, contrast.n_bands, contrast.edges);
    const GenCarve c = gen_carve(g, want_contrast && !narrow, T, n_clips);
    COUGH_REQUIRE(d_workspace && workspace_bytes >= c.total, COUGH_EWORKSPACE,
                  "this featuriser geometry runs on the generic kernel chain and needs a workspace of "
                  "cough_featurizer_workspace_bytes() = %zu bytes (cough_featurize_ws)", c.total);
    COUGH_REQUIRE((reinterpret_cast<size_t>(d_workspace) & 255) == 0, COUGH_EINVAL, "workspace must be 256-byte aligned");
    char* ws = static_cast<char*>(d_workspace);
    float* peaks = reinterpret_cast<float*>(ws + c.o_peaks);
    float* stat = reinterpret_cast<float*>(ws + c.o_stat);
    float* P = reinterpret_cast<float*>(ws + c.o_P);
    float* M = reinterpret_cast<float*>(ws + c.o_M);
    float* mel = reinterpret_cast<float*>(ws + c.o_mel);
    const int n_mels = g->n_mels, n_mfcc = g->n_mfcc;
    for (int c0 = 0; c0 < n_clips; c0 += c.sub) {
        const int nc = n_clips - c0 < c.sub ? n_clips - c0 : c.sub;
        const float* w = d_wav + (long long)c0 * wav_stride;
        float* feat = d_feat + (long long)c0 * nfeat * T;
        // contrast_rows_only: the featurise kernel has left every clip's peak at the head of the workspace (a NaN sample is not in
        // it -- v_max drops NaNs -- but poisons its frames, and the joint z-score of the rows spreads that over the clip)
        const float* pk = normalize ? (contrast_rows_only ? peaks + c0 : peaks) : nullptr;
        if (normalize && !contrast_rows_only)
            hipLaunchKernelGGL(gen_peak_kernel, dim3(nc), dim3(256), 0, stream, w, wav_stride, N, peaks);
        const dim3 gt((T + G_TT - 1) / G_TT, nc);
        // STFT + mel projection in one kernel: the power spectrogram of the (pre-emphasised) signal is never materialised
        const GenMel gm{n_mels, g->mel_lo, g->mel_hi, g->mel_off, g->mel_w, g->n_taps}, none{0, nullptr, nullptr, nullptr, nullptr, 0};
        if (!contrast_rows_only) {   // (else the one-launch kernel has written rows [0, nbase): featurize.hip, run-time geometry)
        gen_launch_stft<false, true>(g, N, T, w, wav_stride, nc, g->win, pk, cfg.use_pre_emphasis, cfg.pre_emphasis_coef, mel, gm, stream);
        hipLaunchKernelGGL(gen_dbstat_kernel, dim3(nc), dim3(256), 0, stream, mel, T, n_mels, cfg.use_pcen, stat);
        hipLaunchKernelGGL(gen_rows_kernel, gt, dim3(256), size_t(n_mels) * G_TT * sizeof(float), stream, mel, T, n_mels, cfg.use_mfcc ? n_mfcc : 0, cfg.use_pcen, stat,
                           g->dct_t, feat, nfeat);
        if (cfg.use_mfcc) {
            hipLaunchKernelGGL(gen_zscore_kernel, dim3(nc), dim3(256), 0, stream, feat, nfeat, T, n_mels, n_mfcc);
            hipLaunchKernelGGL(gen_delta_kernel, dim3((n_mfcc * T + 255) / 256, nc), dim3(256), 0, stream, feat, nfeat, T, n_mels,
                               n_mfcc, cfg.use_delta_delta);
        }
        }
        if (want_contrast) {
            // from the un-emphasised (normalised) signal (:476-478)
            if (narrow) {   // n_fft = 512: rows straight out of the STFT workgroups' tiles
                const long long n_rows = (long long)nc * T;
                const dim3 gs((unsigned)((n_rows + G_FPB * G_CHUNK - 1) / (G_FPB * G_CHUNK)));
                const GenTail tail{contrast, g->freqs, float(g->sample_rate) / 2.0f, nfeat, nbase};
                hipLaunchKernelGGL((gen_stft_kernel<false, false, 1>), gs, dim3(256), 0, stream, w, wav_stride, N, g->hop, T, g->win,
                                   g->tw256, g->tw512, pk, 0.f, feat, none, n_rows, tail);
                hipLaunchKernelGGL((gen_stft_kernel<true, false, 2>), gs, dim3(256), 0, stream, w, wav_stride, N, g->hop, T, g->win_full,
                                   g->tw256, g->tw512, pk, 0.f, feat, none, n_rows, tail);
                hipLaunchKernelGGL(gen_zscore_kernel, dim3(nc), dim3(256), 0, stream, feat, nfeat, T, nbase, contrast.n_bands + 1);
                COUGH_HIP_CHECK(hipGetLastError());
                continue;
            }
            gen_launch_stft<false, false>(g, N, T, w, wav_stride, nc, g->win, pk, 0, 0.f, P, none, stream);
            gen_launch_stft<true, false>(g, N, T, w, wav_stride, nc, g->win_full, pk, 0, 0.f, M, none, stream);
            bool wide = false;   // a band of more than 128 bins (n_fft = 2048 only) ranks its bins out of LDS
            for (int i = 0; i < contrast.n_bands; ++i) wide |= contrast.edges[i + 1] - contrast.edges[i] > G_CT_BINS;
            hipLaunchKernelGGL(gen_contrast_kernel, dim3(gt.x, gt.y, contrast.n_bands + 1), dim3(64),
                               wide ? size_t(G_CT_BINS / 4 * G_TT) * sizeof(float4) : 0, stream, P, M, T, g->nfreq, contrast, g->freqs,
                               float(g->sample_rate) / 2.0f, feat, nfeat, nbase);
            hipLaunchKernelGGL(gen_zscore_kernel, dim3(nc), dim3(256), 0, stream, feat, nfeat, T, nbase, contrast.n_bands + 1);
        }
        COUGH_HIP_CHECK(hipGetLastError());
    }
    return COUGH_OK;
}

}  // namespace cough

extern "C" int cough_pre_emphasis(const float* d_in, long long in_stride, float* d_out, long long out_stride, int n_rows,
                                  int n, float coef, void* stream) {
    using namespace cough;
    COUGH_REQUIRE(d_in && d_out, COUGH_EINVAL, "cough_pre_emphasis: NULL argument");
    COUGH_REQUIRE(n_rows >= 0 && n >= 0 && in_stride >= n && out_stride >= n, COUGH_EINVAL, "cough_pre_emphasis: bad shape");
    COUGH_REQUIRE(d_in != d_out, COUGH_EINVAL, "cough_pre_emphasis: in-place operation is not supported");
    if (n_rows == 0 || n == 0) return COUGH_OK;
    COUGH_REQUIRE(n_rows <= 65535, COUGH_EUNSUPPORTED, "cough_pre_emphasis: at most 65535 rows per call");
    hipLaunchKernelGGL(pre_emphasis_kernel, dim3((n + 255) / 256, n_rows), dim3(256), 0, static_cast<hipStream_t>(stream), d_in,
                       in_stride, d_out, out_stride, n, coef);
    COUGH_HIP_CHECK(hipGetLastError());
    return COUGH_OK;
}

extern "C" int cough_compute_deltas(const float* d_in, float* d_out, long long n_rows, int n_frames, void* stream) {
    using namespace cough;
    COUGH_REQUIRE(d_in && d_out, COUGH_EINVAL, "cough_compute_deltas: NULL argument");
    COUGH_REQUIRE(n_rows >= 0 && n_frames >= 0 && d_in != d_out, COUGH_EINVAL, "cough_compute_deltas: bad shape / in-place");
    const long long total = n_rows * n_frames;
    if (total == 0) return COUGH_OK;
    hipLaunchKernelGGL(deltas_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), d_in,
                       d_out, total, n_frames);
    COUGH_HIP_CHECK(hipGetLastError());
    return COUGH_OK;
}

extern "C" int cough_pcen(const float* d_mel, float* d_out, long long n_rows, int n_frames, float alpha, float delta, float r,
                          float eps, void* stream) {
    using namespace cough;
    COUGH_REQUIRE(d_mel && d_out, COUGH_EINVAL, "cough_pcen: NULL argument");
    COUGH_REQUIRE(n_rows >= 0 && n_frames >= 0 && d_mel != d_out, COUGH_EINVAL, "cough_pcen: bad shape / in-place");
    const long long total = n_rows * n_frames;
    if (total == 0) return COUGH_OK;
    const float dpr = float(std::pow(double(delta), double(r)));   // `delta ** r` is a Python float in the reference
    hipLaunchKernelGGL(pcen_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), d_mel,
                       d_out, total, n_frames, alpha, delta, r, eps, dpr);
    COUGH_HIP_CHECK(hipGetLastError());
    return COUGH_OK;
}
