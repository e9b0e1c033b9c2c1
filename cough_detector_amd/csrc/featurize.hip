// K1 -- fused featuriser for gfx950: waveform -> log-mel / PCEN rows + MFCC + delta [+ delta-delta] [+ the classifier's stem].
//
// Replaces AudioPreprocessor.normalize / extract_features
// (/root/reference/src/preprocessing.py:199-212, :432-489) and the torchaudio transforms it calls
// (T.MelSpectrogram :94-106, T.AmplitudeToDB :109-112, T.MFCC :116-127) at the shipped STFT geometry (16 kHz, n_fft 512,
// hop 160, window 400, 1 s = 101 frames) for ANY filterbank.  The reference runs the STFT->mel->dB chain twice per clip (:398
// and :425); here one pass feeds both branches.
//
// One 4-wave workgroup per clip, 45.5 KB of LDS, so three clips are in flight per CU and their phases overlap (one loads from
// HBM while another runs FFTs and a third stores).  All three per-clip reductions (peak for normalize, dB max for top_db,
// MFCC mean / std) are block reductions, so batches keep the reference's per-clip semantics.
//
//   P0  none.  Peak normalisation (x / max|x|) commutes with everything up to the dB stage: power scales by 1/peak^2, so
//       dB(x/peak) = max(dB(x) - 20*log10(peak), -100) (the -100 is the reference's amin = 1e-10 clamp).  The peak is collected
//       from the samples the frames load anyway, so the clip is read from HBM exactly once.  A clip whose peak lies outside
//       2^-50 .. 2^50 is transformed a second time with scaled window taps (a cold copy of P1); a NaN / Inf sample makes the
//       whole image NaN, as in the reference.
//   P1  per frame (16 lanes each, 4 frames per wave pass): samples straight from global one group ahead (8-byte loads; the 4
//       edge frames take a reflected-index path = torch.stft center / reflect), window taps from registers, packed as 256
//       complex points; 256-pt FFT = radix-16 (registers) -> twiddle (LDS table) -> 16x16 transpose through LDS -> radix-16;
//       real-input split; |X|^2 -> LDS; mel (lane = band); 10*log10 on the hardware log2 -> LDS
//   P2  block max -> top_db floor -> mel rows out (or PCEN); DCT with wave-uniform (scalar) coefficients; mean / unbiased std;
//       z-score, deltas out; STEM: bf16 hi / lo feature images in LDS -> conv7x7 + BN + ReLU + maxpool on the matrix cores
//
// Template parameters (23 instantiations): PRE_EMPH (pre-emphasis while loading), STEM (0 none / 1 bf16 / 2 split-bf16 stem
// fused), FULL (full-band: all 257 bins, CSR filterbank in LDS, run-time n_mels / n_mfcc; otherwise the shipped sparse bank:
// <= 8 register taps per band below bin 128, only bins 0..127 formed), TALL (103-row delta-delta image, stem in two halves),
// PCS (PCEN values feed the stem), GEO (run-time STFT geometry at n_fft 512: hop, window, waveform length and frame count are
// kernel arguments -- other sample rates / hops / windows / segment durations, waveforms of other lengths through any handle, and
// the filterbanks the fixed-geometry kernels do not take: odd band counts, more than 20 MFCCs, PCEN off 64 bands).
#include <cmath>
#include <cstddef>
#include <cstring>
#include <type_traits>
#include <vector>

#include <hip/hip_bf16.h>

#include <cstdlib>

#include "common.h"
#include "fft256.h"
#include "internal.h"

namespace cough {
namespace {

constexpr int NS = 16000, NFFT = 512, HOP = 160, WIN = 400, NFRAMES = 101, NMEL = 64, NMFCC = 13;
constexpr int PADL = NFFT / 2;
constexpr int NBIN = 128;   // spectrum bins formed (0..127)
constexpr int MAXW = 8;     // max non-zero taps of one mel band
constexpr int THREADS = 256, WAVES = THREADS / 64;
constexpr int FPW = 4;      // frames per wave pass
constexpr int NGROUP = (NFRAMES + FPW - 1) / FPW;
constexpr int XROW = 17;    // floats per transpose row (16 + 1 pad: conflict-free ds_read_b32 / ds_write_b32)
constexpr int XFRAME = 16 * XROW;
constexpr int NMF = NMFCC * NFRAMES;  // 1313
constexpr int FIRST_PLAIN = 2, LAST_PLAIN = 98;   // frames whose 400 live taps lie inside the clip

struct FeatTables {
    float win[NFFT];          // periodic Hann(400) zero-padded 56|400|56
    float2 tw256[16][16];     // W256^(j*k1), [j][k1]
    float2 tw512[NBIN];       // W512^k
    int mel_start[NMEL];      // first bin of band m
    float mel_w[NMEL][MAXW];  // taps of band m from mel_start
    float dct_t[NMFCC + 1][NMEL];   // DCT-II ortho, [coeff][mel]; row 13 = zeros (the 7th, unused accumulator of the
                                    // second thread half: keeps the DCT loop free of branches)
};

constexpr size_t LDS_XCH = size_t(WAVES) * FPW * XFRAME * 4;   // 17408
constexpr size_t LDS_MEL = size_t(NMEL) * NFRAMES * 4;        // 25856
constexpr size_t LDS_RED = 16 * 4;
constexpr size_t LDS_TW = size_t(16) * XROW * 8;              // W256^(j*k1) table, 17-float2 row pitch
constexpr size_t LDS_TOTAL = LDS_XCH + LDS_MEL + LDS_RED + LDS_TW;
static_assert(LDS_TOTAL * 3 <= 160 * 1024, "three workgroups per CU");
static_assert(LDS_XCH >= size_t(2) * NMF * 4, "MFCC / delta buffers alias the transpose scratch");

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

#ifdef COUGH_K1_STAMPS
// Diagnostic build only (tools/k1_stamps.py): per-workgroup s_memtime at phase boundaries, written
// to a buffer of its own that no other code reads.  Never compiled into libcough_amd.so.
__device__ unsigned long long* g_stamp_buf = nullptr;
#define K1_STAMP(slot)                                                                      \
    do {                                                                                    \
        if (g_stamp_buf && threadIdx.x == 0)                                                \
            g_stamp_buf[(size_t)blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memtime();    \
    } while (0)
#else
#define K1_STAMP(slot) do { } while (0)
#endif

#ifdef COUGH_K1_MARKERS
// Instruction-budget build only (tools/k1_isa_budget.py): phase names as comments in the generated ISA.
#define K1_MARK(text)                             \
    do {                                          \
        __builtin_amdgcn_sched_barrier(0);        \
        asm volatile("; K1MARK " text);           \
        __builtin_amdgcn_sched_barrier(0);        \
    } while (0)
#else
#define K1_MARK(text) do { } while (0)
#endif

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
__device__ __forceinline__ uint16_t f2bf(float f) {
    __hip_bfloat16 b = __float2bfloat16(f);
    return *reinterpret_cast<uint16_t*>(&b);
}
// geometry of the fused stem for the 90x101 feature image (resnet.hip: stem_bf16_kernel / stem_lds)
constexpr int ST_H = 90, ST_P1H = 22, ST_P1W = 25, ST_ROWS = 94, ST_PITCH = 106;   // image width = NFRAMES
constexpr int ST_PER = ST_P1H * ST_P1W, ST_TILES = (ST_PER + 7) / 8;
constexpr size_t ST_IMG = size_t(ST_ROWS) * ST_PITCH * 2;   // bytes of one bf16 image
static_assert(ST_IMG <= LDS_MEL, "the bf16 feature image aliases the dB buffer");
// split-bf16 stem: two images (hi, lo) at the end of the workgroup's LDS, over everything but the z-scored MFCC rows
constexpr size_t ST_X3_OFF = LDS_TOTAL - 2 * ST_IMG;
static_assert(ST_X3_OFF % 16 == 0 && ST_IMG % 4 == 0 && ST_X3_OFF >= size_t(NMF) * 4, "hi / lo images vs the MFCC buffer");

// kernel argument of the FULL instantiations: device pointers into the featuriser's full-band blob
struct FullBank {
    int n_mels, n_mfcc, n_taps;
    int maxw[2];             // widest band of bands 0..63 / 64..127 (taps)
    int n_frames, n_samples, hop;   // GEO instantiations: frames per clip, samples per clip, hop length
    int cph, cph_pad, cw;    // MFCCs of the first thread half (ceil(n_mfcc / 2)); table rows per half (whole chunks); chunk width 4..7
    const int *lo, *hi, *off;   // [n_mels] first bin, end bin, offset of the band's taps in w
    const float* w;          // CSR taps, times 1/4 (the spectrum is formed as 2X)
    const float* dct;        // [2][cph_pad / cw chunks][n_mels][8] DCT-II (ortho): the cw coefficients of a chunk per mel band,
                             // zero-padded to 8 floats
};
constexpr size_t full_mel_bytes(int n_mels, int n_frames = NFRAMES) { return (size_t(n_mels) * n_frames * 4 + 15) & ~size_t(15); }
// FULL: a wave's transpose scratch interleaves its four frames -- element (row k, frame f, column n) at 65 k + 16 f + n, the
// layout of the stand-alone STFT kernel (spectrogram.hip) -- in 1040 floats instead of 4 x 272; afterwards it holds the four
// frames' power rows of 257 (+ 1 padding) bins at a pitch of 260.  768 bytes less per workgroup: an 80-band bank then still fits
// three workgroups per CU (the hardware grants three up to 53 760 B each -- 1280-byte granules, tools/micro/lds_occupancy.hip)
constexpr int FX_ROW = FPW * 16 + 1, FX_WAVE = 16 * FX_ROW, FX_PROW = 260;
static_assert(FPW * FX_PROW <= FX_WAVE && FX_PROW >= NFFT / 2 + 2 && FX_PROW % 2 == 0, "four power rows replace the transposes");
constexpr size_t LDS_XCH_FULL = size_t(WAVES) * FX_WAVE * 4;   // 16640
constexpr int FULL_MAX_MFCC = int(LDS_XCH_FULL / (size_t(2) * NFRAMES * 4));   // 20: MFCC + delta buffers alias the scratch
constexpr size_t full_lds_bytes(int n_mels, int n_taps, int n_frames = NFRAMES) {
    return LDS_XCH_FULL + full_mel_bytes(n_mels, n_frames) + LDS_RED + LDS_TW + ((size_t(n_taps) * 4 + 15) & ~size_t(15));
}

// PRE_EMPH: pre-emphasis is a separate instantiation (it never costs the shipped path registers).
// STEM: the classifier's stem (conv7x7 s2 + BN + ReLU + maxpool, model.py:227-232) runs at the end of the kernel on
// the matrix cores out of a bf16 copy of the feature image in LDS; `out` may then be nullptr.  1: plain bf16 operands,
// bf16 output; 2: split-bf16 (image and weights as hi + lo, hi*hi + lo*hi + hi*lo per k-step), f32 output.
// FULL: the full-band instantiations -- the same one-launch kernel for ANY filterbank at the shipped STFT geometry (f_max up to the
// Nyquist bin, bands of any width, 2..128 mel bands, up to 20 MFCCs): all 257 bins are formed (X[k] and X[256 - k] from the same
// butterfly, as the stand-alone STFT does), the filterbank is a CSR table in LDS (lane = band, a band's taps in ascending bin
// order), n_mels / n_mfcc are run-time sizes and the dB buffer is sized by them.  The shipped instantiations (FULL = false) keep
// their <= 8 register taps per band below bin 128 and compile-time sizes.
// TALL: the fused split-bf16 stem for the 103-row image of the reference's delta-delta flag (64 mel + 13 MFCC + 13 delta + 13
// delta-delta; src/preprocessing.py:43-49, :471-474).  Its hi + lo images (2 x 23 KB) do not fit beside the MFCC / delta
// buffers in a workgroup's 45.5 KB, so the stem runs in two halves of 13 pooled rows out of a 58-row image each -- three
// workgroups per CU as for the shipped layout.
// PCS: the fused stem's mel rows are the PCEN values (use_pcen with a fused stem) -- a template parameter because the values
// wait in 26 registers from the PCEN branch to the image build, which costs the shipped instantiation 1.3 % when it is a run-time
// choice (same-box A/B, profiles/r05_bench_flags.txt).
// GEO (with FULL, no stem): run-time STFT geometry at n_fft = 512 -- any hop <= 256 (up to 512 when the spans still cover the segment), window <= 512 (all 16 sample pairs of a
// lane are live), segment length (the dB buffer of n_mels x frames must fit the LDS of two workgroups per CU: 201 frames of 64 bands), i.e. other
// sample rates / window durations on the one-launch kernel
// instead of the generic chain.  Samples arrive as 4-byte loads (no alignment contract), two workgroups per CU.
#ifndef COUGH_GEO_WG_PER_CU
#define COUGH_GEO_WG_PER_CU 2   // run-time geometry: 188 VGPRs; three per CU spill 21-33 (measured: profiles/r05_runtime_geometry_featuriser.txt)
#endif
template <bool PRE_EMPH, int STEM, bool FULL = false, bool TALL = false, bool PCS = false, bool GEO = false>
__global__ __launch_bounds__(THREADS, GEO ? COUGH_GEO_WG_PER_CU : 3) void featurize_kernel(
    const float* __restrict__ wav, long long wav_stride, float* __restrict__ out, int nfeat,
    const FeatTables* __restrict__ tb, int normalize, float pre_coef, int delta_delta /* 0: MFCC + delta rows,
    1: + delta-delta, 2: no MFCC rows */, int pcen, StemFuse stem, FullBank fbk,
    const float* __restrict__ full_dct /* = fbk.dct: read-only for the kernel's lifetime, so its wave-uniform loads are scalar */,
    float* __restrict__ peak_out /* [n] or nullptr: the clip's max |sample| under the fused normalise (the contrast path's scale) */) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(!GEO || (FULL && STEM == 0 && !TALL && !PCS), "run-time geometry: full-band instantiations without a stem");
    // frames per clip, four-frame groups, samples per clip, hop: compile-time constants unless GEO
    const int NF = GEO ? fbk.n_frames : NFRAMES, NGR = GEO ? (fbk.n_frames + FPW - 1) / FPW : NGROUP;
    const int NSMP = GEO ? fbk.n_samples : NS, HP = GEO ? fbk.hop : HOP;
    constexpr int N1A = GEO ? 0 : 1, N1B = GEO ? 16 : 15;   // live sample pairs of a lane (window 400 in 512: pairs 1..14)
    const int nmel = FULL ? fbk.n_mels : NMEL, nmfcc = FULL ? fbk.n_mfcc : NMFCC, nmf = nmfcc * NF;
    const size_t lds_mel = FULL ? full_mel_bytes(nmel, NF) : LDS_MEL;
    constexpr size_t lds_xch = FULL ? LDS_XCH_FULL : LDS_XCH;
    float* xs = reinterpret_cast<float*>(smem);
    float* melbuf = reinterpret_cast<float*>(smem + lds_xch);
    float* red = reinterpret_cast<float*>(smem + lds_xch + lds_mel);
    float2* twl = reinterpret_cast<float2*>(smem + lds_xch + lds_mel + LDS_RED);   // [16][XROW]
    // FULL: the taps of the CSR filterbank behind the twiddle table (a band's first bin / width / tap offset live in registers)
    float* c_w = reinterpret_cast<float*>(smem + lds_xch + lds_mel + LDS_RED + LDS_TW);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long clip = blockIdx.x;
    const float* x = wav + clip * wav_stride;
    float* o = out + clip * (long long)nfeat * NF;
    const bool wr = out != nullptr;   // features are materialised (always, unless the fused pipeline asks not to)

    K1_STAMP(0);
    K1_STAMP(1);
    K1_MARK("PHASE prologue (window taps, twiddle table, mel taps to registers)");

    // ---------------- P1: STFT power -> mel -> dB ----------------
    const int j = lane & 15, fsub = lane >> 4;
    twl[(tid >> 4) * XROW + (tid & 15)] = tb->tw256[tid >> 4][tid & 15];   // 256 threads = 16 x 16 entries
    // FULL: lane = band.  Pass 0 takes bands 0..63 with the wave's four frames per lane; the LAST pass of a bank with more than 64
    // bands maps its r = n_mels - 64 bands as (band, frames): all four frames per lane for r > 32, two for r > 16, one otherwise --
    // so 80 bands do not leave 48 of 64 lanes idle through the widest bands of the bank
    int b_lo[2] = {0, 0}, b_w[2] = {0, 0}, b_band[2] = {0, 0};
    const float* b_taps[2] = {c_w, c_w};
    int fpl1 = FPW;   // frames per lane in pass 1
    if constexpr (FULL) {
        for (int i = tid; i < fbk.n_taps; i += THREADS) c_w[i] = fbk.w[i];
        const int r = nmel - 64;
        fpl1 = r > 32 ? 4 : r > 16 ? 2 : 1;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int mb = ps == 0 ? lane : 64 + (fpl1 == 4 ? lane : fpl1 == 2 ? (lane & 31) : (lane & 15));
            if (mb < nmel) {
                b_band[ps] = mb;
                b_lo[ps] = fbk.lo[mb];
                b_w[ps] = fbk.hi[mb] - b_lo[ps];
                b_taps[ps] = c_w + fbk.off[mb];
            } else {
                b_band[ps] = -1;   // idle lane
            }
        }
    }
    __syncthreads();
    const float2* tw_row = twl + j * XROW;   // row pitch 17 float2: the 16 lanes of a frame hit 16 distinct banks
    const float2 tw_j = tb->tw512[j];   // W512^j; W512^(j+16*k2) = W512^j * W32^k2
    float mw[MAXW];
#pragma unroll
    for (int q = 0; q < MAXW; ++q) mw[q] = tb->mel_w[lane][q];
    const int mstart = tb->mel_start[lane];

    float* myx = xs + (wave * FPW + fsub) * XFRAME;                 // shipped: the frame's own 16 x 17 scratch
    float* myw = xs + wave * FX_WAVE;                               // FULL: the wave's interleaved scratch
    float* fxw = myw + 16 * fsub + j;                               //   transpose write base: row k1 at fxw[k1 * FX_ROW]
    const float* fxr = myw + FX_ROW * j + 16 * fsub;                //   transpose read base: column n2 at fxr[n2]
    float* prow = myw + FX_PROW * fsub;                             //   afterwards: this frame's power row
    float run_max = -INFINITY;   // max raw dB seen by this lane
    float peak = 0.f;            // max |sample| seen by this lane
    float chk = 0.f;             // stays 0 while every mel power of this lane is finite, NaN otherwise (acc * 0)
    // Frame samples are fetched one group AHEAD of their use (28 VGPRs): the HBM/L2 latency of a
    // group's 14 eight-byte loads hides behind the ~600 VALU instructions of the previous group, and every
    // byte of the clip is requested from HBM once.
    // GEO: the four frames of group g lie inside the clip with their whole 512-sample spans (wave-uniform)
    auto geo_plain = [&](int g) {
        return HP * (FPW * g) - PADL >= 0 && FPW * g + FPW - 1 < NF && HP * (FPW * g + FPW - 1) - PADL + NFFT <= NSMP;
    };
    [[maybe_unused]] float raw_left = 0.f;   // GEO with pre-emphasis: the sample in front of the frame's first one (0 at the clip's start)
    auto load_group = [&](int g, float2 (&raw)[16]) {
        const int t_raw = FPW * g + fsub;
        const int t = t_raw < NF ? t_raw : NF - 1;             // idle sub-frames redo the last frame
        const int s0 = HP * t - PADL + 2 * j;                  // clip index of padded sample 2j of frame t
        if constexpr (GEO) {
            // frames whose whole 512-sample span lies inside the clip load directly, the others by reflected index (torch.stft
            // center / reflect); 4-byte loads: hop and row stride are arbitrary
            if (geo_plain(g)) {   // wave-uniform
#pragma unroll
                for (int n1 = 0; n1 < 16; ++n1) raw[n1] = make_float2(x[s0 + 32 * n1], x[s0 + 32 * n1 + 1]);
                if constexpr (PRE_EMPH) raw_left = s0 - 2 * j > 0 ? x[s0 - 2 * j - 1] : 0.f;
            } else {
#pragma unroll
                for (int n1 = 0; n1 < 16; ++n1) {
                    int i0 = s0 + 32 * n1, i1 = i0 + 1;
                    i0 = i0 < 0 ? -i0 : (i0 >= NSMP ? 2 * (NSMP - 1) - i0 : i0);
                    i1 = i1 < 0 ? -i1 : (i1 >= NSMP ? 2 * (NSMP - 1) - i1 : i1);
                    raw[n1] = make_float2(x[i0], x[i1]);
                }
            }
        } else
        if (FPW * g >= FIRST_PLAIN && FPW * g + FPW - 1 <= LAST_PLAIN) {   // wave-uniform
            K1_MARK("WEIGHT 0.885 (23 of 26 groups: frames inside the clip)");
#pragma unroll
            for (int n1 = 1; n1 < 15; ++n1) raw[n1] = *reinterpret_cast<const float2*>(x + s0 + 32 * n1);
            K1_MARK("ENDWEIGHT");
        } else {   // frames 0,1,99,100 reach into the reflect padding of torch.stft(center=True)
            K1_MARK("WEIGHT 0.115 (3 of 26 groups: reflected edge frames)");
#pragma unroll
            for (int n1 = 1; n1 < 15; ++n1) {
                int i0 = s0 + 32 * n1, i1 = i0 + 1;
                i0 = i0 < 0 ? -i0 : (i0 >= NS ? 2 * (NS - 1) - i0 : i0);
                i1 = i1 < 0 ? -i1 : (i1 >= NS ? 2 * (NS - 1) - i1 : i1);
                raw[n1] = make_float2(x[i0], x[i1]);
            }
            K1_MARK("ENDWEIGHT");
        }
    };
    float2 raw[16];
    float peak_m = 0.f;   // block peak of the (possibly rescaled) samples, fused normalise only
    // P1 as a body that can run twice: once for every clip, and a second time -- window taps scaled by a power of two -- for a
    // clip whose peak lies outside 2^-50 .. 2^50 under the fused normalise (never audio; see below)
    auto p1_pass = [&](const float ws1, const float ws2) {
        float w_re[16], w_im[16];   // window taps of this lane's samples (zero taps of the padded window are never loaded)
        const int jw = 2 * j;
#pragma unroll
        for (int n1 = N1A; n1 < N1B; ++n1) {
            w_re[n1] = tb->win[32 * n1 + jw] * ws1 * ws2;
            w_im[n1] = tb->win[32 * n1 + jw + 1] * ws1 * ws2;
        }
        load_group(wave, raw);

        K1_MARK("LOOP 6.5 P1 four-frame groups per wave");
        for (int g = wave; g < NGR; g += WAVES) {
            K1_MARK("PHASE P1 peak + window multiply + next group's loads");
            const int t_raw = FPW * g + fsub;
            const int t = t_raw < NF ? t_raw : NF - 1;             // idle sub-frames redo the last frame
            const int s0 = HP * t - PADL + 2 * j;                  // clip index of padded sample 2j of frame t
            float2 a[16];
            if constexpr (!GEO) {
                a[0] = make_float2(0.f, 0.f);    // window is zero on samples [0,56) and [456,512)
                a[15] = make_float2(0.f, 0.f);
            }
            if constexpr (!PRE_EMPH) {
#pragma unroll
                for (int n1 = N1A; n1 < N1B; ++n1) {
                    peak = fmaxf(peak, fmaxf(fabsf(raw[n1].x), fabsf(raw[n1].y)));   // one v_max3_f32 with |.| modifiers
                    a[n1] = make_float2(raw[n1].x * w_re[n1], raw[n1].y * w_im[n1]);
                }
            }
            if constexpr (PRE_EMPH) {
                // y[n] = x[n] - coef*x[n-1], y[0] = x[0] (preprocessing.py:235-238), applied before the reflect padding as the
                // reference does; no FMA contraction (mul_rn, __fsub_rn).  The peak is of x, not of the emphasised signal.
                if (GEO ? geo_plain(g) : (FPW * g >= FIRST_PLAIN && FPW * g + FPW - 1 <= LAST_PLAIN)) {   // wave-uniform: frames inside the clip
                    // x[i0 - 1] is the second sample of the lane to the left (row_ror:1); lane 0 takes lane 15's pair of the
                    // previous n1 (n1 = 1: a zero tap of the padded window, any finite value does; GEO: every tap is live and the
                    // sample in front of the frame arrives with the group's loads)
                    float carry = GEO ? raw_left : 0.f;
#pragma unroll
                    for (int n1 = N1A; n1 < N1B; ++n1) {
                        const float rot = dpp_mov<0x121>(raw[n1].y);
                        const float left = j == 0 ? carry : rot;
                        carry = rot;
                        const float x0 = raw[n1].x, x1 = raw[n1].y;
                        peak = fmaxf(peak, fmaxf(fabsf(x0), fabsf(x1)));
                        a[n1] = make_float2(__fsub_rn(x0, mul_rn(pre_coef, left)) * w_re[n1],
                                            __fsub_rn(x1, mul_rn(pre_coef, x0)) * w_im[n1]);
                    }
                } else {   // reflected edge frames (3 of 26 groups; GEO: every group): the left neighbour in CLIP order, by re-gather
#pragma unroll
                    for (int n1 = N1A; n1 < N1B; ++n1) {
                        int i0 = s0 + 32 * n1, i1 = i0 + 1;
                        i0 = i0 < 0 ? -i0 : (i0 >= NSMP ? 2 * (NSMP - 1) - i0 : i0);
                        i1 = i1 < 0 ? -i1 : (i1 >= NSMP ? 2 * (NSMP - 1) - i1 : i1);
                        const float p0 = i0 > 0 ? mul_rn(pre_coef, x[i0 - 1]) : 0.f;
                        const float p1 = i1 > 0 ? mul_rn(pre_coef, x[i1 - 1]) : 0.f;
                        const float x0 = raw[n1].x, x1 = raw[n1].y;
                        peak = fmaxf(peak, fmaxf(fabsf(x0), fabsf(x1)));
                        a[n1] = make_float2(__fsub_rn(x0, p0) * w_re[n1], __fsub_rn(x1, p1) * w_im[n1]);
                    }
                }
            }
            // the raw registers are free again: the next group's samples start moving now and land while
            // this group's FFT / mel / log run
            if (g + WAVES < NGR) load_group(g + WAVES, raw);
            K1_MARK("PHASE P1 radix-16 #1");
            dft16(a);
            K1_MARK("PHASE P1 twiddle (LDS table) complex multiply");
#pragma unroll
            for (int k1 = 1; k1 < 16; ++k1) a[k1] = cmul(a[k1], tw_row[k1]);
            K1_MARK("PHASE P1 16x16 transpose through LDS");
            // 16x16 transpose through LDS: real parts, then imaginary parts through the same scratch
            float2 z[16];
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) (FULL ? fxw[k1 * FX_ROW] : myx[k1 * XROW + j]) = a[k1].x;
            wave_lds_fence();
#pragma unroll
            for (int n2 = 0; n2 < 16; ++n2) z[n2].x = FULL ? fxr[n2] : myx[j * XROW + n2];
            wave_lds_fence();
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) (FULL ? fxw[k1 * FX_ROW] : myx[k1 * XROW + j]) = a[k1].y;
            wave_lds_fence();
#pragma unroll
            for (int n2 = 0; n2 < 16; ++n2) z[n2].y = FULL ? fxr[n2] : myx[j * XROW + n2];
            K1_MARK("PHASE P1 radix-16 #2");
            dft16(z);   // z[k2] = Z[j + 16*k2]
            K1_MARK("PHASE P1 real-input split + |X|^2 -> LDS");

            // partner Z[256-k] lives in lane (16-j)&15, register 15-k2 (j>=1) or 16-k2 (j==0)
            float2 rv[8];   // z[8 + r] of lane (16 - j) & 15 of the same frame: row_mirror (j -> 15 - j), then rotate right by one
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                rv[r].x = dpp_mov<0x121>(dpp_mov<0x140>(z[8 + r].x));
                rv[r].y = dpp_mov<0x121>(dpp_mov<0x140>(z[8 + r].y));
            }
            wave_lds_fence();
            if constexpr (!FULL) {
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) {
                    const float2 zk = z[k2];
                    const float2 zp0 = (k2 == 0) ? z[0] : rv[8 - k2];   // j == 0
                    const float2 zp = (j == 0) ? zp0 : rv[7 - k2];
                    // 2*X[k] = (Zk + conj Zp) - i*W512^k*(Zk - conj Zp); the factor 1/2 (1/4 in power) is folded
                    // into the mel taps at create time (exact: a power of two)
                    const float ex = zk.x + zp.x, ey = zk.y - zp.y;
                    const float ox = zk.y + zp.y, oy = zp.x - zk.x;
                    const float qx = W32C[k2] * ox - W32S[k2] * oy, qy = W32C[k2] * oy + W32S[k2] * ox;   // W32^k2 * O
                    const float xr = ex + tw_j.x * qx - tw_j.y * qy;
                    const float xi = ey + tw_j.x * qy + tw_j.y * qx;
                    myx[j + 16 * k2] = xr * xr + xi * xi;
                }
                wave_lds_fence();
                K1_MARK("PHASE P1 sparse mel + log2 -> LDS");
                // sparse mel: lane = band, the wave's 4 frames
#pragma unroll
                for (int f = 0; f < FPW; ++f) {
                    const float* p = xs + (wave * FPW + f) * XFRAME + mstart;
                    float acc = 0.f;
#pragma unroll
                    for (int q = 0; q < MAXW; ++q) acc += mw[q] * p[q];
                    const int tf = FPW * g + f;
                    if (tf < NF) {
                        // raw dB = 10*log10(acc) = 3.0103*log2(acc) on the hardware log2 (|error| ~1e-6 dB);
                        // -inf for 0: amin and normalisation are applied in P2
                        const float db = 3.01029995663981195f * __log2f(acc);
                        melbuf[lane * NF + tf] = db;
                        run_max = fmaxf(run_max, db);
                        chk = fmaf(acc, 0.f, chk);   // NaN / Inf power (a non-finite sample under the frame, f32 overflow) -> NaN, sticky
                    }
                }
            } else {
                // all 257 bins: 2X[k] = 2E + W^k 2O and 2X[256 - k] = conj(2E - W^k 2O) from the same butterfly (as spectrogram.hip);
                // the four power rows of the wave's frames replace its transpose scratch
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) {
                    const float2 zk = z[k2];
                    const float2 zp0 = (k2 == 0) ? z[0] : rv[8 - k2];   // j == 0
                    const float2 zp = (j == 0) ? zp0 : rv[7 - k2];
                    const float ex = zk.x + zp.x, ey = zk.y - zp.y;
                    const float ox = zk.y + zp.y, oy = zp.x - zk.x;
                    const float qx = W32C[k2] * ox - W32S[k2] * oy, qy = W32C[k2] * oy + W32S[k2] * ox;
                    const float px = tw_j.x * qx - tw_j.y * qy, py = tw_j.x * qy + tw_j.y * qx;
                    const float ar = ex + px, ai = ey + py, br = ex - px, bi = ey - py;
                    prow[j + 16 * k2] = ar * ar + ai * ai;
                    prow[NFFT / 2 - (j + 16 * k2)] = br * br + bi * bi;
                }
                if (j == 0) prow[NFFT / 4] = 4.0f * (z[8].x * z[8].x + z[8].y * z[8].y);   // X[128] = conj Z[128]
                wave_lds_fence();
                K1_MARK("PHASE P1 CSR mel + log2 -> LDS");
                // CSR mel: lane = band (64 bands per pass), the wave's 4 frames share every tap read; a band's taps run in ascending
                // bin order up to the widest band of the pass (wave-uniform bound, narrower bands idle)
                auto mel_pass = [&](auto fpl_tag, int ps) {
                    constexpr int FPL = decltype(fpl_tag)::value;   // frames per lane: 4, 2 or 1
                    const int f0 = FPL == 4 ? 0 : FPL == 2 ? 2 * (lane >> 5) : (lane >> 4);
                    const int mb = b_band[ps], wdt = mb >= 0 ? b_w[ps] : 0;
                    const float* wm = b_taps[ps];
                    const float* p0 = myw + f0 * FX_PROW + b_lo[ps];
                    float acc[FPL];
#pragma unroll
                    for (int f = 0; f < FPL; ++f) acc[f] = 0.f;
                    // Branch-free, two PAIRS of bins per step, every access an aligned 8-byte LDS read: a band's taps start at an even bin
                    // and have an even count (zero taps as padding, built at create time), so the taps and the four frames' powers of a bin
                    // pair are one ds_read_b64 each and the reads of a step are all in flight together.  (One tap per iteration behind a
                    // divergent branch serialised the loop on the LDS latency -- 19 round trips for the widest band of the 64-band / 8 kHz
                    // bank; 4-byte reads with clamped indices cost 100 LDS instructions per four-frame group.)  A lane past its band's end
                    // re-reads the band's last pair with weight 0: no read leaves the band.
                    const int wmax = fbk.maxw[ps];   // a multiple of 4
                    const int lastp = wdt > 2 ? wdt - 2 : 0;
                    for (int k0 = 0; k0 < wmax; k0 += 4) {
                        float2 w[2], pw[2][FPL];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int k = k0 + 2 * u, kk = k < lastp ? k : lastp;
                            w[u] = *reinterpret_cast<const float2*>(wm + kk);
#pragma unroll
                            for (int f = 0; f < FPL; ++f) pw[u][f] = *reinterpret_cast<const float2*>(p0 + f * FX_PROW + kk);
                            if (k >= wdt) w[u] = make_float2(0.f, 0.f);
                        }
#pragma unroll
                        for (int u = 0; u < 2; ++u)
#pragma unroll
                            for (int f = 0; f < FPL; ++f) acc[f] = fmaf(w[u].y, pw[u][f].y, fmaf(w[u].x, pw[u][f].x, acc[f]));
                    }
#pragma unroll
                    for (int f = 0; f < FPL; ++f) {
                        const int tf = FPW * g + f0 + f;
                        if (mb >= 0 && tf < NF) {
                            const float db = 3.01029995663981195f * __log2f(acc[f]);   // as above
                            melbuf[mb * NF + tf] = db;
                            run_max = fmaxf(run_max, db);
                            chk = fmaf(acc[f], 0.f, chk);
                        }
                    }
                };
                mel_pass(std::integral_constant<int, 4>{}, 0);
                if (nmel > 64) {   // workgroup-uniform
                    if (fpl1 == 4) mel_pass(std::integral_constant<int, 4>{}, 1);
                    else if (fpl1 == 2) mel_pass(std::integral_constant<int, 2>{}, 1);
                    else mel_pass(std::integral_constant<int, 1>{}, 1);
                }
            }
            wave_lds_fence();
        }
    };
    p1_pass(1.f, 1.f);

    K1_MARK("ENDLOOP");
    K1_MARK("PHASE P2 block max / peak, shift and floor");
    K1_STAMP(2);   // wave 0 finished its frames
    // ---------------- P2: top_db floor, mel rows, DCT, z-score, deltas ----------------
    if (normalize) {
        peak_m = block_max(peak, red, tid);
        if (peak_out != nullptr && tid == 0) peak_out[clip] = peak_m;
        // Peak normalisation is applied as a shift in dB, which needs the power of the RAW samples to be representable: a clip
        // whose peak lies outside 2^-50 .. 2^50 (never audio) is transformed again with the window taps scaled by a power of
        // two (exact), so that the reference's `waveform / max` (preprocessing.py:209-212) holds for denormal and huge peaks too
        const int pe = (__float_as_int(peak_m) >> 23) & 0xff;   // biased exponent: 0 = zero / denormal, 255 = Inf
        // silence stays as it is; an Inf sample is caught below
        if ((pe < 127 - 50 || pe > 127 + 50) && peak_m != 0.f && pe != 255) {   // workgroup-uniform
            const int e = pe ? pe - 127 : -127 - __builtin_clz(__float_as_int(peak_m) << 9);   // floor(log2(peak))
            const int k = (e > 0 ? 40 : -40) - e;                                                // peak * 2^k ~ 2^+-40
            const float ws1 = __int_as_float((127 + k / 2) << 23);   // 2^k in two exact factors
            const float ws2 = __int_as_float((127 + k - k / 2) << 23);
            peak_m = peak_m * ws1 * ws2;
            run_max = -INFINITY;
            chk = 0.f;
            __syncthreads();   // every wave is through its mel reads before the scratch / dB buffer are written again
            p1_pass(ws1, ws2);
        }
    }
    if (chk != chk) run_max = INFINITY;   // finite powers give a finite or -inf dB: +inf marks the clip
    const float raw_max = block_max(run_max, red, tid);
    // A NaN or Inf sample (or an f32 overflow of the power) makes the reference's WHOLE image NaN: the frames over it are NaN in
    // every bin, AmplitudeToDB's per-clip `amax` is then NaN and its floor poisons every cell, the z-scores and deltas follow
    // (preprocessing.py:405-410, :428; normalize() leaves a clip with a NaN maximum alone, :209-212)
    if (raw_max == INFINITY) {   // workgroup-uniform
        const float nanv = __builtin_nanf("");
        const int rows = nmel + (delta_delta == 2 ? 0 : (delta_delta == 1 ? 3 : 2) * nmfcc);
        if (wr)
            for (int i = tid; i < rows * NF; i += THREADS) o[i] = nanv;
        if constexpr (STEM != 0)
            if (tid == 0) stem.nanflag[clip] = 1;   // a1 is left unwritten: the head overwrites this clip's logits
        return;
    }
    if constexpr (STEM != 0)
        if (tid == 0) stem.nanflag[clip] = 0;
    float shift = 0.f;   // 20*log10(peak): waveform / waveform.abs().max() if max > 0 (preprocessing.py:209-212)
    if (normalize && peak_m > 0.f) shift = 20.0f * log10f(peak_m);
    // AmplitudeToDB('power', top_db=80): amin = 1e-10 <=> -100 dB; the floor is relative to the per-clip max
    const float floor_db = fmaxf(raw_max - shift, -100.0f) - 80.0f;
    K1_STAMP(3);   // all waves finished P1
    bool wr_mel = wr;
    // STEM == 2 with PCEN: the thread's 26 normalised PCEN values stay in registers until the feature image is built (the dB
    // buffer is still needed for the MFCC branch in between)
    [[maybe_unused]] float pcv[PCS ? 26 : 1];
    K1_MARK("SKIP PCEN branch (off in the shipped configuration)");
    if (GEO && pcen) {
        // Run-time frame count (<= 208) and band count (<= 128): thread = (band, quarter of the frames) walks its quarter in up to
        // two chunks of 26 frames -- the body of the fixed-geometry branch below.  Bands 0..63 keep their values in registers (two
        // workgroups per CU: 256 VGPRs) until the clip's minimum and maximum are known; bands 64..127 (a second round) leave theirs
        // un-normalised in the output rows and the same thread rescales them there.  Threads of a band past the last one idle (an
        // empty frame range; they read band 0).
        const int mt = tid >> 2, quarter = (NF + 3) >> 2;
        const int q0 = (tid & 3) * quarter;
        float lmin = INFINITY, lmax = -INFINITY;
        // chunk c of band mm: values of frames q0 + 26 c + k; [q0, qe) is the thread's live frame range
        auto pcen_chunk = [&](int mm, int c, int qe, float (&pv)[26]) {
            const int t0 = q0 + 26 * c;
            float p[36], s2[35];
#pragma unroll
            for (int k = 0; k < 36; ++k) {
                const int u = t0 - 5 + k;
                p[k] = (u >= 0 && u < NF) ? exp2f((melbuf[mm * NF + u] - shift) * 0.33219280948873623f) : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 35; ++k) s2[k] = p[k] + p[k + 1];
#pragma unroll
            for (int k = 0; k < 26; ++k) {
                const float sm = ((s2[k] + s2[k + 2]) + (s2[k + 4] + s2[k + 6])) + s2[k + 8];
                const float gain = __builtin_amdgcn_exp2f(-0.98f * __builtin_amdgcn_logf(1e-6f + sm * 0.1f));
                pv[k] = __builtin_amdgcn_sqrtf(fmaf(p[k + 5], gain, 2.0f)) - 1.41421356237309515f;
                if (t0 + k < qe) {
                    lmin = fminf(lmin, pv[k]);
                    lmax = fmaxf(lmax, pv[k]);
                }
            }
        };
        const int m = mt < nmel ? mt : 0;
        const int q1 = mt >= nmel ? q0 : q0 + quarter < NF ? q0 + quarter : NF;
        float pv[2][26];
        pcen_chunk(m, 0, q1, pv[0]);
        if (quarter > 26) pcen_chunk(m, 1, q1, pv[1]);   // workgroup-uniform
        const int m2 = mt + 64 < nmel ? mt + 64 : 0;
        const int q2 = mt + 64 >= nmel ? q0 : q0 + quarter < NF ? q0 + quarter : NF;
        if (nmel > 64) {   // workgroup-uniform: the second round
#pragma unroll 1
            for (int c = 0; c < 2; ++c) {
                if (c == 0 || quarter > 26) {
                    float pw[26];
                    pcen_chunk(m2, c, q2, pw);
#pragma unroll
                    for (int k = 0; k < 26; ++k)
                        if (q0 + 26 * c + k < q2) o[m2 * NF + q0 + 26 * c + k] = pw[k];
                }
            }
        }
        const float mn = -block_max(-lmin, red, tid), mx = block_max(lmax, red, tid);
        const float rng = 1.0f / (mx - mn + 1e-8f);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int k = 0; k < 26; ++k)
                if (q0 + 26 * c + k < q1) o[m * NF + q0 + 26 * c + k] = (pv[c][k] - mn) * rng;
        for (int t = q0; t < q2; ++t) o[m2 * NF + t] = (o[m2 * NF + t] - mn) * rng;
        wr_mel = false;
    } else
    if (pcen) {
        // PCEN branch of extract_mel_spectrogram (preprocessing.py:305-340, :400-404): mel rows =
        // min-max-normalised (mel / (1e-6 + smooth)^0.98 + 2)^0.5 - 2^0.5, smooth = 10-frame moving average
        // (zero padded, always / 10).  Works on the (peak-normalised) mel POWER, rebuilt from the raw dB
        // buffer; thread = (band, quarter of the frames) slides the window over its own 36 powers.
        const int m = tid >> 2, t0 = (tid & 3) * 26, t1 = t0 + 26 < NF ? t0 + 26 : NF;   // NF <= 104
        float p[36], pv[26];
#pragma unroll
        for (int k = 0; k < 36; ++k) {
            const int u = t0 - 5 + k;
            p[k] = (u >= 0 && u < NF) ? exp2f((melbuf[m * NF + u] - shift) * 0.33219280948873623f) : 0.f;
        }
        float lmin = INFINITY, lmax = -INFINITY;
        float s2[35];   // pair sums shared by neighbouring windows: 5 adds per window instead of 10
#pragma unroll
        for (int k = 0; k < 35; ++k) s2[k] = p[k] + p[k + 1];
#pragma unroll
        for (int k = 0; k < 26; ++k) {
            const float sm = ((s2[k] + s2[k + 2]) + (s2[k + 4] + s2[k + 6])) + s2[k + 8];
            // mel / (eps + smooth)^0.98 as mel * 2^(-0.98 log2(eps + smooth)): hardware log2 / exp2 / sqrt (1 ulp each; the
            // argument eps + smooth >= 1e-6 is a normal number, the result feeds a min-max normalisation)
            const float gain = __builtin_amdgcn_exp2f(-0.98f * __builtin_amdgcn_logf(1e-6f + sm * 0.1f));
            pv[k] = __builtin_amdgcn_sqrtf(fmaf(p[k + 5], gain, 2.0f)) - 1.41421356237309515f;
            if (t0 + k < t1) { lmin = fminf(lmin, pv[k]); lmax = fmaxf(lmax, pv[k]); }
        }
        const float mn = -block_max(-lmin, red, tid), mx = block_max(lmax, red, tid);
        const float rng = 1.0f / (mx - mn + 1e-8f);
        if (wr) {
#pragma unroll
            for (int k = 0; k < 26; ++k)
                if (t0 + k < t1) o[m * NF + t0 + k] = (pv[k] - mn) * rng;
        }
        if constexpr (PCS) {
#pragma unroll
            for (int k = 0; k < 26; ++k) pcv[k] = (pv[k] - mn) * rng;
        }
        wr_mel = false;   // the log-mel pass below only prepares the floored dB for the MFCC branch
    }
    K1_MARK("ENDSKIP");
    K1_MARK("PHASE P2 floor + mel rows store");
    if constexpr (GEO) {   // n_mels x frames may be odd and the clip's rows 4-byte aligned only: one element per step
        for (int i = tid; i < nmel * NF; i += THREADS) {
            const float d = fmaxf(fmaxf(melbuf[i] - shift, -100.0f), floor_db);
            melbuf[i] = d;
            if (wr_mel) o[i] = fminf(fmaxf((d + 80.0f) * 0.0125f, 0.f), 1.f);
        }
    } else
    for (int i2 = tid; i2 < nmel * NFRAMES / 2; i2 += THREADS) {   // 2 elements / thread: 8-byte stores (n_mels is even)
        float2 d = reinterpret_cast<float2*>(melbuf)[i2];
        d.x = fmaxf(fmaxf(d.x - shift, -100.0f), floor_db);
        d.y = fmaxf(fmaxf(d.y - shift, -100.0f), floor_db);
        reinterpret_cast<float2*>(melbuf)[i2] = d;
        float2 v;
        v.x = fminf(fmaxf((d.x + 80.0f) * 0.0125f, 0.f), 1.f);       // (dB + 80) / 80, preprocessing.py:409-410
        v.y = fminf(fmaxf((d.y + 80.0f) * 0.0125f, 0.f), 1.f);       // (x * 1/80 is within 1 ulp of x / 80)
        if (wr_mel) reinterpret_cast<float2*>(o)[i2] = v;
    }
    __syncthreads();
    K1_STAMP(4);   // mel rows written
    if (delta_delta == 2) return;   // use_mfcc = False: mel rows only (workgroup-uniform)
    K1_MARK("PHASE P2 DCT 13x64");
    // DCT: thread = (frame t, coefficient half); coefficients are wave-uniform -> scalar loads
    float* mf = xs;              // [13][101] z-scored MFCC
    // delta (needed in LDS only for delta-delta); GEO: over the dB buffer, which is dead after the DCT -- the scratch then only has
    // to hold the MFCC rows (n_mfcc x frames x 4 <= 16 640 B: 20 MFCCs of 208 frames)
    float* dl = GEO ? melbuf : mf + nmf;
    const int tt0 = tid & 127;
    const int chalf = __builtin_amdgcn_readfirstlane(tid >> 7);   // waves 0,1: c 0..6; waves 2,3: c 7..12
    if constexpr (FULL) {
        // run-time n_mels / n_mfcc: the half's coefficients in chunks of fbk.cw (the table is [half][cph_pad][n_mels] with zero rows
        // up to whole chunks, so the loop stays free of branches and its coefficients scalar loads); raw MFCCs go to LDS, the
        // z-score follows out of LDS
        const int cph = fbk.cph, nch = chalf ? nmfcc - cph : cph, cbase = chalf * cph;
        auto dct_chunks = [&](auto cw_tag) {
            constexpr int CW = decltype(cw_tag)::value;   // coefficients per chunk
            // GEO: more than 128 frames take further rounds of (frame, coefficient half) threads (workgroup-uniform trip count)
            for (int tt = tt0; tt < (GEO ? ((NF + 127) & ~127) : 128); tt += 128)
            for (int cq = 0; cq < nch; cq += CW) {
                float acc[CW];
#pragma unroll
                for (int cc = 0; cc < CW; ++cc) acc[cc] = 0.f;
                if (tt < NF) {
                    // the chunk's coefficients of one mel band are 8 consecutive floats: one s_load_dwordx8 per band (rows at a
                    // run-time stride of n_mels made every coefficient a scalar load of its own: 3x the shipped DCT phase)
                    const float* drow = full_dct + (size_t(chalf) * fbk.cph_pad + cq) / CW * size_t(nmel) * 8;
#pragma unroll 8
                    for (int m = 0; m < nmel; ++m) {
                        const float v = melbuf[m * NF + tt];
#pragma unroll
                        for (int cc = 0; cc < CW; ++cc) acc[cc] = fmaf(drow[m * 8 + cc], v, acc[cc]);
                    }
#pragma unroll
                    for (int cc = 0; cc < CW; ++cc)
                        if (cq + cc < nch) mf[(cbase + cq + cc) * NF + tt] = acc[cc];
                }
            }
        };
        // the chunk width that wastes the fewest zero rows: 7 for up to seven coefficients per half (13 MFCCs), 5 for ten (20), ...
        if (fbk.cw == 7) dct_chunks(std::integral_constant<int, 7>{});
        else if (fbk.cw == 6) dct_chunks(std::integral_constant<int, 6>{});
        else if (fbk.cw == 5) dct_chunks(std::integral_constant<int, 5>{});
        else dct_chunks(std::integral_constant<int, 4>{});
        __syncthreads();
        float ls = 0.f;
        for (int i = tid; i < nmf; i += THREADS) ls += mf[i];
        const float mean = block_sum(ls, red, tid) / float(nmf);
        float lq = 0.f;
        for (int i = tid; i < nmf; i += THREADS) {
            const float d = mf[i] - mean;
            lq += d * d;
        }
        const float sd = sqrtf(block_sum(lq, red, tid) / float(nmf - 1));   // torch.std: unbiased
        const float rdenom = 1.0f / (sd + 1e-8f);                            // (x - mean) / (std + 1e-8), :428
        K1_STAMP(5);   // DCT + mean + std done
        for (int i = tid; i < nmf; i += THREADS) mf[i] = (mf[i] - mean) * rdenom;
    } else {
        const int c0 = chalf * 7, nc = chalf ? 6 : 7, tt = tt0;
        float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (tt < NFRAMES) {
            const float* drow = &tb->dct_t[c0][0];
            // seven coefficients for both halves (the second half's seventh is the table's zero row): no branch in the
            // loop, so the wave-uniform coefficients arrive as s_load_dwordx8 blocks -- a conditional seventh coefficient
            // made every one of its 64 products wait out a scalar load of its own
#pragma unroll 8
            for (int m = 0; m < NMEL; ++m) {
                const float v = melbuf[m * NFRAMES + tt];
#pragma unroll
                for (int cc = 0; cc < 7; ++cc) acc[cc] = fmaf(drow[cc * NMEL + m], v, acc[cc]);
            }
        }
        K1_MARK("PHASE P2 mean / std / z-score");
        float lsum = 0.f;
#pragma unroll
        for (int cc = 0; cc < 7; ++cc) lsum += (tt < NFRAMES && cc < nc) ? acc[cc] : 0.f;
        const float mean = block_sum(lsum, red, tid) / float(NMF);
        K1_STAMP(5);   // DCT done
        float lsq = 0.f;
#pragma unroll
        for (int cc = 0; cc < 7; ++cc) {
            const float d = acc[cc] - mean;
            lsq += (tt < NFRAMES && cc < nc) ? d * d : 0.f;
        }
        const float sd = sqrtf(block_sum(lsq, red, tid) / float(NMF - 1));   // torch.std: unbiased
        const float rdenom = 1.0f / (sd + 1e-8f);                              // (x - mean) / (std + 1e-8), :428
        if (tt < NFRAMES) {
#pragma unroll
            for (int cc = 0; cc < 7; ++cc)
                if (cc < nc) mf[(c0 + cc) * NFRAMES + tt] = (acc[cc] - mean) * rdenom;
        }
    }
    __syncthreads();
    K1_MARK("PHASE P2 feature image (stem) + MFCC / delta rows");
    // STEM: bf16 feature image(s), row = f + 3, col = t + 3
    uint16_t* img = STEM == 2 ? reinterpret_cast<uint16_t*>(smem + ST_X3_OFF) : reinterpret_cast<uint16_t*>(melbuf);
    uint16_t* img_lo = img + ST_ROWS * ST_PITCH;   // STEM == 2
    auto put = [&](int idx, float v) {
        const uint16_t hi = f2bf(v);
        img[idx] = hi;
        if constexpr (STEM == 2) img_lo[idx] = f2bf(v - __uint_as_float(uint32_t(hi) << 16));
    };
    if constexpr (TALL) {
        static_assert(!TALL || STEM == 2, "the 103-row stem: split-bf16 operands");
        // ---- 103-row image (delta-delta on), stem in two halves.  Half h covers pooled rows [13 h, 13 h + 13) = conv rows
        // [26 h, 26 h + 26) = image rows (feature row + 3) [52 h, 52 h + 57): half 0 is mel rows 0..53 under the top border, half 1
        // mel rows 49..63, the MFCC / delta / delta-delta rows and the bottom border.  LDS: MFCC + delta buffers (10.5 KB) | hi + lo
        // partial images (2 x 12 296 B) | the 15 mel rows half 1 needs again (6 KB), all inside the workgroup's 45.5 KB.
        constexpr int TL_P1H = 26, TL_HALF = 13, TL_PER = TL_HALF * ST_P1W, TL_ROWS = 58, TL_PLANE = TL_ROWS * ST_PITCH;
        constexpr size_t TL_IMG_OFF = (size_t(2) * NMF * 4 + 15) & ~size_t(15), TL_STASH_OFF = TL_IMG_OFF + size_t(2) * TL_PLANE * 2;
        constexpr int TL_KEEP0 = 2 * 2 * TL_HALF - 3;   // 49: first feature row of half 1
        static_assert(TL_STASH_OFF % 16 == 0 && TL_STASH_OFF + size_t(NMEL - TL_KEEP0) * NFRAMES * 4 <= LDS_TOTAL, "103-row stem LDS");
        static_assert(2 * (2 * (TL_HALF - 1) + 1) + 1 + 6 < TL_ROWS && TL_P1H == 2 * TL_HALF, "partial image rows");
        constexpr int NDV = (NMEL * NFRAMES / 2 + THREADS - 1) / THREADS;
        float2 dv[NDV];   // the floored dB values move to registers: the dB buffer becomes image / stash space
#pragma unroll
        for (int it = 0; it < NDV; ++it) {
            const int i2 = tid + it * THREADS;
            dv[it] = i2 < NMEL * NFRAMES / 2 ? reinterpret_cast<const float2*>(melbuf)[i2] : make_float2(0.f, 0.f);
        }
        float* o_mfcc = o + NMEL * NFRAMES;
        float* o_delta = o_mfcc + NMF;
        for (int item = tid; item < NMF; item += THREADS) {
            const int c = item / NFRAMES, t = item - c * NFRAMES;
            const float* row = mf + c * NFRAMES;
            const float d = (row[t < NFRAMES - 1 ? t + 1 : t] - row[t > 0 ? t - 1 : 0]) / 2.0f;   // :353-355
            if (wr) {
                o_mfcc[item] = row[t];
                o_delta[item] = d;
            }
            dl[item] = d;
        }
        __syncthreads();
        auto ddelta = [&](int c, int t) {
            const float* row = dl + c * NFRAMES;
            return (row[t < NFRAMES - 1 ? t + 1 : t] - row[t > 0 ? t - 1 : 0]) / 2.0f;
        };
        if (wr) {
            float* o_dd = o_delta + NMF;
            for (int item = tid; item < NMF; item += THREADS) o_dd[item] = ddelta(item / NFRAMES, item % NFRAMES);
        }
        uint16_t* timg = reinterpret_cast<uint16_t*>(smem + TL_IMG_OFF);
        float* stash = reinterpret_cast<float*>(smem + TL_STASH_OFF);
        auto tput = [&](int idx, float v) {
            const uint16_t hi = f2bf(v);
            timg[idx] = hi;
            timg[TL_PLANE + idx] = f2bf(v - __uint_as_float(uint32_t(hi) << 16));
        };
        auto zero_image = [&]() {
            for (int i = tid; i < 2 * TL_PLANE / 8; i += THREADS) reinterpret_cast<uint4*>(timg)[i] = make_uint4(0, 0, 0, 0);
            for (int i = (2 * TL_PLANE / 8) * 8 + tid; i < 2 * TL_PLANE; i += THREADS) timg[i] = 0;
            __syncthreads();
        };
        zero_image();
        if constexpr (PCS) {   // mel rows = the PCEN values this thread formed for (band tid / 4, quarter tid % 4 of the frames)
            const int m = tid >> 2, t0 = (tid & 3) * 26;
#pragma unroll
            for (int k = 0; k < 26; ++k) {
                if (t0 + k < NFRAMES) {
                    if (m + 3 < TL_ROWS - 1) tput((m + 3) * ST_PITCH + t0 + k + 3, pcv[k]);
                    if (m >= TL_KEEP0) stash[(m - TL_KEEP0) * NFRAMES + t0 + k] = pcv[k];
                }
            }
        } else {
            {
#pragma unroll
                for (int it = 0; it < NDV; ++it) {
                    const int e = 2 * (tid + it * THREADS);
                    if (e < NMEL * NFRAMES) {
                        const int m0 = e / NFRAMES, t0 = e - m0 * NFRAMES;
                        const int m1 = t0 + 1 < NFRAMES ? m0 : m0 + 1, t1 = t0 + 1 < NFRAMES ? t0 + 1 : 0;
                        const float v0 = fminf(fmaxf((dv[it].x + 80.0f) * 0.0125f, 0.f), 1.f);
                        const float v1 = fminf(fmaxf((dv[it].y + 80.0f) * 0.0125f, 0.f), 1.f);
                        if (m0 + 3 < TL_ROWS - 1) tput((m0 + 3) * ST_PITCH + t0 + 3, v0);
                        if (m1 + 3 < TL_ROWS - 1) tput((m1 + 3) * ST_PITCH + t1 + 3, v1);
                        if (m0 >= TL_KEEP0) stash[(m0 - TL_KEEP0) * NFRAMES + t0] = v0;
                        if (m1 >= TL_KEEP0) stash[(m1 - TL_KEEP0) * NFRAMES + t1] = v1;
                    }
                }
            }
        }
        // the stem's weight fragments arrive only now: the 26 registers of floored dB values are dead
        const int sr = lane & 31, sh = lane >> 5;
        bf16x8 bw[2][4];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int st = 0; st < 4; ++st)
                bw[pl][st] = *reinterpret_cast<const bf16x8*>(stem.wfrag + pl * 2048 + ((st * 2 + sh) * 32 + sr) * 8);
        const float bn = stem.bias[sr];
        const int q = sr >> 2, dy = (sr >> 1) & 1, dx = sr & 1;
        float* oa = reinterpret_cast<float*>(stem.a1) + clip * (long long)(TL_P1H * ST_P1W) * 32;
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            if (half == 1) {
                zero_image();
                // image row = feature row + 3 - 52: mel 49..63 -> 0..14, MFCC -> 15.., delta -> 28.., delta-delta -> 41..53
                for (int i = tid; i < (NMEL - TL_KEEP0) * NFRAMES; i += THREADS) {
                    const int m = i / NFRAMES, t = i - m * NFRAMES;
                    tput(m * ST_PITCH + t + 3, stash[i]);
                }
                for (int item = tid; item < NMF; item += THREADS) {
                    const int c = item / NFRAMES, t = item - c * NFRAMES;
                    tput((NMEL - TL_KEEP0 + c) * ST_PITCH + t + 3, mf[item]);
                    tput((NMEL - TL_KEEP0 + NMFCC + c) * ST_PITCH + t + 3, dl[item]);
                    tput((NMEL - TL_KEEP0 + 2 * NMFCC + c) * ST_PITCH + t + 3, ddelta(c, t));
                }
            }
            __syncthreads();
            // stem tiles of this half: 325 pooled positions = 41 tiles of 8; a wave takes two tiles at a time so that one tile's LDS
            // latency and epilogue overlap the other's MFMAs (see the 90-row stem below for the fragment layout)
            auto tl_base = [&](int tile) -> const uint32_t* {
                int P = tile * 8 + q;
                if (P >= TL_PER) P = TL_PER - 1;
                const int ph = P / ST_P1W, pw = P - ph * ST_P1W;
                return reinterpret_cast<const uint32_t*>(timg + (2 * (2 * ph + dy) + sh) * ST_PITCH + 2 * (2 * pw + dx));
            };
            auto tl_load = [&](const uint32_t* base, bf16x8 (&fa)[2][4]) {
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                    for (int st = 0; st < 4; ++st) {
                        const uint32_t* pp = base + pl * (TL_PLANE / 2) + st * ST_PITCH;
                        union { uint32_t u[4]; bf16x8 v; } tv;
                        tv.u[0] = pp[0]; tv.u[1] = pp[1]; tv.u[2] = pp[2]; tv.u[3] = pp[3];
                        fa[pl][st] = tv.v;
                    }
            };
            auto tl_mma = [&](const bf16x8 (&fa)[2][4], f32x16& c) {
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][st], bw[0][st], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][st], bw[0][st], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][st], bw[1][st], c, 0, 0, 0);
                }
            };
            auto tl_store = [&](int tile, const f32x16& acc2) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int Po = tile * 8 + 2 * g + sh;
                    float v = fmaxf(fmaxf(acc2[4 * g], acc2[4 * g + 1]), fmaxf(acc2[4 * g + 2], acc2[4 * g + 3])) + bn;
                    v = fmaxf(v, 0.f);
                    if (Po < TL_PER) oa[(half * TL_PER + Po) * 32 + sr] = v;
                }
            };
            constexpr int TL_TILES = (TL_PER + 7) / 8;
            for (int ta = wave; ta < TL_TILES; ta += 2 * WAVES) {
                const int tb = ta + WAVES;
                const bool two = tb < TL_TILES;   // wave-uniform
                bf16x8 fa[2][4], fb[2][4];
                tl_load(tl_base(ta), fa);
                tl_load(tl_base(two ? tb : ta), fb);
                f32x16 ca = {0}, cb = {0};
                tl_mma(fa, ca);
                tl_mma(fb, cb);
                tl_store(ta, ca);
                if (two) tl_store(tb, cb);
            }
            __syncthreads();   // the next half re-uses the image
        }
        return;
    }
    if constexpr (STEM != 0) {
        // the floored dB values move to registers, then the dB buffer becomes the zero-bordered bf16 image
        float2 dv[(NMEL * NFRAMES / 2 + THREADS - 1) / THREADS];
#pragma unroll
        for (int it = 0; it < (NMEL * NFRAMES / 2 + THREADS - 1) / THREADS; ++it) {
            const int i2 = tid + it * THREADS;
            dv[it] = i2 < NMEL * NFRAMES / 2 ? reinterpret_cast<const float2*>(melbuf)[i2] : make_float2(0.f, 0.f);
        }
        __syncthreads();
        constexpr int NIMG = STEM == 2 ? 2 : 1;     // x3: hi and lo images are adjacent
        for (int i = tid; i < NIMG * ST_ROWS * ST_PITCH / 8; i += THREADS) reinterpret_cast<uint4*>(img)[i] = make_uint4(0, 0, 0, 0);
        for (int i = (NIMG * ST_ROWS * ST_PITCH / 8) * 8 + tid; i < NIMG * ST_ROWS * ST_PITCH; i += THREADS) img[i] = 0;
        __syncthreads();
        if constexpr (PCS) {   // mel rows = the PCEN values this thread formed for (band tid / 4, quarter tid % 4 of the frames)
            const int m = tid >> 2, t0 = (tid & 3) * 26;
#pragma unroll
            for (int k = 0; k < 26; ++k)
                if (t0 + k < NFRAMES) put((m + 3) * ST_PITCH + t0 + k + 3, pcv[k]);
        } else {
#pragma unroll
            for (int it = 0; it < (NMEL * NFRAMES / 2 + THREADS - 1) / THREADS; ++it) {
                const int e = 2 * (tid + it * THREADS);
                if (e < NMEL * NFRAMES) {
                    const int m0 = e / NFRAMES, t0 = e - m0 * NFRAMES;
                    const int m1 = t0 + 1 < NFRAMES ? m0 : m0 + 1, t1 = t0 + 1 < NFRAMES ? t0 + 1 : 0;
                    put((m0 + 3) * ST_PITCH + t0 + 3, fminf(fmaxf((dv[it].x + 80.0f) * 0.0125f, 0.f), 1.f));
                    put((m1 + 3) * ST_PITCH + t1 + 3, fminf(fmaxf((dv[it].y + 80.0f) * 0.0125f, 0.f), 1.f));
                }
            }
        }
    }
    float* o_mfcc = o + nmel * NF;
    float* o_delta = o_mfcc + nmf;
    for (int item = tid; item < nmf; item += THREADS) {
        const int c = item / NF, t = item - c * NF;
        const float* row = mf + c * NF;
        const float d = (row[t < NF - 1 ? t + 1 : t] - row[t > 0 ? t - 1 : 0]) / 2.0f;   // :353-355
        if (wr) {
            o_mfcc[item] = row[t];
            o_delta[item] = d;
        }
        if (delta_delta == 1) dl[item] = d;
        if constexpr (STEM != 0) {
            put((NMEL + c + 3) * ST_PITCH + t + 3, row[t]);
            put((NMEL + NMFCC + c + 3) * ST_PITCH + t + 3, d);
        }
    }
    K1_STAMP(6);
    if (delta_delta == 1 && wr) {
        __syncthreads();
        float* o_dd = o_delta + nmf;
        for (int item = tid; item < nmf; item += THREADS) {
            const int c = item / NF, t = item - c * NF;
            const float* row = dl + c * NF;
            o_dd[item] = (row[t < NF - 1 ? t + 1 : t] - row[t > 0 ? t - 1 : 0]) / 2.0f;
        }
    }
    if constexpr (STEM != 0) {
        // ---- K2 fused: conv7x7 s2 p3 (1->32) + BN + ReLU + maxpool2 on v_mfma_f32_32x32x16_bf16.  K is 8 kernel
        // rows x 8 taps (7 + a zero tap; row 8 all zero): MFMA step st, lane half h <-> kernel row 2*st+h,
        // 8 consecutive image pixels per lane = 4 aligned ds_read_b32.  GEMM rows are (pool window, dy, dx), so
        // the 2x2 max is a max over 4 accumulator registers of one lane.  STEM == 2: three MFMAs per step
        // (image_hi * w_hi + image_lo * w_hi + image_hi * w_lo), f32 output. ---------------------------------
        __syncthreads();
        K1_MARK("PHASE K2 stem MFMA + pool + store");
        constexpr int NP = STEM == 2 ? 2 : 1;   // operand planes
        const int sr = lane & 31, sh = lane >> 5;
        bf16x8 bw[NP][4];
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int st = 0; st < 4; ++st)
                bw[pl][st] = *reinterpret_cast<const bf16x8*>(stem.wfrag + pl * 2048 + ((st * 2 + sh) * 32 + sr) * 8);
        const float bn = stem.bias[sr];
        const int q = sr >> 2, dy = (sr >> 1) & 1, dx = sr & 1;
        using out_t = std::conditional_t<STEM == 2, float, uint16_t>;
        out_t* oa = reinterpret_cast<out_t*>(stem.a1) + clip * (long long)ST_PER * 32;
        auto frag_base = [&](int tile) -> const uint32_t* {
            int P = tile * 8 + q;
            if (P >= ST_PER) P = ST_PER - 1;
            const int ph = P / ST_P1W, pw = P - ph * ST_P1W;
            return reinterpret_cast<const uint32_t*>(img + (2 * (2 * ph + dy) + sh) * ST_PITCH + 2 * (2 * pw + dx));
        };
        auto load_frags = [&](const uint32_t* base, bf16x8 (&a)[NP][4]) {
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const uint32_t* p = base + pl * (ST_ROWS * ST_PITCH / 2) + st * ST_PITCH;   // +2 image rows per step = ST_PITCH dwords
                    union { uint32_t u[4]; bf16x8 v; } t;
                    t.u[0] = p[0]; t.u[1] = p[1]; t.u[2] = p[2]; t.u[3] = p[3];
                    a[pl][st] = t.v;
                }
        };
        auto mma = [&](const bf16x8 (&a)[NP][4], f32x16& c) {
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][st], bw[0][st], c, 0, 0, 0);
                if constexpr (STEM == 2) {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][st], bw[0][st], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][st], bw[1][st], c, 0, 0, 0);
                }
            }
        };
        auto epilogue = [&](int tile, const f32x16& acc2, bool guard) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int Po = tile * 8 + 2 * g + sh;
                float v = fmaxf(fmaxf(acc2[4 * g], acc2[4 * g + 1]), fmaxf(acc2[4 * g + 2], acc2[4 * g + 3])) + bn;
                v = fmaxf(v, 0.f);
                if (!guard || Po < ST_PER) {
                    if constexpr (STEM == 2) oa[Po * 32 + sr] = v;
                    else oa[Po * 32 + sr] = f2bf(v);
                }
            }
        };
        // ST_TILES = 69: every wave owns 17 full tiles (wave, wave+4, ..., wave+64), taken two at a time so one
        // tile's LDS latency and epilogue overlap the other's MFMAs; wave 0 finishes the partial tile 68.
        static_assert(ST_TILES == 17 * WAVES + 1, "stem tile split");
        for (int it = 0; it < 16; it += 2) {
            const int ta = wave + WAVES * it, tbb = ta + WAVES;
            bf16x8 fa[NP][4], fb[NP][4];
            load_frags(frag_base(ta), fa);
            load_frags(frag_base(tbb), fb);
            f32x16 ca = {0}, cb = {0};
            mma(fa, ca);
            mma(fb, cb);
            epilogue(ta, ca, false);
            epilogue(tbb, cb, false);
        }
        {
            const int ta = wave + WAVES * 16;               // 17th full tile
            const bool last = wave == 0;                    // wave 0 also takes the partial tile 68
            bf16x8 fa[NP][4], fb[NP][4];
            load_frags(frag_base(ta), fa);
            load_frags(frag_base(last ? ST_TILES - 1 : ta), fb);
            f32x16 ca = {0}, cb = {0};
            mma(fa, ca);
            mma(fb, cb);
            epilogue(ta, ca, false);
            if (last) epilogue(ST_TILES - 1, cb, true);
        }
        K1_STAMP(7);   // stem done (wave 0)
    }
}

}  // namespace
}  // namespace cough

// ------------------------------------------------------------------------------------------ C-ABI
struct cough_featurizer {
    cough_feat_config cfg;
    cough::FeatTables* d_tables;
    float* d_win_full;   // periodic Hann(n_fft), for cough_spectrogram(COUGH_SPEC_FULL_WINDOW)
    int nfeat;           // rows of the feature image
    int nbase;           // rows the featurise kernel writes (mel [+ MFCC, delta, delta-delta])
    cough::ContrastCfg contrast;   // n_bands == 0: no spectral-contrast rows
    int n_cus;           // compute units of the device the featuriser was created on
    int kind;            // the one-launch kernel that serves the constructor's segment length: 0 none (generic chain), 1 the
                         // shipped sparse-filterbank instantiations, 2 the full-band ones (any filterbank, run-time n_mels / n_mfcc),
                         // 3 the full-band ones with a run-time STFT geometry (n_fft 512, hop <= 256)
    char* d_full;        // kind 2: CSR filterbank + DCT rows (one blob)
    cough::FullBank full;
    size_t full_lds;     // kind 2 / 3: dynamic LDS of a workgroup at the segment length
    bool geo_any;        // the CSR tables exist and the run-time-geometry kernel can serve this configuration: waveforms of OTHER lengths
                         // that fit its limits (geo_fits) take it too, whatever `kind` serves the segment
    cough::GenFeat* gen; // the generic kernel chain's tables (featurize_generic.hip): every geometry the tuned kernel does not
                         // cover, and -- for every featuriser -- waveforms of any other length (extract_features of any N)
};

namespace cough {
namespace {
// What the run-time-geometry instantiations (featurize_kernel<..., GEO>) take, whatever the waveform length ...
bool geo_basic(const cough_feat_config& c) {
    return c.n_fft == NFFT && c.win_length >= 1 && c.win_length <= NFFT && c.hop_length >= 1 && c.n_mels >= 2 && c.n_mels <= 128 &&
           (!c.use_mfcc || (c.n_mfcc >= 1 && c.n_mfcc <= c.n_mels));
}
// ... and for waveforms of n samples.  The frames' 512-sample spans must cover every sample (the fused normalise collects the peak
// from them): always so for hop <= 256, for a hop up to 512 when the last span reaches the end of the waveform.  PCEN: a thread keeps
// its quarter of a band's frames in 52 registers.  The MFCC rows lie in the 16 640-byte transpose scratch (the deltas over the dead
// dB buffer), the dB buffer of n_mels x frames in the LDS of two workgroups per CU.
bool geo_fits(const cough_feat_config& c, int n_taps, int n) {
    if (n <= NFFT / 2) return false;   // reflect padding
    const long long T = n / c.hop_length + 1;
    const bool covered = c.hop_length <= NFFT / 2 || (c.hop_length <= NFFT && (T - 1) * c.hop_length + NFFT / 2 >= n);
    return covered && T <= 1024 && (!c.use_pcen || T <= 208) && (!c.use_mfcc || size_t(c.n_mfcc) * T * 4 <= LDS_XCH_FULL) &&
           full_lds_bytes(c.n_mels, n_taps, int(T)) <= 80 * 1024;
}
}  // namespace
}  // namespace cough

extern "C" int cough_featurizer_create(cough_featurizer** out, const cough_feat_config* cfg,
                                       const float* window, const float* mel_fb, const float* dct) {
    using namespace cough;
    COUGH_REQUIRE(out && cfg && window && mel_fb && dct, COUGH_EINVAL, "cough_featurizer_create: NULL argument");
    // The one-launch kernel serves the shipped STFT geometry (16 kHz, n_fft 512, hop 160, window 400, 1 s).  kind 1: the shipped
    // 64-mel / 13-MFCC layout with a filterbank of <= 8 taps per band below bin 128 (any f_max <= sample_rate / 4) -- taps in
    // registers, half the spectrum formed; kind 2: every other filterbank at that geometry (f_max up to the Nyquist bin, 2..128
    // mel bands, up to 20 MFCCs) -- all 257 bins, CSR filterbank in LDS.  Everything else goes to the generic kernel chain
    // (featurize_generic.hip), whose tables every featuriser carries for waveforms of other lengths.
    const int nfreq0 = NFFT / 2 + 1;
    // (the sample rate only shapes the filterbank, which arrives as a table: 2 s at 8 kHz or 0.5 s at 32 kHz are the same STFT)
    const bool stft_ok = cfg->n_fft == NFFT && cfg->hop_length == HOP && cfg->win_length == WIN &&
                         cfg->segment_samples == NS;
    bool tuned = stft_ok && cfg->n_mels == NMEL && cfg->n_mfcc == NMFCC;
    if (tuned) {
        for (int m = 0; m < NMEL && tuned; ++m) {
            int first = -1, last = -1;
            for (int k = 0; k < nfreq0; ++k)
                if (mel_fb[k * NMEL + m] != 0.f) { if (first < 0) first = k; last = k; }
            if (first >= 0 && (last >= NBIN || last - first >= MAXW)) tuned = false;
        }
    }
    // full-band CSR tables (kind 2)
    std::vector<int> f_lo, f_hi, f_off;
    std::vector<float> f_taps, f_dct;
    FullBank fb{};
    // kind 3: the full-band kernel with a run-time STFT geometry at n_fft = 512 -- other sample rates / hops / windows / segment
    // lengths (geo_basic / geo_fits above), and at the shipped STFT what the fixed-geometry full-band kernels do not take: an odd
    // number of mel bands (they store the mel rows in pairs), more than 20 MFCCs (they keep MFCC and delta rows side by side in the
    // scratch), PCEN with another band count than 64.  Contrast rows come from the generic chain's kernels behind it.
    const int geo_frames = cfg->hop_length > 0 ? cfg->segment_samples / cfg->hop_length + 1 : 0;
    const bool basic = geo_basic(*cfg);
    if (basic) {   // the CSR tables: for kinds 2 and 3, and for waveforms of other lengths on any kind
        const int nm = cfg->n_mels, nc = cfg->use_mfcc ? cfg->n_mfcc : 1;
        fb.n_mels = nm;
        fb.n_mfcc = nc;
        for (int m = 0; m < nm; ++m) {
            int first = -1, last = -1;
            for (int k = 0; k < nfreq0; ++k)
                if (mel_fb[k * nm + m] != 0.f) { if (first < 0) first = k; last = k; }
            if (first < 0) { first = 0; last = -1; }   // empty band: no taps, mel power 0
            // the kernel reads taps and powers as aligned pairs: the band starts at an even bin and has an even number of taps
            // (zero taps as padding; bin 257 of a power row is scratch of the frame's own finite transform, times 0)
            const int lo2 = first & ~1, hi2 = last >= first ? (last + 2) & ~1 : lo2;
            f_lo.push_back(lo2);
            f_hi.push_back(hi2);
            f_off.push_back(hi2 > lo2 ? int(f_taps.size()) : 0);   // an empty band points at a valid pair (weight 0 in the kernel)
            for (int k = lo2; k < hi2; ++k) f_taps.push_back(k >= first && k <= last ? 0.25f * mel_fb[k * nm + m] : 0.f);   // |2X|^2 / 4
            const int w = (hi2 - lo2 + 3) / 4 * 4;   // the kernel walks a band's taps four at a time
            if (w > fb.maxw[m >> 6]) fb.maxw[m >> 6] = w;
        }
        if (f_taps.size() < 2) f_taps.assign(2, 0.f);
        fb.n_taps = int(f_taps.size());
        fb.cph = (nc + 1) / 2;
        const int n_chunks = (fb.cph + 6) / 7;
        fb.cw = (fb.cph + n_chunks - 1) / n_chunks;   // 7 coefficients per half -> one chunk of 7; 10 -> two of 5; 11 -> two of 6
        if (fb.cw < 4) fb.cw = 4;
        fb.cph_pad = (fb.cph + fb.cw - 1) / fb.cw * fb.cw;
        const int chunks_per_half = fb.cph_pad / fb.cw;
        f_dct.assign(size_t(2) * chunks_per_half * nm * 8, 0.f);
        if (cfg->use_mfcc)
            for (int c = 0; c < nc; ++c) {
                const int half = c >= fb.cph, r = half ? c - fb.cph : c;
                for (int m = 0; m < nm; ++m)
                    f_dct[((size_t(half) * chunks_per_half + r / fb.cw) * nm + m) * 8 + r % fb.cw] = dct[m * nc + c];
            }
        fb.n_samples = cfg->segment_samples;
        fb.hop = cfg->hop_length;
    }
    bool fixed_full = !tuned && basic && stft_ok && cfg->n_mels % 2 == 0 && (!cfg->use_mfcc || cfg->n_mfcc <= FULL_MAX_MFCC) &&
                      (!cfg->use_pcen || cfg->n_mels == NMEL) &&
                      full_lds_bytes(cfg->n_mels, fb.n_taps, NFRAMES) <= 80 * 1024;   // at least two workgroups per CU
    if (cfg->use_spectral_contrast) {
        COUGH_REQUIRE(cfg->n_contrast_bands >= 1 && cfg->n_contrast_bands <= COUGH_MAX_CONTRAST_BANDS, COUGH_EUNSUPPORTED,
                      "n_contrast_bands = %d: the HIP path takes 1..%d", cfg->n_contrast_bands, COUGH_MAX_CONTRAST_BANDS);
        for (int i = 0; i <= cfg->n_contrast_bands; ++i) {
            const int lo = cfg->contrast_edges[i], hi = cfg->contrast_edges[i + 1];
            COUGH_REQUIRE(lo >= 0 && lo < cfg->n_fft / 2 + 1 && (i == cfg->n_contrast_bands || hi - lo <= 1024),
                          COUGH_EUNSUPPORTED, "spectral-contrast band %d = bins [%d, %d): the HIP path takes bands of <= 1024 bins "
                          "inside the spectrum", i, lo, hi);
            // the contrast kernel behind the persistent STFT passes (shipped geometry) selects out of bands of <= 128 bins
            if (i < cfg->n_contrast_bands && hi - lo > 128 && stft_ok) tuned = fixed_full = false;
        }
    }
    const bool geo_seg = !tuned && !fixed_full && basic && geo_fits(*cfg, fb.n_taps, cfg->segment_samples);
    fb.n_frames = geo_seg ? geo_frames : NFRAMES;
    std::vector<FeatTables> host(1);
    FeatTables& t = host[0];
    std::memset(&t, 0, sizeof(t));
    const double PI = 3.14159265358979323846;
    if (stft_ok || basic) {   // the STFT tables of the one-launch kernels and of the persistent STFT kernel (spectrogram.hip)
        const int left = (NFFT - cfg->win_length) / 2;   // torch.stft centres a short window in the frame
        for (int n = 0; n < cfg->win_length; ++n) t.win[left + n] = window[n];
        for (int jj = 0; jj < 16; ++jj)
            for (int k1 = 0; k1 < 16; ++k1) {
                const double a = -2.0 * PI * double(jj * k1) / 256.0;
                t.tw256[jj][k1] = make_float2(float(std::cos(a)), float(std::sin(a)));
            }
        for (int k = 0; k < NBIN; ++k) {
            const double a = -2.0 * PI * double(k) / 512.0;
            t.tw512[k] = make_float2(float(std::cos(a)), float(std::sin(a)));
        }
    }
    if (tuned) {
        const int nfreq = NFFT / 2 + 1;
        for (int m = 0; m < NMEL; ++m) {
            int first = -1;
            for (int k = 0; k < nfreq && first < 0; ++k)
                if (mel_fb[k * NMEL + m] != 0.f) first = k;
            if (first < 0) first = 0;   // empty band: all-zero taps
            if (first > NBIN - MAXW) first = NBIN - MAXW;   // keep start+8 inside the power buffer
            t.mel_start[m] = first;
            for (int q = 0; q < MAXW; ++q) t.mel_w[m][q] = 0.25f * mel_fb[(first + q) * NMEL + m];   // |2X|^2 / 4
        }
        for (int c = 0; c < NMFCC; ++c)
            for (int m = 0; m < NMEL; ++m) t.dct_t[c][m] = dct[m * NMFCC + c];
        for (int m = 0; m < NMEL; ++m) t.dct_t[NMFCC][m] = 0.f;
    }
    GenFeat* gen = nullptr;
    if (int e = gen_feat_create(&gen, cfg, window, mel_fb, dct)) return e;

    cough_featurizer* f = new cough_featurizer();
    f->cfg = *cfg;
    f->gen = gen;
    f->kind = tuned ? 1 : fixed_full ? 2 : geo_seg ? 3 : 0;
    f->d_full = nullptr;
    f->full = fb;
    f->full_lds = fixed_full || geo_seg ? full_lds_bytes(fb.n_mels, fb.n_taps, fb.n_frames) : 0;
    f->geo_any = basic;
    f->nbase = cfg->use_mfcc ? cfg->n_mels + 2 * cfg->n_mfcc + (cfg->use_delta_delta ? cfg->n_mfcc : 0) : cfg->n_mels;
    f->nfeat = f->nbase + (cfg->use_spectral_contrast ? cfg->n_contrast_bands + 1 : 0);
    f->contrast.n_bands = cfg->use_spectral_contrast ? cfg->n_contrast_bands : 0;
    for (int i = 0; i < 18; ++i) f->contrast.edges[i] = cfg->contrast_edges[i];
    f->d_tables = nullptr;
    f->d_win_full = nullptr;
    std::vector<float> hann(NFFT);   // torch.hann_window(n_fft, periodic=True)
    for (int n = 0; n < NFFT; ++n) hann[n] = float(0.5 - 0.5 * std::cos(2.0 * PI * double(n) / double(NFFT)));
    hipError_t e = hipMalloc(&f->d_tables, sizeof(FeatTables));
    if (e == hipSuccess) e = hipMemcpy(f->d_tables, &t, sizeof(FeatTables), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&f->d_win_full, NFFT * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(f->d_win_full, hann.data(), NFFT * sizeof(float), hipMemcpyHostToDevice);
    // per-device kernel attributes (the persistent STFT kernel's 162 KB of dynamic LDS) on the creator's device
    if (e == hipSuccess && stft_prepare_device(&f->n_cus) != COUGH_OK) e = hipErrorUnknown;
    if (e == hipSuccess && basic) {
        // one blob: lo | hi | off (ints), taps, DCT rows; every piece 16-byte aligned
        auto al = [](size_t v) { return (v + 15) & ~size_t(15); };
        const size_t nm = size_t(fb.n_mels), o_hi = al(nm * 4), o_off = o_hi + al(nm * 4), o_w = o_off + al(nm * 4),
                     o_dct = o_w + al(f_taps.size() * 4), total = o_dct + al(f_dct.size() * 4);
        std::vector<char> blob(total, 0);
        std::memcpy(blob.data(), f_lo.data(), nm * 4);
        std::memcpy(blob.data() + o_hi, f_hi.data(), nm * 4);
        std::memcpy(blob.data() + o_off, f_off.data(), nm * 4);
        std::memcpy(blob.data() + o_w, f_taps.data(), f_taps.size() * 4);
        std::memcpy(blob.data() + o_dct, f_dct.data(), f_dct.size() * 4);
        e = hipMalloc(&f->d_full, total);
        if (e == hipSuccess) e = hipMemcpy(f->d_full, blob.data(), total, hipMemcpyHostToDevice);
        f->full.lo = reinterpret_cast<const int*>(f->d_full);
        f->full.hi = reinterpret_cast<const int*>(f->d_full + o_hi);
        f->full.off = reinterpret_cast<const int*>(f->d_full + o_off);
        f->full.w = reinterpret_cast<const float*>(f->d_full + o_w);
        f->full.dct = reinterpret_cast<const float*>(f->d_full + o_dct);
        // more than 64 KB of dynamic LDS with many mel bands: per-device attribute, set here (not lazily at launch)
        const void* fns[] = {reinterpret_cast<const void*>(featurize_kernel<false, 0, true>),
                             reinterpret_cast<const void*>(featurize_kernel<true, 0, true>),
                             reinterpret_cast<const void*>(featurize_kernel<false, 2, true, false>),
                             reinterpret_cast<const void*>(featurize_kernel<false, 2, true, true>),
                             reinterpret_cast<const void*>(featurize_kernel<true, 2, true, false>),
                             reinterpret_cast<const void*>(featurize_kernel<true, 2, true, true>),
                             reinterpret_cast<const void*>(featurize_kernel<false, 2, true, false, true>),
                             reinterpret_cast<const void*>(featurize_kernel<false, 2, true, true, true>),
                             reinterpret_cast<const void*>(featurize_kernel<true, 2, true, false, true>),
                             reinterpret_cast<const void*>(featurize_kernel<true, 2, true, true, true>),
                             reinterpret_cast<const void*>(featurize_kernel<false, 0, true, false, false, true>),
                             reinterpret_cast<const void*>(featurize_kernel<true, 0, true, false, false, true>)};
        for (const void* fn : fns)
            if (e == hipSuccess) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    }
    if (e != hipSuccess) {
        set_error("cough_featurizer_create: %s", hipGetErrorString(e));
        if (f->d_full) (void)hipFree(f->d_full);
        if (f->d_tables) (void)hipFree(f->d_tables);
        if (f->d_win_full) (void)hipFree(f->d_win_full);
        gen_feat_destroy(f->gen);
        delete f;
        return COUGH_EHIP;
    }
    *out = f;
    return COUGH_OK;
}

#ifdef COUGH_K1_STAMPS
extern "C" __attribute__((visibility("default"))) int cough_debug_set_stamp_buffer(void* d_buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(cough::g_stamp_buf), &d_buf, sizeof(d_buf)) == hipSuccess ? 0 : 3;
}
#endif

extern "C" void cough_featurizer_destroy(cough_featurizer* f) {
    if (!f) return;
    if (f->d_full) (void)hipFree(f->d_full);
    if (f->d_tables) (void)hipFree(f->d_tables);
    if (f->d_win_full) (void)hipFree(f->d_win_full);
    cough::gen_feat_destroy(f->gen);
    delete f;
}

extern "C" int cough_featurizer_num_features(const cough_featurizer* f) { return f ? f->nfeat : -1; }
extern "C" int cough_featurizer_path(const cough_featurizer* f) { return f ? f->kind : -1; }
extern "C" int cough_featurizer_num_frames(const cough_featurizer* f) { return !f ? -1 : cough::gen_frames(f->gen, 0); }
extern "C" int cough_featurizer_num_frames_for(const cough_featurizer* f, int n_samples) {
    return !f || n_samples < 0 ? -1 : cough::gen_frames(f->gen, n_samples);
}

namespace cough {
StftView featurizer_stft_view(const cough_featurizer* f) {
    const char* base = reinterpret_cast<const char*>(f->d_tables);
    return StftView{reinterpret_cast<const float*>(base + offsetof(FeatTables, win)), f->d_win_full,
                    reinterpret_cast<const float2*>(base + offsetof(FeatTables, tw256)),
                    reinterpret_cast<const float2*>(base + offsetof(FeatTables, tw512)), f->n_cus};
}
int featurizer_num_features(const cough_featurizer* f) { return f->nfeat; }
const GenFeat* featurizer_generic(const cough_featurizer* f) { return f->gen; }
bool featurizer_tuned(const cough_featurizer* f, int n_samples) {
    return f->kind != 0 && (n_samples <= 0 || n_samples == f->cfg.segment_samples);
}
// Which kernels featurise waveforms of n_samples (0: the segment): 1 the handle's own one-launch kernel (kind 1 / 2 / 3 at the segment
// length), 3 the run-time-geometry instantiation at ANOTHER length that fits its limits, 0 the generic kernel chain.
static int featurizer_route(const cough_featurizer* f, int n_samples) {
    if (n_samples <= 0 || n_samples == f->cfg.segment_samples) return f->kind != 0 ? 1 : 0;
    return f->geo_any && geo_fits(f->cfg, f->full.n_taps, n_samples) ? 3 : 0;
}
bool featurizer_shipped_stft(const cough_featurizer* f, int n_samples) {   // the persistent STFT kernel's geometry
    return (f->kind == 1 || f->kind == 2) && (n_samples <= 0 || n_samples == NS);
}
bool featurizer_stem_fusable(const cough_featurizer* f, bool x3) {
    // a one-launch kernel writing the whole image (no contrast rows), 64 mel + 13 MFCC rows
    if (f->kind == 0 || f->kind == 3 || f->cfg.n_mels != NMEL || f->cfg.n_mfcc != NMFCC || !f->cfg.use_mfcc || f->nfeat != f->nbase)
        return false;
    // split-bf16 stem: the 90-row layout, or the 103-row layout of the delta-delta flag (two halves); shipped or full-band
    // filterbank, with or without pre-emphasis / PCEN.  The approximate single-bf16 stem exists for the shipped 90-row set only.
    if (x3) return f->nfeat == ST_H || (f->nfeat == ST_H + NMFCC && f->cfg.use_delta_delta);
    return f->nfeat == ST_H && f->kind == 1 && !f->cfg.use_pre_emphasis && !f->cfg.use_pcen;
}
size_t featurizer_workspace_bytes(const cough_featurizer* f, int n_clips, int n_samples) {
    const int route = featurizer_route(f, n_samples);
    if (route == 0) return gen_workspace_bytes(f->gen, f->cfg, n_samples, n_clips);
    if (f->contrast.n_bands == 0 || n_clips <= 0) return 0;
    // contrast rows: behind the persistent STFT passes (shipped geometry), else by the generic chain's kernels
    return route == 3 || f->kind == 3 ? gen_workspace_bytes(f->gen, f->cfg, n_samples, n_clips) : contrast_workspace_bytes(n_clips);
}

int launch_featurize(const cough_featurizer* f, const float* d_wav, long long wav_stride, float* d_feat, int n_clips,
                     int flags, const StemFuse* stem, hipStream_t stream, void* d_workspace, size_t workspace_bytes,
                     int n_samples) {
    COUGH_REQUIRE(f && d_wav && (d_feat || stem), COUGH_EINVAL, "cough_featurize: NULL argument");
    COUGH_REQUIRE(n_clips >= 0 && n_samples >= 0, COUGH_EINVAL, "cough_featurize: n_clips < 0 or n_samples < 0");
    const int route = featurizer_route(f, n_samples);
    const bool geo_launch = route == 3 || (route == 1 && f->kind == 3);   // the run-time-geometry instantiation runs
    if (route == 0) {
        COUGH_REQUIRE(!stem, COUGH_EUNSUPPORTED, "the fused stem needs the shipped 90-row feature layout");
        if (n_clips == 0) return COUGH_OK;
        return gen_featurize(f->gen, f->cfg, f->contrast, d_wav, wav_stride, n_samples, d_feat, f->nfeat, f->nbase, n_clips,
                             (flags & COUGH_FEAT_NORMALIZE) ? 1 : 0, d_workspace, workspace_bytes, stream);
    }
    const int n_wave = route == 3 ? n_samples : f->cfg.segment_samples;   // samples per clip of this launch
    if (geo_launch)
        COUGH_REQUIRE(wav_stride >= n_wave, COUGH_EINVAL, "cough_featurize: row stride %lld < %d samples", wav_stride, n_wave);
    else
        COUGH_REQUIRE(wav_stride >= NS && (wav_stride & 3) == 0 && (reinterpret_cast<size_t>(d_wav) & 15) == 0,
                      COUGH_EINVAL, "cough_featurize: d_wav must be 16-byte aligned with a row stride >= 16000, multiple of 4");
    COUGH_REQUIRE(!stem || (route == 1 && featurizer_stem_fusable(f, stem->x3 != 0)), COUGH_EUNSUPPORTED,
                  "the fused stem needs the shipped 90-row feature layout");
    if (n_clips == 0) return COUGH_OK;
    const int norm = (flags & COUGH_FEAT_NORMALIZE) ? 1 : 0;
    const int rows = (f->cfg.use_mfcc ? (f->cfg.use_delta_delta ? 1 : 0) : 2);   // kernel row selector
    const dim3 grid(n_clips), block(THREADS);
    const StemFuse none{nullptr, nullptr, nullptr, 0, nullptr};
    const FullBank nofb{};
    // spectral-contrast rows under the fused normalise: the featurise kernel leaves every clip's peak for the contrast path
    float* peak_out = nullptr;
    if (f->contrast.n_bands > 0 && norm && !geo_launch) {
        COUGH_REQUIRE(d_workspace && workspace_bytes >= contrast_workspace_bytes(n_clips), COUGH_EWORKSPACE,
                      "spectral contrast needs a workspace of cough_featurizer_workspace_bytes() bytes (cough_featurize_ws)");
        peak_out = contrast_peaks(d_workspace, n_clips);
    } else if (f->contrast.n_bands > 0 && norm) {   // run-time geometry: the generic chain's layout, peaks of the call at its head
        COUGH_REQUIRE(d_workspace && workspace_bytes >= gen_workspace_bytes(f->gen, f->cfg, n_samples, n_clips) &&
                          (reinterpret_cast<size_t>(d_workspace) & 255) == 0,
                      COUGH_EWORKSPACE, "spectral contrast needs a 256-byte aligned workspace of "
                      "cough_featurizer_workspace_bytes() bytes (cough_featurize_ws)");
        peak_out = static_cast<float*>(d_workspace);
    }
    // one instantiation per (pre-emphasis, stem, full-band filterbank, 103-row stem); everything else is a run-time argument
    const bool full = f->kind >= 2 || geo_launch, pe = f->cfg.use_pre_emphasis != 0;
    FullBank geo_bank = f->full;   // another waveform length: the same tables, its own frame count
    if (route == 3) {
        geo_bank.n_samples = n_samples;
        geo_bank.n_frames = n_samples / f->cfg.hop_length + 1;
    }
    const size_t lds = route == 3 ? full_lds_bytes(geo_bank.n_mels, geo_bank.n_taps, geo_bank.n_frames) : full ? f->full_lds : LDS_TOTAL;
    const FullBank& fbk = route == 3 ? geo_bank : full ? f->full : nofb;
    const float* fdct = full ? f->full.dct : nullptr;
    auto go = [&](auto kernel, const StemFuse& sf, int pcen) {
        hipLaunchKernelGGL(kernel, grid, block, lds, stream, d_wav, wav_stride, d_feat, f->nfeat, f->d_tables, norm,
                           f->cfg.pre_emphasis_coef, rows, pcen, sf, fbk, fdct, peak_out);
    };
    if (stem && stem->x3) {   // split-bf16 stem fused (90-row image, or the 103-row image of the delta-delta flag in two halves)
        const bool tall = f->cfg.use_delta_delta != 0;
        const int sel = (f->cfg.use_pcen ? 8 : 0) | (pe ? 4 : 0) | (full ? 2 : 0) | (tall ? 1 : 0);
        switch (sel) {
#define K1_CASE(N, PE, FU, TA, PC) case N: go(featurize_kernel<PE, 2, FU, TA, PC>, *stem, PC ? 1 : 0); break;
            K1_CASE(0, false, false, false, false) K1_CASE(1, false, false, true, false) K1_CASE(2, false, true, false, false)
            K1_CASE(3, false, true, true, false) K1_CASE(4, true, false, false, false) K1_CASE(5, true, false, true, false)
            K1_CASE(6, true, true, false, false) K1_CASE(7, true, true, true, false) K1_CASE(8, false, false, false, true)
            K1_CASE(9, false, false, true, true) K1_CASE(10, false, true, false, true) K1_CASE(11, false, true, true, true)
            K1_CASE(12, true, false, false, true) K1_CASE(13, true, false, true, true) K1_CASE(14, true, true, false, true)
            default: go(featurize_kernel<true, 2, true, true, true>, *stem, 1); break;
#undef K1_CASE
        }
    } else if (stem) {        // approximate single-bf16 stem: shipped filterbank, no pre-emphasis (featurizer_stem_fusable)
        go(featurize_kernel<false, 1>, *stem, 0);
    } else if (geo_launch) {   // run-time STFT geometry
        if (pe) go(featurize_kernel<true, 0, true, false, false, true>, none, f->cfg.use_pcen);
        else go(featurize_kernel<false, 0, true, false, false, true>, none, f->cfg.use_pcen);
    } else if (full) {
        if (pe) go(featurize_kernel<true, 0, true>, none, f->cfg.use_pcen);
        else go(featurize_kernel<false, 0, true>, none, f->cfg.use_pcen);
    } else {
        if (pe) go(featurize_kernel<true, 0>, none, f->cfg.use_pcen);
        else go(featurize_kernel<false, 0>, none, f->cfg.use_pcen);
    }
    COUGH_HIP_CHECK(hipGetLastError());
    if (f->contrast.n_bands > 0 && geo_launch)   // run-time geometry: the generic chain's STFT + contrast kernels add the rows
        return gen_featurize(f->gen, f->cfg, f->contrast, d_wav, wav_stride, n_samples, d_feat, f->nfeat, f->nbase, n_clips, norm,
                             d_workspace, workspace_bytes, stream, /*contrast_rows_only=*/true);
    if (f->contrast.n_bands > 0)   // rows [nbase, nfeat): from the un-emphasised signal (preprocessing.py:476-478)
        return launch_contrast(featurizer_stft_view(f), f->contrast, d_wav, wav_stride, d_feat, f->nfeat, f->nbase,
                               n_clips, norm, d_workspace, workspace_bytes, stream);
    return COUGH_OK;
}
}  // namespace cough

extern "C" size_t cough_featurizer_workspace_bytes(const cough_featurizer* f, int n_clips) {
    return f ? cough::featurizer_workspace_bytes(f, n_clips) : 0;
}
extern "C" size_t cough_featurizer_workspace_bytes_for(const cough_featurizer* f, int n_samples, int n_clips) {
    return f && n_samples >= 0 ? cough::featurizer_workspace_bytes(f, n_clips, n_samples) : 0;
}
extern "C" int cough_featurize_any(const cough_featurizer* f, const float* d_wav, long long wav_stride, int n_samples,
                                   float* d_feat, int n_clips, int flags, void* d_workspace, size_t workspace_bytes, void* stream) {
    COUGH_REQUIRE(d_feat, COUGH_EINVAL, "cough_featurize_any: NULL argument");
    return cough::launch_featurize(f, d_wav, wav_stride, d_feat, n_clips, flags, nullptr, static_cast<hipStream_t>(stream),
                                   d_workspace, workspace_bytes, n_samples);
}

extern "C" int cough_featurize_ws(const cough_featurizer* f, const float* d_wav, long long wav_stride, float* d_feat,
                                  int n_clips, int flags, void* d_workspace, size_t workspace_bytes, void* stream) {
    COUGH_REQUIRE(d_feat, COUGH_EINVAL, "cough_featurize_ws: NULL argument");
    return cough::launch_featurize(f, d_wav, wav_stride, d_feat, n_clips, flags, nullptr,
                                   static_cast<hipStream_t>(stream), d_workspace, workspace_bytes);
}

extern "C" int cough_featurize(const cough_featurizer* f, const float* d_wav, long long wav_stride,
                               float* d_feat, int n_clips, int flags, void* stream) {
    COUGH_REQUIRE(d_feat, COUGH_EINVAL, "cough_featurize: NULL argument");
    return cough::launch_featurize(f, d_wav, wav_stride, d_feat, n_clips, flags, nullptr, static_cast<hipStream_t>(stream));
}
