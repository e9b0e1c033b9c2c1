// Stand-alone STFT for gfx950: waveform -> |X|^2 (or |X|) spectrogram, (n, 257, 101) float32.
//
// Replaces the T.Spectrogram(n_fft=512, win_length=400, hop_length=160, power=2) transform of
// /root/reference/src/preprocessing.py:131-136 (torch.stft(center=True, pad_mode="reflect", onesided=True),
// periodic Hann(400) zero-padded to 512) and, with COUGH_SPEC_FULL_WINDOW | COUGH_SPEC_MAGNITUDE, the magnitude
// spectrogram T.SpectralCentroid computes internally with its default Hann(n_fft) window (:137-141).
//
// The fused featuriser (featurize.hip) never materialises the spectrogram; this file is the same FFT with all 257 bins
// formed and stored.  Algorithmic bytes per clip: 64 000 read + 257*101*4 = 103 828 written.
//
// The kernel that runs is stft3_kernel (below): one persistent 12-wave workgroup per CU owns a clip's whole spectrogram,
// stages it in LDS as the linear image it is in memory and writes it with aligned 16-byte stores; samples arrive by
// LDS-DMA.  0.169 ms for 4096 clips = 51 % of 8 TB/s, PMC traffic 1.001x (profiles/r03_stft_experiments.txt).
//
//
// A wave transforms 4 frames at a time (16 lanes each): lane j of a frame ends up with Z[j+16*k2]; with its partner's
// Z[256-k] it forms both X[k] (k = j+16*k2 < 128) and X[256-k] -- conj(E - W^k O) -- so every bin 0..256 comes out of
// the same 8 butterflies.  (The round-1/2 kernel -- one 4-wave workgroup per (clip, frame chunk), sibling chunks' row
// fragments meeting in L2, 0.2445 ms = 35 % -- is in the git history; profiles/r03_stft_experiments.txt has the A/B.)
#include "common.h"
#include "contrast_rank.h"
#include "fft256.h"
#include "internal.h"

namespace cough {
namespace {

constexpr int NS = 16000, NFFT = 512, HOP = 160, NFRAMES = 101, NFREQ = 257;
constexpr int PADL = NFFT / 2;
constexpr int FPW = 4;
constexpr int NGROUP = (NFRAMES + FPW - 1) / FPW;           // 26 groups of 4 frames
constexpr int XROW = 17;
constexpr size_t LDS_TW = size_t(16) * XROW * 8;               // 2176: [16][XROW] float2 twiddles

#ifdef COUGH_K1_STAMPS
// Diagnostic build only (tools/stft_stamps.py): s_memtime of wave 0 and wave 3 at phase boundaries, into a buffer of
// its own that no other code reads.  Never compiled into libcough_amd.so.
__device__ unsigned long long* g_stft_stamp_buf = nullptr;
#define STFT_STAMP(slot)                                                                              \
    do {                                                                                              \
        if (g_stft_stamp_buf && (threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) == 0 || (threadIdx.x >> 6) == 3)) \
            g_stft_stamp_buf[(size_t(blockIdx.x) * 2 + (threadIdx.x >> 7)) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define STFT_STAMP(slot) do {} while (0)
#endif

// ---------------------------------------------------------------------------------------------------------------
// stft3_kernel: one PERSISTENT 12-wave workgroup per CU, one clip at a time, whole spectrogram staged in LDS.
//
// Why: the (bin, time) output is time-minor with 404-byte rows, so a workgroup that owns a range of frames writes
// 60-110-byte row fragments at 4-byte alignment and relies on sibling workgroups' fragments meeting in L2 (stft_kernel
// above).  Here a workgroup owns the clip's WHOLE 257 x 101 spectrogram -- one contiguous 103 828-byte range of the
// output -- builds it in LDS as the linear image it is in memory, and writes it with 16-byte-per-lane stores at
// 16-byte-aligned addresses: every 128-byte line leaves the CU whole, once, in one burst.
//   * 26 four-frame groups over 12 waves: two each, waves 0 and 1 (on different SIMDs) a third -- r05: three waves per SIMD and
//     the worst SIMD at 7 group-rounds measured 1.5 % faster than 13 waves (SIMD 0 hosting four, 8 group-rounds; r03 / r04),
//     same-box A/B in profiles/r05_stft_waves_ab.txt; COUGH_STFT_WAVES selects.  The clip's image (103.8 KB) + 12 transpose scratches of
//     4 160 B + twiddle and window tables = 157 984 B of the CU's 160 KiB (163 840 B).
//   * A wave's 992 samples per group come straight from HBM into its transpose scratch by `global_load_lds_dwordx4`
//     (no data VGPRs; reflected edges: dword DMAs with per-lane reflected source addresses), and the NEXT group's
//     samples (the next clip's, in round 1) are requested as soon as the second transpose has been read back, so
//     their latency runs under the second radix-16, the real-input split, the barriers and the flush.
//   * The transpose scratch interleaves the wave's four frames: element (row k, frame f, column n) at 65 k + 16 f + n.
//     Writes (lane = (f, j), row k1) are 64 consecutive floats, reads (row j, column n2) hit 32 distinct banks per
//     half-wave, every address is a per-lane base + an immediate, and it takes 1040 floats instead of 4 x 272.
//   * Raw `s_barrier`s behind `lgkmcnt(0)` (a `__syncthreads()` would drain the DMA); the samples are awaited with a
//     counted `vmcnt` that leaves the flush stores issued after the DMA in flight.
#ifndef COUGH_STFT_WAVES
#define COUGH_STFT_WAVES 12
#endif
#ifndef COUGH_STFT_WIN_REGS
#define COUGH_STFT_WIN_REGS 0
#endif
constexpr int W3 = COUGH_STFT_WAVES, THREADS3 = W3 * 64;                // 768 threads
constexpr int ROUNDS3 = (NGROUP + W3 - 1) / W3;                         // four-frame groups of a clip per wave, at most
constexpr int GSPAN = (FPW - 1) * HOP + NFFT;                           // 992 samples feed one four-frame group
constexpr int X3ROW = FPW * 16 + 1, X3WAVE = 16 * X3ROW;                // transpose scratch: [16 rows][4 frames x 16 + 1 pad]
constexpr int IMG = NFREQ * NFRAMES;                                    // 25 957 floats
constexpr int IMG_PIECES = (IMG + 3 + 3) / 4;                           // 16-byte pieces incl. up to 3 floats of lead-in
constexpr size_t LDS3_IMG = size_t(IMG_PIECES) * 16;                    // 103 840
constexpr size_t LDS3_XCH = size_t(W3) * X3WAVE * 4;                    // 49 920
constexpr size_t LDS3_WIN = size_t(NFFT) * 4;                           // 2 048
constexpr size_t LDS3_TOTAL = LDS3_IMG + LDS3_XCH + LDS_TW + LDS3_WIN;  // 157 984
static_assert(LDS3_TOTAL <= 160 * 1024, "one workgroup owns the CU's LDS");
static_assert(W3 * ROUNDS3 >= NGROUP && W3 <= 16, "every group of a clip has a wave");
static_assert(GSPAN <= X3WAVE && GSPAN % 4 == 0, "a group's samples land in the wave's transpose scratch");
static_assert(LDS3_IMG % 16 == 0 && LDS3_XCH % 16 == 0 && LDS_TW % 16 == 0, "16-byte aligned regions");

typedef __attribute__((address_space(3))) void* lptr_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));

// LDS-DMA by inline asm: the destination is M0 (wave-uniform LDS byte address) + lane * size.  hipcc does not count an
// asm statement's memory operation, so it adds no `vmcnt(0)` of its own in front of the LDS reads -- the kernel waits
// with a COUNTED `vmcnt` that leaves the flush stores issued after the DMA in flight (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void glds16(const float* gsrc, unsigned lds_dst) {
    unsigned keep;
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);   // wave-uniform by construction; pins it to an SGPR
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds4(const float* gsrc, unsigned lds_dst) {
    unsigned keep;
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void lds_barrier() {   // LDS hand-off between waves; leaves vector-memory ops in flight
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#ifdef COUGH_K1_STAMPS
#define STFT3_STAMP(slot) do { if (iter_no == 5) STFT_STAMP(slot); } while (0)
#else
#define STFT3_STAMP(slot) do {} while (0)
#endif

// CENT: centroid mode (the spectral-contrast path): no image, no flush -- every frame reduces sum(k 31.25 Hz |X[k]|) / sum |X[k]| /
// 8000 over its 16 lanes and `out` is [n_clips][101] (torchaudio.functional.spectral_centroid / (sample_rate / 2),
// /root/reference/src/preprocessing.py:295-298, with FULLWIN + MAG).  VMW: flush-store instructions every wave is guaranteed to
// issue behind its DMA (the counted wait below); 7 for the whole image, less when only the first `rows_out` bins are flushed
// (the contrast path needs the bins below its last band edge; VMW = 7 always flushes all 257).  PEAK: `peaks` [n_clips] holds every
// clip's max |sample| (left by the featurise kernel under the fused normalise, else nullptr): a clip whose peak lies outside
// 2^+-50 is transformed with its samples scaled by a power of two (exact), so that its power neither underflows nor overflows
// before contrast_kernel applies 1 / peak^2 -- the reference divides the samples by the peak first (:199-212).
// 2^k that brings a peak outside 2^-50 .. 2^50 to about 2^+-40 (exact in float32 as a factor), 1 for ordinary, zero or
// non-finite peaks
__device__ __forceinline__ float extreme_peak_scale(float m) {
    const int pe = (__float_as_int(m) >> 23) & 0xff;
    if ((pe >= 127 - 50 && pe <= 127 + 50) || m == 0.f || pe == 255) return 1.0f;
    const int e = pe ? pe - 127 : -127 - __builtin_clz(__float_as_int(m) << 9);   // floor(log2(m))
    int k = (e > 0 ? 40 : -40) - e;
    k = k > 126 ? 126 : k;   // one finite factor: a denormal peak below 2^-166 + 40 cannot occur (2^-149 is the smallest)
    return __int_as_float((127 + k) << 23);
}

template <bool FULLWIN, bool MAG, bool CENT = false, int VMW = 7, bool PEAK = false>
__global__ __launch_bounds__(THREADS3) void stft3_kernel(const float* __restrict__ wav, long long wav_stride,
                                                         float* __restrict__ out, const float* __restrict__ win,
                                                         const float2* __restrict__ tw256,
                                                         const float2* __restrict__ tw512, int n_clips, int rows_out,
                                                         const float* __restrict__ peaks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* img0 = reinterpret_cast<float*>(smem);                                       // the clip's image (+ lead-in)
    float* xs = reinterpret_cast<float*>(smem + LDS3_IMG);                              // [W3][X3WAVE]
    float2* twl = reinterpret_cast<float2*>(smem + LDS3_IMG + LDS3_XCH);                // [16][XROW]
    float* winl = reinterpret_cast<float*>(smem + LDS3_IMG + LDS3_XCH + LDS_TW);        // [512]
    constexpr int N0 = FULLWIN ? 0 : 1, N1 = FULLWIN ? 16 : 15;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, fsub = lane >> 4;
    float* myw = xs + wave * X3WAVE;              // this wave's scratch: DMA landing zone, then the two transposes
    float* xw = myw + 16 * fsub + j;              // transpose write base: row k1 at xw[k1 * X3ROW]
    const float* xr = myw + X3ROW * j + 16 * fsub;   // transpose read base: column n2 at xr[n2]
    const float2* tw_row = twl + j * XROW;
    const unsigned myw_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)myw);

    // request the 992 samples of group g of the clip at xc: scratch[i] = x[reflect(640 g - 256 + i)]
    auto dma_group = [&](const float* xc, int g) {
        const int s0 = FPW * HOP * g - PADL;
        int ln = lane;
        asm volatile("" : "+v"(ln));   // the per-lane source offsets are loop-invariant across clips: keep hipcc from
                                       // hoisting ~40 of them out of the persistent loop into VGPRs (it spilled 100+)
        if (s0 >= 0 && s0 + GSPAN <= NS) {   // wave-uniform: groups 1..23
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int i = it * 64 + ln;   // 16-byte piece
                if (i < GSPAN / 4) glds16(xc + s0 + 4 * i, myw_lds + it * 1024);
            }
        } else {   // groups 0, 24, 25: reflect padding of torch.stft(center=True)
            // the in-range part of the span still moves as 16-byte pieces (span start and clip length are multiples
            // of 4 samples); only the reflected samples need per-lane source addresses, one dword each.  Group 25
            // holds frame 100 alone: its first 512 samples are enough
            const int lo = s0 < 0 ? -s0 : 0, hi = NS - s0 < GSPAN ? NS - s0 : GSPAN;   // direct samples: [lo, hi)
            const int need = g == NGROUP - 1 ? NFFT : GSPAN;
#pragma unroll 1
            for (int p0 = lo / 4; p0 < (hi < need ? hi : need) / 4; p0 += 64) {
                const int pc = p0 + ln;
                if (4 * pc < hi) glds16(xc + s0 + 4 * pc, myw_lds + p0 * 16);
            }
#pragma unroll 1
            for (int i0 = 0; i0 < lo; i0 += 64) glds4(xc - (s0 + i0 + ln), myw_lds + i0 * 4);   // lo is a multiple of 64
#pragma unroll 1
            for (int i0 = hi; i0 < need; i0 += 64) {                                             // so is hi
                const int i = i0 + ln;
                if (i < need) glds4(xc + 2 * (NS - 1) - (s0 + i), myw_lds + i0 * 4);
            }
        }
    };

    long long clip = blockIdx.x;
    if (clip >= n_clips) return;
    dma_group(wav + clip * wav_stride, wave);
    const float2 tw_j = tw512[j];
    if (tid < 256) twl[(tid >> 4) * XROW + (tid & 15)] = tw256[tid];
    if (tid < 256) reinterpret_cast<float2*>(winl)[tid] = reinterpret_cast<const float2*>(win)[tid];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the first group's samples too (hipcc does not see the asm DMA)
    __syncthreads();
#ifdef COUGH_K1_STAMPS
    int iter_no = 0;
#endif

    // window taps from LDS; COUGH_STFT_WIN_REGS keeps them in registers (three waves per SIMD leave 168 VGPRs): measured no faster
    constexpr bool WIN_REGS = COUGH_STFT_WIN_REGS != 0;
    [[maybe_unused]] float2 wreg[16];
    if constexpr (WIN_REGS) {
#pragma unroll
        for (int n1 = N0; n1 < N1; ++n1) wreg[n1] = *reinterpret_cast<const float2*>(winl + 2 * j + 32 * n1);
    }
    bool first = true;
    while (true) {
        STFT3_STAMP(0);
        const long long clip_n = clip + gridDim.x;
        // the image sits `lead` floats into its region so that LDS and global addresses agree modulo 16 bytes
        float* oc = out + clip * (long long)(CENT ? NFRAMES : IMG);
        const int lead = CENT ? 0 : int((reinterpret_cast<size_t>(oc) >> 2) & 3);
        float* img = img0 + lead;
        [[maybe_unused]] float cs = 1.0f;   // PEAK: power of two this clip's samples are scaled by (1 unless its peak is extreme)
        if constexpr (PEAK) {
            if (peaks != nullptr) cs = extreme_peak_scale(peaks[clip]);
        }
#pragma unroll 1
        for (int rd = 0; rd < ROUNDS3; ++rd) {
            const int g = wave + W3 * rd;
            if (g >= NGROUP) break;   // wave-uniform (12 waves: waves 0 and 1 take a third group)
            // this group's samples are in the scratch.  Round 0: everything but the flush stores issued after the DMA
            // (at least 7 per wave) has completed; round 1: nothing was issued after its DMA
            // The 7: every wave issues >= 7 flush stores behind the DMA -- the flush below hands thread t pieces t,
            // t + THREADS3, ... of the IMG_PIECES - 2 interior pieces, so the wave with the fewest gets
            // (IMG_PIECES - 2) / THREADS3 of them.  A build without the stores has nothing behind the DMA: vmcnt(0).
            static_assert((IMG_PIECES - 2) / THREADS3 >= 7, "round 0 waits with vmcnt(7): every wave must issue >= 7 flush stores after its DMA");
#ifndef COUGH_STFT_NO_STORE
            // CENT: the one vector-memory operation behind this group's DMA is the previous round's centroid store (every group
            // holds at least one valid frame, so that store always issues): waiting for it as well would expose its latency
            if constexpr (CENT) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else if (rd == 0 && VMW > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            if (rd == 0) STFT3_STAMP(1);
            float2 a[16], z[16];
            if constexpr (!FULLWIN) {
                a[0] = make_float2(0.f, 0.f);
                a[15] = make_float2(0.f, 0.f);
            }
            const float* sp = myw + HOP * fsub + 2 * j;
            [[maybe_unused]] const float* wp = winl + 2 * j;
#pragma unroll
            for (int n1 = N0; n1 < N1; ++n1) {
                const float2 r = *reinterpret_cast<const float2*>(sp + 32 * n1);
                const float2 w = WIN_REGS ? wreg[n1] : *reinterpret_cast<const float2*>(wp + 32 * n1);
                if constexpr (PEAK) a[n1] = make_float2((r.x * cs) * w.x, (r.y * cs) * w.y);
                else a[n1] = make_float2(r.x * w.x, r.y * w.y);
            }
            wave_lds_fence();
            dft16(a);
#pragma unroll
            for (int k1 = 1; k1 < 16; ++k1) a[k1] = cmul(a[k1], tw_row[k1]);
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) xw[k1 * X3ROW] = a[k1].x;
            wave_lds_fence();
#pragma unroll
            for (int n2 = 0; n2 < 16; ++n2) z[n2].x = xr[n2];
            wave_lds_fence();
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) xw[k1 * X3ROW] = a[k1].y;
            wave_lds_fence();
#pragma unroll
            for (int n2 = 0; n2 < 16; ++n2) z[n2].y = xr[n2];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the scratch is dead: every read of it has returned
            if (rd == 0) STFT3_STAMP(2);
            if (g + W3 < NGROUP) dma_group(wav + clip * wav_stride, g + W3);
            else if (clip_n < n_clips) dma_group(wav + clip_n * wav_stride, wave);
            if (rd == 0) STFT3_STAMP(3);
            dft16(z);   // z[k2] = Z[j + 16*k2]
            float2 rv[8];   // z[8 + r] of lane (16 - j) & 15 of the same frame: row_mirror, then rotate right by one
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                rv[r].x = dpp_mov<0x121>(dpp_mov<0x140>(z[8 + r].x));
                rv[r].y = dpp_mov<0x121>(dpp_mov<0x140>(z[8 + r].y));
            }
            float* col = img + FPW * g + fsub;
            auto put = [&](int bin, float pwr4) {   // pwr4 = |2X|^2
                col[bin * NFRAMES] = MAG ? 0.5f * sqrtf(pwr4) : 0.25f * pwr4;
            };
            // the previous clip's flush has read the image: that barrier sits HERE, not behind the flush, so a wave
            // that finished flushing early is already through the window / first radix-16 / transposes of this clip
            if (!CENT && rd == 0 && !first) lds_barrier();
            if constexpr (CENT) {
                // centroid of this frame: lane j holds bins j + 16 k2, 256 - (j + 16 k2) (and lane 0 bin 128); freqs =
                // torch.linspace(0, 8000, 257) = 31.25 k exactly
                float num = 0.f, den = 0.f;
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) {
                    const float2 zk = z[k2];
                    const float2 zp0 = (k2 == 0) ? z[0] : rv[8 - k2];
                    const float2 zp = (j == 0) ? zp0 : rv[7 - k2];
                    const float ex = zk.x + zp.x, ey = zk.y - zp.y;
                    const float ox = zk.y + zp.y, oy = zp.x - zk.x;
                    const float qx = W32C[k2] * ox - W32S[k2] * oy, qy = W32C[k2] * oy + W32S[k2] * ox;
                    const float px = tw_j.x * qx - tw_j.y * qy, py = tw_j.x * qy + tw_j.y * qx;
                    const float ar = ex + px, ai = ey + py, br = ex - px, bi = ey - py;
                    const int k = j + 16 * k2;
                    const float ma = 0.5f * sqrtf(ar * ar + ai * ai), mb = 0.5f * sqrtf(br * br + bi * bi);
                    num += (31.25f * float(k)) * ma + (31.25f * float(NFFT / 2 - k)) * mb;
                    den += ma + mb;
                }
                if (j == 0) {
                    const float m128 = sqrtf(z[8].x * z[8].x + z[8].y * z[8].y);   // |X[128]| = |Z[128]|
                    num += 4000.0f * m128;
                    den += m128;
                }
                // totals of the frame's 16 lanes (a DPP row)
                num += dpp_mov<0xB1>(num); den += dpp_mov<0xB1>(den);
                num += dpp_mov<0x4E>(num); den += dpp_mov<0x4E>(den);
                num += dpp_mov<0x141>(num); den += dpp_mov<0x141>(den);
                num += dpp_mov<0x140>(num); den += dpp_mov<0x140>(den);
                if (j == 0 && FPW * g + fsub < NFRAMES) oc[FPW * g + fsub] = (num / den) / 8000.0f;   // / (sample_rate / 2), :297
            } else
            if (FPW * g + fsub < NFRAMES) {   // idle sub-frames of the last group store nothing
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) {
                    const float2 zk = z[k2];
                    const float2 zp0 = (k2 == 0) ? z[0] : rv[8 - k2];   // j == 0
                    const float2 zp = (j == 0) ? zp0 : rv[7 - k2];
                    // 2E = Zk + conj Zp, 2O = -i (Zk - conj Zp); 2X[k] = 2E + W^k 2O, 2X[256-k] = conj(2E - W^k 2O)
                    const float ex = zk.x + zp.x, ey = zk.y - zp.y;
                    const float ox = zk.y + zp.y, oy = zp.x - zk.x;
                    const float qx = W32C[k2] * ox - W32S[k2] * oy, qy = W32C[k2] * oy + W32S[k2] * ox;
                    const float px = tw_j.x * qx - tw_j.y * qy, py = tw_j.x * qy + tw_j.y * qx;
                    const float ar = ex + px, ai = ey + py, br = ex - px, bi = ey - py;
                    const int k = j + 16 * k2;
                    put(k, ar * ar + ai * ai);
                    put(NFFT / 2 - k, br * br + bi * bi);
                }
                if (j == 0) put(128, 4.0f * (z[8].x * z[8].x + z[8].y * z[8].y));   // X[128] = conj Z[128]
            }
            if (rd == 0) STFT3_STAMP(4);
        }
        STFT3_STAMP(5);
        if constexpr (!CENT) lds_barrier();

        STFT3_STAMP(6);
#ifndef COUGH_STFT_NO_STORE
        if constexpr (!CENT) {
            // flush: the image is the clip's output range verbatim; piece p = LDS floats [4p, 4p + 4) = image elements
            // [4p - lead, 4p - lead + 4) -> one 16-byte store at a 16-byte-aligned address.  The first and last piece
            // may hold elements of the neighbouring clips' ranges: those two go out float by float.
            // rows_out < 257: only bins [0, rows_out) leave the CU (the contrast path reads the bins below its last band edge)
            const int n_out = (VMW == 7 ? NFREQ : rows_out) * NFRAMES;
            const f32x4* src = reinterpret_cast<const f32x4*>(img0);
            f32x4* dst = reinterpret_cast<f32x4*>(oc - lead);
            const int last = (n_out + lead + 3) / 4 - 1;
#pragma unroll
            for (int it = 0; it < (IMG_PIECES + THREADS3 - 1) / THREADS3; ++it) {
                const int p = tid + THREADS3 * it;
                if (p > 0 && p < last) dst[p] = src[p];
            }
            if (tid < 8) {   // element e of the first (tid < 4) or last piece
                const int e = (tid < 4 ? 0 : 4 * last) + (tid & 3) - lead;
                if (e >= 0 && e < n_out) oc[e] = img[e];
            }
        }
#endif
        first = false;
        STFT3_STAMP(7);
#ifdef COUGH_K1_STAMPS
        ++iter_no;
#endif
        if (clip_n >= n_clips) break;
        clip = clip_n;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Spectral contrast + centroid rows (AudioPreprocessor.extract_spectral_contrast, preprocessing.py:242-303), out of
// the two spectrograms the STFT kernel above leaves in a workspace.  One 128-thread workgroup per clip, thread =
// frame.  Per band: its rows go to LDS, then each frame ranks its bins (rank = number of smaller values, ties by
// index -- the position torch.sort would give) and sums the ranks >= top_idx and < bot_idx: the means of the
// reference's sorted slices without sorting.  An empty top slice (one-bin band) divides 0 by 0, as the reference's
// mean of an empty tensor does, and the z-score over all rows then turns every row into NaN, as in the reference.
constexpr int CT_THREADS = 128, CT_MAX_BINS = 128;
static_assert(CT_THREADS >= NFRAMES, "thread = frame");

__device__ __forceinline__ float ct_block_sum(float v, float* red, int tid) {
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return red[0] + red[1];
}

// One 128-thread workgroup per clip, thread = frame: band contrasts out of the power rows the STFT kernel left in the workspace
// (read once, straight from L2: consecutive frames are consecutive addresses), the centroid row from the centroid-mode STFT
// pass, the joint z-score (:300), rows stored behind the MFCC block.
__global__ __launch_bounds__(CT_THREADS) void contrast_kernel(const float* __restrict__ power, const float* __restrict__ centroid,
                                                             const float* __restrict__ peaks, float* __restrict__ feat,
                                                             int nfeat, int row0, ContrastCfg cfg, int normalize) {
    __shared__ float cr[(COUGH_MAX_CONTRAST_BANDS + 1) * NFRAMES];
    __shared__ float red[2];
    const int tid = threadIdx.x;
    const long long clip = blockIdx.x;
    const float* P = power + clip * (long long)NFREQ * NFRAMES;
    const int nb_rows = cfg.n_bands + 1;
    // normalize() (preprocessing.py:199-212) scales the waveform by 1/peak, i.e. the power by 1/peak^2; the centroid is a ratio
    // and does not change.  A NaN peak cannot occur (v_max drops it): such a clip is NaN through its powers.
    float scale = 1.0f;
    if (normalize) {
        const float m = peaks[clip];
        if (m > 0.f) {   // the power rows are those of samples * extreme_peak_scale(m): divide by (m * that)^2
            const float inv = 1.0f / (m * extreme_peak_scale(m));
            scale = inv * inv;
        }
    }
    for (int i = 0; i < cfg.n_bands; ++i) {   // workgroup-uniform control flow
        int low = cfg.edges[i], high = cfg.edges[i + 1];   // :272-278
        if (high <= low) high = low + 1;
        if (high > NFREQ) high = NFREQ;
        const int nb = high - low;
        if (tid < NFRAMES) {
            float pk, valleys, chk = 0.f;
            contrast_select(P + low * NFRAMES + tid, NFRAMES, nb, scale, pk, valleys, chk);
            cr[i * NFRAMES + tid] = (log1pf(pk) - log1pf(valleys)) + chk;
        }
    }
    if (tid < NFRAMES) cr[cfg.n_bands * NFRAMES + tid] = centroid[clip * NFRAMES + tid];
    __syncthreads();
    // (contrast - mean) / (std + 1e-8) over all rows, std unbiased (:300)
    const int total = nb_rows * NFRAMES;
    float ls = 0.f;
    for (int i = tid; i < total; i += CT_THREADS) ls += cr[i];
    const float mean = ct_block_sum(ls, red, tid) / float(total);
    float lq = 0.f;
    for (int i = tid; i < total; i += CT_THREADS) {
        const float d = cr[i] - mean;
        lq += d * d;
    }
    const float sd = sqrtf(ct_block_sum(lq, red, tid) / float(total - 1));
    const float rden = 1.0f / (sd + 1e-8f);
    float* o = feat + (clip * nfeat + row0) * (long long)NFRAMES;
    for (int i = tid; i < total; i += CT_THREADS) o[i] = (cr[i] - mean) * rden;
}

constexpr int CT_SUB_BATCH = 2048;   // clips per workspace fill: 2048 x 103 828 B = 213 MB of power rows (Infinity-Cache sized)

struct CtCarve {
    size_t sub, o_cent, o_peak, total;
};
CtCarve ct_carve(int n_clips) {
    CtCarve c;
    c.sub = size_t(n_clips < CT_SUB_BATCH ? n_clips : CT_SUB_BATCH);
    auto al = [](size_t v) { return (v + 255) & ~size_t(255); };
    c.o_cent = al(c.sub * NFREQ * NFRAMES * sizeof(float));
    c.o_peak = c.o_cent + al(c.sub * NFRAMES * sizeof(float));
    c.total = c.o_peak + al(size_t(n_clips) * sizeof(float));   // peaks: every clip of the call (the featurise kernel writes them)
    return c;
}

}  // namespace

size_t contrast_workspace_bytes(int n_clips) { return ct_carve(n_clips).total; }
float* contrast_peaks(void* d_workspace, int n_clips) {
    return reinterpret_cast<float*>(static_cast<char*>(d_workspace) + ct_carve(n_clips).o_peak);
}

int launch_contrast(const StftView& v, const ContrastCfg& cfg, const float* d_wav, long long wav_stride, float* d_feat,
                    int nfeat, int row0, int n_clips, int normalize, void* d_workspace, size_t workspace_bytes,
                    hipStream_t stream) {
    const CtCarve c = ct_carve(n_clips);
    COUGH_REQUIRE(d_workspace && workspace_bytes >= c.total, COUGH_EWORKSPACE,
                  "spectral contrast needs a workspace of cough_featurizer_workspace_bytes() bytes (cough_featurize_ws)");
    COUGH_REQUIRE((reinterpret_cast<size_t>(d_workspace) & 255) == 0, COUGH_EINVAL, "workspace must be 256-byte aligned");
    float* pw = static_cast<float*>(d_workspace);
    float* cent = reinterpret_cast<float*>(static_cast<char*>(d_workspace) + c.o_cent);
    float* peaks = reinterpret_cast<float*>(static_cast<char*>(d_workspace) + c.o_peak);
    int widest = 1, rows_out = 1;   // rows_out: bins below the last band's upper edge -- the only power rows anybody reads
    for (int i = 0; i < cfg.n_bands; ++i) {
        int lo = cfg.edges[i], hi = cfg.edges[i + 1];
        if (hi <= lo) hi = lo + 1;
        if (hi > NFREQ) hi = NFREQ;
        if (hi - lo > widest) widest = hi - lo;
        if (hi > rows_out) rows_out = hi;
    }
    COUGH_REQUIRE(widest <= CT_MAX_BINS, COUGH_EUNSUPPORTED, "spectral-contrast band of %d bins (<= %d)", widest, CT_MAX_BINS);
    // flush-store instructions per wave the power pass is sure to issue (its counted wait, stft3_kernel VMW)
    const int pieces = (rows_out * NFRAMES + 3 + 3) / 4, sure = (pieces - 2) / THREADS3;
    for (int c0 = 0; c0 < n_clips; c0 += int(c.sub)) {
        const int nc = n_clips - c0 < int(c.sub) ? n_clips - c0 : int(c.sub);
        const float* w = d_wav + (long long)c0 * wav_stride;
        const dim3 grid3(nc < v.n_cus ? nc : v.n_cus), block3(THREADS3);
        const float* pk = normalize ? peaks + c0 : nullptr;   // left by the featurise kernel (launch_featurize)
#define COUGH_STFT_POWER(V) hipLaunchKernelGGL((stft3_kernel<false, false, false, V, true>), grid3, block3, LDS3_TOTAL, stream, w, \
                                               wav_stride, pw, v.win, v.tw256, v.tw512, nc, rows_out, pk)
        if (sure >= 7) COUGH_STFT_POWER(7);
        else if (sure >= 3) COUGH_STFT_POWER(3);
        else if (sure >= 2) COUGH_STFT_POWER(2);
        else if (sure >= 1) COUGH_STFT_POWER(1);
        else COUGH_STFT_POWER(0);
#undef COUGH_STFT_POWER
        // the centroid is a ratio of magnitudes, scale-invariant -- but sqrt(re^2 + im^2) is not safe from under- / overflow:
        // the same power-of-two prescale of extreme clips
        hipLaunchKernelGGL((stft3_kernel<true, true, true, 7, true>), grid3, block3, LDS3_TOTAL, stream, w, wav_stride, cent,
                           v.win_full, v.tw256, v.tw512, nc, 0, pk);
        hipLaunchKernelGGL(contrast_kernel, dim3(nc), dim3(CT_THREADS), 0, stream, pw, cent, pk,
                           d_feat + (long long)c0 * nfeat * NFRAMES, nfeat, row0, cfg, normalize);
        COUGH_HIP_CHECK(hipGetLastError());
    }
    return COUGH_OK;
}

int launch_stft(const StftView& v, const float* d_wav, long long wav_stride, float* d_spec, int n_clips, int flags,
                hipStream_t stream) {
    const bool full = flags & COUGH_SPEC_FULL_WINDOW, mag = flags & COUGH_SPEC_MAGNITUDE;
    const float* win = full ? v.win_full : v.win;
    const dim3 grid3(n_clips < v.n_cus ? n_clips : v.n_cus), block3(THREADS3);   // one persistent workgroup per CU
    // the > 64 KB dynamic-LDS attribute of the four instantiations is set per device by stft_prepare_device (called from
    // cough_featurizer_create on the featuriser's device), not lazily here: the attribute is per device and a launch
    // path must not carry process-wide mutable state
    if (full && mag) hipLaunchKernelGGL((stft3_kernel<true, true>), grid3, block3, LDS3_TOTAL, stream, d_wav, wav_stride, d_spec, win, v.tw256, v.tw512, n_clips, NFREQ, static_cast<const float*>(nullptr));
    else if (full) hipLaunchKernelGGL((stft3_kernel<true, false>), grid3, block3, LDS3_TOTAL, stream, d_wav, wav_stride, d_spec, win, v.tw256, v.tw512, n_clips, NFREQ, static_cast<const float*>(nullptr));
    else if (mag) hipLaunchKernelGGL((stft3_kernel<false, true>), grid3, block3, LDS3_TOTAL, stream, d_wav, wav_stride, d_spec, win, v.tw256, v.tw512, n_clips, NFREQ, static_cast<const float*>(nullptr));
    else hipLaunchKernelGGL((stft3_kernel<false, false>), grid3, block3, LDS3_TOTAL, stream, d_wav, wav_stride, d_spec, win, v.tw256, v.tw512, n_clips, NFREQ, static_cast<const float*>(nullptr));
    COUGH_HIP_CHECK(hipGetLastError());
    return COUGH_OK;
}

int stft_prepare_device(int* n_cus) {
    int dev = 0, cus = 0;
    COUGH_HIP_CHECK(hipGetDevice(&dev));
    COUGH_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    *n_cus = cus > 0 ? cus : 256;
    const void* fns[] = {reinterpret_cast<const void*>(stft3_kernel<false, false>), reinterpret_cast<const void*>(stft3_kernel<false, true>),
                         reinterpret_cast<const void*>(stft3_kernel<true, false>), reinterpret_cast<const void*>(stft3_kernel<true, true>),
                         reinterpret_cast<const void*>(stft3_kernel<false, false, false, 7, true>),
                         reinterpret_cast<const void*>(stft3_kernel<false, false, false, 3, true>),
                         reinterpret_cast<const void*>(stft3_kernel<false, false, false, 2, true>),
                         reinterpret_cast<const void*>(stft3_kernel<false, false, false, 1, true>),
                         reinterpret_cast<const void*>(stft3_kernel<false, false, false, 0, true>),
                         reinterpret_cast<const void*>(stft3_kernel<true, true, true, 7, true>)};
    for (const void* fn : fns)
        COUGH_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS3_TOTAL));
    return COUGH_OK;
}

}  // namespace cough

#ifdef COUGH_K1_STAMPS
extern "C" __attribute__((visibility("default"))) int cough_debug_set_stft_stamp_buffer(void* d_buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(cough::g_stft_stamp_buf), &d_buf, sizeof(d_buf)) == hipSuccess ? 0 : 3;
}
#endif

extern "C" int cough_spectrogram_any(const cough_featurizer* f, const float* d_wav, long long wav_stride, int n_samples,
                                     float* d_spec, int n_clips, int flags, void* stream) {
    using namespace cough;
    COUGH_REQUIRE(f && d_wav && d_spec, COUGH_EINVAL, "cough_spectrogram: NULL argument");
    COUGH_REQUIRE(n_clips >= 0 && n_samples >= 0, COUGH_EINVAL, "cough_spectrogram: n_clips < 0 or n_samples < 0");
    COUGH_REQUIRE((flags & ~(COUGH_SPEC_MAGNITUDE | COUGH_SPEC_FULL_WINDOW)) == 0, COUGH_EINVAL,
                  "cough_spectrogram: unknown flag bits 0x%x", flags);
    if (!featurizer_shipped_stft(f, n_samples)) {   // a geometry / a waveform length the persistent kernel is not built for
        if (n_clips == 0) return COUGH_OK;
        return gen_spectrogram(featurizer_generic(f), d_wav, wav_stride, n_samples, d_spec, n_clips, flags,
                               static_cast<hipStream_t>(stream));
    }
    COUGH_REQUIRE(wav_stride >= NS && (wav_stride & 3) == 0 && (reinterpret_cast<size_t>(d_wav) & 15) == 0,
                  COUGH_EINVAL, "cough_spectrogram: d_wav must be 16-byte aligned with a row stride >= 16000, multiple of 4");
    if (n_clips == 0) return COUGH_OK;
    return launch_stft(featurizer_stft_view(f), d_wav, wav_stride, d_spec, n_clips, flags,
                       static_cast<hipStream_t>(stream));
}

extern "C" int cough_spectrogram(const cough_featurizer* f, const float* d_wav, long long wav_stride, float* d_spec,
                                 int n_clips, int flags, void* stream) {
    return cough_spectrogram_any(f, d_wav, wav_stride, 0, d_spec, n_clips, flags, stream);
}
