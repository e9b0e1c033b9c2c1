// Sequential conv-block classifiers for gfx950 (eval mode): CoughDetector ("standard") and CoughDetectorSmall.
//
// Replaces /root/reference/src/model.py:11-40 (ConvBlock), :43-141 (CoughDetector.forward / predict) and
// :144-207 (CoughDetectorSmall.forward / predict).  Both networks are a stack of
//     y = maxpool2?( ReLU( BN( conv(x) ) ) )        conv = 3x3 pad 1, or depthwise 3x3 pad 1 -> pointwise 1x1
// followed by a global mean and Linear -> ReLU -> Linear.  At create time BatchNorm (running stats) is folded into
// the conv, and a depthwise/pointwise pair -- there is nothing between the two convolutions -- is composed into
// the dense 3x3 convolution it equals, W[n][c][tap] = Wpw[n][c] * Wdw[c][tap], b[n] = bpw[n] + sum_c Wpw[n][c]*bdw[c]
// (in double): 9x the separable MACs, but a real dense contraction for the matrix cores instead of a
// bandwidth-bound elementwise pass plus a skinny GEMM.  Dropout / Dropout2d are the identity in eval mode.
//
//   first block (Cin = 1)  cnn_first_kernel: direct f32 conv out of the (H, W) feature image, bias, ReLU, 2x2 max,
//                          NHWC store; weights are wave-uniform (scalar loads)
//   other blocks           bf16 with >= 32 input / 64 output channels: conv_gemm_bf16_kernel<NT, POOL> (conv_gemm.h,
//                          LDS-staged 128 x N tiles); otherwise cnn_conv_kernel<T, NT, POOL>: implicit GEMM on
//                          v_mfma_f32_32x32x2_f32 (T = float) or v_mfma_f32_32x32x16_bf16 with operands straight
//                          from global / L2, NHWC activations.  With POOL the GEMM rows are ordered
//                          (pool window, dy, dx), so the 2x2 max is a max over 4 accumulator registers of one lane
//                          and the un-pooled conv output never exists (floor mode: the odd last row / column is
//                          never computed)
//   head                   cnn_tail_kernel: mean over H x W -> Linear -> ReLU -> Linear -> softmax / argmax
#include <cmath>
#include <utility>
#include <vector>

#include "common.h"
#include "internal.h"
#include "nn_common.h"
#include "conv_gemm.h"

namespace cough {
namespace {

// ------------------------------------------------------------------------------------------ first block
// thread = one (pooled) output pixel, all N channels; the 4x4 (3x3) input patch sits in registers
template <typename T, bool POOL>
__global__ __launch_bounds__(256) void cnn_first_kernel(const float* __restrict__ feat, int H, int W, int OH, int OW,
                                                        long long n_out, const float* __restrict__ wk /* [9][N] */,
                                                        const float* __restrict__ bias, int N, T* __restrict__ out) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_out) return;
    const int per = OH * OW;
    const long long b = idx / per;
    const int rem = int(idx - b * per), oh = rem / OW, ow = rem - oh * OW;
    constexpr int P = POOL ? 4 : 3, S = POOL ? 2 : 1;
    const float* src = feat + b * (long long)H * W;
    float v[P][P];
#pragma unroll
    for (int i = 0; i < P; ++i)
#pragma unroll
        for (int jx = 0; jx < P; ++jx) {
            const int ih = S * oh - 1 + i, iw = S * ow - 1 + jx;
            v[i][jx] = (ih >= 0 && ih < H && iw >= 0 && iw < W) ? src[ih * W + iw] : 0.f;
        }
    T* o = out + idx * N;
    for (int n0 = 0; n0 < N; n0 += 8) {   // wk / bias addresses are wave-uniform: scalar loads
        float res[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int n = n0 + u;
            float w9[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) w9[k] = wk[k * N + n];
            float best = -INFINITY;
#pragma unroll
            for (int dy = 0; dy < S; ++dy)
#pragma unroll
                for (int dx = 0; dx < S; ++dx) {
                    float acc = 0.f;
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) acc = fmaf(v[dy + kh][dx + kw], w9[kh * 3 + kw], acc);
                    best = fmaxf(best, acc);
                }
            res[u] = fmaxf(best + bias[n], 0.f);
        }
        if constexpr (sizeof(T) == 4) {   // 8 consecutive channels of one pixel: 32 / 16 contiguous bytes per lane
            *reinterpret_cast<float4*>(o + n0) = make_float4(res[0], res[1], res[2], res[3]);
            *reinterpret_cast<float4*>(o + n0 + 4) = make_float4(res[4], res[5], res[6], res[7]);
        } else {
            uint4 pk;
            pk.x = uint32_t(f2bf(res[0])) | (uint32_t(f2bf(res[1])) << 16);
            pk.y = uint32_t(f2bf(res[2])) | (uint32_t(f2bf(res[3])) << 16);
            pk.z = uint32_t(f2bf(res[4])) | (uint32_t(f2bf(res[5])) << 16);
            pk.w = uint32_t(f2bf(res[6])) | (uint32_t(f2bf(res[7])) << 16);
            *reinterpret_cast<uint4*>(o + n0) = pk;
        }
    }
}

// ------------------------------------------------------------------------------------------ first block, split-bf16
// The first ConvBlock (1 input channel, 3x3, pad 1, BN, ReLU, 2x2 max) on the matrix cores with the stem's image trick
// (resnet.hip: stem_bf16_kernel<true>): one workgroup per clip stages the (H, W) feature image ONCE as two bf16 images
// (hi, lo) with a zero border in LDS; the 9 taps fill ONE 16-wide k-step -- k = 4 kh + kw', kw' = 0..3 (the 4th tap and
// the 4th kernel row carry zero weights), so a lane's fragment is 4 consecutive image pixels of two rows: three aligned
// ds_read_b32 + two v_alignbyte per row.  GEMM rows are ordered (pool window, dy, dx): the 2x2 max is a max over 4
// accumulator registers.  Three MFMAs per tile (hi*hi + lo*hi + hi*lo), f32 NHWC output.  r03's f32 VALU kernel
// (cnn_first_kernel) spent 0.27-0.54 ms per 4096 clips here; it stays for the other dtypes / shapes.
constexpr int CNN_FIRST_MAXL = 22;   // float2 loads per thread: H * W <= 2 * 256 * 22 = 11 264 (the 110 x 101 image fits)
struct FirstLds { int nrows, pitch; size_t bytes; };
inline FirstLds first_lds(int H, int W) {
    FirstLds l;
    l.nrows = H + 3;                       // rows -1 .. H + 1 (the row below the last conv row is read with zero weights)
    l.pitch = (W + 7) & ~1;                // cols -1 .. W + 4, even (dword-aligned pairs)
    l.bytes = size_t(l.nrows) * l.pitch * 2;
    return l;
}

__global__ __launch_bounds__(256) void cnn_first_x3_kernel(const float* __restrict__ feat, int H, int W, int OH, int OW,
                                                           int nrows, int pitch, const bf16_t* __restrict__ wfrag /* [2][64][8] */,
                                                           const float* __restrict__ bias, int N, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem_f[];
    bf16_t* img = reinterpret_cast<bf16_t*>(smem_f);
    const int plane = nrows * pitch;   // even
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const long long clip = blockIdx.x;
    const float* src = feat + clip * (long long)H * W;
    const int npairs = (H * W + 1) / 2;
    const bool even = ((H * W) & 1) == 0;
    float2 pv[CNN_FIRST_MAXL];
#pragma unroll
    for (int u = 0; u < CNN_FIRST_MAXL; ++u) {   // all loads in flight before the first conversion
        const int p = tid + u * 256;
        pv[u] = make_float2(0.f, 0.f);
        if (p < npairs) {
            if (even) pv[u] = reinterpret_cast<const float2*>(src)[p];
            else { pv[u].x = src[2 * p]; if (2 * p + 1 < H * W) pv[u].y = src[2 * p + 1]; }
        }
    }
    for (int i = tid; i < 2 * plane / 8; i += 256) reinterpret_cast<uint4*>(img)[i] = make_uint4(0, 0, 0, 0);
    for (int i = (2 * plane / 8) * 8 + tid; i < 2 * plane; i += 256) img[i] = 0;
    __syncthreads();
    auto put = [&](int idx, float v) {
        const bf16_t hi = f2bf(v);
        img[idx] = hi;
        img[plane + idx] = f2bf(v - bf2f(hi));
    };
    const float inv_w = 1.0f / float(W);
#pragma unroll
    for (int u = 0; u < CNN_FIRST_MAXL; ++u) {
        const int e = 2 * (tid + u * 256);
        if (e < H * W) {
            int ih = __float2int_rz(__int2float_rn(e) * inv_w);
            ih -= (ih * W > e);
            ih += ((ih + 1) * W <= e);
            const int iw = e - ih * W;
            put((ih + 1) * pitch + iw + 1, pv[u].x);
            if (e + 1 < H * W) {
                const int ih1 = iw + 1 < W ? ih : ih + 1, iw1 = iw + 1 < W ? iw + 1 : 0;
                put((ih1 + 1) * pitch + iw1 + 1, pv[u].y);
            }
        }
    }
    const bf16x8 bw_hi = *reinterpret_cast<const bf16x8*>(wfrag + lane * 8);
    const bf16x8 bw_lo = *reinterpret_cast<const bf16x8*>(wfrag + 512 + lane * 8);
    const float bn = r < N ? bias[r] : 0.f;
    __syncthreads();

    const int n_win = OH * OW, n_tiles = (n_win + 7) / 8;
    const int q = r >> 2, dy = (r >> 1) & 1, dx = r & 1;
    const float inv_ow = 1.0f / float(OW);
    float* o = out + clip * (long long)n_win * N;
    for (int tile = wave; tile < n_tiles; tile += 4) {
        int win = tile * 8 + q;
        if (win >= n_win) win = n_win - 1;
        int py = __float2int_rz(__int2float_rn(win) * inv_ow);
        py -= (py * OW > win);
        py += ((py + 1) * OW <= win);
        const int px = win - py * OW;
        const int y = 2 * py + dy, x = 2 * px + dx;             // conv pixel; bordered image: row y + kh, col x + kw'
        const int rowA = y + 2 * h, rowB = h ? rowA : rowA + 1;  // h = 0: kernel rows 0, 1; h = 1: kernel row 2 (+ a zero-weight row)
        const int cb = x & ~1;
        const unsigned sh = 2u * unsigned(x & 1);
        const uint32_t* pa = reinterpret_cast<const uint32_t*>(img + rowA * pitch + cb);
        const uint32_t* pb = reinterpret_cast<const uint32_t*>(img + rowB * pitch + cb);
        union { uint32_t u[4]; bf16x8 v; } ah, al;
        {
            const uint32_t a0 = pa[0], a1 = pa[1], a2 = pa[2], b0 = pb[0], b1 = pb[1], b2 = pb[2];
            ah.u[0] = __builtin_amdgcn_alignbyte(a1, a0, sh); ah.u[1] = __builtin_amdgcn_alignbyte(a2, a1, sh);
            ah.u[2] = __builtin_amdgcn_alignbyte(b1, b0, sh); ah.u[3] = __builtin_amdgcn_alignbyte(b2, b1, sh);
        }
        {
            const uint32_t* qa = pa + plane / 2;
            const uint32_t* qb = pb + plane / 2;
            const uint32_t a0 = qa[0], a1 = qa[1], a2 = qa[2], b0 = qb[0], b1 = qb[1], b2 = qb[2];
            al.u[0] = __builtin_amdgcn_alignbyte(a1, a0, sh); al.u[1] = __builtin_amdgcn_alignbyte(a2, a1, sh);
            al.u[2] = __builtin_amdgcn_alignbyte(b1, b0, sh); al.u[3] = __builtin_amdgcn_alignbyte(b2, b1, sh);
        }
        f32x16 acc = {0};
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, bw_hi, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al.v, bw_hi, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, bw_lo, acc, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int wo = tile * 8 + 2 * g + h;
            const float v = fmaxf(fmaxf(acc[4 * g], acc[4 * g + 1]), fmaxf(acc[4 * g + 2], acc[4 * g + 3]));
            if (wo < n_win && r < N) o[(long long)wo * N + r] = fmaxf(v + bn, 0.f);
        }
    }
}

// ------------------------------------------------------------------------------------------ dense blocks
template <typename T>
struct CnnConvArgs {
    const T* in;      // NHWC [B][H][W][C]
    int H, W, C, KS;  // KS = 1 or 3 (pad (KS-1)/2, stride 1)
    const T* wp;      // [N][KS*KS*C], k = (kh*KS + kw)*C + c
    const float* bias;
    T* out;           // NHWC [B][OH][OW][N]
    int OH, OW, N;
    long long M;      // GEMM rows: B*OH*OW (x4 with POOL)
};

// K chunking (A and B use the same channel <-> k-slot map):
//   f32 : 8 channels per chunk, four v_mfma_f32_32x32x2_f32; k-slot h of step e <-> channel 4h+e
//   bf16: 16 channels per chunk, one v_mfma_f32_32x32x16_bf16; lane half h holds channels 8h..8h+7
template <typename T, int NT, bool POOL>
__global__ __launch_bounds__(256) void cnn_conv_kernel(CnnConvArgs<T> a) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const long long m0 = ((long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 32;
    if (m0 >= a.M) return;
    const int n_base = blockIdx.y * (NT * 32);
    const long long m = m0 + r;
    const bool rowok = m < a.M;
    const long long mc = rowok ? m : 0;
    const int per = a.OH * a.OW;
    int b, oh, ow;   // conv output pixel of this GEMM row
    if constexpr (POOL) {
        const long long pix = mc >> 2;
        const int q = int(mc & 3);
        b = int(pix / per);
        const int rem = int(pix - (long long)b * per), ph = rem / a.OW, pw = rem - ph * a.OW;
        oh = 2 * ph + (q >> 1);
        ow = 2 * pw + (q & 1);
    } else {
        b = int(mc / per);
        const int rem = int(mc - (long long)b * per);
        oh = rem / a.OW;
        ow = rem - oh * a.OW;
    }
    const int pad = (a.KS - 1) >> 1, Ktot = a.KS * a.KS * a.C;

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x16{0};
    const T* wrow[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wrow[nt] = a.wp + (long long)(n_base + nt * 32 + r) * Ktot;

    int kbase = 0;
    for (int kh = 0; kh < a.KS; ++kh)
        for (int kw = 0; kw < a.KS; ++kw) {
            const int ih = oh - pad + kh, iw = ow - pad + kw;
            const bool ok = rowok && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
            const T* ap = a.in + (((long long)b * a.H + (ok ? ih : 0)) * a.W + (ok ? iw : 0)) * a.C;
            if constexpr (sizeof(T) == 4) {
                for (int c0 = 0; c0 < a.C; c0 += 8) {
                    float4 av = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (ok) av = *reinterpret_cast<const float4*>(ap + c0 + 4 * h);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const float4 bv = *reinterpret_cast<const float4*>(wrow[nt] + kbase + c0 + 4 * h);
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[nt], 0, 0, 0);
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[nt], 0, 0, 0);
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[nt], 0, 0, 0);
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[nt], 0, 0, 0);
                    }
                }
            } else {
                for (int c0 = 0; c0 < a.C; c0 += 16) {
                    bf16x8 av = {0};
                    if (ok) av = *reinterpret_cast<const bf16x8*>(ap + c0 + 8 * h);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const bf16x8 bv = *reinterpret_cast<const bf16x8*>(wrow[nt] + kbase + c0 + 8 * h);
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[nt], 0, 0, 0);
                    }
                }
            }
            kbase += a.C;
        }
    // epilogue: accumulator register `reg` of lane (r, h) is GEMM row (reg & 3) + 8 * (reg >> 2) + 4 * h, column r
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = n_base + nt * 32 + r;
        const float bn = a.bias[n];
        if constexpr (POOL) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {   // rows 8g + 4h .. +3 = the four positions of pool window 2g + h
                const long long pix = (m0 >> 2) + 2 * g + h;
                const float v = fmaxf(fmaxf(acc[nt][4 * g], acc[nt][4 * g + 1]), fmaxf(acc[nt][4 * g + 2], acc[nt][4 * g + 3]));
                if (pix * 4 < a.M) a.out[pix * a.N + n] = from_f32<T>(fmaxf(v + bn, 0.f));
            }
        } else {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const long long mo = m0 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                if (mo < a.M) a.out[mo * a.N + n] = from_f32<T>(fmaxf(acc[nt][reg] + bn, 0.f));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ dense blocks from an LDS image
// bf16, CIN in {16, 32, 64}: one workgroup = one band of conv-output rows of one clip.  The band's input rows (+ one
// row / column of zero border each side) are staged ONCE into an XOR-swizzled LDS image, so every input pixel leaves
// L2 once instead of nine times (the im2col GEMM above re-gathers it per tap, and with N = 64 that traffic, not
// the MFMAs, sets its speed); all 9 * CIN / 16 k-steps then run out of LDS.  Weights arrive as MFMA fragments
// [k-step][n-tile][64 lanes][8] straight from L2 (every wave of every workgroup walks the same 36-74 KB stream).
// GEMM rows are ordered (pool window, dy, dx) as in cnn_conv_kernel; wave w owns tiles w*MW .. w*MW + MW - 1.
constexpr int CNN_LDS_UNP = 12;   // 16-byte pieces of the LDS image per thread (image <= 48 KB)
struct CnnLdsArgs {
    const bf16_t* in;     // NHWC [B][H][W][CIN]
    const bf16_t* wf;     // [9 * CIN / 16][NT][64][8]
    const float* bias;
    bf16_t* out;          // NHWC [B][OH][OW][32 * NT]
    int H, W, OH, OW;     // OH x OW: output size (pooled when POOL)
    int band_rows;        // output rows per workgroup
    int n_bands;          // workgroups per clip
};

template <int CIN, int NT, int MW, bool POOL>
__global__ __launch_bounds__(256) void cnn_conv_lds_kernel(CnnLdsArgs a) {
    constexpr int CH = CIN / 8, KSTEPS = 9 * CIN / 16, N = 32 * NT, S = POOL ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t* img = reinterpret_cast<bf16_t*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int clip = blockIdx.x / a.n_bands, band = blockIdx.x - clip * a.n_bands;
    const int o0 = band * a.band_rows;                                   // first output row of the band
    const int orows = a.OH - o0 < a.band_rows ? a.OH - o0 : a.band_rows; // output rows in this band
    const int crow0 = S * o0, crows = S * orows;                         // conv rows [crow0, crow0 + crows)
    const int Wb = a.W + 2, irows = crows + 2;                           // image: conv rows - 1 .. + crows, cols -1 .. W

    // ---- stage (zero outside the input): every 16-byte piece is requested before the first one is written to
    // LDS, so the band pays one global-memory latency, not one per piece (host: <= UNP * 256 pieces) ----
    {
        const bf16_t* src = a.in + (long long)clip * a.H * a.W * CIN;
        const int total = irows * Wb * CH;
        const float inv_wb = 1.0f / float(Wb);
        uint4 v[CNN_LDS_UNP];
        int dst[CNN_LDS_UNP];
#pragma unroll
        for (int u = 0; u < CNN_LDS_UNP; ++u) {
            const int i = tid + u * 256;
            const int j = i & (CH - 1), P = i / CH;
            int rr = int(float(P) * inv_wb);
            rr -= (rr * Wb > P);
            rr += ((rr + 1) * Wb <= P);
            const int cc = P - rr * Wb;
            const int iy = crow0 - 1 + rr, ix = cc - 1;
            v[u] = make_uint4(0, 0, 0, 0);
            dst[u] = i < total ? swz_off<CIN>(P, j) : -1;
            if (i < total && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
                v[u] = *reinterpret_cast<const uint4*>(src + ((long long)iy * a.W + ix) * CIN + 8 * j);
        }
#pragma unroll
        for (int u = 0; u < CNN_LDS_UNP; ++u)
            if (dst[u] >= 0) *reinterpret_cast<uint4*>(img + dst[u]) = v[u];
    }
    __syncthreads();

    const int M = orows * a.OW * (POOL ? 4 : 1);   // GEMM rows of the band
    if (wave * MW * 32 >= M) return;               // whole wave beyond the band (no barrier follows)
    int pix[MW];                                   // LDS pixel of tap (0, 0) of this lane's row in each tile
#pragma unroll
    for (int mt = 0; mt < MW; ++mt) {
        int m = (wave * MW + mt) * 32 + r;
        if (m >= M) m = M - 1;
        int y, x;
        if constexpr (POOL) {
            const int win = m >> 2, q = m & 3, py = win / a.OW, px = win - py * a.OW;
            y = 2 * py + (q >> 1);
            x = 2 * px + (q & 1);
        } else {
            y = m / a.OW;
            x = m - y * a.OW;
        }
        pix[mt] = y * Wb + x;                      // image row y = conv row crow0 + y - 1 + kh, column x - 1 + kw
    }
    f32x16 acc[MW][NT];
#pragma unroll
    for (int mt = 0; mt < MW; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x16{0};

    // Software pipeline over the fully unrolled k-steps (compile-time taps): weight fragments run D k-steps ahead in
    // a register ring, activation fragments one k-step ahead; each (ds_read, MFMA group) pair is pinned with
    // scheduling barriers -- left alone hipcc issues every load right before its use and each k-step pays the
    // full L2 / LDS latency.
    const bf16_t* wl = a.wf + lane * 8;
    constexpr int D = (NT <= 2) ? 6 : 3;
    bf16x8 wring[D][NT], af[2][MW];
    auto wload = [&](int s, bf16x8 (&dst)[NT]) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) dst[nt] = *reinterpret_cast<const bf16x8*>(wl + (size_t(s) * NT + nt) * 512);
    };
    auto aload = [&](auto sc, int mt) -> bf16x8 {
        constexpr int s = decltype(sc)::value, tap = s / (CIN / 16), c16 = s % (CIN / 16), kh = tap / 3, kw = tap % 3;
        const int P = pix[mt] + kh * Wb + kw;
        return *reinterpret_cast<const bf16x8*>(img + swz_off<CIN>(P, 2 * c16 + h));
    };
#pragma unroll
    for (int i = 0; i < D; ++i) wload(i, wring[i]);
#pragma unroll
    for (int mt = 0; mt < MW; ++mt) af[0][mt] = aload(std::integral_constant<int, 0>{}, mt);
    auto step = [&]<int s>() {
        bf16x8 bw[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bw[nt] = wring[s % D][nt];
#pragma unroll
        for (int mt = 0; mt < MW; ++mt) {
            if constexpr (s + 1 < KSTEPS) af[(s + 1) & 1][mt] = aload(std::integral_constant<int, s + 1>{}, mt);
            if constexpr (s + D < KSTEPS) {
                if (mt == 0) wload(s + D, wring[s % D]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s & 1][mt], bw[nt], acc[mt][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    [&]<int... Ss>(std::integer_sequence<int, Ss...>) {
        (step.template operator()<Ss>(), ...);
    }(std::make_integer_sequence<int, KSTEPS>{});
    static_assert(KSTEPS == 9 * (CIN / 16), "k-steps");

    // epilogue: register `reg` of lane (r, h) is GEMM row (reg & 3) + 8 * (reg >> 2) + 4 * h of the tile, column r
    bf16_t* o = a.out + ((long long)clip * a.OH + o0) * a.OW * N;
#pragma unroll
    for (int mt = 0; mt < MW; ++mt) {
        const int m0 = (wave * MW + mt) * 32;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = nt * 32 + r;
            const float bn = a.bias[n];
            if constexpr (POOL) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int win = (m0 >> 2) + 2 * g + h;
                    const float v = fmaxf(fmaxf(acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1]),
                                          fmaxf(acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]));
                    if (win * 4 < M) o[(long long)win * N + n] = f2bf(fmaxf(v + bn, 0.f));
                }
            } else {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int mo = m0 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                    if (mo < M) o[(long long)mo * N + n] = f2bf(fmaxf(acc[mt][nt][reg] + bn, 0.f));
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ split-bf16 blocks
// cnn_conv_lds_x3_kernel: the LDS-image convolution above with every operand as a pair x = hi + lo of bf16 values
// (hi = bf16(x), lo = bf16(x - hi): 16 significant bits) and three MFMAs per k-step, hi*hi + lo*hi + hi*lo, into one
// f32 accumulator -- the scheme of resblock_x3.h, whose logits stay within 1e-3 of the f32 reference where single bf16
// operands are 0.1-0.3 off.  Activations cross HBM as f32 (NHWC) and are split while they are staged into TWO swizzled
// LDS images; weights arrive as (hi, lo) MFMA fragments [k-step][n-tile][2][64 lanes][8]; blockIdx.y selects a group
// of NT 32-channel tiles (128 -> 256 channels: two groups).
struct CnnX3Args {
    const float* in;      // NHWC [B][H][W][CIN] f32
    const bf16_t* wf;     // [9 * CIN / 16][ntot][2 = hi, lo][64][8]
    const float* bias;
    float* out;           // NHWC [B][OH][OW][32 * ntot] f32
    int H, W, OH, OW;     // OH x OW: output size (pooled when POOL)
    int band_rows;        // output rows per workgroup
    int n_bands;          // workgroups per clip
    int ntot;             // 32-channel tiles of the layer
    float* mean_out;      // last block, one band per clip: [B][32 * ntot] channel means over OH x OW (the head's global
                          // average pool, model.py:118-121 / :198-200) INSTEAD of `out` -- the activation is neither written
                          // nor re-read by the head kernel.  nullptr: store `out`
};

// The workgroup is WM x WN waves: wave (wm, wn) owns MW 32-row tiles x NT / WN 32-channel tiles of the band, so the band
// image is staged ONCE for all NT tiles while a wave's accumulators are MW * NT / WN * 16 registers (r03: one N-wave, 128
// AGPRs + 140-156 VGPRs = one wave per SIMD).  PIECES: 16-byte f32 pieces (4 channels of a pixel) the band may hold =
// LDS bytes of its two images / 16.
template <int CIN, int NT, int MW, bool POOL, int WM, int WN, int PIECES>
__global__ __launch_bounds__(64 * WM * WN) void cnn_conv_lds_x3_kernel(CnnX3Args a) {
    constexpr int QP = CIN / 4, KSTEPS = 9 * CIN / 16, S = POOL ? 2 : 1, THREADS = 64 * WM * WN, NTW = NT / WN;
    constexpr int NPIECES = PIECES < 0 ? -PIECES : PIECES;   // (a negative PIECES selects the LEAN fragment pipeline)
    constexpr int UNP = (NPIECES + THREADS - 1) / THREADS;   // 16-byte pieces per thread
    static_assert(NT % WN == 0, "channel tiles split evenly over the N-waves");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int wm = wave % WM, wn = wave / WM;
    const int clip = blockIdx.x / a.n_bands, band = blockIdx.x - clip * a.n_bands;
    const int nt0 = blockIdx.y * NT + wn * NTW, N = 32 * a.ntot;
    const int o0 = band * a.band_rows;
    const int orows = a.OH - o0 < a.band_rows ? a.OH - o0 : a.band_rows;
    const int crow0 = S * o0, crows = S * orows;
    const int Wb = a.W + 2, irows = crows + 2;
    bf16_t* img = reinterpret_cast<bf16_t*>(smem_raw);                       // hi image
    bf16_t* img_lo = img + ((irows * Wb * CIN + 7) & ~7);                    // lo image (16-byte aligned)

    // ---- stage: every 16-byte f32 piece (4 channels of one pixel) is requested before the first one is split ----
    {
        const float* src = a.in + (long long)clip * a.H * a.W * CIN;
        const int total = irows * Wb * QP;
        const float inv_wb = 1.0f / float(Wb);
        float4 v[UNP];
        int dst[UNP];
#pragma unroll
        for (int u = 0; u < UNP; ++u) {
            const int i = tid + u * THREADS;
            const int q = i & (QP - 1), P = i / QP;
            int rr = int(float(P) * inv_wb);
            rr -= (rr * Wb > P);
            rr += ((rr + 1) * Wb <= P);
            const int cc = P - rr * Wb;
            const int iy = crow0 - 1 + rr, ix = cc - 1;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            dst[u] = i < total ? swz_off<CIN>(P, q >> 1) + 4 * (q & 1) : -1;
            if (i < total && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
                v[u] = *reinterpret_cast<const float4*>(src + ((long long)iy * a.W + ix) * CIN + 4 * q);
        }
#pragma unroll
        for (int u = 0; u < UNP; ++u)
            if (dst[u] >= 0) {
                uint2 hi, lo;
                split4(v[u].x, v[u].y, v[u].z, v[u].w, hi, lo);
                *reinterpret_cast<uint2*>(img + dst[u]) = hi;
                *reinterpret_cast<uint2*>(img_lo + dst[u]) = lo;
            }
    }
    __syncthreads();

    const int M = orows * a.OW * (POOL ? 4 : 1);
    if (wm * MW * 32 >= M && a.mean_out == nullptr) return;   // whole wave beyond the band (no barrier follows; the fused
                                                               // mean has two: there such a wave computes masked rows)
    int pix[MW];
#pragma unroll
    for (int mt = 0; mt < MW; ++mt) {
        int m = (wm * MW + mt) * 32 + r;
        if (m >= M) m = M - 1;
        int y, x;
        if constexpr (POOL) {
            const int win = m >> 2, q = m & 3, py = win / a.OW, px = win - py * a.OW;
            y = 2 * py + (q >> 1);
            x = 2 * px + (q & 1);
        } else {
            y = m / a.OW;
            x = m - y * a.OW;
        }
        pix[mt] = y * Wb + x;
    }
    f32x16 acc[MW][NTW];
#pragma unroll
    for (int mt = 0; mt < MW; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = f32x16{0};

    const bf16_t* wl = a.wf + (size_t(nt0) * 2) * 512 + lane * 8;
    const size_t wstep = size_t(a.ntot) * 2 * 512;
    // activation fragments: a whole k-step of tiles ahead (af[2][MW]: 2 * MW * 8 registers), or -- LEAN -- ONE tile ahead
    // through two rotating sets (16 registers) and a 2-deep weight ring: with MW = 5 that is what brings the kernel under the
    // 168 registers of THREE waves per SIMD, i.e. three workgroups per CU whose staging / MFMA phases interleave
    constexpr bool LEAN = PIECES < 0;
    constexpr int D = LEAN ? 2 : (NTW <= 2) ? 4 : 2;
    constexpr int AFN = LEAN ? 1 : MW;
    bf16x8 wring[D][NTW][2], af[2][AFN][2];
    auto wload = [&](int s, bf16x8 (&dst)[NTW][2]) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            dst[nt][0] = *reinterpret_cast<const bf16x8*>(wl + size_t(s) * wstep + size_t(nt) * 1024);
            dst[nt][1] = *reinterpret_cast<const bf16x8*>(wl + size_t(s) * wstep + size_t(nt) * 1024 + 512);
        }
    };
    auto aload = [&](auto sc, int mt, bf16x8 (&dst)[2]) {
        constexpr int s = decltype(sc)::value, tap = s / (CIN / 16), c16 = s % (CIN / 16), kh = tap / 3, kw = tap % 3;
        const int off = swz_off<CIN>(pix[mt] + kh * Wb + kw, 2 * c16 + h);
        dst[0] = *reinterpret_cast<const bf16x8*>(img + off);
        dst[1] = *reinterpret_cast<const bf16x8*>(img_lo + off);
    };
#pragma unroll
    for (int i = 0; i < D; ++i) wload(i, wring[i]);
    if constexpr (LEAN) {
        aload(std::integral_constant<int, 0>{}, 0, af[0][0]);
    } else {
#pragma unroll
        for (int mt = 0; mt < MW; ++mt) aload(std::integral_constant<int, 0>{}, mt, af[0][mt]);
    }
    auto step = [&]<int s>() {
        bf16x8 bw[NTW][2];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) { bw[nt][0] = wring[s % D][nt][0]; bw[nt][1] = wring[s % D][nt][1]; }
#pragma unroll
        for (int mt = 0; mt < MW; ++mt) {
            bf16x8 cur0, cur1;
            if constexpr (LEAN) {
                const int slot = (s * MW + mt) & 1;      // compile-time after unrolling
                cur0 = af[slot][0][0];
                cur1 = af[slot][0][1];
                if (mt + 1 < MW) aload(std::integral_constant<int, s>{}, mt + 1, af[slot ^ 1][0]);
                else if constexpr (s + 1 < KSTEPS) aload(std::integral_constant<int, s + 1>{}, 0, af[slot ^ 1][0]);
            } else {
                cur0 = af[s & 1][mt][0];
                cur1 = af[s & 1][mt][1];
                if constexpr (s + 1 < KSTEPS) aload(std::integral_constant<int, s + 1>{}, mt, af[(s + 1) & 1][mt]);
            }
            if constexpr (s + D < KSTEPS) {
                if (mt == 0) wload(s + D, wring[s % D]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur0, bw[nt][0], acc[mt][nt], 0, 0, 0);
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur1, bw[nt][0], acc[mt][nt], 0, 0, 0);
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur0, bw[nt][1], acc[mt][nt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    [&]<int... Ss>(std::integer_sequence<int, Ss...>) {
        (step.template operator()<Ss>(), ...);
    }(std::make_integer_sequence<int, KSTEPS>{});

    // epilogue: register `reg` of lane (r, h) is GEMM row (reg & 3) + 8 * (reg >> 2) + 4 * h of the tile, column r
    float* o = a.out + ((long long)clip * a.OH + o0) * a.OW * N;
    const bool to_mean = a.mean_out != nullptr;   // workgroup-uniform
    float csum[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) csum[nt] = 0.f;
#pragma unroll
    for (int mt = 0; mt < MW; ++mt) {
        const int m0 = (wm * MW + mt) * 32;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int n = (nt0 + nt) * 32 + r;
            const float bn = a.bias[n];
            if constexpr (POOL) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int win = (m0 >> 2) + 2 * g + h;
                    const float v = fmaxf(fmaxf(fmaxf(acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1]),
                                                fmaxf(acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3])) + bn, 0.f);
                    if (win * 4 < M) {
                        if (to_mean) csum[nt] += v;
                        else o[(long long)win * N + n] = v;
                    }
                }
            } else {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int mo = m0 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                    const float v = fmaxf(acc[mt][nt][reg] + bn, 0.f);
                    if (mo < M) {
                        if (to_mean) csum[nt] += v;
                        else o[(long long)mo * N + n] = v;
                    }
                }
            }
        }
    }
    if (to_mean) {
        // channel means of the clip (one band = the whole image): lane rows, then the two lane halves, then the M-waves, in
        // that fixed order -- deterministic and independent of the batch
        __syncthreads();                                   // every wave has finished reading the images
        float* red = reinterpret_cast<float*>(smem_raw);   // [WM][NT * 32]
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const float t = csum[nt] + __shfl_xor(csum[nt], 32, 64);
            if (h == 0) red[wm * (NT * 32) + (wn * NTW + nt) * 32 + r] = t;
        }
        __syncthreads();
        if (tid < NT * 32) {
            float t = red[tid];
#pragma unroll
            for (int w = 1; w < WM; ++w) t += red[w * (NT * 32) + tid];
            a.mean_out[(long long)clip * N + blockIdx.y * (NT * 32) + tid] = t / float(a.OH * a.OW);
        }
    }
}

// ------------------------------------------------------------------------------------------ head
// one workgroup per clip: thread c < C averages channel c; thread j < HID forms hidden unit j; wave 0 the logits
template <typename T>
__global__ __launch_bounds__(256) void cnn_tail_kernel(const T* __restrict__ act, int HW, int C, int HID,
                                                       const float* __restrict__ w1, const float* __restrict__ b1,
                                                       const float* __restrict__ w2, const float* __restrict__ b2,
                                                       float* __restrict__ logits, float* __restrict__ probs,
                                                       int* __restrict__ preds) {
    __shared__ float v[256], hid[256];
    const int tid = threadIdx.x;
    const long long b = blockIdx.x;
    if (tid < C) {
        const T* p = act + b * (long long)HW * C + tid;
        float s = 0.f;
        for (int i = 0; i < HW; ++i) s += to_f32<T>(p[(long long)i * C]);
        v[tid] = s / float(HW);
    }
    __syncthreads();
    if (tid < HID) {   // w1 is stored transposed, [C][HID]: the lanes of a wave read consecutive addresses
        float s = 0.f;
        for (int c = 0; c < C; ++c) s = fmaf(w1[c * HID + tid], v[c], s);
        hid[tid] = fmaxf(s + b1[tid], 0.f);
    }
    __syncthreads();
    if (tid < 64) {
        float l0 = 0.f, l1 = 0.f;
        for (int jx = tid; jx < HID; jx += 64) {
            l0 = fmaf(w2[jx], hid[jx], l0);
            l1 = fmaf(w2[HID + jx], hid[jx], l1);
        }
        l0 = wave_sum(l0) + b2[0];
        l1 = wave_sum(l1) + b2[1];
        if (tid == 0) {
            logits[b * 2] = l0;
            logits[b * 2 + 1] = l1;
            if (probs) {
                const float mx = fmaxf(l0, l1), e0 = expf(l0 - mx), e1 = expf(l1 - mx), inv = 1.0f / (e0 + e1);
                probs[b * 2] = e0 * inv;
                probs[b * 2 + 1] = e1 * inv;
            }
            if (preds) preds[b] = (l1 > l0) ? 1 : 0;   // argmax returns the first maximal index on ties
        }
    }
}

template <typename T>
__global__ void cnn_nhwc_to_nchw_kernel(const T* __restrict__ in, float* __restrict__ out, int C, int HW, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int p = int(idx % HW);
    const long long bc = idx / HW;
    const int c = int(bc % C);
    const long long b = bc / C;
    out[idx] = to_f32<T>(in[(b * HW + p) * C + c]);
}

size_t align256(size_t v) { return (v + 255) & ~size_t(255); }

}  // namespace
}  // namespace cough

struct cough_cnn {
    struct Layer {
        int cin, cout, ks, pool;
        void* d_w;      // first layer: float [9][N]; others: T [N][ktot]
        float* d_b;
        int ktot;       // row pitch of d_w: ks*ks*cin, padded to a multiple of 64 for the LDS-staged bf16 GEMM
        bool gemm;      // bf16, cin % 32 == 0, cout % 64 == 0: conv_gemm_bf16_kernel
        cough::bf16_t* d_wfrag;   // bf16 (16->32), (32->64), (64->128) blocks: MFMA fragments for cnn_conv_lds_kernel;
                                  // bf16x3: (hi, lo) fragments of every block after the first for cnn_conv_lds_x3_kernel
    };
    int dtype;
    size_t esize;
    std::vector<Layer> layers;
    int feat_c, hidden;
    float *d_w1, *d_b1, *d_w2, *d_b2;
};

namespace cough {
namespace {

struct CnnShape { int h, w, c; };

// output shape of every block for an (H, W) input; false if the image vanishes
bool cnn_shapes(const cough_cnn* m, int H, int W, std::vector<CnnShape>& out) {
    int h = H, w = W;
    out.clear();
    for (const auto& l : m->layers) {
        if (l.pool == 2) { h /= 2; w /= 2; }
        if (h < 1 || w < 1) return false;
        out.push_back({h, w, l.cout});
    }
    return true;
}

template <typename T>
int cnn_upload(void** dst, const std::vector<T>& v) {
    COUGH_HIP_CHECK(hipMalloc(dst, v.size() * sizeof(T)));
    COUGH_HIP_CHECK(hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return COUGH_OK;
}

// Tile configuration of the split-bf16 LDS-image convolution per input width (cin): NT 32-channel tiles per workgroup,
// WM x WN waves, MW 32-row tiles per wave, PIECES = LDS capacity of the band's two images in 16-byte units.
struct X3Cfg { int nt, mw, wm, wn, pieces; };
#ifndef CNN_X3_CFG32
// measured (profiles/r04_cnn_x3_experiments.txt): 4-wave workgroups of 2 x 2 waves with bands small enough for TWO per CU
// (one's staging overlaps the other's MFMA phase) for the 64-channel input and THREE (lean fragment pipeline) for the
// 32-channel one; the 128-channel input (one clip =
// 120 GEMM rows, 86 KB of images: one workgroup per CU) as 2 x 4 waves so that all eight waves have rows
#define CNN_X3_CFG32 2, 5, 2, 2, -3328   /* negative: LEAN fragment pipeline, <= 168 registers, bands <= 52 KB: THREE workgroups per CU */
#define CNN_X3_CFG64 4, 3, 2, 2, 5056
#define CNN_X3_CFG128 4, 2, 2, 4, 6144
#endif
#define CNN_X3_CFG16 1, 4, 4, 1, 6144
constexpr X3Cfg X3_C16{CNN_X3_CFG16}, X3_C32{CNN_X3_CFG32}, X3_C64{CNN_X3_CFG64}, X3_C128{CNN_X3_CFG128};
inline const X3Cfg* x3_cfg(const cough_cnn::Layer& l) {
    const int nt = l.cout >= 128 ? 4 : l.cout / 32;
    if (l.cin == 16 && nt == X3_C16.nt) return &X3_C16;
    if (l.cin == 32 && nt == X3_C32.nt) return &X3_C32;
    if (l.cin == 64 && nt == X3_C64.nt) return &X3_C64;
    if (l.cin == 128 && nt == X3_C128.nt) return &X3_C128;
    return nullptr;
}
// output rows per workgroup (0: the layer does not fit, use the f32 kernel): the kernel's tile shape must exist, the
// band's GEMM rows must fit WM waves x MW tiles and its two bf16 images the configuration's LDS capacity
inline int x3_band(const cough_cnn::Layer& l, const CnnShape& s, int in_w) {
    const X3Cfg* c = x3_cfg(l);
    if (!c) return 0;
    const int per_out = l.pool == 2 ? 4 : 1;
    int band = (c->wm * c->mw * 32) / (per_out * s.w);
    if (band > s.h) band = s.h;
    while (band >= 1 && size_t((l.pool == 2 ? 2 : 1) * band + 2) * (in_w + 2) * (l.cin / 4) > size_t(c->pieces < 0 ? -c->pieces : c->pieces)) --band;
    return band;
}
template <int CIN, int NT, int MW, int WM, int WN, int PIECES>
void x3_launch(bool pool, dim3 grid, size_t lds, hipStream_t st, const CnnX3Args& a) {
    if (pool) hipLaunchKernelGGL((cnn_conv_lds_x3_kernel<CIN, NT, MW, true, WM, WN, PIECES>), grid, dim3(64 * WM * WN), lds, st, a);
    else hipLaunchKernelGGL((cnn_conv_lds_x3_kernel<CIN, NT, MW, false, WM, WN, PIECES>), grid, dim3(64 * WM * WN), lds, st, a);
}
template <int CIN, int NT, int MW, int WM, int WN, int PIECES>
hipError_t x3_set_lds() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cnn_conv_lds_x3_kernel<CIN, NT, MW, true, WM, WN, PIECES>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (PIECES < 0 ? -PIECES : PIECES) * 16 + 64);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(cnn_conv_lds_x3_kernel<CIN, NT, MW, false, WM, WN, PIECES>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (PIECES < 0 ? -PIECES : PIECES) * 16 + 64);
    return e;
}

template <typename T>
int cnn_forward_impl(const cough_cnn* m, const float* d_feat, int n, int H, int W, float* d_logits, float* d_probs,
                     int* d_preds, char* ws, hipStream_t st, int tap_layer, float* d_tap) {
    std::vector<CnnShape> shp;
    cnn_shapes(m, H, W, shp);
    size_t buf = 0;
    for (const auto& s : shp) buf = std::max(buf, align256(size_t(n) * s.h * s.w * s.c * m->esize));
    T* ping[2] = {reinterpret_cast<T*>(ws), reinterpret_cast<T*>(ws + buf)};
    const T* cur = nullptr;
    int ch = H, cw = W;
    bool means_in_dst = false;
    for (size_t i = 0; i < m->layers.size(); ++i) {
        const auto& l = m->layers[i];
        const CnnShape& s = shp[i];
        T* dst = ping[i & 1];
        if (i == 0) {
            const long long n_out = (long long)n * s.h * s.w;
            const dim3 grid((unsigned)((n_out + 255) / 256));
            const FirstLds fl = first_lds(ch, cw);
            if (sizeof(T) == 4 && m->dtype == COUGH_DTYPE_BF16X3 && l.d_wfrag && l.pool == 2 && ch * cw <= 2 * 256 * CNN_FIRST_MAXL &&
                2 * fl.bytes <= 64 * 1024) {
                if constexpr (sizeof(T) == 4)
                    hipLaunchKernelGGL(cnn_first_x3_kernel, dim3(n), dim3(256), 2 * fl.bytes, st, d_feat, ch, cw, s.h, s.w, fl.nrows,
                                       fl.pitch, l.d_wfrag, l.d_b, l.cout, dst);
            } else if (l.pool == 2)
                hipLaunchKernelGGL((cnn_first_kernel<T, true>), grid, dim3(256), 0, st, d_feat, ch, cw, s.h, s.w, n_out,
                                   static_cast<const float*>(l.d_w), l.d_b, l.cout, dst);
            else
                hipLaunchKernelGGL((cnn_first_kernel<T, false>), grid, dim3(256), 0, st, d_feat, ch, cw, s.h, s.w, n_out,
                                   static_cast<const float*>(l.d_w), l.d_b, l.cout, dst);
        } else if (l.d_wfrag && sizeof(T) == 2) {
            if constexpr (sizeof(T) == 2) {
                // band of output rows per workgroup: 4 waves x MW tiles x 32 GEMM rows (x4 rows per output when pooled)
                const int mw = l.cin == 64 ? 2 : 4, per_out = l.pool == 2 ? 4 : 1;
                int band = (4 * mw * 32) / (per_out * s.w);
                if (band > s.h) band = s.h;
                const int n_bands = (s.h + band - 1) / band;
                const size_t lds = size_t((l.pool == 2 ? 2 : 1) * band + 2) * (cw + 2) * l.cin * 2;
                if (band >= 1 && lds <= size_t(CNN_LDS_UNP) * 256 * 16) {
                    CnnLdsArgs a{cur, l.d_wfrag, l.d_b, dst, ch, cw, s.h, s.w, band, n_bands};
                    const dim3 grid((unsigned)(n * n_bands));
#define COUGH_LDS_LAUNCH(CIN, NT, MW)                                                                            \
    do {                                                                                                         \
        if (l.pool == 2) hipLaunchKernelGGL((cnn_conv_lds_kernel<CIN, NT, MW, true>), grid, dim3(256), lds, st, a);  \
        else hipLaunchKernelGGL((cnn_conv_lds_kernel<CIN, NT, MW, false>), grid, dim3(256), lds, st, a);             \
    } while (0)
                    if (l.cin == 16) COUGH_LDS_LAUNCH(16, 1, 4);
                    else if (l.cin == 32) COUGH_LDS_LAUNCH(32, 2, 4);
                    else COUGH_LDS_LAUNCH(64, 4, 2);
#undef COUGH_LDS_LAUNCH
                } else {
                    set_error("cough_cnn_forward: image %dx%d too wide for the LDS-image convolution", ch, cw);
                    return COUGH_EUNSUPPORTED;
                }
            }
        } else if (l.gemm) {
            if constexpr (sizeof(T) == 2) {
                ConvArgs<bf16_t> a{};
                a.in = cur; a.H = ch; a.W = cw; a.C = l.cin; a.KH = a.KW = 3; a.stride = 1; a.pad = 1;
                a.in2 = nullptr; a.H2 = a.W2 = 0; a.C2 = 0; a.stride2 = 1;
                a.wp = static_cast<const bf16_t*>(l.d_w); a.bias = l.d_b;
                a.out = dst; a.OH = s.h; a.OW = s.w; a.N = l.cout; a.Ktot = l.ktot;
                a.M = (long long)n * s.h * s.w * (l.pool == 2 ? 4 : 1);
                const int nt = l.cout % 128 == 0 ? 4 : 2;
                const dim3 grid((unsigned)((a.M + CG_BM - 1) / CG_BM), (unsigned)(l.cout / (32 * nt)));
                if (nt == 4 && l.pool == 2) hipLaunchKernelGGL((conv_gemm_bf16_kernel<4, true>), grid, dim3(256), 0, st, a);
                else if (nt == 4) hipLaunchKernelGGL((conv_gemm_bf16_kernel<4, false>), grid, dim3(256), 0, st, a);
                else if (l.pool == 2) hipLaunchKernelGGL((conv_gemm_bf16_kernel<2, true>), grid, dim3(256), 0, st, a);
                else hipLaunchKernelGGL((conv_gemm_bf16_kernel<2, false>), grid, dim3(256), 0, st, a);
            }
        } else if (sizeof(T) == 4 && m->dtype == COUGH_DTYPE_BF16X3 && l.d_wfrag && x3_band(l, s, cw) >= 1) {
            if constexpr (sizeof(T) == 4) {
                // split-bf16 LDS-image convolution: band of output rows per workgroup, as many as 4 waves x MW tiles cover
                // and as two band images of <= 96 KB hold
                const int band = x3_band(l, s, cw), n_bands = (s.h + band - 1) / band;
                const size_t lds = (size_t((l.pool == 2 ? 2 : 1) * band + 2) * (cw + 2) * l.cin * 2 + 15) / 16 * 16 * 2;
                CnnX3Args a{cur, l.d_wfrag, l.d_b, dst, ch, cw, s.h, s.w, band, n_bands, l.cout / 32, nullptr};
                // last block, whole image in one band, logits wanted, nobody taps the activation: its epilogue forms the
                // head's channel means ([n][cout] floats in `dst`) and the activation never crosses HBM
                if (i + 1 == m->layers.size() && n_bands == 1 && d_logits && !(d_tap && tap_layer == int(i))) {
                    a.mean_out = reinterpret_cast<float*>(dst);
                    means_in_dst = true;
                }
                const int nt = l.cout >= 128 ? 4 : l.cout / 32;
                const dim3 grid((unsigned)(n * n_bands), (unsigned)(l.cout / (32 * nt)));
                if (l.cin == 16) x3_launch<16, CNN_X3_CFG16>(l.pool == 2, grid, lds, st, a);
                else if (l.cin == 32) x3_launch<32, CNN_X3_CFG32>(l.pool == 2, grid, lds, st, a);
                else if (l.cin == 64) x3_launch<64, CNN_X3_CFG64>(l.pool == 2, grid, lds, st, a);
                else x3_launch<128, CNN_X3_CFG128>(l.pool == 2, grid, lds, st, a);
            }
        } else {
            CnnConvArgs<T> a{};
            a.in = cur; a.H = ch; a.W = cw; a.C = l.cin; a.KS = l.ks;
            a.wp = static_cast<const T*>(l.d_w); a.bias = l.d_b;
            a.out = dst; a.OH = s.h; a.OW = s.w; a.N = l.cout;
            a.M = (long long)n * s.h * s.w * (l.pool == 2 ? 4 : 1);
            const int nt = l.cout >= 128 ? 4 : l.cout / 32;
            const dim3 grid((unsigned)(((a.M + 31) / 32 + 3) / 4), (unsigned)(l.cout / (32 * nt)));
#define COUGH_CNN_LAUNCH(NT)                                                                              \
    do {                                                                                                 \
        if (l.pool == 2) hipLaunchKernelGGL((cnn_conv_kernel<T, NT, true>), grid, dim3(256), 0, st, a);   \
        else hipLaunchKernelGGL((cnn_conv_kernel<T, NT, false>), grid, dim3(256), 0, st, a);              \
    } while (0)
            if (nt == 1) COUGH_CNN_LAUNCH(1);
            else if (nt == 2) COUGH_CNN_LAUNCH(2);
            else COUGH_CNN_LAUNCH(4);
#undef COUGH_CNN_LAUNCH
        }
        COUGH_HIP_CHECK(hipGetLastError());
        if (d_tap && tap_layer == int(i)) {
            const long long total = (long long)n * s.c * s.h * s.w;
            hipLaunchKernelGGL(cnn_nhwc_to_nchw_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dst,
                               d_tap, s.c, s.h * s.w, total);
            COUGH_HIP_CHECK(hipGetLastError());
        }
        cur = dst;
        ch = s.h;
        cw = s.w;
    }
    if (d_logits) {
        // (means_in_dst: `cur` holds the channel means, i.e. a 1 x 1 "image" per clip)
        hipLaunchKernelGGL(cnn_tail_kernel<T>, dim3(n), dim3(256), 0, st, cur, means_in_dst ? 1 : ch * cw, m->feat_c, m->hidden,
                           m->d_w1, m->d_b1, m->d_w2, m->d_b2, d_logits, d_probs, d_preds);
        COUGH_HIP_CHECK(hipGetLastError());
    }
    return COUGH_OK;
}

}  // namespace
}  // namespace cough

extern "C" void cough_cnn_destroy(cough_cnn* m) {
    if (!m) return;
    for (auto& l : m->layers) {
        (void)hipFree(l.d_w);
        (void)hipFree(l.d_b);
        (void)hipFree(l.d_wfrag);
    }
    (void)hipFree(m->d_w1);
    (void)hipFree(m->d_b1);
    (void)hipFree(m->d_w2);
    (void)hipFree(m->d_b2);
    delete m;
}

extern "C" int cough_cnn_create(cough_cnn** out, const cough_cnn_weights* w, int dtype) {
    using namespace cough;
    COUGH_REQUIRE(out && w && w->blocks && w->fc1_w && w->fc1_b && w->fc2_w && w->fc2_b, COUGH_EINVAL,
                  "cough_cnn_create: NULL argument");
    COUGH_REQUIRE(dtype == COUGH_DTYPE_FP32 || dtype == COUGH_DTYPE_BF16 || dtype == COUGH_DTYPE_BF16X3, COUGH_EINVAL,
                  "cough_cnn_create: unknown dtype %d", dtype);
    COUGH_REQUIRE(w->n_blocks >= 2 && w->n_blocks <= 16, COUGH_EUNSUPPORTED, "cough_cnn_create: %d blocks (need 2..16)", w->n_blocks);
    COUGH_REQUIRE(w->hidden >= 1 && w->hidden <= 256, COUGH_EUNSUPPORTED, "cough_cnn_create: hidden = %d (need 1..256)", w->hidden);
    // bf16x3 keeps f32 activations: a layer without a split-bf16 instantiation (x3_band) runs the exact-f32 kernel, so only
    // that kernel's 8-channel chunk is required; the single-bf16 mode needs 16
    const int chunk = dtype == COUGH_DTYPE_BF16 ? 16 : 8;
    for (int i = 0; i < w->n_blocks; ++i) {
        const cough_cnn_block& bk = w->blocks[i];
        const cough_conv_bn& p = bk.conv;
        COUGH_REQUIRE(p.w && p.b && p.bn_w && p.bn_b && p.bn_mean && p.bn_var, COUGH_EINVAL,
                      "cough_cnn_create: NULL weight pointer in block %d", i);
        COUGH_REQUIRE(bk.pool == 1 || bk.pool == 2, COUGH_EUNSUPPORTED, "cough_cnn_create: block %d pool = %d", i, bk.pool);
        const bool sep = bk.dw_w != nullptr;
        COUGH_REQUIRE(sep ? (bk.ksize == 1 && bk.dw_b) : bk.ksize == 3, COUGH_EUNSUPPORTED,
                      "cough_cnn_create: block %d: need a 3x3 conv, or a depthwise 3x3 followed by a 1x1", i);
        if (i == 0) {
            COUGH_REQUIRE(bk.cin == 1 && !sep && bk.cout >= 8 && bk.cout <= 64 && bk.cout % chunk == 0, COUGH_EUNSUPPORTED,
                          "cough_cnn_create: the first block must be a dense 3x3 conv of 1 input channel to 16..64 outputs");
        } else {
            COUGH_REQUIRE(bk.cin == w->blocks[i - 1].cout, COUGH_EINVAL, "cough_cnn_create: block %d cin != previous cout", i);
            COUGH_REQUIRE(bk.cin % chunk == 0 && bk.cout % 32 == 0 && (bk.cout <= 128 || bk.cout % 128 == 0), COUGH_EUNSUPPORTED,
                          "cough_cnn_create: block %d (%d -> %d channels): need cin %% %d == 0 and cout in {32, 64, 128k}", i,
                          bk.cin, bk.cout, chunk);
        }
    }
    COUGH_REQUIRE(w->blocks[w->n_blocks - 1].cout <= 256, COUGH_EUNSUPPORTED, "cough_cnn_create: last block > 256 channels");

    cough_cnn* m = new cough_cnn();
    m->dtype = dtype;
    m->esize = dtype == COUGH_DTYPE_BF16 ? 2 : 4;
    m->d_w1 = m->d_b1 = m->d_w2 = m->d_b2 = nullptr;
    int err = COUGH_OK;
    for (int i = 0; i < w->n_blocks && !err; ++i) {
        const cough_cnn_block& bk = w->blocks[i];
        const cough_conv_bn& p = bk.conv;
        const int N = bk.cout, Cc = bk.cin, K = 9 * Cc;
        std::vector<double> wd(size_t(N) * K), bd(N);   // dense 3x3, k = (kh*3 + kw)*C + c
        for (int n = 0; n < N; ++n) {
            double bacc = p.b[n];
            for (int c = 0; c < Cc; ++c) {
                if (bk.dw_w) {   // depthwise 3x3 then 1x1: one dense 3x3
                    const double pw = p.w[size_t(n) * Cc + c];
                    for (int t = 0; t < 9; ++t) wd[size_t(n) * K + t * Cc + c] = pw * double(bk.dw_w[size_t(c) * 9 + t]);
                    bacc += pw * double(bk.dw_b[c]);
                } else {
                    for (int t = 0; t < 9; ++t) wd[size_t(n) * K + t * Cc + c] = p.w[(size_t(n) * Cc + c) * 9 + t];
                }
            }
            const double scale = double(p.bn_w[n]) / std::sqrt(double(p.bn_var[n]) + double(w->bn_eps));
            for (int k = 0; k < K; ++k) wd[size_t(n) * K + k] *= scale;
            bd[n] = (bacc - double(p.bn_mean[n])) * scale + double(p.bn_b[n]);
        }
        cough_cnn::Layer l{Cc, N, 3, bk.pool, nullptr, nullptr, K, false, nullptr};
        l.gemm = i > 0 && m->esize == 2 && Cc % 32 == 0 && N % 64 == 0;
        if (l.gemm) l.ktot = ((K + 63) / 64) * 64;
        std::vector<float> bf(N);
        for (int n = 0; n < N; ++n) bf[n] = float(bd[n]);
        if (i == 0) {
            std::vector<float> wk(size_t(9) * N);
            for (int n = 0; n < N; ++n)
                for (int t = 0; t < 9; ++t) wk[size_t(t) * N + n] = float(wd[size_t(n) * 9 + t]);
            err = cnn_upload(&l.d_w, wk);
        } else if (m->esize == 4) {
            std::vector<float> wf(wd.size());
            for (size_t k = 0; k < wd.size(); ++k) wf[k] = float(wd[k]);
            err = cnn_upload(&l.d_w, wf);
        } else {
            std::vector<bf16_t> wb(size_t(N) * l.ktot, 0);
            for (int n = 0; n < N; ++n)
                for (int k = 0; k < K; ++k) wb[size_t(n) * l.ktot + k] = f2bf_host(float(wd[size_t(n) * K + k]));
            err = cnn_upload(&l.d_w, wb);
        }
        if (!err && i > 0 && m->esize == 2 && ((Cc == 16 && N == 32) || (Cc == 32 && N == 64) || (Cc == 64 && N == 128))) {
            // fragment order of cnn_conv_lds_kernel: lane (r, h) of (k-step s, n-tile t) holds W[32t + r][16s + 8h ..+7]
            const int ks = K / 16, nt = N / 32;
            std::vector<bf16_t> wfr(size_t(ks) * nt * 512);
            for (int st = 0; st < ks; ++st)
                for (int t = 0; t < nt; ++t)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int jj = 0; jj < 8; ++jj)
                            wfr[((size_t(st) * nt + t) * 64 + lane) * 8 + jj] =
                                f2bf_host(float(wd[size_t(32 * t + (lane & 31)) * K + 16 * st + 8 * (lane >> 5) + jj]));
            err = cnn_upload(reinterpret_cast<void**>(&l.d_wfrag), wfr);
        }
        if (!err && i > 0 && dtype == COUGH_DTYPE_BF16X3 && (Cc == 16 || Cc == 32 || Cc == 64 || Cc == 128) &&
            (N == 32 || N == 64 || N % 128 == 0)) {
            // fragment order of cnn_conv_lds_x3_kernel: [k-step][n-tile][hi, lo][64 lanes][8], lo = bf16(w - hi)
            const int ks = K / 16, nt = N / 32;
            std::vector<bf16_t> wfr(size_t(ks) * nt * 2 * 512);
            for (int st = 0; st < ks; ++st)
                for (int t = 0; t < nt; ++t)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int jj = 0; jj < 8; ++jj) {
                            const float v = float(wd[size_t(32 * t + (lane & 31)) * K + 16 * st + 8 * (lane >> 5) + jj]);
                            const bf16_t hi = f2bf_host(v);
                            const uint32_t hb = uint32_t(hi) << 16;
                            float hf;
                            std::memcpy(&hf, &hb, 4);
                            const size_t base = ((size_t(st) * nt + t) * 2) * 512 + size_t(lane) * 8 + jj;
                            wfr[base] = hi;
                            wfr[base + 512] = f2bf_host(v - hf);
                        }
            err = cnn_upload(reinterpret_cast<void**>(&l.d_wfrag), wfr);
        }
        if (!err && i == 0 && dtype == COUGH_DTYPE_BF16X3 && bk.pool == 2 && N <= 32) {
            // B fragments of cnn_first_x3_kernel: lane (n, h), element jj <-> k = 8h + jj = 4 kh + kw' (kw' = 3 and kh = 3: zero)
            std::vector<bf16_t> wfr(2 * 512, 0);
            for (int lane = 0; lane < 64; ++lane)
                for (int jj = 0; jj < 8; ++jj) {
                    const int n = lane & 31, k = 8 * (lane >> 5) + jj, kh = k >> 2, kw = k & 3;
                    if (n >= N || kh >= 3 || kw >= 3) continue;
                    const float v = float(wd[size_t(n) * 9 + kh * 3 + kw]);
                    const bf16_t hi = f2bf_host(v);
                    const uint32_t hb = uint32_t(hi) << 16;
                    float hf;
                    std::memcpy(&hf, &hb, 4);
                    wfr[size_t(lane) * 8 + jj] = hi;
                    wfr[512 + size_t(lane) * 8 + jj] = f2bf_host(v - hf);
                }
            err = cnn_upload(reinterpret_cast<void**>(&l.d_wfrag), wfr);
        }
        if (!err) err = cnn_upload(reinterpret_cast<void**>(&l.d_b), bf);
        m->layers.push_back(l);
    }
    m->feat_c = w->blocks[w->n_blocks - 1].cout;
    m->hidden = w->hidden;
    if (!err && dtype == COUGH_DTYPE_BF16X3) {   // the two LDS images of a band can exceed 64 KB
        hipError_t e = x3_set_lds<16, CNN_X3_CFG16>();
        if (e == hipSuccess) e = x3_set_lds<32, CNN_X3_CFG32>();
        if (e == hipSuccess) e = x3_set_lds<64, CNN_X3_CFG64>();
        if (e == hipSuccess) e = x3_set_lds<128, CNN_X3_CFG128>();
        if (e != hipSuccess) {
            set_error("cough_cnn_create: %s", hipGetErrorString(e));
            err = COUGH_EHIP;
        }
    }
    if (!err) {   // fc1 transposed to [C][hidden] for coalesced reads in the head kernel
        std::vector<float> w1t(size_t(m->hidden) * m->feat_c);
        for (int jx = 0; jx < m->hidden; ++jx)
            for (int c = 0; c < m->feat_c; ++c) w1t[size_t(c) * m->hidden + jx] = w->fc1_w[size_t(jx) * m->feat_c + c];
        err = cnn_upload(reinterpret_cast<void**>(&m->d_w1), w1t);
    }
    if (!err) err = cnn_upload(reinterpret_cast<void**>(&m->d_b1), std::vector<float>(w->fc1_b, w->fc1_b + m->hidden));
    if (!err) err = cnn_upload(reinterpret_cast<void**>(&m->d_w2), std::vector<float>(w->fc2_w, w->fc2_w + size_t(2) * m->hidden));
    if (!err) err = cnn_upload(reinterpret_cast<void**>(&m->d_b2), std::vector<float>(w->fc2_b, w->fc2_b + 2));
    if (err) {
        cough_cnn_destroy(m);
        return err;
    }
    *out = m;
    return COUGH_OK;
}

extern "C" size_t cough_cnn_workspace_bytes(const cough_cnn* m, int n_clips, int height, int width) {
    using namespace cough;
    std::vector<CnnShape> shp;
    if (!m || n_clips < 0 || height < 1 || width < 1 || !cnn_shapes(m, height, width, shp)) return 0;
    size_t buf = 0;
    for (const auto& s : shp) buf = std::max(buf, align256(size_t(n_clips) * s.h * s.w * s.c * m->esize));
    return 2 * buf;
}

namespace {
int cnn_run(const cough_cnn* m, const float* d_feat, int n_clips, int height, int width, float* d_logits, float* d_probs,
            int* d_preds, void* d_workspace, size_t workspace_bytes, void* stream, int tap_layer, float* d_tap) {
    using namespace cough;
    COUGH_REQUIRE(m && d_feat && d_workspace, COUGH_EINVAL, "cough_cnn_forward: NULL argument");
    COUGH_REQUIRE(n_clips >= 0 && height >= 1 && width >= 1, COUGH_EINVAL, "cough_cnn_forward: bad shape");
    std::vector<CnnShape> shp;
    COUGH_REQUIRE(cnn_shapes(m, height, width, shp), COUGH_EINVAL, "cough_cnn_forward: input %dx%d too small for the network",
                  height, width);
    COUGH_REQUIRE((reinterpret_cast<size_t>(d_workspace) & 255) == 0, COUGH_EINVAL, "cough_cnn_forward: workspace must be 256-byte aligned");
    COUGH_REQUIRE(workspace_bytes >= cough_cnn_workspace_bytes(m, n_clips, height, width), COUGH_EWORKSPACE,
                  "cough_cnn_forward: workspace too small");
    if (n_clips == 0) return COUGH_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(d_workspace);
    const int e = m->esize == 4
        ? cnn_forward_impl<float>(m, d_feat, n_clips, height, width, d_logits, d_probs, d_preds, ws, st, tap_layer, d_tap)
        : cnn_forward_impl<bf16_t>(m, d_feat, n_clips, height, width, d_logits, d_probs, d_preds, ws, st, tap_layer, d_tap);
    if (e == COUGH_OK && d_logits) {   // a NaN pixel -> NaN logits (nn_common.h: torch's ReLU / max-pool propagate it)
        hipLaunchKernelGGL(nan_rule_kernel, dim3(n_clips), dim3(256), 0, st, d_feat, (long long)height * width, nullptr, d_logits,
                           d_probs, d_preds);
        COUGH_HIP_CHECK(hipGetLastError());
    }
    return e;
}
}  // namespace

extern "C" int cough_cnn_forward(const cough_cnn* m, const float* d_feat, int n_clips, int height, int width,
                                 float* d_logits, float* d_probs, int* d_preds, void* d_workspace,
                                 size_t workspace_bytes, void* stream) {
    COUGH_REQUIRE(d_logits, COUGH_EINVAL, "cough_cnn_forward: NULL argument");
    return cnn_run(m, d_feat, n_clips, height, width, d_logits, d_probs, d_preds, d_workspace, workspace_bytes, stream, -1, nullptr);
}

extern "C" int cough_cnn_conv_output(const cough_cnn* m, const float* d_feat, int n_clips, int height, int width,
                                     float* d_out, void* d_workspace, size_t workspace_bytes, void* stream) {
    COUGH_REQUIRE(m && d_out, COUGH_EINVAL, "cough_cnn_conv_output: NULL argument");
    return cnn_run(m, d_feat, n_clips, height, width, nullptr, nullptr, nullptr, d_workspace, workspace_bytes, stream,
                   int(m->layers.size()) - 1, d_out);
}
