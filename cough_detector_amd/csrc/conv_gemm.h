// LDS-staged bf16 implicit-GEMM convolution shared by the residual net (resnet.hip, fallback path of the fused
// block kernels) and the conv-block classifiers (cnn.hip).
#pragma once
#include "nn_common.h"

namespace cough {
namespace {

template <typename T>
struct ConvArgs {
    const T* in;      // NHWC [B][H][W][C]
    int H, W, C, KH, KW, stride, pad;
    const T* in2;     // optional fused 1x1 projection input NHWC [B][H2][W2][C2] (C2 = 0: none)
    int H2, W2, C2, stride2;
    const T* wp;      // [N][Ktot], k = ((kh*KW + kw)*C + c), then C2 skip channels
    const float* bias;
    T* out;           // NHWC [B][OH][OW][N]
    int OH, OW, N, Ktot;
    long long M;      // B*OH*OW
};

// bf16 path: LDS-staged implicit GEMM.  Workgroup tile = 128 output pixels x all N channels; K is the
// flat index k = (kh*KW + kw)*C + c followed by the C2 channels of the fused 1x1 projection, zero-padded
// to a multiple of 64 in the packed weights.  Per 64-wide chunk every thread gathers 64 contiguous bytes
// (32 channels of one tap of one output pixel, zero outside the image) of the A tile and 64 bytes of a
// weight row into registers TWO chunks ahead of the multiply; chunks pass through two LDS stages with one
// barrier each.  LDS rows are 144 B (128 + 16 pad): ds_read_b128 of 16 different rows is conflict-free.
// Waves form a 2(M) x 2(N) grid: each owns 64 rows x N/2 columns.
constexpr int CG_BM = 128, CG_BK = 64, CG_PITCH = 72;   // pitch in bf16 elements

struct CgRegs {
    uint4 a[4], b[4];
};

// POOL: the GEMM rows are ordered (pool window, dy, dx) -- a.OH x a.OW is the POOLED size, a.M = 4 * B * OH * OW --
// and the epilogue takes the 2x2 max (4 accumulator registers of one lane) before bias + ReLU: conv + BN + ReLU +
// MaxPool2d(2) without the un-pooled tensor.  blockIdx.y selects a slice of N output channels (a.N = all of them).
template <int NT, bool POOL = false>   // N = 32 * NT channels per workgroup
__global__ __launch_bounds__(256, 2) void conv_gemm_bf16_kernel(ConvArgs<bf16_t> a) {
    constexpr int N = 32 * NT;
    const int n_base = blockIdx.y * N;
    constexpr int A_ELEMS = CG_BM * CG_PITCH, B_ELEMS = N * CG_PITCH, STAGE = A_ELEMS + B_ELEMS;
    __shared__ __attribute__((aligned(16))) bf16_t lds[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const long long m0 = (long long)blockIdx.x * CG_BM;
    const int per = a.OH * a.OW;

    // this thread's A segment: row tid>>1, channels (tid&1)*32 .. +31 of each chunk
    const int arow = tid >> 1, seg = (tid & 1) * 32;
    const long long am = m0 + arow;
    const bool aok = am < a.M;
    const long long amc = aok ? am : 0;
    int ab, aoh, aow;   // conv output pixel of this thread's A row
    if constexpr (POOL) {
        const long long pix = amc >> 2;
        const int q = int(amc & 3);
        ab = int(pix / per);
        const int arem = int(pix - (long long)ab * per), ph = arem / a.OW, pw = arem - ph * a.OW;
        aoh = 2 * ph + (q >> 1);
        aow = 2 * pw + (q & 1);
    } else {
        ab = int(amc / per);
        const int arem = int(amc - (long long)ab * per);
        aoh = arem / a.OW;
        aow = arem - aoh * a.OW;
    }
    const int kmain = a.KH * a.KW * a.C, kreal = kmain + a.C2;
    const int n_chunks = a.Ktot / CG_BK;   // Ktot is padded to a multiple of 64
    const bool bload = arow < N;           // N = 64: half the threads carry no B segment
    const bf16_t* brow = a.wp + (long long)(n_base + (bload ? arow : 0)) * a.Ktot + seg;

    auto load_chunk = [&](int kc, CgRegs& rg) {
        const int k0 = kc * CG_BK + seg;
        const bf16_t* p = a.in;   // always readable; `ok` decides whether the data is used
        bool ok = false;
        if (k0 < kmain) {
            const int tap = k0 / a.C, c0 = k0 - tap * a.C;
            const int kh = tap / a.KW, kw = tap - kh * a.KW;
            const int ih = aoh * a.stride - a.pad + kh, iw = aow * a.stride - a.pad + kw;
            if (aok && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W) {
                p = a.in + (((long long)ab * a.H + ih) * a.W + iw) * a.C + c0;
                ok = true;
            }
        } else if (k0 < kreal && aok) {
            p = a.in2 + (((long long)ab * a.H2 + aoh * a.stride2) * a.W2 + aow * a.stride2) * a.C2 + (k0 - kmain);
            ok = true;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 v = reinterpret_cast<const uint4*>(p)[q];
            rg.a[q] = ok ? v : make_uint4(0, 0, 0, 0);
        }
        if (bload) {
#pragma unroll
            for (int q = 0; q < 4; ++q) rg.b[q] = reinterpret_cast<const uint4*>(brow + kc * CG_BK)[q];
        }
    };
    auto store_chunk = [&](int buf, const CgRegs& rg) {
        bf16_t* la = lds + buf * STAGE;
        bf16_t* lb = la + A_ELEMS;
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<uint4*>(la + arow * CG_PITCH + seg + 8 * q) = rg.a[q];
        if (bload) {
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<uint4*>(lb + arow * CG_PITCH + seg + 8 * q) = rg.b[q];
        }
    };

    f32x16 acc[2][NT / 2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT / 2; ++nt) acc[mt][nt] = f32x16{0};

    auto compute = [&](int buf) {
        const bf16_t* la = lds + buf * STAGE;
        const bf16_t* lb = la + A_ELEMS;
#pragma unroll
        for (int ks = 0; ks < CG_BK / 16; ++ks) {
            bf16x8 af[2], bfr[NT / 2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                af[mt] = *reinterpret_cast<const bf16x8*>(la + (64 * wm + 32 * mt + r) * CG_PITCH + ks * 16 + 8 * h);
#pragma unroll
            for (int nt = 0; nt < NT / 2; ++nt)
                bfr[nt] = *reinterpret_cast<const bf16x8*>(lb + (wn * (N / 2) + 32 * nt + r) * CG_PITCH + ks * 16 + 8 * h);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT / 2; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt], bfr[nt], acc[mt][nt], 0, 0, 0);
        }
    };

    CgRegs r0, r1;
    load_chunk(0, r0);
    if (n_chunks > 1) load_chunk(1, r1);
    store_chunk(0, r0);
    __syncthreads();
    for (int kc = 0; kc < n_chunks; kc += 2) {
        if (kc + 2 < n_chunks) load_chunk(kc + 2, r0);
        compute(0);
        if (kc + 1 < n_chunks) store_chunk(1, r1);
        __syncthreads();
        if (kc + 1 >= n_chunks) break;
        if (kc + 3 < n_chunks) load_chunk(kc + 3, r1);
        compute(1);
        if (kc + 2 < n_chunks) store_chunk(0, r0);
        __syncthreads();
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT / 2; ++nt) {
            const int n = n_base + wn * (N / 2) + 32 * nt + r;
            const float bn = a.bias[n];
            if constexpr (POOL) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {   // rows 8g + 4h .. +3 of the tile = the four positions of one window
                    const long long pix = ((m0 + 64 * wm + 32 * mt) >> 2) + 2 * g + h;
                    const float v = fmaxf(fmaxf(acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1]),
                                          fmaxf(acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]));
                    if (pix * 4 < a.M) a.out[pix * a.N + n] = f2bf(fmaxf(v + bn, 0.f));
                }
            } else {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const long long mo = m0 + 64 * wm + 32 * mt + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                    if (mo < a.M) a.out[mo * a.N + n] = f2bf(fmaxf(acc[mt][nt][reg] + bn, 0.f));
                }
            }
        }
}

}  // namespace
}  // namespace cough
