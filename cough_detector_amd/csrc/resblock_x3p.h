// EXPERIMENT (round 3, not on the product path unless COUGH_RBXP=1): block 0 of the split-bf16 classifier as ONE
// PERSISTENT 9-wave workgroup per CU -- the structure that took the STFT stage from 35 % to 51 % of its roofline.
//
// resblock_x3_kernel (resblock_x3.h) spends 30 % of a workgroup's life staging its clip: 70 KB of f32 activations arrive
// at the CU's ingest rate behind a 6 k-cycle first-byte latency, get split into hi / lo bf16 and written to LDS, and only
// the partner workgroup's MFMA phases cover it.  Two resident workgroups are all that fit (74 KB each), so nothing can
// be prefetched.  Here the whole CU belongs to one workgroup:
//   * the clip's LDS image (hi + lo planes, 73 728 B) is prepared in HBM in exactly the layout the k-loop reads (here by
//     rbxp_planes_kernel; in a product version the stem's epilogue would write it), so staging is a linear copy;
//   * a LOADER wave (wave 8) copies the NEXT clip's image into the other of two LDS buffers by LDS-DMA
//     (72 x global_load_lds_dwordx4) while the eight compute waves work on the current one -- its vmcnt is its own, so
//     the compute waves' weight stream never queues behind the prefetch;
//   * eight compute waves = 4 pixel tiles x 2 channel tiles of the SAME clip (one 32 x 32 tile each: acc1 + acc2 = 32
//     accumulator registers); waves 0..3 also own 16 channels of the 16-row tail tile.
// Same arithmetic, same operand order per accumulator as resblock_x3_kernel: the outputs are bit-identical.
#pragma once
#include "resblock_x3.h"

namespace cough {
namespace {

// f32 NHWC activations of one clip -> the two chunk-planar, parity-split bf16 planes (hi, lo) of RbxCfg, as bytes
template <int CIN, int COUT, int XH, int XW>
__global__ __launch_bounds__(256) void rbxp_planes_kernel(const float* __restrict__ x, unsigned char* __restrict__ planes,
                                                          int n_clips) {
    using Cfg = RbxCfg<CIN, COUT, 1, XH, XW>;
    constexpr int OW = Cfg::OW, NPP = Cfg::NPP, CHI = Cfg::CHI, CPX = Cfg::CPX, PL = Cfg::PL;
    const long long clip = blockIdx.x;
    const float* src = x + clip * (long long)(XH * XW * CIN);
    unsigned char* dst = planes + clip * (long long)(2 * PL);
    for (int item = threadIdx.x; item < (NPP + 1) * CHI; item += 256) {
        const int c = item / CHI, q = item % CHI;   // cell, 8-channel chunk
        int ih = -1, iw = -1;
        if (c < NPP) {
            const int sub = c < Cfg::PB01 ? 0 : c < Cfg::PB10 ? 1 : c < Cfg::PB11 ? 2 : 3;
            const int base = sub == 0 ? 0 : sub == 1 ? Cfg::PB01 : sub == 2 ? Cfg::PB10 : Cfg::PB11;
            const int i = (c - base) / OW, j = (c - base) % OW;
            ih = 2 * i + (sub >> 1);
            iw = 2 * j + (sub & 1);
        }
        uint2 hi0 = make_uint2(0, 0), lo0 = hi0, hi1 = hi0, lo1 = hi0;
        if (ih >= 0 && ih < XH && iw < XW) {
            const float4 v0 = *reinterpret_cast<const float4*>(src + (ih * XW + iw) * CIN + 8 * q);
            const float4 v1 = *reinterpret_cast<const float4*>(src + (ih * XW + iw) * CIN + 8 * q + 4);
            split4(v0.x, v0.y, v0.z, v0.w, hi0, lo0);
            split4(v1.x, v1.y, v1.z, v1.w, hi1, lo1);
        }
        *reinterpret_cast<uint4*>(dst + q * CPX + c * 16) = make_uint4(hi0.x, hi0.y, hi1.x, hi1.y);
        *reinterpret_cast<uint4*>(dst + PL + q * CPX + c * 16) = make_uint4(lo0.x, lo0.y, lo1.x, lo1.y);
    }
    (void)n_clips;
}

__device__ __forceinline__ void rbxp_glds16(const unsigned char* gsrc, unsigned lds_dst) {
    unsigned keep;
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int CIN, int COUT, int XH, int XW>
struct RbxpCfg {
    using Base = RbxCfg<CIN, COUT, 1, XH, XW>;
    static constexpr int CWAVES = 8, THREADS = (CWAVES + 1) * 64;       // 8 compute waves + the loader
    static constexpr int BUF = 2 * Base::PL;                             // one clip's hi + lo planes
    static constexpr int BIAS = 2 * BUF;
    static constexpr int LDS = BIAS + 2 * COUT * 4;
    static_assert(Base::NT == 2 && Base::FULL == 4 && Base::TAIL, "4 full tiles x 2 channel tiles + the 16-row tile");
    static_assert(BUF % 1024 == 0, "whole LDS-DMA pieces");
    static_assert(LDS <= 160 * 1024, "one workgroup per CU");
};

template <int CIN, int COUT, int XH, int XW>
__global__ __launch_bounds__(576) void resblock_x3p_kernel(RbxArgs a, const unsigned char* __restrict__ planes) {
    using Cfg = RbxCfg<CIN, COUT, 1, XH, XW>;
    using PCfg = RbxpCfg<CIN, COUT, XH, XW>;
    constexpr int OH = Cfg::OH, OW = Cfg::OW, PER = Cfg::PER, M = Cfg::M;
    constexpr int NT = Cfg::NT, KS1 = Cfg::KS1, KSP = Cfg::KSP, KS = Cfg::KS;
    using f32x4 = __attribute__((ext_vector_type(4))) float;
    constexpr int CHO = Cfg::CHO, PL = Cfg::PL, OP = Cfg::OP, CPX = Cfg::CPX, CPH = Cfg::CPH, NPP = Cfg::NPP;
    constexpr int ZX = NPP * 16, ZH = M * 16;
    constexpr int D = 3, DT = 1, BUF = PCfg::BUF, CT = PCfg::CWAVES * 64;
    extern __shared__ __attribute__((aligned(256))) char smem[];
    float* lbias = reinterpret_cast<float*>(smem + PCfg::BIAS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const bool loader = wave == PCfg::CWAVES;
    const int ng = wave & 1, mg = (wave >> 1) & 3;
    const bool tailw = wave < 4;                       // waves 0..3: 16 channels each of the 16-row tile
    const unsigned smem_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);

    auto dma_clip = [&](long long c, int buf) {        // loader wave: 72 KB, 1 KB per instruction
        const unsigned char* src = planes + c * (long long)BUF + lane * 16;
#pragma unroll 8
        for (int it = 0; it < BUF / 1024; ++it) rbxp_glds16(src + it * 1024, smem_lds + buf * BUF + it * 1024);
    };

    long long clip = blockIdx.x;
    if (clip >= a.n_clips) return;
    if (loader) {
        dma_clip(clip, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (tid < COUT) {
        lbias[tid] = a.b1[tid];
        lbias[COUT + tid] = a.b2[tid];
    }
    __syncthreads();

    // ---- per-lane geometry (compute waves): lane r owns output pixel R = 32 mg + r of its tile ----
    const int R = mg * 32 + r;                         // < 128 <= M: always a real row
    const int goh = R / OW, gow = R % OW;
    const int px1 = R * 16 + h * CPX;                  // x cell (oh, ow) of sub-image (0, 0), this lane's chunk (G = 1: rem = R)
    const int ph1 = R * 16 + h * CPH;
    auto tapx = [](int kh, int kw) constexpr -> int {
        const int aa = (kh + 1) & 1, bb = (kw + 1) & 1;
        return (aa ? (bb ? Cfg::PB11 : Cfg::PB10) : (bb ? Cfg::PB01 : 0)) - (kh == 0 ? Cfg::OW : 0) - (kw == 0 ? 1 : 0);
    };
    const int tq = lane >> 4;
    const int Rt = Cfg::FULL * 32 + (lane & 15);
    const bool rokt = Rt < M;
    int toh = -4, tow = 0, tpx1 = 0, tph1 = 0;
    if (rokt) {
        toh = Rt / OW;
        tow = Rt % OW;
        tpx1 = Rt * 16 + tq * CPX;
        tph1 = Rt * 16 + tq * CPH;
    }
    const bf16_t* wbase0 = a.wf + size_t(ng) * 1024 + lane * 8;
    const bf16_t* wtbase0 = a.wt + size_t(wave & 3) * 1024 + lane * 8;

    int cur = 0;
#pragma unroll 1
    while (true) {
        const long long next = clip + gridDim.x;
        char* const sm = smem + cur * BUF;
        asm volatile("" ::: "memory");
        if (loader) {
            // ---- the next clip's image moves into the other buffer while this one is computed; five barriers ----
            if (next < a.n_clips) dma_clip(next, cur ^ 1);
            asm volatile("s_barrier" ::: "memory");    // x dead
            asm volatile("s_barrier" ::: "memory");    // h written
            asm volatile("s_barrier" ::: "memory");    // conv2 done
            asm volatile("s_barrier" ::: "memory");    // output tile written
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");    // buffer free, next image landed
        } else {
            // the fragment addresses are the same for every clip: hidden from loop-invariant code motion, which would
            // otherwise keep ~170 of them in registers across the persistent loop (347 spilled VGPRs)
            const bf16_t* wbase = wbase0;
            const bf16_t* wtbase = wtbase0;
            asm volatile("" : "+v"(wbase), "+v"(wtbase));
            auto wfrag = [&](int s, int plane) -> bf16x8 {
                return *reinterpret_cast<const bf16x8*>(wbase + (size_t(s) * NT * 2 + plane) * 512);
            };
            auto wtfrag = [&](int q, int plane) -> bf16x8 {
                return *reinterpret_cast<const bf16x8*>(wtbase + (size_t(q) * 4 * 2 + plane) * 512);
            };
            f32x16 acc1 = f32x16{0}, acc2 = f32x16{0};
            f32x4 tacc1 = {0.f, 0.f, 0.f, 0.f}, tacc2 = {0.f, 0.f, 0.f, 0.f};
            bf16x8 ring[D][2], tring[DT][2], taf[2], af[2][2];
#pragma unroll
            for (int i = 0; i < D; ++i) { ring[i][0] = wfrag(i, 0); ring[i][1] = wfrag(i, 1); }
            if (tailw) {
#pragma unroll
                for (int i = 0; i < DT; ++i) { tring[i][0] = wtfrag(i, 0); tring[i][1] = wtfrag(i, 1); }
            }
            int ttadr = 0, tadr = 0;
            auto tfrag = [&](auto qc) {
                constexpr int q = decltype(qc)::value;
                constexpr bool conv1 = q < KS1 / 2, proj = !conv1 && q < (KS1 + KSP) / 2;
                constexpr int kt = conv1 ? q * 32 : proj ? (q - KS1 / 2) * 32 : (q - (KS1 + KSP) / 2) * 32;
                constexpr int C = (conv1 || proj) ? CIN : COUT, CP = (conv1 || proj) ? CPX : CPH;
                constexpr int tap = proj ? 4 : kt / C, c32 = (kt % C) / 32, kh = tap / 3, kw = tap % 3;
                if constexpr (c32 == 0) {
                    if constexpr (conv1 || proj) {
                        const int ih = 2 * toh - 1 + kh, iw = 2 * tow - 1 + kw;
                        const bool ok = unsigned(ih) < unsigned(XH) && unsigned(iw) < unsigned(XW);
                        ttadr = ok ? tpx1 + tapx(kh, kw) * 16 : ZX + tq * CPX;
                    } else {
                        const int ih = toh - 1 + kh, iw = tow - 1 + kw;
                        const bool ok = unsigned(ih) < unsigned(OH) && unsigned(iw) < unsigned(OW);
                        ttadr = ok ? tph1 + ((kh - 1) * OW + kw - 1) * 16 : ZH + tq * CPH;
                    }
                }
                const char* p = sm + ttadr + 4 * c32 * CP;
                taf[0] = *reinterpret_cast<const bf16x8*>(p);
                taf[1] = *reinterpret_cast<const bf16x8*>(p + PL);
            };
            auto aaddr = [&](auto sc) -> const char* {
                constexpr int s = decltype(sc)::value;
                constexpr bool conv1 = s < KS1, proj = !conv1 && s < KS1 + KSP;
                constexpr int kt = conv1 ? s * 16 : proj ? (s - KS1) * 16 : (s - KS1 - KSP) * 16;
                constexpr int C = (conv1 || proj) ? CIN : COUT, CP = (conv1 || proj) ? CPX : CPH;
                constexpr int tap = proj ? 4 : kt / C, c16 = (kt % C) / 16, kh = tap / 3, kw = tap % 3;
                if constexpr (c16 == 0) {
                    if constexpr (conv1 || proj) {
                        const int ih = 2 * goh - 1 + kh, iw = 2 * gow - 1 + kw;
                        const bool ok = unsigned(ih) < unsigned(XH) && unsigned(iw) < unsigned(XW);
                        tadr = ok ? px1 + tapx(kh, kw) * 16 : ZX + h * CPX;
                    } else {
                        const int ih = goh - 1 + kh, iw = gow - 1 + kw;
                        const bool ok = unsigned(ih) < unsigned(OH) && unsigned(iw) < unsigned(OW);
                        tadr = ok ? ph1 + ((kh - 1) * OW + kw - 1) * 16 : ZH + h * CPH;
                    }
                }
                return sm + tadr + 2 * c16 * CP;
            };
            {
                const char* p0 = aaddr(std::integral_constant<int, 0>{});
                af[0][0] = *reinterpret_cast<const bf16x8*>(p0);
                af[0][1] = *reinterpret_cast<const bf16x8*>(p0 + PL);
            }
            auto step = [&]<int s>() {
                if constexpr (s == KS1 + KSP) {
                    // ---- x is dead: h = ReLU(conv1 + b1), split, goes over the x planes ----
                    __syncthreads();
                    if (tid < 2 * CHO) *reinterpret_cast<uint4*>(sm + (tid / CHO) * PL + (tid % CHO) * CPH + ZH) = make_uint4(0, 0, 0, 0);
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int n0 = ng * 32 + 8 * gq + 4 * h;
                        const float4 bb = *reinterpret_cast<const float4*>(lbias + n0);
                        uint2 hi, lo;
                        split4(fmaxf(acc1[4 * gq] + bb.x, 0.f), fmaxf(acc1[4 * gq + 1] + bb.y, 0.f),
                               fmaxf(acc1[4 * gq + 2] + bb.z, 0.f), fmaxf(acc1[4 * gq + 3] + bb.w, 0.f), hi, lo);
                        const int off = (n0 >> 3) * CPH + R * 16 + h * 8;
                        *reinterpret_cast<uint2*>(sm + off) = hi;
                        *reinterpret_cast<uint2*>(sm + off + PL) = lo;
                    }
                    if (tailw) {
                        const int n0 = 16 * wave + 4 * tq;
                        const float4 bb = *reinterpret_cast<const float4*>(lbias + n0);
                        uint2 hi, lo;
                        split4(fmaxf(tacc1[0] + bb.x, 0.f), fmaxf(tacc1[1] + bb.y, 0.f), fmaxf(tacc1[2] + bb.z, 0.f),
                               fmaxf(tacc1[3] + bb.w, 0.f), hi, lo);
                        if (rokt) {
                            const int off = (n0 >> 3) * CPH + Rt * 16 + ((n0 >> 2) & 1) * 8;
                            *reinterpret_cast<uint2*>(sm + off) = hi;
                            *reinterpret_cast<uint2*>(sm + off + PL) = lo;
                        }
                    }
                    __syncthreads();
                    const char* p0 = aaddr(std::integral_constant<int, s>{});
                    af[s & 1][0] = *reinterpret_cast<const bf16x8*>(p0);
                    af[s & 1][1] = *reinterpret_cast<const bf16x8*>(p0 + PL);
                }
                const bf16x8 whi = ring[s % D][0], wlo = ring[s % D][1];
                {
                    constexpr bool pf = s + 1 < KS && s + 1 != KS1 + KSP;
                    const bf16x8 cur_hi = af[s & 1][0], cur_lo = af[s & 1][1];
                    const char* np = nullptr;
                    if constexpr (pf) np = aaddr(std::integral_constant<int, s + 1>{});
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (s < KS1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi, cur_hi, acc1, 0, 0, 0);
                    else acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi, cur_hi, acc2, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (pf) af[(s + 1) & 1][0] = *reinterpret_cast<const bf16x8*>(np);
                    if constexpr (s + D < KS) ring[s % D][0] = wfrag(s + D, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (s < KS1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi, cur_lo, acc1, 0, 0, 0);
                    else acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi, cur_lo, acc2, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (pf) af[(s + 1) & 1][1] = *reinterpret_cast<const bf16x8*>(np + PL);
                    if constexpr (s + D < KS) ring[s % D][1] = wfrag(s + D, 1);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (s < KS1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo, cur_hi, acc1, 0, 0, 0);
                    else acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo, cur_hi, acc2, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (tailw) {   // wave-uniform
                    constexpr int q = s / 2;
                    if constexpr (s % 2 == 0) {
                        tfrag(std::integral_constant<int, q>{});
                    } else {
                        const bf16x8 twhi = tring[q % DT][0], twlo = tring[q % DT][1];
                        if constexpr (q + DT < KS / 2) { tring[q % DT][0] = wtfrag(q + DT, 0); tring[q % DT][1] = wtfrag(q + DT, 1); }
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (s < KS1) {
                            tacc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(twhi, taf[0], tacc1, 0, 0, 0);
                            tacc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(twhi, taf[1], tacc1, 0, 0, 0);
                            tacc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(twlo, taf[0], tacc1, 0, 0, 0);
                        } else {
                            tacc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(twhi, taf[0], tacc2, 0, 0, 0);
                            tacc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(twhi, taf[1], tacc2, 0, 0, 0);
                            tacc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(twlo, taf[0], tacc2, 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            };
            [&]<int... Ss>(std::integer_sequence<int, Ss...>) {
                (step.template operator()<Ss>(), ...);
            }(std::make_integer_sequence<int, KS>{});

            // ---- epilogue: out = ReLU(conv2 + projection + b2) -> f32 [pixel][COUT] tile over the (dead) planes ----
            __syncthreads();
            float* otile = reinterpret_cast<float*>(sm);
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int n0 = ng * 32 + 8 * gq + 4 * h;
                const float4 bb = *reinterpret_cast<const float4*>(lbias + COUT + n0);
                *reinterpret_cast<float4*>(otile + R * OP + n0) =
                    make_float4(fmaxf(acc2[4 * gq] + bb.x, 0.f), fmaxf(acc2[4 * gq + 1] + bb.y, 0.f),
                                fmaxf(acc2[4 * gq + 2] + bb.z, 0.f), fmaxf(acc2[4 * gq + 3] + bb.w, 0.f));
            }
            if (tailw) {
                const int n0 = 16 * wave + 4 * tq;
                const float4 bb = *reinterpret_cast<const float4*>(lbias + COUT + n0);
                if (rokt)
                    *reinterpret_cast<float4*>(otile + Rt * OP + n0) =
                        make_float4(fmaxf(tacc2[0] + bb.x, 0.f), fmaxf(tacc2[1] + bb.y, 0.f), fmaxf(tacc2[2] + bb.z, 0.f),
                                    fmaxf(tacc2[3] + bb.w, 0.f));
            }
            __syncthreads();
            {
                constexpr int nvec = PER * (COUT / 4);
                float4* o = reinterpret_cast<float4*>(a.out + clip * (long long)(PER * COUT));
                for (int p = tid; p < nvec; p += CT) {
                    const int row = p / (COUT / 4), c4 = p % (COUT / 4);
                    o[p] = *reinterpret_cast<const float4*>(otile + row * OP + 4 * c4);
                }
            }
            __syncthreads();   // the buffer is free; the loader has seen the next image land
        }
        if (next >= a.n_clips) break;
        clip = next;
        cur ^= 1;
    }
}

}  // namespace
}  // namespace cough
