// K3/K4, split-bf16 ("bf16x3") -- fused residual block for gfx950 whose logits stay within 1e-3 of the f32
// reference at a trained head's scale (plain bf16 operands: 5.5e-2, profiles/r02_precision_bf16_baseline.txt).
//
// Replaces ResidualBlock.forward (/root/reference/src/model.py:285-293) with the projection skip of :280-283,
// BatchNorm folded.  Every operand of every product -- activations and BN-folded weights -- is carried as a
// pair of bf16 values x = hi + lo (hi = bf16(x), lo = bf16(x - hi): 16 significant bits) and every k-step is
// three v_mfma_f32_32x32x16_bf16 into one f32 accumulator: hi*hi + hi*lo + lo*hi (the lo*lo term, 2^-18 relative,
// is dropped).  Activations live in HBM as f32 (NHWC) and are split when they enter LDS.
//
// One 4-wave workgroup = G clips, <= 75 KB of LDS, so TWO workgroups share a CU and one's staging / epilogue
// overlaps the other's MFMA phases.
//   * x is staged once into two un-bordered LDS planes (hi, lo) of 16-byte channel chunks, XOR-swizzled so that
//     ds_read_b128 of 16 pixels is conflict-free for stride-1 AND stride-2 taps; taps outside the image read a
//     zero pixel kept at the end of each plane (no border, no predicated fragments).
//   * conv1 (3x3 s2) accumulates into acc1; the 1x1 s2 projection of x then opens conv2's accumulator acc2, so x is
//     dead afterwards and h = ReLU(conv1 + b1) is written (split) OVER the x planes; conv2 (3x3 s1) runs out of h.
//   * wave (mg, ng) owns up to MW 32-pixel tiles x one 32-channel tile; weight fragments (hi, lo) stream from L2 in
//     fragment order through a register ring; activation fragments are fetched one k-step ahead, pinned in front of
//     their MFMAs.  MFMA operands are swapped (weights = A): a lane owns one pixel x 4 consecutive channels per
//     accumulator quad, so h and the output tile are written with 8 / 16-byte LDS stores.
//   * epilogue: ReLU(acc2 + b2) -> f32 [pixel][COUT] tile in LDS -> one contiguous run of 16-byte global stores;
//     block 1 also finishes the head (global mean -> Linear(128, 2) -> softmax / argmax, model.py:242-265).
#pragma once
#include <utility>

#include "common.h"
#include "nn_common.h"

namespace cough {
namespace {

struct RbxArgs {
    const float* x;       // [B][XH][XW][CIN] f32 NHWC
    int n_clips;
    const bf16_t* wf;     // MFMA fragments [KS][NT][2 = hi, lo][64 lanes][8]; k-steps: conv1 (9*CIN/16), projection (CIN/16),
                          // conv2 (9*COUT/16); lane (r, h) of (step s, tile t) holds W[32t + r][16s + 8h .. +7] of its operand
    const float* b1;      // [COUT] folded conv1 bias
    const float* b2;      // [COUT] folded conv2 bias + projection bias
    float* out;           // [B][OH][OW][COUT] f32 NHWC, or nullptr (pipeline: only the fused head reads block 1's output)
    const float* fcw;     // fused head (block 1; nullptr: none): [2][COUT]
    const float* fcb;     // [2]
    float* logits;        // [B][2]
    float* probs;         // [B][2] or nullptr
    int* preds;           // [B] or nullptr
};

template <int CIN, int COUT, int G, int XH, int XW>
struct RbxCfg {
    static constexpr int WAVES = 4, THREADS = 256;
    static constexpr int OH = (XH - 1) / 2 + 1, OW = (XW - 1) / 2 + 1;
    static constexpr int NPX = XH * XW, PER = OH * OW, M = G * PER;
    static constexpr int NT = COUT / 32, MG = WAVES / NT, TILES = (M + 31) / 32, MW = (TILES + MG - 1) / MG;
    static constexpr int KS1 = 9 * CIN / 16, KSP = CIN / 16, KS2 = 9 * COUT / 16, KS = KS1 + KSP + KS2;
    static constexpr int CHI = CIN / 8, CHO = COUT / 8;
    static constexpr int DATA = ((G * NPX * CIN * 2 + 255) / 256) * 256;   // bytes of one x plane
    static constexpr int PL = DATA + 256;                                    // plane pitch: data + the zero pixel
    static constexpr int ZOFF = DATA;                                        // zero pixel (same offset in both planes)
    static constexpr int BIAS = 2 * PL;                                      // b1[COUT], b2[COUT] f32
    static constexpr int HRED = BIAS + 2 * COUT * 4;                         // head reduction scratch [WAVES][2] f32
    static constexpr int LDS = HRED + WAVES * 2 * 4;
    static constexpr int OP = COUT + 4;                                      // floats per row of the f32 output tile
    static_assert(WAVES % NT == 0 && MG * MW * 32 >= M, "tile split");
    static_assert(G * PER * COUT * 2 <= DATA, "the h planes lie over the x planes");
    static_assert(M * OP * 4 <= 2 * PL, "the output tile lies over the planes");
    static_assert(PL < 65536, "the lo plane is addressed by a 16-bit immediate offset");
    static_assert((G * NPX * CIN / 4 + THREADS - 1) / THREADS <= 18, "staging registers");
    static_assert(LDS * 2 <= 160 * 1024, "two workgroups per CU");
};

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
// byte offset of 16-byte chunk L of a plane: chunks are laid out linearly in 256-byte bank rows and the position
// inside a row is XORed with the row index -- 16 pixels at stride 1 or 2 then touch 16 different 16-byte slots
__device__ __forceinline__ int swz16(int L) { return ((L & ~15) | ((L ^ (L >> 4)) & 15)) << 4; }

template <int CIN, int COUT, int G, int XH, int XW>
__global__ __launch_bounds__(256, 2) void resblock_x3_kernel(RbxArgs a) {
    using Cfg = RbxCfg<CIN, COUT, G, XH, XW>;
    constexpr int THREADS = Cfg::THREADS, OH = Cfg::OH, OW = Cfg::OW, NPX = Cfg::NPX, PER = Cfg::PER, M = Cfg::M;
    constexpr int NT = Cfg::NT, MW = Cfg::MW, KS1 = Cfg::KS1, KSP = Cfg::KSP, KS = Cfg::KS;
    constexpr int CHI = Cfg::CHI, CHO = Cfg::CHO, PL = Cfg::PL, ZOFF = Cfg::ZOFF, OP = Cfg::OP;
    constexpr int D = 4;   // weight prefetch depth (k-steps; one k-step = MW x 3 MFMAs >= 192 cycles)
    extern __shared__ __attribute__((aligned(256))) char smem[];
    float* lbias = reinterpret_cast<float*>(smem + Cfg::BIAS);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int ng = wave % NT, mg = wave / NT;
    const int clip0 = blockIdx.x * G;
    const int nvalid = a.n_clips - clip0 < G ? a.n_clips - clip0 : G;

    // ---- weight fragment stream (hi, lo per k-step) -------------------------------------------------------
    const bf16_t* wbase = a.wf + size_t(ng) * 1024 + lane * 8;
    auto wfrag = [&](int s, int plane) -> bf16x8 {
        return *reinterpret_cast<const bf16x8*>(wbase + (size_t(s) * NT * 2 + plane) * 512);
    };
    bf16x8 ring[D][2];
#pragma unroll
    for (int i = 0; i < D; ++i) { ring[i][0] = wfrag(i, 0); ring[i][1] = wfrag(i, 1); }

    // ---- stage: the clips' x (f32) is one linear run of 16-byte pieces = 4 channels of one pixel; all loads are
    // issued first, then each piece is split and its hi / lo halves go to the swizzled chunk of the two planes ----
    {
        constexpr int NPIECE = G * NPX * CIN / 4, UN = (NPIECE + THREADS - 1) / THREADS, QP = CIN / 4;
        const int valid = nvalid * NPX * QP;
        const float4* src = reinterpret_cast<const float4*>(a.x + (long long)clip0 * NPX * CIN);
        float4 v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int i = tid + u * THREADS;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < valid) v[u] = src[i];
        }
        if (tid < COUT) { lbias[tid] = a.b1[tid]; lbias[COUT + tid] = a.b2[tid]; }
        if (tid < 32) {   // the zero pixel of both planes (256 B each)
            *reinterpret_cast<uint4*>(smem + ZOFF + (tid & 15) * 16 + (tid >> 4) * PL) = make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int i = tid + u * THREADS;
            if (i < NPIECE) {
                const int P = i / QP, q = i % QP;
                uint2 hi, lo;
                split4(v[u].x, v[u].y, v[u].z, v[u].w, hi, lo);
                const int off = swz16(P * CHI + (q >> 1)) + (q & 1) * 8;
                *reinterpret_cast<uint2*>(smem + off) = hi;
                *reinterpret_cast<uint2*>(smem + off + PL) = lo;
            }
        }
    }
    float fw0 = 0.f, fw1 = 0.f, fb0 = 0.f, fb1 = 0.f;   // fused head (block 1): Linear(128, 2) weights of channel tid & 127
    if constexpr (COUT == 128) {
        if (a.fcw != nullptr) {
            fw0 = a.fcw[tid & 127]; fw1 = a.fcw[128 + (tid & 127)];
            fb0 = a.fcb[0]; fb1 = a.fcb[1];
        }
    }

    // ---- per-lane geometry: lane r owns output pixel R of each of its tiles ---------------------------------
    int goh[MW], gow[MW], px1[MW], ph1[MW];
    bool rok[MW];
#pragma unroll
    for (int mt = 0; mt < MW; ++mt) {
        const int R = (mg * MW + mt) * 32 + r;
        rok[mt] = R < M;
        const int Rc = rok[mt] ? R : 0;
        const int g = Rc / PER, rem = Rc % PER;
        goh[mt] = rok[mt] ? rem / OW : -4;   // -4: every tap of a padding row is out of range -> the zero pixel
        gow[mt] = rem % OW;
        px1[mt] = (g * XH + 2 * goh[mt] - 1) * XW + 2 * gow[mt] - 1;   // x pixel of conv1 tap (0, 0)
        ph1[mt] = (g * OH + goh[mt] - 1) * OW + gow[mt] - 1;           // h pixel of conv2 tap (0, 0)
    }
    __syncthreads();

    auto body = [&]<int MWX>() {
        f32x16 acc1[MWX], acc2[MWX];
#pragma unroll
        for (int mt = 0; mt < MWX; ++mt) { acc1[mt] = f32x16{0}; acc2[mt] = f32x16{0}; }

        // Activation fragments of k-step s (compile-time at every call site).  The pixel-dependent part of the
        // swizzled address is computed when the first k-step of a tap is fetched; further k-steps of the tap cost
        // one xor.  The lo plane is the same address + PL (an immediate).
        int tadr[MWX];
        auto afrag = [&](auto sc, int mt, bf16x8& fhi, bf16x8& flo) {
            constexpr int s = decltype(sc)::value;
            constexpr bool conv1 = s < KS1, proj = !conv1 && s < KS1 + KSP;
            constexpr int kt = conv1 ? s * 16 : proj ? (s - KS1) * 16 : (s - KS1 - KSP) * 16;   // k inside this operand
            constexpr int C = (conv1 || proj) ? CIN : COUT, CH = C / 8;
            constexpr int tap = proj ? 4 : kt / C, c16 = (kt % C) / 16, kh = tap / 3, kw = tap % 3;
            if constexpr (c16 == 0) {
                int P;
                bool ok;
                if constexpr (conv1 || proj) {
                    const int ih = 2 * goh[mt] - 1 + kh, iw = 2 * gow[mt] - 1 + kw;
                    ok = unsigned(ih) < unsigned(XH) && unsigned(iw) < unsigned(XW);
                    P = px1[mt] + kh * XW + kw;
                } else {
                    const int ih = goh[mt] - 1 + kh, iw = gow[mt] - 1 + kw;
                    ok = unsigned(ih) < unsigned(OH) && unsigned(iw) < unsigned(OW);
                    P = ph1[mt] + kh * OW + kw;
                }
                tadr[mt] = ok ? swz16(P * CH + h) : ZOFF + h * 16;
            }
            const char* p = smem + (tadr[mt] ^ (32 * c16));
            fhi = *reinterpret_cast<const bf16x8*>(p);
            flo = *reinterpret_cast<const bf16x8*>(p + PL);
        };

        bf16x8 af[2][MWX][2];
#pragma unroll
        for (int mt = 0; mt < MWX; ++mt) afrag(std::integral_constant<int, 0>{}, mt, af[0][mt][0], af[0][mt][1]);

        auto step = [&]<int s>() {
            if constexpr (s == KS1 + KSP) {
                // ---- x is dead: h = ReLU(conv1 + b1), split, goes over the x planes ----
                __syncthreads();
#pragma unroll
                for (int mt = 0; mt < MWX; ++mt) {
                    const int R = (mg * MW + mt) * 32 + r;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int n0 = ng * 32 + 8 * gq + 4 * h;
                        const float4 bb = *reinterpret_cast<const float4*>(lbias + n0);
                        uint2 hi, lo;
                        split4(fmaxf(acc1[mt][4 * gq] + bb.x, 0.f), fmaxf(acc1[mt][4 * gq + 1] + bb.y, 0.f),
                               fmaxf(acc1[mt][4 * gq + 2] + bb.z, 0.f), fmaxf(acc1[mt][4 * gq + 3] + bb.w, 0.f), hi, lo);
                        if (rok[mt]) {
                            const int off = swz16(R * CHO + (n0 >> 3)) + h * 8;
                            *reinterpret_cast<uint2*>(smem + off) = hi;
                            *reinterpret_cast<uint2*>(smem + off + PL) = lo;
                        }
                    }
                }
                __syncthreads();
#pragma unroll
                for (int mt = 0; mt < MWX; ++mt)
                    afrag(std::integral_constant<int, s>{}, mt, af[s & 1][mt][0], af[s & 1][mt][1]);
            }
            // Software pipeline, pinned with scheduling barriers: ahead of the three MFMAs of tile mt sit the address
            // math + two ds_reads of the NEXT step's fragments of tile mt (and, once per step, the weight loads D steps
            // ahead); left alone, the scheduler sinks every ds_read next to its MFMA and each pays the LDS latency.
            const bf16x8 whi = ring[s % D][0], wlo = ring[s % D][1];
#pragma unroll
            for (int mt = 0; mt < MWX; ++mt) {
                if constexpr (s + 1 < KS && s + 1 != KS1 + KSP)
                    afrag(std::integral_constant<int, s + 1>{}, mt, af[(s + 1) & 1][mt][0], af[(s + 1) & 1][mt][1]);
                if constexpr (s + D < KS) {
                    if (mt == 0) { ring[s % D][0] = wfrag(s + D, 0); ring[s % D][1] = wfrag(s + D, 1); }
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (s < KS1) {
                    acc1[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi, af[s & 1][mt][0], acc1[mt], 0, 0, 0);
                    acc1[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi, af[s & 1][mt][1], acc1[mt], 0, 0, 0);
                    acc1[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo, af[s & 1][mt][0], acc1[mt], 0, 0, 0);
                } else {
                    acc2[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi, af[s & 1][mt][0], acc2[mt], 0, 0, 0);
                    acc2[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi, af[s & 1][mt][1], acc2[mt], 0, 0, 0);
                    acc2[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo, af[s & 1][mt][0], acc2[mt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        [&]<int... Ss>(std::integer_sequence<int, Ss...>) {
            (step.template operator()<Ss>(), ...);
        }(std::make_integer_sequence<int, KS>{});

        // ---- epilogue: out = ReLU(conv2 + projection + b2) -> f32 [pixel][COUT] tile over the (dead) planes ----
        __syncthreads();
        float* otile = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int mt = 0; mt < MWX; ++mt) {
            const int R = (mg * MW + mt) * 32 + r;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int n0 = ng * 32 + 8 * gq + 4 * h;
                const float4 bb = *reinterpret_cast<const float4*>(lbias + COUT + n0);
                const float4 o = make_float4(fmaxf(acc2[mt][4 * gq] + bb.x, 0.f), fmaxf(acc2[mt][4 * gq + 1] + bb.y, 0.f),
                                             fmaxf(acc2[mt][4 * gq + 2] + bb.z, 0.f), fmaxf(acc2[mt][4 * gq + 3] + bb.w, 0.f));
                if (rok[mt]) *reinterpret_cast<float4*>(otile + R * OP + n0) = o;
            }
        }
    };
    // a wave whose last tile lies entirely beyond the M valid rows runs the shorter body (one wave-uniform choice)
    constexpr int TILES = Cfg::TILES;
    const int mytiles = TILES - mg * MW < MW ? TILES - mg * MW : MW;
    if constexpr (MW > 1 && TILES % MW != 0) {
        if (mytiles < MW) body.template operator()<(TILES % MW)>();
        else body.template operator()<MW>();
    } else {
        body.template operator()<MW>();
    }
    __syncthreads();
    const float* otile = reinterpret_cast<const float*>(smem);
    if (a.out != nullptr) {
        const int nvec = nvalid * PER * (COUT / 4);
        float4* o = reinterpret_cast<float4*>(a.out + (long long)clip0 * PER * COUT);
        for (int p = tid; p < nvec; p += THREADS) {
            const int row = p / (COUT / 4), c4 = p % (COUT / 4);
            o[p] = *reinterpret_cast<const float4*>(otile + row * OP + 4 * c4);
        }
    }
    if constexpr (COUT == 128) {
        // ---- fused head (model.py:242-247, :257-265): same summation order as tail_kernel (pixels in order, then
        // the lanes of a wave, then the two waves of a clip) ----
        static_assert(G * 128 == THREADS, "one thread per (clip, channel)");
        if (a.fcw != nullptr) {
            float* hred = reinterpret_cast<float*>(smem + Cfg::HRED);
            const int c = tid & 127, g = tid >> 7;
            float sum = 0.f;
#pragma unroll 6
            for (int i = 0; i < PER; ++i) sum += otile[(g * PER + i) * OP + c];
            const float mean = sum / float(PER);
            float l0 = wave_sum(mean * fw0), l1 = wave_sum(mean * fw1);
            if (lane == 0) { hred[wave * 2] = l0; hred[wave * 2 + 1] = l1; }
            __syncthreads();
            if (c == 0 && g < nvalid) {
                const int w0 = g * 2;   // the clip's two waves
                l0 = hred[w0 * 2] + hred[(w0 + 1) * 2] + fb0;
                l1 = hred[w0 * 2 + 1] + hred[(w0 + 1) * 2 + 1] + fb1;
                const long long b = clip0 + g;
                a.logits[b * 2] = l0;
                a.logits[b * 2 + 1] = l1;
                if (a.probs) {
                    const float mx = fmaxf(l0, l1), e0 = expf(l0 - mx), e1 = expf(l1 - mx), inv = 1.0f / (e0 + e1);
                    a.probs[b * 2] = e0 * inv;
                    a.probs[b * 2 + 1] = e1 * inv;
                }
                if (a.preds) a.preds[b] = (l1 > l0) ? 1 : 0;
            }
        }
    }
}

// Host: folded [N][K] weights of conv1, projection and conv2 -> split-bf16 MFMA fragments in stream order.
inline void pack_x3_fragments(std::vector<bf16_t>& wf, const std::vector<float>& w1, int K1, const std::vector<float>& wp,
                              int KP, const std::vector<float>& w2, int K2, int N) {
    const int nt = N / 32, ks1 = K1 / 16, ksp = KP / 16, ks2 = K2 / 16, ks = ks1 + ksp + ks2;
    wf.assign(size_t(ks) * nt * 2 * 512, 0);
    for (int s = 0; s < ks; ++s) {
        const std::vector<float>& w = s < ks1 ? w1 : s < ks1 + ksp ? wp : w2;
        const int K = s < ks1 ? K1 : s < ks1 + ksp ? KP : K2;
        const int sl = s < ks1 ? s : s < ks1 + ksp ? s - ks1 : s - ks1 - ksp;
        for (int t = 0; t < nt; ++t)
            for (int lane = 0; lane < 64; ++lane)
                for (int jj = 0; jj < 8; ++jj) {
                    const float v = w[size_t(32 * t + (lane & 31)) * K + 16 * sl + 8 * (lane >> 5) + jj];
                    const bf16_t hi = f2bf_host(v);
                    uint32_t hb = uint32_t(hi) << 16;
                    float hf;
                    std::memcpy(&hf, &hb, 4);
                    const size_t base = ((size_t(s) * nt + t) * 2) * 512 + size_t(lane) * 8 + jj;
                    wf[base] = hi;
                    wf[base + 512] = f2bf_host(v - hf);
                }
    }
}

}  // namespace
}  // namespace cough
