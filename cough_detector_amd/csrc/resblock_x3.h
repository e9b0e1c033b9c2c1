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
//   * x is staged once into two un-bordered LDS planes (hi, lo).  A plane is chunk-planar -- [8-channel chunk][pixel]
//     of 16-byte cells -- and its pixels are stored parity-split (even / odd image rows x even / odd columns, each
//     sub-image with the row pitch of the OUTPUT image), so the 32 output pixels of a tile read, for any tap of the
//     stride-2 conv1 as for the stride-1 conv2 over h, 32 CONSECUTIVE cells of one chunk plane: the 16-lane groups of
//     a ds_read_b128 cover all 16 residues mod 16, i.e. every fragment read is bank-conflict-free at any alignment
//     (the XOR-swizzled pixel-major layout measured 2.2x / 1.9x the conflict-free LDS cycles on conv1), and a tap's
//     address is the tile's base + a compile-time constant.  Taps outside the image read a zero cell kept at the end
//     of each chunk plane (no border, no predicated fragments).
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

#ifdef COUGH_K1_STAMPS
// diagnostic build only (tools/rb_stamps.py, tools/rbx_stamps.py): per-workgroup s_memtime at phase boundaries, written to
// a buffer of its own that no other code reads.  Never compiled into libcough_amd.so.
__device__ unsigned long long* g_rb_stamp_buf = nullptr;
#define RB_STAMP(slot)                                                                        \
    do {                                                                                      \
        if (g_rb_stamp_buf && threadIdx.x == 0)                                               \
            g_rb_stamp_buf[(size_t)blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memtime();   \
    } while (0)
#else
#define RB_STAMP(slot) do { } while (0)
#endif

struct RbxArgs {
    const float* x;       // [B][XH][XW][CIN] f32 NHWC
    int n_clips;
    const bf16_t* wf;     // MFMA fragments [KS][NT][2 = hi, lo][64 lanes][8]; k-steps: conv1 (9*CIN/16), projection (CIN/16),
                          // conv2 (9*COUT/16); lane (r, h) of (step s, tile t) holds W[32t + r][16s + 8h .. +7] of its operand
    const bf16_t* wt;     // TAIL: the same weights as 16x16x32 fragments [KS/2][COUT/16][2][64 lanes][8]: lane l of (k32-step q,
                          // tile t) holds W[16t + (l & 15)][32q + 8(l >> 4) .. +7]
    const float* b1;      // [COUT] folded conv1 bias
    const float* b2;      // [COUT] folded conv2 bias + projection bias
    float* out;           // [B][OH][OW][COUT] f32 NHWC, or nullptr (pipeline: only the fused head reads block 1's output)
    const float* fcw;     // fused head (block 1; nullptr: none): [2][COUT]
    const float* fcb;     // [2]
    float* logits;        // [B][2]
    float* probs;         // [B][2] or nullptr
    int* preds;           // [B] or nullptr
    const int* nanflag;   // fused head: [B] or nullptr; 1 = the clip's image holds a NaN -> NaN logits (nn_common.h: NaN rule)
};

template <int CIN, int COUT, int G, int XH, int XW>
struct RbxCfg {
    static constexpr int WAVES = 4, THREADS = 256;
    static constexpr int OH = (XH - 1) / 2 + 1, OW = (XW - 1) / 2 + 1;
    static constexpr int NPX = XH * XW, PER = OH * OW, M = G * PER;
    static constexpr int NT = COUT / 32, MG = WAVES / NT, TILES = (M + 31) / 32;
    // TAIL: the last M % 32 <= 16 rows are not padded to a 32-row tile (block 0: 143 rows = 4 tiles + 15 rows would
    // waste 17 of 160 rows AND leave the waves with 3 / 3 / 2 / 2 tiles); they form one 16-row tile that the four waves
    // share by channels (16 each) on v_mfma_f32_16x16x32_bf16: every wave then owns FULL / MG 32x32 tiles + one
    // 16x16 tile -- the same work for each, 143 of 144 rows useful.
    static constexpr bool TAIL = M % 32 != 0 && M % 32 <= 16 && COUT / 16 == WAVES && (M / 32) % MG == 0 &&
                                 (9 * CIN / 16) % 2 == 0 && (CIN / 16) % 2 == 0;
    static constexpr int FULL = TAIL ? M / 32 : TILES;                 // 32-row tiles
    static constexpr int MW = (FULL + MG - 1) / MG;
    static constexpr int KS1 = 9 * CIN / 16, KSP = CIN / 16, KS2 = 9 * COUT / 16, KS = KS1 + KSP + KS2;
    static constexpr int CHI = CIN / 8, CHO = COUT / 8;
    // x planes: parity-split pixel order.  Sub-image (row parity a, column parity b) holds pixels (2i + a, 2j + b) at
    // i * OW + j; RE / RO = number of even / odd rows; the odd-column sub-images of an odd XW carry one unused column.
    static constexpr int RE = (XH + 1) / 2, RO = XH / 2;
    static constexpr int NPP = 2 * XH * OW;                                  // cells of one clip in one chunk plane
    static constexpr int PB01 = RE * OW, PB10 = 2 * RE * OW, PB11 = 2 * RE * OW + RO * OW;   // sub-image bases (PB00 = 0)
    static constexpr int CPX = (G * NPP + 1) * 16;                           // bytes of one x chunk plane (+ the zero cell)
    static constexpr int CPH = (M + 1) * 16;                                 // bytes of one h chunk plane (+ the zero cell)
    static constexpr int XBYTES = CHI * CPX, HBYTES = CHO * CPH;
    static constexpr int PL = (((XBYTES > HBYTES ? XBYTES : HBYTES) + 255) / 256) * 256;   // pitch between the hi and lo planes
    static constexpr int BIAS = 2 * PL;                                      // b1[COUT], b2[COUT] f32
    static constexpr int HRED = BIAS + 2 * COUT * 4;                         // head reduction scratch [WAVES][2] f32
    static constexpr int LDS = HRED + WAVES * 2 * 4;
    static constexpr int OP = COUT + 4;                                      // floats per row of the f32 output tile
    static_assert(WAVES % NT == 0 && MG * MW * 32 + (TAIL ? 16 : 0) >= M, "tile split");
    static_assert(OW == (XW + 1) / 2, "sub-image row pitch");
    static_assert(M * OP * 4 <= 2 * PL, "the output tile lies over the planes");
    static constexpr int STAGE_MAX = 18;                                     // 16-byte pieces per thread staged in one batch
    static_assert((G * NPX * CIN / 4 + THREADS - 1) / THREADS <= 2 * STAGE_MAX, "staging registers (f32 input, two batches)");
    static_assert(LDS <= 160 * 1024, "one workgroup must fit a CU (two do for the shipped image)");
};

template <int CIN, int COUT, int G, int XH, int XW>
__global__ __launch_bounds__(256, 2) void resblock_x3_kernel(RbxArgs a) {
    using Cfg = RbxCfg<CIN, COUT, G, XH, XW>;
    constexpr int THREADS = Cfg::THREADS, OH = Cfg::OH, OW = Cfg::OW, NPX = Cfg::NPX, PER = Cfg::PER, M = Cfg::M;
    constexpr int NT = Cfg::NT, MW = Cfg::MW, KS1 = Cfg::KS1, KSP = Cfg::KSP, KS = Cfg::KS;
    constexpr bool TAIL = Cfg::TAIL;
    using f32x4 = __attribute__((ext_vector_type(4))) float;
    constexpr int CHI = Cfg::CHI, CHO = Cfg::CHO, PL = Cfg::PL, OP = Cfg::OP, CPX = Cfg::CPX, CPH = Cfg::CPH, NPP = Cfg::NPP;
    constexpr int ZX = G * NPP * 16, ZH = M * 16;   // byte offset of the zero cell inside an x / h chunk plane
#ifndef RBX_D
#define RBX_D 4
#endif
#ifndef RBX_PIN
#define RBX_PIN 1   // (the 16-row tile's pins; the 32-row tiles are always pinned, see the k-loop)
#endif
#ifndef RBX_PRIO
#define RBX_PRIO 0
#endif
    constexpr int D = RBX_D;   // weight prefetch depth (k-steps; one k-step = MW x 3 MFMAs >= 192 cycles)
    extern __shared__ __attribute__((aligned(256))) char smem[];
    float* lbias = reinterpret_cast<float*>(smem + Cfg::BIAS);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int ng = wave % NT, mg = wave / NT;
    const int clip0 = blockIdx.x * G;
    const int nvalid = a.n_clips - clip0 < G ? a.n_clips - clip0 : G;
    RB_STAMP(0);

    // biases first: vector-memory loads return in issue order, so a bias loaded after the staging loads and stored
    // to LDS straight away would drain every load of wave 0 before its first piece is split
    float bv1 = 0.f, bv2 = 0.f;
    if (tid < COUT) { bv1 = a.b1[tid]; bv2 = a.b2[tid]; }

    // ---- weight fragment stream (hi, lo per k-step) -------------------------------------------------------
    const bf16_t* wbase = a.wf + size_t(ng) * 1024 + lane * 8;
    auto wfrag = [&](int s, int plane) -> bf16x8 {
        return *reinterpret_cast<const bf16x8*>(wbase + (size_t(s) * NT * 2 + plane) * 512);
    };
    bf16x8 ring[D][2];
#pragma unroll
    for (int i = 0; i < D; ++i) { ring[i][0] = wfrag(i, 0); ring[i][1] = wfrag(i, 1); }
    // TAIL: this wave's 16 channels (tile `wave`) of the 16-row tile, in k32-steps
    constexpr int DT = D / 2 > 0 ? D / 2 : 1;
    const bf16_t* wtbase = TAIL ? a.wt + size_t(wave) * 1024 + lane * 8 : nullptr;
    auto wtfrag = [&](int q, int plane) -> bf16x8 {
        return *reinterpret_cast<const bf16x8*>(wtbase + (size_t(q) * Cfg::WAVES * 2 + plane) * 512);
    };
    bf16x8 tring[DT][2];
    if constexpr (TAIL) {
#pragma unroll
        for (int i = 0; i < DT; ++i) { tring[i][0] = wtfrag(i, 0); tring[i][1] = wtfrag(i, 1); }
    }
    const int tch = 16 * wave;   // TAIL: the 16 output channels this wave owns in the 16-row tile

    // ---- stage: the clips' x (f32) is one linear run of 16-byte pieces = 4 channels of one pixel; all loads are
    // issued first, then each piece is split and its hi / lo halves go to the swizzled chunk of the two planes ----
    {
        constexpr int NPIECE = G * NPX * CIN / 4, UN = (NPIECE + THREADS - 1) / THREADS, QP = CIN / 4;
        // the shipped image stages in ONE batch (all loads in flight before the first split); a larger image (103- / 110-row
        // features) in two, so that the staging registers stay within the accumulators' budget
        constexpr int NB = UN <= Cfg::STAGE_MAX ? 1 : 2, UNB = (UN + NB - 1) / NB;
        const int valid = nvalid * NPX * QP;
        const float4* src = reinterpret_cast<const float4*>(a.x + (long long)clip0 * NPX * CIN);
        if (tid < 2 * CHI) {   // the zero cell of every chunk plane, hi and lo
            *reinterpret_cast<uint4*>(smem + (tid / CHI) * PL + (tid % CHI) * CPX + ZX) = make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int b0 = 0; b0 < UN; b0 += UNB) {
            float4 v[UNB];
#pragma unroll
            for (int u = 0; u < UNB; ++u) {
                // unconditional, index-clamped loads: a load under a branch makes the compiler drain vmcnt to 0 before the
                // first piece is split; clamped, the pieces are processed as they arrive (vmcnt(UNB - 1 - u))
                const int i = tid + (b0 + u) * THREADS;
                v[u] = src[i < valid ? i : valid - 1];
            }
#pragma unroll
            for (int u = 0; u < UNB; ++u) {
                const int i = tid + (b0 + u) * THREADS;
                if (i < NPIECE) {
                    const int P = i / QP, q = i % QP;                      // raster pixel (clip, row, column), quarter-chunk
                    const int g = P / NPX, rem = P % NPX, ih = rem / XW, iw = rem % XW;
                    const int cell = g * NPP + ((ih & 1) ? ((iw & 1) ? Cfg::PB11 : Cfg::PB10) : ((iw & 1) ? Cfg::PB01 : 0)) +
                                     (ih >> 1) * OW + (iw >> 1);
                    uint2 hi, lo;
                    if (i >= valid) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);   // clips beyond the batch read as zeros
                    split4(v[u].x, v[u].y, v[u].z, v[u].w, hi, lo);
                    const int off = (q >> 1) * CPX + cell * 16 + (q & 1) * 8;
                    *reinterpret_cast<uint2*>(smem + off) = hi;
                    *reinterpret_cast<uint2*>(smem + off + PL) = lo;
                }
            }
        }
    }
    if (tid < COUT) { lbias[tid] = bv1; lbias[COUT + tid] = bv2; }
    float fw0 = 0.f, fw1 = 0.f, fb0 = 0.f, fb1 = 0.f;   // fused head (block 1): Linear(128, 2) weights of channel tid & 127
    if constexpr (COUT == 128) {
        if (a.fcw != nullptr) {
            fw0 = a.fcw[tid & 127]; fw1 = a.fcw[128 + (tid & 127)];
            fb0 = a.fcb[0]; fb1 = a.fcb[1];
        }
    }
    int nan_clip = 0;   // NaN rule (nn_common.h): the verdict on this thread's clip, fetched here so that its latency is long gone
    if constexpr (COUT == 128) {
        if (a.fcw != nullptr && a.nanflag != nullptr && (tid & 127) == 0 && (tid >> 7) < nvalid) nan_clip = a.nanflag[clip0 + (tid >> 7)];
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
        px1[mt] = (g * NPP + rem) * 16 + h * CPX;   // byte offset of x cell (clip, oh, ow) of sub-image (0, 0), this lane's chunk
        ph1[mt] = Rc * 16 + h * CPH;                // byte offset of h cell R, this lane's chunk
    }
    // cell offset of conv1 tap (kh, kw) from cell (oh, ow) of sub-image (0, 0): input (2oh - 1 + kh, 2ow - 1 + kw) lies in
    // the sub-image of parities ((kh + 1) & 1, (kw + 1) & 1) at (oh - [kh == 0], ow - [kw == 0])
    auto tapx = [](int kh, int kw) constexpr -> int {
        const int a = (kh + 1) & 1, b = (kw + 1) & 1;
        return (a ? (b ? Cfg::PB11 : Cfg::PB10) : (b ? Cfg::PB01 : 0)) - (kh == 0 ? Cfg::OW : 0) - (kw == 0 ? 1 : 0);
    };
    // TAIL: lane l owns pixel Rt of the 16-row tile and the 8-channel chunk (l >> 4) of every 32-wide k-step
    const int tq = lane >> 4;
    const int Rt = Cfg::FULL * 32 + (lane & 15);
    const bool rokt = TAIL && Rt < M;
    int toh = -4, tow = 0, tpx1 = 0, tph1 = 0;
    if (rokt) {
        const int g = Rt / PER, rem = Rt % PER;
        toh = rem / OW;
        tow = rem % OW;
        tpx1 = (g * NPP + rem) * 16 + tq * CPX;
        tph1 = Rt * 16 + tq * CPH;
    }
    RB_STAMP(1);
    __syncthreads();
    RB_STAMP(2);
#if RBX_PRIO
    __builtin_amdgcn_s_setprio(1);   // MFMA phases outrank the co-resident workgroup's staging / epilogue VALU work
#endif

    auto body = [&]<int MWX>() {
        f32x16 acc1[MWX], acc2[MWX];
#pragma unroll
        for (int mt = 0; mt < MWX; ++mt) { acc1[mt] = f32x16{0}; acc2[mt] = f32x16{0}; }
        f32x4 tacc1 = {0.f, 0.f, 0.f, 0.f}, tacc2 = {0.f, 0.f, 0.f, 0.f};
        bf16x8 taf[2];
        int ttadr = 0;
        // TAIL fragments of k32-step q (q counts 32-wide steps through conv1, projection, conv2)
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
            const char* p = smem + ttadr + 4 * c32 * CP;
            taf[0] = *reinterpret_cast<const bf16x8*>(p);
            taf[1] = *reinterpret_cast<const bf16x8*>(p + PL);
        };

        // Activation fragments of k-step s (compile-time at every call site).  A tap's address -- the tile's base cell +
        // a compile-time cell offset, or the zero cell -- is chosen when the first k-step of the tap is fetched; the
        // further k-steps (chunk planes) and the lo plane are immediate offsets.
        int tadr[MWX];
        auto afrag = [&](auto sc, int mt, bf16x8& fhi, bf16x8& flo) {
            constexpr int s = decltype(sc)::value;
            constexpr bool conv1 = s < KS1, proj = !conv1 && s < KS1 + KSP;
            constexpr int kt = conv1 ? s * 16 : proj ? (s - KS1) * 16 : (s - KS1 - KSP) * 16;   // k inside this operand
            constexpr int C = (conv1 || proj) ? CIN : COUT, CP = (conv1 || proj) ? CPX : CPH;
            constexpr int tap = proj ? 4 : kt / C, c16 = (kt % C) / 16, kh = tap / 3, kw = tap % 3;
            if constexpr (c16 == 0) {
                if constexpr (conv1 || proj) {
                    const int ih = 2 * goh[mt] - 1 + kh, iw = 2 * gow[mt] - 1 + kw;
                    const bool ok = unsigned(ih) < unsigned(XH) && unsigned(iw) < unsigned(XW);
                    tadr[mt] = ok ? px1[mt] + tapx(kh, kw) * 16 : ZX + h * CPX;
                } else {
                    const int ih = goh[mt] - 1 + kh, iw = gow[mt] - 1 + kw;
                    const bool ok = unsigned(ih) < unsigned(OH) && unsigned(iw) < unsigned(OW);
                    tadr[mt] = ok ? ph1[mt] + ((kh - 1) * OW + kw - 1) * 16 : ZH + h * CPH;
                }
            }
            const char* p = smem + tadr[mt] + 2 * c16 * CP;
            fhi = *reinterpret_cast<const bf16x8*>(p);
            flo = *reinterpret_cast<const bf16x8*>(p + PL);
        };
        // the address part of afrag alone (the k-loop issues the two reads separately, one in front of each MFMA)
        auto aaddr = [&](auto sc, int mt) -> const char* {
            constexpr int s = decltype(sc)::value;
            constexpr bool conv1 = s < KS1, proj = !conv1 && s < KS1 + KSP;
            constexpr int kt = conv1 ? s * 16 : proj ? (s - KS1) * 16 : (s - KS1 - KSP) * 16;
            constexpr int C = (conv1 || proj) ? CIN : COUT, CP = (conv1 || proj) ? CPX : CPH;
            constexpr int tap = proj ? 4 : kt / C, c16 = (kt % C) / 16, kh = tap / 3, kw = tap % 3;
            if constexpr (c16 == 0) {
                if constexpr (conv1 || proj) {
                    const int ih = 2 * goh[mt] - 1 + kh, iw = 2 * gow[mt] - 1 + kw;
                    const bool ok = unsigned(ih) < unsigned(XH) && unsigned(iw) < unsigned(XW);
                    tadr[mt] = ok ? px1[mt] + tapx(kh, kw) * 16 : ZX + h * CPX;
                } else {
                    const int ih = goh[mt] - 1 + kh, iw = gow[mt] - 1 + kw;
                    const bool ok = unsigned(ih) < unsigned(OH) && unsigned(iw) < unsigned(OW);
                    tadr[mt] = ok ? ph1[mt] + ((kh - 1) * OW + kw - 1) * 16 : ZH + h * CPH;
                }
            }
            return smem + tadr[mt] + 2 * c16 * CP;
        };

        bf16x8 af[2][MWX][2];
#pragma unroll
        for (int mt = 0; mt < MWX; ++mt) afrag(std::integral_constant<int, 0>{}, mt, af[0][mt][0], af[0][mt][1]);

        auto step = [&]<int s>() {
            if constexpr (s == KS1 + KSP) {
                // ---- x is dead: h = ReLU(conv1 + b1), split, goes over the x planes ----
                RB_STAMP(3);
#if RBX_PRIO
                __builtin_amdgcn_s_setprio(0);
#endif
                __syncthreads();
                if (tid < 2 * CHO)   // the zero cell of every h chunk plane, hi and lo
                    *reinterpret_cast<uint4*>(smem + (tid / CHO) * PL + (tid % CHO) * CPH + ZH) = make_uint4(0, 0, 0, 0);
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
                            const int off = (n0 >> 3) * CPH + R * 16 + h * 8;
                            *reinterpret_cast<uint2*>(smem + off) = hi;
                            *reinterpret_cast<uint2*>(smem + off + PL) = lo;
                        }
                    }
                }
                if constexpr (TAIL) {
                    const int n0 = tch + 4 * tq;
                    const float4 bb = *reinterpret_cast<const float4*>(lbias + n0);
                    uint2 hi, lo;
                    split4(fmaxf(tacc1[0] + bb.x, 0.f), fmaxf(tacc1[1] + bb.y, 0.f), fmaxf(tacc1[2] + bb.z, 0.f),
                           fmaxf(tacc1[3] + bb.w, 0.f), hi, lo);
                    if (rokt) {
                        const int off = (n0 >> 3) * CPH + Rt * 16 + ((n0 >> 2) & 1) * 8;
                        *reinterpret_cast<uint2*>(smem + off) = hi;
                        *reinterpret_cast<uint2*>(smem + off + PL) = lo;
                    }
                }
                __syncthreads();
                RB_STAMP(4);
#if RBX_PRIO
                __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
                for (int mt = 0; mt < MWX; ++mt)
                    afrag(std::integral_constant<int, s>{}, mt, af[s & 1][mt][0], af[s & 1][mt][1]);
            }
            // Software pipeline, pinned with scheduling barriers: ONE filler sits in front of EACH MFMA of tile mt -- the
            // address math of the next step's fragment, then its hi read (+ the hi weight load D steps ahead), then its lo
            // read (+ the lo weight load) -- so that no filler group is longer than the ~24 issue cycles an MFMA leaves
            // free.  Left alone, the scheduler sinks every ds_read next to its MFMA and each pays the LDS latency; one
            // filler group per MFMA triple (the first version) left the pipe idle behind every third MFMA (+2.3 %).
            const bf16x8 whi = ring[s % D][0], wlo = ring[s % D][1];
#pragma unroll
            for (int mt = 0; mt < MWX; ++mt) {
                constexpr bool pf = s + 1 < KS && s + 1 != KS1 + KSP;
                const bf16x8 cur_hi = af[s & 1][mt][0], cur_lo = af[s & 1][mt][1];
                const char* np = nullptr;
                if constexpr (pf) np = aaddr(std::integral_constant<int, s + 1>{}, mt);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (s < KS1) acc1[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi, cur_hi, acc1[mt], 0, 0, 0);
                else acc2[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi, cur_hi, acc2[mt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (pf) af[(s + 1) & 1][mt][0] = *reinterpret_cast<const bf16x8*>(np);
                if constexpr (s + D < KS) { if (mt == 0) ring[s % D][0] = wfrag(s + D, 0); }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (s < KS1) acc1[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi, cur_lo, acc1[mt], 0, 0, 0);
                else acc2[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi, cur_lo, acc2[mt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (pf) af[(s + 1) & 1][mt][1] = *reinterpret_cast<const bf16x8*>(np + PL);
                if constexpr (s + D < KS) { if (mt == 0) ring[s % D][1] = wfrag(s + D, 1); }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (s < KS1) acc1[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo, cur_hi, acc1[mt], 0, 0, 0);
                else acc2[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo, cur_hi, acc2[mt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (TAIL) {
                // the 16-row tile advances one 32-wide k-step per two 16-wide steps: fragments are fetched in the even
                // step, the three 16x16x32 MFMAs issue in the odd one
                constexpr int q = s / 2;
                if constexpr (s % 2 == 0) {
                    tfrag(std::integral_constant<int, q>{});
                } else {
                    const bf16x8 twhi = tring[q % DT][0], twlo = tring[q % DT][1];
                    if constexpr (q + DT < KS / 2) { tring[q % DT][0] = wtfrag(q + DT, 0); tring[q % DT][1] = wtfrag(q + DT, 1); }
#if RBX_PIN
                    __builtin_amdgcn_sched_barrier(0);
#endif
                    if constexpr (s < KS1) {
                        tacc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(twhi, taf[0], tacc1, 0, 0, 0);
                        tacc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(twhi, taf[1], tacc1, 0, 0, 0);
                        tacc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(twlo, taf[0], tacc1, 0, 0, 0);
                    } else {
                        tacc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(twhi, taf[0], tacc2, 0, 0, 0);
                        tacc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(twhi, taf[1], tacc2, 0, 0, 0);
                        tacc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(twlo, taf[0], tacc2, 0, 0, 0);
                    }
#if RBX_PIN
                    __builtin_amdgcn_sched_barrier(0);
#endif
                }
            }
        };
        [&]<int... Ss>(std::integer_sequence<int, Ss...>) {
            (step.template operator()<Ss>(), ...);
        }(std::make_integer_sequence<int, KS>{});

        // ---- epilogue: out = ReLU(conv2 + projection + b2) -> f32 [pixel][COUT] tile over the (dead) planes ----
        RB_STAMP(5);
#if RBX_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
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
        if constexpr (TAIL) {
            const int n0 = tch + 4 * tq;
            const float4 bb = *reinterpret_cast<const float4*>(lbias + COUT + n0);
            const float4 o = make_float4(fmaxf(tacc2[0] + bb.x, 0.f), fmaxf(tacc2[1] + bb.y, 0.f),
                                         fmaxf(tacc2[2] + bb.z, 0.f), fmaxf(tacc2[3] + bb.w, 0.f));
            if (rokt) *reinterpret_cast<float4*>(otile + Rt * OP + n0) = o;
        }
    };
    // a wave whose last tile lies entirely beyond the M valid rows runs the shorter body (one wave-uniform choice)
    constexpr int TILES = Cfg::FULL;
    const int mytiles = TILES - mg * MW < MW ? TILES - mg * MW : MW;
    if constexpr (MW > 1 && TILES % MW != 0) {
        if (mytiles < MW) body.template operator()<(TILES % MW)>();
        else body.template operator()<MW>();
    } else {
        body.template operator()<MW>();
    }
    __syncthreads();
    RB_STAMP(6);
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
        static_assert(G * 128 <= THREADS, "one thread per (clip, channel)");
        if (a.fcw != nullptr) {
            float* hred = reinterpret_cast<float*>(smem + Cfg::HRED);
            const int c = tid & 127, g = tid >> 7;
            float sum = 0.f;
            if (g < G) {
#pragma unroll 6
                for (int i = 0; i < PER; ++i) sum += otile[(g * PER + i) * OP + c];
            }
            const float mean = sum / float(PER);
            float l0 = wave_sum(mean * fw0), l1 = wave_sum(mean * fw1);
            if (lane == 0) { hred[wave * 2] = l0; hred[wave * 2 + 1] = l1; }
            __syncthreads();
            if (c == 0 && g < nvalid) {
                const int w0 = g * 2;   // the clip's two waves
                l0 = hred[w0 * 2] + hred[(w0 + 1) * 2] + fb0;
                l1 = hred[w0 * 2 + 1] + hred[(w0 + 1) * 2 + 1] + fb1;
                const long long b = clip0 + g;
                if (nan_clip) l0 = l1 = __builtin_nanf("");   // probs NaN, argmax 0 as torch.argmax
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
    RB_STAMP(7);
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

// Host: the same weights as 16x16x32 fragments for the TAIL tile ([KS/2][N/16][2][64 lanes][8], resblock_x3_kernel).
inline void pack_x3_tail_fragments(std::vector<bf16_t>& wt, const std::vector<float>& w1, int K1, const std::vector<float>& wp,
                                   int KP, const std::vector<float>& w2, int K2, int N) {
    const int nt = N / 16, q1 = K1 / 32, qp = KP / 32, q2 = K2 / 32, nq = q1 + qp + q2;
    wt.assign(size_t(nq) * nt * 2 * 512, 0);
    for (int q = 0; q < nq; ++q) {
        const std::vector<float>& w = q < q1 ? w1 : q < q1 + qp ? wp : w2;
        const int K = q < q1 ? K1 : q < q1 + qp ? KP : K2;
        const int ql = q < q1 ? q : q < q1 + qp ? q - q1 : q - q1 - qp;
        for (int t = 0; t < nt; ++t)
            for (int lane = 0; lane < 64; ++lane)
                for (int jj = 0; jj < 8; ++jj) {
                    const float v = w[size_t(16 * t + (lane & 15)) * K + 32 * ql + 8 * (lane >> 4) + jj];
                    const bf16_t hi = f2bf_host(v);
                    const uint32_t hb = uint32_t(hi) << 16;
                    float hf;
                    std::memcpy(&hf, &hb, 4);
                    const size_t base = ((size_t(q) * nt + t) * 2) * 512 + size_t(lane) * 8 + jj;
                    wt[base] = hi;
                    wt[base + 512] = f2bf_host(v - hf);
                }
    }
}

}  // namespace
}  // namespace cough
