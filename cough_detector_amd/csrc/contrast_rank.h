// Sorted-slice sums of one frame's band without sorting (AudioPreprocessor.extract_spectral_contrast,
// /root/reference/src/preprocessing.py:279-290): every bin's rank = the position torch.sort would give it (number of smaller
// values, ties by index), summed where rank >= top_idx (the "peaks" slice) and where rank < bot_idx (the "valleys" slice).
//
// LDS layout: quad-planar, band4[q * FT + frame] = bins 4 q .. 4 q + 3 of that frame as one float4 (16-byte reads, consecutive
// lanes consecutive addresses: conflict-free); bins past the band's end hold NaN, which compares false against everything.
// Four values stay in registers while the band streams past them once: 16 comparisons per LDS read.  The tie rule splits by
// position -- quads BEFORE the four count (u <= v), quads AFTER count (u < v), the quad itself is spelled out -- so a
// comparison is one v_cmp + one add-with-carry.
#pragma once
#include <hip/hip_runtime.h>

namespace cough {

// `col`: the band's first bin of this frame, bin k at col[k * pitch].  Sums of the CAP-limited largest / smallest values of one frame's band, i.e. of the reference's sorted slices
// `sorted_band[top_idx:]` / `sorted_band[:bot_idx]` (:279-290) without sorting: the band streams past once while the frame keeps
// its k largest and k smallest values so far in registers (an insertion network of max / min pairs, 4 CAP operations per value;
// the rank count below, kept for the bands of more than 128 bins of n_fft = 2048, costs 2 nb per value).  Equal values give the
// same sums whatever their order, so torch.sort's tie rule is immaterial; max / min drop a NaN, so non-finite powers are
// tracked in `chk` (v * 0) and poison the result as the reference's NaN-last sort + mean would.
template <int CAP>
__device__ __forceinline__ void select_sums(const float* __restrict__ col, int pitch, int nb, int ktop, int kbot, float scale,
                                            float& top, float& bot, float& chk) {
    float hi[CAP], lo[CAP];
#pragma unroll
    for (int i = 0; i < CAP; ++i) {
        hi[i] = -INFINITY;
        lo[i] = INFINITY;
    }
#pragma unroll 4
    for (int k = 0; k < nb; ++k) {
        const float v = col[(long long)k * pitch] * scale;
        chk = fmaf(v, 0.f, chk);
        float x = v, y = v;
#pragma unroll
        for (int i = 0; i < CAP; ++i) {   // hi: descending
            const float m = fmaxf(hi[i], x);
            x = fminf(hi[i], x);
            hi[i] = m;
        }
#pragma unroll
        for (int i = 0; i < CAP; ++i) {   // lo: ascending
            const float m = fminf(lo[i], y);
            y = fmaxf(lo[i], y);
            lo[i] = m;
        }
    }
    top = 0.f;
    bot = 0.f;
#pragma unroll
    for (int i = CAP - 1; i >= 0; --i)   // ascending, as the sorted slice is summed
        if (i < ktop) top += hi[i];
#pragma unroll
    for (int i = 0; i < CAP; ++i)
        if (i < kbot) bot += lo[i];
}

// The slice sizes of a band of nb bins (python int(n_bins * 0.8) / int(n_bins * 0.2), each at least 1, :279-284) and the selection
// network that holds the larger of them; nb <= 128 (at most 26 values per slice), or nb <= 64 with MAXCAP = 13 (fewer registers).
template <int MAXCAP = 26>
__device__ __forceinline__ void contrast_select(const float* __restrict__ col, int pitch, int nb, float scale, float& peaks,
                                                float& valleys, float& chk) {
    int top_idx = (int)((double)nb * 0.8), bot_idx = (int)((double)nb * 0.2);
    if (top_idx < 1) top_idx = 1;
    if (bot_idx < 1) bot_idx = 1;
    const int ktop = nb - top_idx, kbot = bot_idx < nb ? bot_idx : nb, cap = ktop > kbot ? ktop : kbot;
    float top, bot;
    if (cap <= 1) select_sums<1>(col, pitch, nb, ktop, kbot, scale, top, bot, chk);
    else if (cap <= 2) select_sums<2>(col, pitch, nb, ktop, kbot, scale, top, bot, chk);
    else if (cap <= 3) select_sums<3>(col, pitch, nb, ktop, kbot, scale, top, bot, chk);
    else if (cap <= 4) select_sums<4>(col, pitch, nb, ktop, kbot, scale, top, bot, chk);
    else if (cap <= 6) select_sums<6>(col, pitch, nb, ktop, kbot, scale, top, bot, chk);
    else if (cap <= 8) select_sums<8>(col, pitch, nb, ktop, kbot, scale, top, bot, chk);
    else if (cap <= 13 || MAXCAP <= 13) select_sums<13>(col, pitch, nb, ktop, kbot, scale, top, bot, chk);
    else if (cap <= 18) select_sums<18>(col, pitch, nb, ktop, kbot, scale, top, bot, chk);
    else select_sums<26>(col, pitch, nb, ktop, kbot, scale, top, bot, chk);
    peaks = top / float(ktop);   // 0 / 0 = NaN when the top slice is empty (one-bin band), as mean() of an empty tensor
    valleys = bot / float(kbot);
}

__device__ __forceinline__ int rank_le4(const float4 u, const float v) {
    return int(u.x <= v) + int(u.y <= v) + int(u.z <= v) + int(u.w <= v);
}
__device__ __forceinline__ int rank_lt4(const float4 u, const float v) {
    return int(u.x < v) + int(u.y < v) + int(u.z < v) + int(u.w < v);
}

// top / bot: sums of the values of rank >= top_idx / rank < bot_idx, added in bin order
__device__ __forceinline__ void contrast_band_sums(const float4* __restrict__ band4, int FT, int frame, int nb, int top_idx,
                                                   int bot_idx, float& top, float& bot) {
    const int nq = (nb + 3) >> 2;
    const float4* col = band4 + frame;
    top = 0.f;
    bot = 0.f;
    for (int eb = 0; eb < nq; ++eb) {
        const float4 v = col[eb * FT];
        int r0 = 0, r1 = 0, r2 = 0, r3 = 0;
#pragma unroll 2
        for (int qb = 0; qb < eb; ++qb) {
            const float4 u = col[qb * FT];
            r0 += rank_le4(u, v.x);
            r1 += rank_le4(u, v.y);
            r2 += rank_le4(u, v.z);
            r3 += rank_le4(u, v.w);
        }
        r0 += int(v.y < v.x) + int(v.z < v.x) + int(v.w < v.x);
        r1 += int(v.x <= v.y) + int(v.z < v.y) + int(v.w < v.y);
        r2 += int(v.x <= v.z) + int(v.y <= v.z) + int(v.w < v.z);
        r3 += int(v.x <= v.w) + int(v.y <= v.w) + int(v.z <= v.w);
#pragma unroll 2
        for (int qb = eb + 1; qb < nq; ++qb) {
            const float4 u = col[qb * FT];
            r0 += rank_lt4(u, v.x);
            r1 += rank_lt4(u, v.y);
            r2 += rank_lt4(u, v.z);
            r3 += rank_lt4(u, v.w);
        }
        const int e = eb * 4;   // the padding (e + j >= nb) is NaN: it is skipped here and never counted above
        if (r0 >= top_idx) top += v.x;
        if (r0 < bot_idx) bot += v.x;
        if (e + 1 < nb) {
            if (r1 >= top_idx) top += v.y;
            if (r1 < bot_idx) bot += v.y;
        }
        if (e + 2 < nb) {
            if (r2 >= top_idx) top += v.z;
            if (r2 < bot_idx) bot += v.z;
        }
        if (e + 3 < nb) {
            if (r3 >= top_idx) top += v.w;
            if (r3 < bot_idx) bot += v.w;
        }
    }
}

}  // namespace cough
