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
