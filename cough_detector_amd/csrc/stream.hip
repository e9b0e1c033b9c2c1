// K6 -- multi-stream ring buffers on device: the batched counterpart of the per-stream FIFO in
// RealtimePreprocessor.add_audio (/root/reference/src/preprocessing.py:597-610: append chunk, cut
// windows of window_samples, drop hop_samples).  Positions are absolute sample counters kept by the
// host; ring index = pos % ring_len.  Pure byte movement, HBM/L2-bound.
#include "common.h"

namespace cough {
namespace {

__global__ void ring_write_kernel(float* __restrict__ rings, int ring_len, const float* __restrict__ chunks,
                                  int chunk_len, const int* __restrict__ ids, const long long* __restrict__ wpos) {
    const int c = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= chunk_len) return;
    long long p = (wpos[c] + i) % ring_len;
    if (p < 0) p += ring_len;
    rings[(long long)ids[c] * ring_len + p] = chunks[(long long)c * chunk_len + i];
}

__global__ void window_gather_kernel(const float* __restrict__ rings, int ring_len, const int* __restrict__ ids,
                                     const long long* __restrict__ spos, int window_len, float* __restrict__ out) {
    const int w = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= window_len) return;
    long long p = (spos[w] + i) % ring_len;
    if (p < 0) p += ring_len;
    out[(long long)w * window_len + i] = rings[(long long)ids[w] * ring_len + p];
}

// Polyphase windowed-sinc resampler: one thread per output sample; consecutive lanes = consecutive phases of
// the same input block, so the input reads are broadcasts and the kernel rows stream from L2.
__global__ void resample_kernel(const float* __restrict__ in, long long in_stride, int in_len,
                                const float* __restrict__ kern, int orig, int nw, int width,
                                float* __restrict__ out, long long out_stride, int out_len) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= out_len) return;
    const float* x = in + (long long)blockIdx.y * in_stride;
    const int blk = n / nw, ph = n - blk * nw, K = 2 * width + orig;
    const float* kr = kern + (long long)ph * K;
    const int base = blk * orig - width;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
        const int i = base + k;
        if (i >= 0 && i < in_len) acc = fmaf(x[i], kr[k], acc);
    }
    out[(long long)blockIdx.y * out_stride + n] = acc;
}

}  // namespace
}  // namespace cough

extern "C" int cough_resample(const float* d_in, long long in_stride, int n_rows, int in_len, const float* d_kernel,
                              int orig, int new_freq, int width, float* d_out, long long out_stride, int out_len,
                              void* stream) {
    using namespace cough;
    COUGH_REQUIRE(d_in && d_kernel && d_out, COUGH_EINVAL, "cough_resample: NULL argument");
    COUGH_REQUIRE(n_rows >= 0 && n_rows <= 65535 && in_len >= 0 && out_len >= 0 && orig > 0 && new_freq > 0 && width > 0,
                  COUGH_EINVAL, "cough_resample: bad sizes");
    if (n_rows == 0 || out_len == 0) return COUGH_OK;
    hipLaunchKernelGGL(resample_kernel, dim3((out_len + 255) / 256, n_rows), dim3(256), 0,
                       static_cast<hipStream_t>(stream), d_in, in_stride, in_len, d_kernel, orig, new_freq, width, d_out,
                       out_stride, out_len);
    COUGH_HIP_CHECK(hipGetLastError());
    return COUGH_OK;
}

extern "C" int cough_ring_write(float* d_rings, int ring_len, const float* d_chunks, int chunk_len,
                                const int* d_stream_ids, const long long* d_write_pos, int n_chunks, void* stream) {
    using namespace cough;
    COUGH_REQUIRE(d_rings && d_chunks && d_stream_ids && d_write_pos, COUGH_EINVAL, "cough_ring_write: NULL argument");
    COUGH_REQUIRE(ring_len > 0 && chunk_len >= 0 && chunk_len <= ring_len && n_chunks >= 0 && n_chunks <= 65535,
                  COUGH_EINVAL, "cough_ring_write: need 0 <= chunk_len <= ring_len, 0 <= n_chunks <= 65535");
    if (n_chunks == 0 || chunk_len == 0) return COUGH_OK;
    hipLaunchKernelGGL(ring_write_kernel, dim3((chunk_len + 255) / 256, n_chunks), dim3(256), 0,
                       static_cast<hipStream_t>(stream), d_rings, ring_len, d_chunks, chunk_len, d_stream_ids,
                       d_write_pos);
    COUGH_HIP_CHECK(hipGetLastError());
    return COUGH_OK;
}

extern "C" int cough_window_gather(const float* d_rings, int ring_len, const int* d_stream_ids,
                                   const long long* d_start_pos, int n_windows, int window_len, float* d_out,
                                   void* stream) {
    using namespace cough;
    COUGH_REQUIRE(d_rings && d_stream_ids && d_start_pos && d_out, COUGH_EINVAL, "cough_window_gather: NULL argument");
    COUGH_REQUIRE(ring_len > 0 && window_len > 0 && window_len <= ring_len && n_windows >= 0 && n_windows <= 65535,
                  COUGH_EINVAL, "cough_window_gather: need 0 < window_len <= ring_len, 0 <= n_windows <= 65535");
    if (n_windows == 0) return COUGH_OK;
    hipLaunchKernelGGL(window_gather_kernel, dim3((window_len + 255) / 256, n_windows), dim3(256), 0,
                       static_cast<hipStream_t>(stream), d_rings, ring_len, d_stream_ids, d_start_pos, window_len,
                       d_out);
    COUGH_HIP_CHECK(hipGetLastError());
    return COUGH_OK;
}

// ------------------------------------------------------------------------------------------ SpecAugment masks
namespace cough {
namespace {
struct MaskSet {
    int n;
    int axis[COUGH_MAX_MASKS], start[COUGH_MAX_MASKS], end[COUGH_MAX_MASKS];
};
// one pass: out = masked ? 0 : in.  4 floats per thread where the row length allows aligned float4 access.
__global__ __launch_bounds__(256) void mask_axes_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                        long long total, int height, int width, MaskSet ms) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int col = int(idx % width);
    const int row = int((idx / width) % height);
    bool hit = false;
    for (int k = 0; k < ms.n; ++k) {
        const int i = ms.axis[k] == 0 ? row : col;
        hit = hit || (i >= ms.start[k] && i < ms.end[k]);
    }
    out[idx] = hit ? 0.f : in[idx];
}
}  // namespace
}  // namespace cough

extern "C" int cough_mask_axes(const float* d_in, float* d_out, long long n_images, int height, int width, int n_masks,
                               const int* axis, const int* start, const int* end, void* stream) {
    using namespace cough;
    COUGH_REQUIRE(d_in && d_out && (n_masks == 0 || (axis && start && end)), COUGH_EINVAL, "cough_mask_axes: NULL argument");
    COUGH_REQUIRE(n_images >= 0 && height >= 1 && width >= 1, COUGH_EINVAL, "cough_mask_axes: bad shape");
    COUGH_REQUIRE(n_masks >= 0 && n_masks <= COUGH_MAX_MASKS, COUGH_EINVAL, "cough_mask_axes: n_masks = %d (0..%d)", n_masks,
                  COUGH_MAX_MASKS);
    MaskSet ms{};
    ms.n = n_masks;
    for (int k = 0; k < n_masks; ++k) {
        COUGH_REQUIRE(axis[k] == 0 || axis[k] == 1, COUGH_EINVAL, "cough_mask_axes: axis[%d] = %d (0 = rows, 1 = columns)", k, axis[k]);
        ms.axis[k] = axis[k]; ms.start[k] = start[k]; ms.end[k] = end[k];
    }
    const long long total = n_images * height * width;
    if (total == 0) return COUGH_OK;
    hipLaunchKernelGGL(mask_axes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       d_in, d_out, total, height, width, ms);
    COUGH_HIP_CHECK(hipGetLastError());
    return COUGH_OK;
}

// ------------------------------------------------------------------------------------------ clip preparation
namespace cough {
namespace {
// one workgroup per clip (a file-level operation): pass 1 takes the peak of the mono signal, pass 2 writes the
// centre-trimmed / zero-padded, normalised window
__global__ __launch_bounds__(1024) void prepare_clip_kernel(const float* __restrict__ in, long long in_stride, int C, int n,
                                                            float* __restrict__ out, int out_len, int normalize) {
    __shared__ float red[16];
    const int tid = threadIdx.x;
    const float fc = float(C);
    auto mono = [&](int i) {   // torch.mean(dim=0): channels summed in order, divided by the count
        float s = in[i];
        for (int c = 1; c < C; ++c) s += in[c * in_stride + i];
        return C == 1 ? s : s / fc;
    };
    float peak = 0.f;
    int isnan = 0;   // `waveform.abs().max()` of a signal holding a NaN is NaN, and `NaN > 0` leaves the signal unscaled (:209-212)
    if (normalize) {
        for (int i = tid; i < n; i += blockDim.x) {
            const float v = mono(i);
            peak = fmaxf(peak, fabsf(v));
            isnan |= v != v;
        }
        peak = wave_max(peak);
        if ((tid & 63) == 0) red[tid >> 6] = peak;
        isnan = __syncthreads_or(isnan);
        peak = red[0];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) peak = fmaxf(peak, red[w]);
    }
    const bool scale = normalize && peak > 0.f && !isnan;   // all-zero input: unchanged (preprocessing.py:209-212)
    // n > out_len: window [start, start + out_len), start = (n - out_len) / 2; n < out_len: left = (out_len - n) / 2
    const int shift = n >= out_len ? (n - out_len) / 2 : -((out_len - n) / 2);
    for (int o = tid; o < out_len; o += blockDim.x) {
        const int i = o + shift;
        float v = 0.f;
        if (i >= 0 && i < n) {
            v = mono(i);
            if (scale) v = v / peak;
        }
        out[o] = v;
    }
}
}  // namespace
}  // namespace cough

extern "C" int cough_prepare_clip(const float* d_in, long long in_stride, int n_channels, int n_samples, float* d_out,
                                  int out_len, int flags, void* stream) {
    using namespace cough;
    COUGH_REQUIRE(d_in && d_out, COUGH_EINVAL, "cough_prepare_clip: NULL argument");
    COUGH_REQUIRE(n_channels >= 1 && n_samples >= 1 && out_len >= 1 && in_stride >= n_samples, COUGH_EINVAL,
                  "cough_prepare_clip: bad shape (%d channels x %d samples, stride %lld, out %d)", n_channels, n_samples,
                  in_stride, out_len);
    hipLaunchKernelGGL(prepare_clip_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), d_in, in_stride,
                       n_channels, n_samples, d_out, out_len, (flags & COUGH_PREP_NORMALIZE) ? 1 : 0);
    COUGH_HIP_CHECK(hipGetLastError());
    return COUGH_OK;
}
