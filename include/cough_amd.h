/*
 * cough_amd.h -- C-ABI of the MI355X (gfx950) cough-detector hot path.
 *
 * The reference (dataexplorations2026/cough_detector) has no FFI: its hot path is
 * plain Python on torch tensors.  This header is therefore the boundary a
 * maintainer binds with ctypes; every entry point names the
 * reference interface it replaces.  Conventions:
 *
 *   - plain pointers and sizes only; `d_` = device (HBM) pointer, otherwise host;
 *   - every call returns COUGH_OK (0) or a COUGH_E* code; cough_amd_last_error()
 *     gives the thread-local message (the Python side maps EINVAL/EUNSUPPORTED to
 *     ValueError as /root/reference/src/model.py:313-314 does, the rest to RuntimeError);
 *   - launches are stream-ordered on `stream` (a hipStream_t; NULL = default
 *     stream); no call synchronises the device or allocates device memory except
 *     the *_create functions;
 *   - the caller owns every input / output / workspace buffer; handles own only
 *     their immutable tables (window, twiddles, sparse filterbank, DCT, folded
 *     and packed conv weights);
 *   - handles are immutable after creation: any number of host threads may launch
 *     with the same handle, each with its own workspace and stream (reference threading: one
 *     consumer thread, /root/reference/src/inference.py:302-335; tests/test_gpu_threads.py).
 */
#ifndef COUGH_AMD_H
#define COUGH_AMD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: exactly the entry points declared in this header are exported
 * (tests/test_host_logic.py compares `nm -D` of the built library with this list). */
#pragma GCC visibility push(default)

#define COUGH_AMD_ABI_VERSION 5

#define COUGH_OK 0
#define COUGH_EINVAL 1        /* bad argument (NULL, negative size, misaligned pointer) */
#define COUGH_EUNSUPPORTED 2  /* a configuration the HIP path does not implement */
#define COUGH_EHIP 3          /* a HIP runtime call failed */
#define COUGH_EWORKSPACE 4    /* workspace smaller than *_workspace_bytes() */

int cough_amd_abi_version(void);
const char* cough_amd_arch(void);        /* "gfx950" */
const char* cough_amd_last_error(void);  /* thread-local, never NULL */

/* ------------------------------------------------------------------ featuriser (K1)
 * Replaces AudioPreprocessor.__init__ / extract_features / normalize
 * (/root/reference/src/preprocessing.py:32-144, :432-489, :199-212) for every flag of the
 * constructor (the shipped set is /root/reference/src/train.py:264-287) and every geometry with
 * 16 <= n_fft <= 2048 (even or odd): any sample_rate, hop_length >= 1, win_length <= n_fft, n_mels <= 256,
 * n_mfcc <= n_mels, any filterbank (f_min / f_max), any segment longer than n_fft / 2 samples (RealtimePreprocessor's
 * window_duration, :559-580; the engine's re-construction from a checkpoint config,
 * /root/reference/src/inference.py:89-108).  The values in the comments below are the shipped geometry.
 * At the shipped STFT geometry (16 kHz, n_fft 512, hop 160, window 400, 1 s) ANY filterbank runs on the one-launch
 * kernel (cough_featurizer_path: the shipped sparse bank with its taps in registers, every other bank -- f_max up to the
 * Nyquist bin, 2..128 bands, <= 20 MFCCs -- on the full-band instantiations); every other geometry runs on a chain of
 * small kernels and NEEDS A WORKSPACE (cough_featurizer_workspace_bytes > 0: use cough_featurize_ws), as do the
 * spectral-contrast rows and waveforms of another length (cough_featurize_any).  n_fft = 512 (the reference's
 * default) uses the register radix-16 x radix-16 FFT, the other powers of two >= 64 a radix-4 Stockham kernel, every
 * other n_fft (400 = torchaudio's own default, odd sizes) a direct DFT on the f32 matrix cores -- exact, O(n_fft^2), 2-3x slower; the frame
 * count is torch.stft's, (N - n_fft % 2) / hop_length + 1; n_fft outside 16 .. 2048 returns COUGH_EUNSUPPORTED.
 * Output row order as the reference concatenates (:456-487): mel[0:n_mels]
 * (log-mel or PCEN), MFCC, delta, (delta-delta), (spectral contrast + centroid). */
#define COUGH_MAX_CONTRAST_BANDS 16
typedef struct cough_feat_config {
    int sample_rate;       /* 16000 */
    int n_fft;             /* 512 */
    int hop_length;        /* 160 */
    int win_length;        /* 400 */
    int n_mels;            /* 64 */
    int n_mfcc;            /* 13 */
    int segment_samples;   /* 16000 */
    int use_pre_emphasis;  /* preprocessing.py:214-240 */
    float pre_emphasis_coef;
    int use_delta_delta;   /* preprocessing.py:471-474 */
    int use_pcen;          /* preprocessing.py:305-340, :400-404: mel rows = min-max normalised PCEN instead of log-mel */
    int use_mfcc;          /* preprocessing.py:458-474: 0 -> mel rows only (no MFCC / delta rows) */
    int use_spectral_contrast; /* preprocessing.py:242-303, :476-480: n_contrast_bands + 1 extra rows (needs a workspace:
                                * cough_featurize_ws).  NB the reference's rows are all NaN for n_contrast_bands >= 5
                                * (its first band is one bin whose "top 20 %" slice is empty); this library reproduces
                                * that. */
    int n_contrast_bands;  /* 1..COUGH_MAX_CONTRAST_BANDS */
    int contrast_edges[COUGH_MAX_CONTRAST_BANDS + 2]; /* torch.logspace(0, log10(n_fft/2+1), n_bands+2).int(), :267-268 */
} cough_feat_config;

typedef struct cough_featurizer cough_featurizer;

/* window[win_length], mel_fb[(n_fft/2+1) * n_mels] row-major (freq, mel), dct[n_mels * n_mfcc]
 * row-major (mel, coeff) are HOST float32 tables built by the caller with the same op
 * sequence torchaudio uses (T.MelSpectrogram / T.MFCC, preprocessing.py:94-127). */
int cough_featurizer_create(cough_featurizer** out, const cough_feat_config* cfg,
                            const float* window, const float* mel_fb, const float* dct);
void cough_featurizer_destroy(cough_featurizer* f);
int cough_featurizer_num_features(const cough_featurizer* f); /* get_num_features(), :536-550 */
int cough_featurizer_num_frames(const cough_featurizer* f);   /* get_expected_time_frames(), :532-534 */
/* Which kernels serve windows of segment_samples: COUGH_PATH_GENERIC (the kernel chain, needs a workspace), COUGH_PATH_TUNED
 * (one launch; the shipped 64-mel / 13-MFCC layout with a filterbank of <= 8 taps per band below bin 128) or
 * COUGH_PATH_TUNED_FULLBAND (one launch; any filterbank at the shipped STFT geometry -- 16 kHz, n_fft 512, hop 160, window 400,
 * 1 s --: f_max up to the Nyquist bin, 2..128 mel bands (even), up to 20 MFCCs; PCEN with 64 bands). */
#define COUGH_PATH_GENERIC 0
#define COUGH_PATH_TUNED 1
#define COUGH_PATH_TUNED_FULLBAND 2
#define COUGH_PATH_TUNED_GEOMETRY 3 /* one launch, run-time STFT geometry: n_fft 512, any window, hop <= 256 (<= 512 while the
                                     * frames' spans still cover the segment), segments whose
                                     * n_mels x frames dB buffer fits the LDS (64 bands: ~220 frames; other sample rates / window
                                     * durations), any filterbank -- also odd band counts, more than 20 MFCCs and PCEN with another band count than 64 at the shipped STFT, while
                                     * n_mfcc x frames x 4 <= 16 640 B --; spectral-contrast rows are added by the generic STFT kernel behind it
                                     * (needs the workspace) */
int cough_featurizer_path(const cough_featurizer* f);

#define COUGH_FEAT_NORMALIZE 1 /* apply normalize() (peak, per clip) before extract_features */

/* d_wav: n_clips rows of segment_samples float32, row i at d_wav + i*wav_stride (elements; for the
 * shipped geometry a multiple of 4 with a 16-byte aligned base, otherwise any stride >= segment_samples).
 * d_feat: [n_clips][num_features][num_frames] float32, num_frames = segment_samples / hop_length + 1.
 * All reductions (peak, top_db floor, MFCC mean/std) are per clip. */
int cough_featurize(const cough_featurizer* f, const float* d_wav, long long wav_stride,
                    float* d_feat, int n_clips, int flags, void* stream);

/* Same, for configurations that need scratch memory (use_spectral_contrast: two spectrograms of a sub-batch;
 * any non-shipped geometry: the power spectrogram and the mel powers of a sub-batch, <= 192 MiB).
 * cough_featurizer_workspace_bytes is 0 for every other configuration, and cough_featurize then equals
 * cough_featurize_ws(..., NULL, 0, ...); cough_featurize on a configuration that needs scratch returns
 * COUGH_EWORKSPACE.  d_workspace: device memory, 256-byte aligned, owned by the caller. */
size_t cough_featurizer_workspace_bytes(const cough_featurizer* f, int n_clips);
int cough_featurize_ws(const cough_featurizer* f, const float* d_wav, long long wav_stride, float* d_feat,
                       int n_clips, int flags, void* d_workspace, size_t workspace_bytes, void* stream);

/* Waveforms of ANY length through one handle.  The reference's extract_features never checks the length of its input
 * (/root/reference/src/preprocessing.py:432-489: T = 1 + N / hop_length frames for whatever N arrives), and none of a
 * featuriser's tables depends on it: n_samples is a launch parameter.  n_samples = 0 or = segment_samples is exactly
 * cough_featurize_ws / cough_featurizer_workspace_bytes / cough_featurizer_num_frames; any other length (> n_fft / 2, the
 * reflect padding of torch.stft) runs on the generic kernel chain and needs cough_featurizer_workspace_bytes_for() bytes.
 * d_feat: [n_clips][num_features][cough_featurizer_num_frames_for(f, n_samples)]. */
int cough_featurizer_num_frames_for(const cough_featurizer* f, int n_samples);
size_t cough_featurizer_workspace_bytes_for(const cough_featurizer* f, int n_samples, int n_clips);
int cough_featurize_any(const cough_featurizer* f, const float* d_wav, long long wav_stride, int n_samples, float* d_feat,
                        int n_clips, int flags, void* d_workspace, size_t workspace_bytes, void* stream);

/* Stand-alone STFT: T.Spectrogram(n_fft, win_length, hop_length, power=2.0) of
 * /root/reference/src/preprocessing.py:131-136 (the "STFT stage" on its own; cough_featurize never
 * materialises it) at the featuriser's geometry.  d_spec: [n_clips][n_fft/2+1][num_frames] float32
 * (shipped: 101 frames, the persistent kernel the 50 % HBM figure is quoted on).
 * flags: COUGH_SPEC_MAGNITUDE -> power=1.0; COUGH_SPEC_FULL_WINDOW -> periodic Hann(n_fft) instead of the
 * featuriser's window (both together = the spectrogram T.SpectralCentroid(sample_rate, n_fft, hop_length)
 * forms internally, :137-141). */
#define COUGH_SPEC_MAGNITUDE 1
#define COUGH_SPEC_FULL_WINDOW 2
int cough_spectrogram(const cough_featurizer* f, const float* d_wav, long long wav_stride,
                      float* d_spec, int n_clips, int flags, void* stream);
/* ... of waveforms of n_samples each (0 = segment_samples): d_spec [n_clips][n_fft/2+1][num_frames_for(n_samples)] */
int cough_spectrogram_any(const cough_featurizer* f, const float* d_wav, long long wav_stride, int n_samples,
                          float* d_spec, int n_clips, int flags, void* stream);

/* The three helper methods of AudioPreprocessor a caller may use on their own (extract_features fuses them):
 *   cough_pre_emphasis   apply_pre_emphasis (/root/reference/src/preprocessing.py:214-240): y[0] = x[0],
 *                        y[n] = x[n] - coef * x[n-1] per row (the product and the difference rounded separately, as torch);
 *   cough_compute_deltas compute_deltas (:342-356): replicate-pad the last axis by one, (x[t+1] - x[t-1]) / 2, over
 *                        n_rows contiguous rows of n_frames values;
 *   cough_pcen           apply_pcen (:305-340): smooth = 10-frame moving average (avg_pool2d kernel (1, 10), padding (0, 5),
 *                        trimmed to n_frames), (mel / (eps + smooth)^alpha + delta)^r - delta^r.
 * Out-of-place only (d_out != d_in). */
int cough_pre_emphasis(const float* d_in, long long in_stride, float* d_out, long long out_stride, int n_rows, int n,
                       float coef, void* stream);
int cough_compute_deltas(const float* d_in, float* d_out, long long n_rows, int n_frames, void* stream);
int cough_pcen(const float* d_mel, float* d_out, long long n_rows, int n_frames, float alpha, float delta, float r,
               float eps, void* stream);

/* ------------------------------------------------------------------ classifier (K2-K5)
 * Replaces CoughDetectorResidual.forward / predict and ResidualBlock.forward
 * (/root/reference/src/model.py:210-293) in eval mode.  Pointers are HOST float32
 * tensors exactly as stored in the reference state_dict. */
typedef struct cough_conv_bn {
    const float* w;       /* conv weight  [Cout][Cin][KH][KW] */
    const float* b;       /* conv bias    [Cout] */
    const float* bn_w;    /* BatchNorm weight / bias / running_mean / running_var, [Cout] each */
    const float* bn_b;
    const float* bn_mean;
    const float* bn_var;
} cough_conv_bn;

typedef struct cough_resblock_weights {
    cough_conv_bn conv1; /* res_blocks.i.conv1 + bn1 : 3x3 s2 p1 */
    cough_conv_bn conv2; /* res_blocks.i.conv2 + bn2 : 3x3 s1 p1 */
    cough_conv_bn skip;  /* res_blocks.i.skip.0 + skip.1 : 1x1 s2 */
} cough_resblock_weights;

typedef struct cough_resnet_weights {
    cough_conv_bn stem;              /* conv1.0 + conv1.1 : 7x7 s2 p3, 1 -> 32 */
    cough_resblock_weights block[2]; /* 32 -> 64, 64 -> 128 */
    const float* fc_w;               /* fc.2.weight [2][128] */
    const float* fc_b;               /* fc.2.bias   [2] */
    float bn_eps;                    /* 1e-5 */
} cough_resnet_weights;

#define COUGH_DTYPE_FP32 0 /* exact-f32 MFMA (v_mfma_f32_*_f32): CPU-reference numerics */
#define COUGH_DTYPE_BF16 1 /* bf16 operands and activations, f32 accumulate: FAST, APPROXIMATE (logit error ~2e-2 of the
                              class-margin spread; outside the 1e-3 parity tolerance for a trained head) */
#define COUGH_DTYPE_BF16X3 3 /* split-bf16: every operand as hi + lo bf16 (16 significant bits), three MFMAs per k-step
                                (hi*hi + hi*lo + lo*hi), f32 accumulate, f32 activations in HBM: logits within 1e-3 of
                                the f32 reference at a trained head's scale.  cough_resnet_create and cough_cnn_create */

typedef struct cough_resnet cough_resnet;

int cough_resnet_create(cough_resnet** out, const cough_resnet_weights* w, int dtype);
/* The fused split-bf16 kernels of cough_resnet_create are compiled for every feature image the reference's own flags
 * produce at 99..102 frames -- 13 instantiations, selected by the stem's output height ((rows - 1) / 2 + 1) / 2:
 * 63..70 rows (use_mfcc = 0: 64 mel rows + contrast rows), 87..98 rows (shipped 90; + contrast rows) and 103..110 rows
 * (delta-delta; + contrast rows: the constructor's defaults); any other image size runs the exact-f32 kernels (same entry
 * points, same results or better).
 *
 * The same for any `channels` tuple of CoughDetectorResidual.__init__ (/root/reference/src/model.py:216-247):
 * channels[0 .. n_blocks] = (stem out, block 0 out, ..., block n_blocks-1 out); blocks[i] holds res_blocks.i;
 * fc_w is fc.2.weight [2][channels[n_blocks]].  A model created here ALWAYS runs on the exact-f32 kernels, whatever
 * `dtype` asks for (channel counts padded to multiples of 32 in device memory); the fused split-bf16 / bf16 kernels
 * are compiled for the shipped (32, 64, 128) and reached through cough_resnet_create. */
int cough_resnet_create_ex(cough_resnet** out, int n_blocks, const int* channels, const cough_conv_bn* stem,
                           const cough_resblock_weights* blocks, const float* fc_w, const float* fc_b, float bn_eps,
                           int dtype);
void cough_resnet_destroy(cough_resnet* m);
size_t cough_resnet_workspace_bytes(const cough_resnet* m, int n_clips, int height, int width);

/* d_feat: [n_clips][1][height][width] float32 (height = num_features, width = num_frames).
 * d_logits: [n_clips][2].  d_probs ([n_clips][2], softmax) and d_preds ([n_clips] int32, argmax)
 * may be NULL (model.py:261-265).  d_workspace: >= cough_resnet_workspace_bytes, 256-byte aligned. */
int cough_resnet_forward(const cough_resnet* m, const float* d_feat, int n_clips, int height, int width,
                         float* d_logits, float* d_probs, int* d_preds,
                         void* d_workspace, size_t workspace_bytes, void* stream);

/* Parity taps: copy the activation after the stem (which=1), block 0 (2), block 1 (3), ... of the
 * LAST forward on this workspace to d_out as [n_clips][C][H][W] float32. */
int cough_resnet_read_activation(const cough_resnet* m, const void* d_workspace, int n_clips,
                                 int height, int width, int which, float* d_out, void* stream);

/* One ResidualBlock as a module of its own (/root/reference/src/model.py:268-293):
 *     y = ReLU( BN2(conv2( ReLU(BN1(conv1(x))) )) + skip(x) ),   conv1 3x3 stride s pad 1, conv2 3x3 stride 1 pad 1,
 * skip = 1x1 stride-s conv + BN (`skip` != NULL; what ResidualBlock builds when s != 1 or in_ch != out_ch, :280-283) or
 * the identity (`skip` == NULL: needs s == 1 and in_ch == out_ch).  d_x: [n][in_ch][H][W] f32, d_y: [n][out_ch][OH][OW]
 * f32 with OH = (H - 1) / s + 1, OW = (W - 1) / s + 1.  Exact-f32 MFMA kernels, eval-mode BatchNorm folded at create. */
typedef struct cough_resblock cough_resblock;
int cough_resblock_create(cough_resblock** out, int in_ch, int out_ch, int stride, const cough_conv_bn* conv1,
                          const cough_conv_bn* conv2, const cough_conv_bn* skip, float bn_eps);
void cough_resblock_destroy(cough_resblock* m);
size_t cough_resblock_workspace_bytes(const cough_resblock* m, int n, int height, int width);
int cough_resblock_out_shape(const cough_resblock* m, int height, int width, int* out_h, int* out_w);
int cough_resblock_forward(const cough_resblock* m, const float* d_x, int n, int height, int width, float* d_y,
                           void* d_workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ conv-block classifiers
 * Replaces CoughDetector.forward / predict (/root/reference/src/model.py:43-141, ConvBlock :11-40) and
 * CoughDetectorSmall.forward / predict (:144-207) in eval mode.  Both are a stack of blocks
 *     y = MaxPool2d(pool)( ReLU( BN( conv(x) ) ) )
 * where conv is a dense 3x3 (padding 1), or -- dw_w != NULL -- a depthwise 3x3 (padding 1, groups = cin)
 * followed by the 1x1 convolution `conv` (nothing in between), then a global mean, Linear(hidden), ReLU,
 * Linear(2).  Pointers are HOST float32 tensors exactly as stored in the reference state_dict. */
typedef struct cough_cnn_block {
    int cin, cout;
    int ksize;             /* of `conv`: 3, or 1 when dw_w is given */
    const float* dw_w;     /* depthwise weight [cin][1][3][3] or NULL */
    const float* dw_b;     /* depthwise bias   [cin] or NULL */
    cough_conv_bn conv;    /* conv weight [cout][cin][ksize][ksize], bias, BatchNorm (running stats) */
    int pool;              /* 2: MaxPool2d(2) after the ReLU; 1: none */
} cough_cnn_block;

typedef struct cough_cnn_weights {
    int n_blocks;
    const cough_cnn_block* blocks;   /* blocks[0].cin == 1 (the feature image) */
    int hidden;
    const float* fc1_w;    /* [hidden][blocks[n-1].cout] */
    const float* fc1_b;
    const float* fc2_w;    /* [2][hidden] */
    const float* fc2_b;
    float bn_eps;
} cough_cnn_weights;

typedef struct cough_cnn cough_cnn;

int cough_cnn_create(cough_cnn** out, const cough_cnn_weights* w, int dtype /* COUGH_DTYPE_* */);
void cough_cnn_destroy(cough_cnn* m);
size_t cough_cnn_workspace_bytes(const cough_cnn* m, int n_clips, int height, int width);
/* d_feat: [n_clips][1][height][width] float32 -> d_logits [n_clips][2]; d_probs / d_preds optional as above */
int cough_cnn_forward(const cough_cnn* m, const float* d_feat, int n_clips, int height, int width,
                      float* d_logits, float* d_probs, int* d_preds, void* d_workspace, size_t workspace_bytes,
                      void* stream);
/* Parity tap: runs the conv stack only and writes its output (before the global mean) to d_out as
 * [n_clips][C][h][w] float32. */
int cough_cnn_conv_output(const cough_cnn* m, const float* d_feat, int n_clips, int height, int width,
                          float* d_out, void* d_workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ fused pipeline (K1 -> K5)
 * waveform -> logits in one call: what CoughDetectorInference.process_audio_chunk does per window with
 * preprocessor.add_audio (/root/reference/src/inference.py:214) followed by predict (:217, :165-189), for a
 * whole batch.  With a bf16 classifier and the shipped 90x101 layout the stem runs inside the featurise kernel
 * and the feature image never leaves the CU; d_feat (nullable) additionally materialises the features
 * [n_clips][num_features][num_frames].  flags: COUGH_FEAT_NORMALIZE.  Results equal cough_featurize followed
 * by cough_resnet_forward.  The two optional events are recorded on `stream` around the featurise launch
 * (profiling hook; hipEventRecord only, no synchronisation). */
size_t cough_pipeline_workspace_bytes(const cough_featurizer* f, const cough_resnet* m, int n_clips);
int cough_pipeline_forward(const cough_featurizer* f, const cough_resnet* m, const float* d_wav,
                           long long wav_stride, int n_clips, int flags, float* d_feat, float* d_logits,
                           float* d_probs, int* d_preds, void* d_workspace, size_t workspace_bytes, void* stream,
                           void* ev_featurize_begin, void* ev_featurize_end /* optional hipEvent_t, may be NULL */);

/* ------------------------------------------------------------------ SpecAugment masking (training-side featurisation)
 * Replaces the masked_fill of T.FrequencyMasking / T.TimeMasking as SpecAugment.__call__ applies them
 * (/root/reference/src/augmentation.py:303-331): every image of the batch gets the same masks.  d_in / d_out:
 * [n_images][height][width] float32 (d_out may equal d_in); mask k zeroes rows (axis 0, frequency) or columns
 * (axis 1, time) start[k] <= i < end[k].  The caller draws the masks (host RNG, as the reference does). */
#define COUGH_MAX_MASKS 16
int cough_mask_axes(const float* d_in, float* d_out, long long n_images, int height, int width, int n_masks,
                    const int* axis, const int* start, const int* end, void* stream);

/* ------------------------------------------------------------------ resampler (front of process())
 * Replaces T.Resample(orig, 16000)(waveform) (/root/reference/src/preprocessing.py:146-183): polyphase
 * windowed-sinc FIR.  d_kernel: device [new][K] float32, K = 2*width + orig, built by the caller the way
 * torchaudio builds it (orig/new already divided by their gcd).  Row r of the input (in_len samples at
 * d_in + r*in_stride) yields out_len = ceil(new*in_len/orig) samples at d_out + r*out_stride:
 * out[n] = sum_k x[(n / new)*orig + k - width] * kernel[n % new][k], x = 0 outside [0, in_len). */
int cough_resample(const float* d_in, long long in_stride, int n_rows, int in_len, const float* d_kernel,
                   int orig, int new_freq, int width, float* d_out, long long out_stride, int out_len,
                   void* stream);

/* Middle of AudioPreprocessor.process (/root/reference/src/preprocessing.py:505-512): to_mono (:185-197, mean over
 * channels), normalize (:199-212, divide by the peak of the WHOLE mono signal when it is > 0) and pad_or_trim
 * (:358-385, centre trim / symmetric zero pad, the odd sample on the right) in one launch.
 * d_in: [n_channels] rows of n_samples float32 at d_in + c*in_stride; d_out: out_len float32.
 * flags: COUGH_PREP_NORMALIZE. */
#define COUGH_PREP_NORMALIZE 1
int cough_prepare_clip(const float* d_in, long long in_stride, int n_channels, int n_samples, float* d_out,
                       int out_len, int flags, void* stream);

/* ------------------------------------------------------------------ synthetic clip source (benchmark input)
 * On-device counterpart of the reference's synthetic data generators
 * (/root/reference/setup_coughvid.py:381-441, /root/reference/prepare_data.py:136-163) for BASELINE.json
 * configs[3] ("1M-clip synthetic stream ... generated on-device from seed"): clip c of the call is a pure
 * function of seed = first_seed + c*seed_stride (mixture by seed % 6: cough-like burst, near silence, white
 * noise, hum, clicks, speech-like sine stack; un-normalised level), written as 16000 float32 at
 * d_out + c*stride.  Host mirror with the same float32 operation sequence: cough_detector_amd/synth.py
 * make_clip_counter. */
int cough_synth_clips(float* d_out, long long stride, int n_clips, long long first_seed, long long seed_stride,
                      void* stream);

/* ------------------------------------------------------------------ streaming windows (K6)
 * Device-side counterpart of RealtimePreprocessor.add_audio's FIFO
 * (/root/reference/src/preprocessing.py:582-616) for many concurrent streams: each stream owns
 * one ring of ring_len float32 in d_rings[stream][ring_len]; positions are absolute sample
 * counters (ring index = pos % ring_len). */
int cough_ring_write(float* d_rings, int ring_len, const float* d_chunks, int chunk_len,
                     const int* d_stream_ids, const long long* d_write_pos, int n_chunks, void* stream);
int cough_window_gather(const float* d_rings, int ring_len, const int* d_stream_ids,
                        const long long* d_start_pos, int n_windows, int window_len,
                        float* d_out, void* stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* COUGH_AMD_H */
