// Cross-translation-unit plumbing of libcough_amd (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

struct cough_featurizer;
struct cough_resnet;

namespace cough {

// Stem (K2) executed at the end of the featurise kernel: the 90x101 feature image never leaves the CU.
struct StemFuse {
    const uint16_t* wfrag;   // [4 steps][2 halves][32 channels][8 taps] bf16 MFMA fragments (resnet.hip); x3: the lo
                             // fragments follow the hi ones
    const float* bias;       // [32]
    void* a1;                // [n][22][25][32] NHWC: bf16 (x3 == 0) or f32 (x3 == 1)
    int x3;                  // 1: split-bf16 operands (feature image and weights as hi + lo, three MFMAs per k-step)
    int* nanflag;            // [n]: 1 = the clip holds a non-finite sample -- the reference's image, and so its logits, are NaN
};

// featurize.hip: d_feat may be nullptr when `stem` is given (features not materialised)
// n_samples: length of every waveform row (0 = the constructor's segment; another length runs on the generic chain)
int launch_featurize(const cough_featurizer* f, const float* d_wav, long long wav_stride, float* d_feat, int n_clips,
                     int flags, const StemFuse* stem, hipStream_t stream, void* d_workspace = nullptr,
                     size_t workspace_bytes = 0, int n_samples = 0);
size_t featurizer_workspace_bytes(const cough_featurizer* f, int n_clips, int n_samples = 0);
int featurizer_num_features(const cough_featurizer* f);
bool featurizer_stem_fusable(const cough_featurizer* f, bool x3);   // 90-row layout on a one-launch kernel, no pre-emphasis / PCEN

// Device tables of a featuriser that the stand-alone STFT (spectrogram.hip) shares.
struct StftView {
    const float* win;        // [512] the caller's window centred in the frame (Hann(400): 56|400|56)
    const float* win_full;   // [512] periodic Hann(512) (T.SpectralCentroid's default window)
    const float2* tw256;     // [16][16] W256^(j*k1)
    const float2* tw512;     // [128] W512^k
    int n_cus;               // compute units of the featuriser's device (grid of the persistent STFT kernel)
};
StftView featurizer_stft_view(const cough_featurizer* f);
// spectrogram.hip: spectral-contrast + centroid rows [row0, row0 + n_bands + 1) of d_feat ([n][nfeat][101])
struct ContrastCfg {
    int n_bands;
    int edges[18];
};
size_t contrast_workspace_bytes(int n_clips);
float* contrast_peaks(void* d_workspace, int n_clips);   // [n_clips] inside that workspace: per-clip max |sample| (written by K1)
int launch_contrast(const StftView& v, const ContrastCfg& cfg, const float* d_wav, long long wav_stride, float* d_feat,
                    int nfeat, int row0, int n_clips, int normalize, void* d_workspace, size_t workspace_bytes,
                    hipStream_t stream);
// spectrogram.hip: per-device set-up of the persistent STFT kernel (its 162 KB dynamic-LDS attribute) on the CURRENT
// device + that device's CU count; cough_featurizer_create calls it
int stft_prepare_device(int* n_cus);
// featurize_generic.hip: the kernel chain for every geometry the tuned featurise kernel does not cover (n_fft = 512)
struct GenFeat;
int gen_feat_create(GenFeat** out, const cough_feat_config* cfg, const float* window, const float* mel_fb, const float* dct);
void gen_feat_destroy(GenFeat* g);
// the tables hold for waveforms of ANY length: n_samples is a launch parameter (0 = the constructor's segment)
int gen_frames(const GenFeat* g, int n_samples);
int gen_segment_samples(const GenFeat* g);
size_t gen_workspace_bytes(const GenFeat* g, const cough_feat_config& cfg, int n_samples, int n_clips);
int gen_spectrogram(const GenFeat* g, const float* d_wav, long long wav_stride, int n_samples, float* d_spec, int n_clips, int flags,
                    hipStream_t stream);
int gen_featurize(const GenFeat* g, const cough_feat_config& cfg, const ContrastCfg& contrast, const float* d_wav,
                  long long wav_stride, int n_samples, float* d_feat, int nfeat, int nbase, int n_clips, int normalize,
                  void* d_workspace, size_t workspace_bytes, hipStream_t stream, bool contrast_rows_only = false);
const GenFeat* featurizer_generic(const cough_featurizer* f);   // every featuriser has the generic chain's tables
bool featurizer_tuned(const cough_featurizer* f, int n_samples = 0);   // the one-launch kernel serves waveforms of this length
bool featurizer_shipped_stft(const cough_featurizer* f, int n_samples = 0);   // ... at the shipped STFT geometry (persistent STFT kernel)
int launch_stft(const StftView& v, const float* d_wav, long long wav_stride, float* d_spec, int n_clips, int flags,
                hipStream_t stream);

}  // namespace cough
