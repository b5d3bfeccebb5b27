// Post-codec audio scaling — the step right after the hot path:
//   normalize_audio   utils/data_utils.py:407-466   ('clip' | 'peak' | 'rms' | 'none')
//   scale_audio       scripts/generate.py:440-461   (called per generated clip, save_results :404)
// Bound: HBM (one read + one write of the waveform; 'peak'/'rms' read it twice).  The statistics are per clip,
// reduced in a fixed order (per-block partials, then one ordered pass) so results do not depend on scheduling.
#include "common.h"

#define POST_BLOCKS 64   // partials per clip

// partial[(clip * POST_BLOCKS + blk) * 2 + {0: max|x|, 1: sum x^2}]
__global__ __launch_bounds__(256) void audio_stats_kernel(const float* __restrict__ wav, float* __restrict__ partial, int64_t n) {
  const int clip = blockIdx.y, blk = blockIdx.x;
  const float* w = wav + (size_t)clip * n;
  float mx = 0.f, ss = 0.f;
  for (int64_t i = (int64_t)blk * 256 + threadIdx.x; i < n; i += (int64_t)POST_BLOCKS * 256) {
    const float x = w[i];
    mx = fmaxf(mx, fabsf(x));
    ss = fmaf(x, x, ss);
  }
  mx = wave_max(mx);
  ss = wave_sum(ss);
  __shared__ float pm[4], ps[4];
  if ((threadIdx.x & 63) == 0) { pm[threadIdx.x >> 6] = mx; ps[threadIdx.x >> 6] = ss; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[((size_t)clip * POST_BLOCKS + blk) * 2 + 0] = fmaxf(fmaxf(pm[0], pm[1]), fmaxf(pm[2], pm[3]));
    partial[((size_t)clip * POST_BLOCKS + blk) * 2 + 1] = ((ps[0] + ps[1]) + ps[2]) + ps[3];
  }
}

// strategy: 0 clip, 1 peak, 2 rms, 3 none
__global__ __launch_bounds__(256) void audio_apply_kernel(const float* __restrict__ wav, float* __restrict__ out,
                                                          const float* __restrict__ partial, int64_t n, int strategy,
                                                          int normalize, float scale_peak, float scale_rms) {
  const int clip = blockIdx.y;
  float gain = 1.f, lo = -INFINITY, hi = INFINITY;
  if (strategy == 0) {
    lo = -scale_peak; hi = scale_peak;                       // wav.clamp(-scale_peak, scale_peak)
  } else if (strategy == 1 || strategy == 2) {
    float mx = 0.f, ss = 0.f;
    for (int b = 0; b < POST_BLOCKS; ++b) {                  // same order in every thread
      mx = fmaxf(mx, partial[((size_t)clip * POST_BLOCKS + b) * 2]);
      ss += partial[((size_t)clip * POST_BLOCKS + b) * 2 + 1];
    }
    // `python_float / tensor` is Tensor.__rtruediv__ = tensor.reciprocal() * fp32(python_float): two roundings
    const float rescaling = strategy == 1 ? (1.0f / mx) * scale_peak : (1.0f / sqrtf(ss / (float)n)) * scale_rms;
    if (normalize || rescaling < 1.f) gain = rescaling;      // data_utils.py:443, :450
    if (strategy == 2) { lo = -1.f; hi = 1.f; }              // _clip_wav
  }
  const float* w = wav + (size_t)clip * n;
  float* o = out + (size_t)clip * n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    o[i] = fminf(fmaxf(w[i] * gain, lo), hi);
}

extern "C" {

size_t vaura_audio_scratch_elems(int n_clips) { return n_clips > 0 ? (size_t)n_clips * POST_BLOCKS * 2 : 0; }

int vaura_audio_normalize(const float* wav, float* out, int n_clips, int64_t n_samples, int strategy, int normalize,
                          float peak_clip_headroom_db, float rms_headroom_db, float* scratch, vaura_stream_t s_) {
  if (!wav || !out || n_clips <= 0 || n_samples <= 0) return VAURA_ERR_ARG;
  if (strategy < 0 || strategy > 3) return VAURA_ERR_ARG;
  if ((strategy == 1 || strategy == 2) && !scratch) return VAURA_ERR_ARG;
  hipStream_t s = as_stream(s_);
  // the reference evaluates 10 ** (-db / 20) in double and hands it to fp32 tensor ops
  const float scale_peak = (float)pow(10.0, -(double)peak_clip_headroom_db / 20.0);
  const float scale_rms = (float)pow(10.0, -(double)rms_headroom_db / 20.0);
  if (strategy == 1 || strategy == 2)
    VA_LAUNCH(audio_stats_kernel, dim3(POST_BLOCKS, n_clips), dim3(256), 0, s, wav, scratch, n_samples);
  const unsigned gx = (unsigned)((n_samples + 256 * 8 - 1) / (256 * 8));
  VA_LAUNCH(audio_apply_kernel, dim3(gx > 1024 ? 1024 : (gx ? gx : 1), n_clips), dim3(256), 0, s, wav, out, (const float*)scratch,
            n_samples, strategy, normalize, scale_peak, scale_rms);
  return 0;
}

}  // extern "C"
