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

// ---- 'loudness' (utils/data_utils.py:347-387 normalize_loudness -> torchaudio.transforms.Loudness, torchaudio 2.2.1: a third-party
// dependency that is absent here — restated from the published algorithm, ITU-R BS.1770-4, as torchaudio.functional.loudness
// implements it: K-weighting = treble shelf (+4 dB, 1500 Hz, Q = 1/sqrt(2)) then high-pass (38 Hz, Q = 0.5), each a biquad run by
// lfilter in direct form I on fp32 with its output clamped to [-1, 1]; mean square over 400 ms blocks with 75 % overlap; absolute gate
// at -70 LKFS, relative gate 10 LU below the absolutely gated mean; LKFS = -0.691 + 10 log10(mean of the gated blocks).  PARITY
// UNPINNED, like DAC.)  One thread per clip: the recursion is sequential (112 640 samples: ~2-3 ms, outside the metric's wall time).
// gains[clip] = 10^((-headroom - LKFS) / 20), or 1 where the reference leaves the clip alone (rms < energy_floor, fewer samples than
// one gating block — its unfold raises and normalize_loudness returns the input —, or no block passes the gates / a non-finite gain).
#define LOUD_MAX_STEPS 4096       // quarter-blocks kept per clip (412 s at 44.1 kHz)
__global__ void audio_loudness_gain_kernel(const float* __restrict__ wav, float* __restrict__ gains, float* __restrict__ steps, int64_t n,
                                           int gate, int step, float headroom_db, float energy_floor,
                                           float tb0, float tb1, float tb2, float ta1, float ta2,
                                           float hb0, float hb1, float hb2, float ha1, float ha2) {
  const int clip = blockIdx.x;
  if (threadIdx.x != 0) return;
  const float* w = wav + (size_t)clip * n;
  float* S = steps + (size_t)clip * LOUD_MAX_STEPS;                // sum of squares of the K-weighted signal per `step` samples
  const int64_t nsteps = n / step;
  float x1 = 0.f, x2 = 0.f, y1 = 0.f, y2 = 0.f, u1 = 0.f, u2 = 0.f, v1 = 0.f, v2 = 0.f;
  double tot = 0.0;
  float acc = 0.f;
  int64_t j = 0;
  int inblk = 0;
  for (int64_t i = 0; i < n; ++i) {
    const float x = w[i];
    tot += (double)x * (double)x;
    float y = ((tb0 * x + tb1 * x1) + tb2 * x2) - ta1 * y1 - ta2 * y2;          // treble shelf
    x2 = x1; x1 = x; y2 = y1; y1 = y;
    const float yc = fminf(fmaxf(y, -1.f), 1.f);                                // lfilter(clamp=True) clamps what it RETURNS; its recursion runs on the unclamped state
    float v = ((hb0 * yc + hb1 * u1) + hb2 * u2) - ha1 * v1 - ha2 * v2;         // high-pass
    u2 = u1; u1 = yc; v2 = v1; v1 = v;
    const float vc = fminf(fmaxf(v, -1.f), 1.f);
    acc = fmaf(vc, vc, acc);
    if (++inblk == step) {
      if (j < LOUD_MAX_STEPS) S[j] = acc;
      ++j; acc = 0.f; inblk = 0;
    }
  }
  float gain = -1.f;                 // negative = "the reference leaves this clip alone" (returns before gain AND compressor); a computed gain is > 0
  const float rms = (float)sqrt(tot / (double)n);
  const int per = gate / step;                                                   // 4 quarter-blocks per gating block
  const int64_t nblk = (n >= gate && nsteps <= LOUD_MAX_STEPS) ? (n - gate) / step + 1 : 0;     // (longer clips are refused by the launcher)
  if (rms >= energy_floor && nblk > 0) {
    auto energy = [&](int64_t b) { float e = 0.f; for (int q = 0; q < per; ++q) e += S[b + q]; return e / (float)gate; };
    float s1 = 0.f; int c1 = 0;
    for (int64_t b = 0; b < nblk; ++b) {
      const float e = energy(b);
      if (-0.691f + 10.f * log10f(e) > -70.f) { s1 += e; ++c1; }
    }
    if (c1 > 0) {
      const float gamma_rel = -0.691f + 10.f * log10f(s1 / (float)c1) - 10.f;
      float s2 = 0.f; int c2 = 0;
      for (int64_t b = 0; b < nblk; ++b) {
        const float e = energy(b);
        const float l = -0.691f + 10.f * log10f(e);
        if (l > -70.f && l > gamma_rel) { s2 += e; ++c2; }
      }
      if (c2 > 0) {
        const float lkfs = -0.691f + 10.f * log10f(s2 / (float)c2);
        const float g = powf(10.f, (-headroom_db - lkfs) / 20.f);
        if (isfinite(g)) gain = g;
      }
    }
  }
  gains[clip] = gain;
}

__global__ __launch_bounds__(256) void audio_gain_clip_kernel(const float* __restrict__ wav, float* __restrict__ out,
                                                              const float* __restrict__ gains, int64_t n, int compressor) {
  const int clip = blockIdx.y;
  const float g0 = gains[clip];
  const bool untouched = g0 < 0.f;                                             // (an explicit flag: a computed gain of exactly 1 still goes through the compressor)
  const float g = untouched ? 1.f : g0;
  const float* w = wav + (size_t)clip * n;
  float* o = out + (size_t)clip * n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float y = w[i] * g;
    if (compressor && !untouched) y = tanhf(y);                                 // (the untouched clips return before the compressor)
    o[i] = fminf(fmaxf(y, -1.f), 1.f);                                          // _clip_wav (data_utils.py:389-404)
  }
}

extern "C" {

size_t vaura_audio_loudness_scratch_elems(int n_clips) { return n_clips > 0 ? (size_t)n_clips * (LOUD_MAX_STEPS + 1) : 0; }

// f3, strategy 'loudness' of normalize_audio (utils/data_utils.py:453-458 -> normalize_loudness :347-387 -> _clip_wav)
int vaura_audio_loudness(const float* wav, float* out, int n_clips, int64_t n_samples, int sample_rate, float loudness_headroom_db,
                         int compressor, float energy_floor, float* scratch, vaura_stream_t s_) {
  if (!wav || !out || !scratch || n_clips <= 0 || n_samples <= 0 || sample_rate <= 0) return VAURA_ERR_ARG;
  hipStream_t s = as_stream(s_);
  // Python's round() (what torchaudio's Loudness uses for its block sizes) rounds halves to EVEN: nearbyint in the default rounding mode
  const int gate = (int)nearbyint(0.4 * (double)sample_rate);
  const int step = (int)nearbyint((double)gate * 0.25);
  // Refused LOUDLY rather than computed wrongly or skipped: a gating block that is not four whole steps (e.g. 11 025 Hz), and
  // clips longer than the LOUD_MAX_STEPS quarter-blocks the scratch holds (412 s at 44.1 kHz) — the reference would normalise those
  if (step <= 0 || gate % step) return VAURA_ERR_SHAPE;
  if (n_samples / step > LOUD_MAX_STEPS) return VAURA_ERR_SHAPE;
  // biquad coefficients as torchaudio.functional.{treble_biquad, highpass_biquad} derive them (double), normalised by a0, rounded to fp32
  const double PI = 3.14159265358979323846;
  double w0 = 2.0 * PI * 1500.0 / sample_rate, A = exp(4.0 / 40.0 * log(10.0)), alpha = sin(w0) / 2.0 / (1.0 / sqrt(2.0));
  double t1 = 2.0 * sqrt(A) * alpha, t2 = (A - 1.0) * cos(w0), t3 = (A + 1.0) * cos(w0);
  double b0 = A * ((A + 1.0) + t2 + t1), b1 = -2.0 * A * ((A - 1.0) + t3), b2 = A * ((A + 1.0) + t2 - t1);
  double a0 = (A + 1.0) - t2 + t1, a1 = 2.0 * ((A - 1.0) - t3), a2 = (A + 1.0) - t2 - t1;
  const float tb0 = (float)(b0 / a0), tb1 = (float)(b1 / a0), tb2 = (float)(b2 / a0), ta1 = (float)(a1 / a0), ta2 = (float)(a2 / a0);
  w0 = 2.0 * PI * 38.0 / sample_rate; alpha = sin(w0) / 2.0 / 0.5;
  b0 = (1.0 + cos(w0)) / 2.0; b1 = -1.0 - cos(w0); b2 = b0; a0 = 1.0 + alpha; a1 = -2.0 * cos(w0); a2 = 1.0 - alpha;
  const float hb0 = (float)(b0 / a0), hb1 = (float)(b1 / a0), hb2 = (float)(b2 / a0), ha1 = (float)(a1 / a0), ha2 = (float)(a2 / a0);
  float* gains = scratch;
  float* steps = scratch + n_clips;
  VA_LAUNCH(audio_loudness_gain_kernel, dim3(n_clips), dim3(64), 0, s, wav, gains, steps, n_samples, gate, step, loudness_headroom_db,
            energy_floor, tb0, tb1, tb2, ta1, ta2, hb0, hb1, hb2, ha1, ha2);
  const unsigned gx = (unsigned)((n_samples + 256 * 8 - 1) / (256 * 8));
  VA_LAUNCH(audio_gain_clip_kernel, dim3(gx > 1024 ? 1024 : (gx ? gx : 1), n_clips), dim3(256), 0, s, wav, out, (const float*)gains,
            n_samples, compressor);
  return 0;
}

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
