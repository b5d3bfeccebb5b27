// DAC 44.1 kHz decode: codes -> latent -> waveform.
//   call site          models/modules/dac/model.py:41-48  (quantizer.from_codes, model.decode)
//   arithmetic         descript-audio-codec==1.0.0 (un-vendored; conda_env_cuda12.1.yaml:298):
//                      Decoder / DecoderBlock / ResidualUnit / Snake1d / WNConv1d / WNConvTranspose1d
//
// Bound: MFMA.  Every Conv1d / ConvTranspose1d is one "multi-tap GEMM"
//     out[j*ostride + oshift][co] = bias[co] + sum_t sum_ci  act[j + off_t][ci] * W_t[co][ci]
// over channels-last activations (B, L, C): a k=7 dilated conv has 7 taps (off_t = (t-3)*dil), a k=1
// conv one, and a stride-r transposed conv (k = 2r, pad = r/2) is r independent 2-tap problems, one
// per output phase (off = {0, -1}, ostride = r, oshift = phase - r/2).  Products run on
// v_mfma_f32_16x16x4_f32 (exact fp32), with A = weight tile (rows = co) and B = activation tile
// (cols = position), so each lane ends up holding 4 consecutive channels of one position and the
// epilogue is float4: + bias, + residual, raw store, and Snake with the NEXT layer's alpha — the
// activation is evaluated once per element by its producer, never per tap by the consumer.
#include "common.h"

#define BM 128   // positions per workgroup tile
#define BN 96    // output channels per workgroup tile (divides 1536, 768, 384, 192, 96)
#define BK 32    // input channels per k-tile

struct ConvArgs {
  const float* in;     // (B, Lin, Cin) already activated
  const float* w;      // [phase][tap][Cout][Cin]
  const float* bias;   // (Cout)
  const float* res;    // optional (B, Lout, Cout)
  const float* alpha;  // (Cout) Snake alpha for out_act
  float* out_raw;      // optional (B, Lout, Cout)
  float* out_act;      // optional (B, Lout, Cout)
  int Lin, Lout, Cin, Cout, NT;
  int off_base, off_step;   // off_t = off_base + t*off_step
  int ostride, oshift0;     // output row = j*ostride + oshift0 + phase
  int jcount;               // j in [0, jcount)
};

// mx8 activation buffers keep their scale words behind the e4m3 bytes of the whole (B, L, C) tensor
__host__ __device__ __forceinline__ size_t mx8_scale_offset(size_t elems) { return (elems + 15) & ~(size_t)15; }

#ifndef VA_CONV_ABL
#define VA_CONV_ABL 0      // timing ablations, see conv_pair_kernel
#endif
// sin^2(x) for Snake.  The library sinf (Payne-Hanek path, sign / quadrant selects; inlined at every call site) was 2.3 ms of the 11.2 ms
// decode of 8 clips (profiles/r04_codec_ablations.txt: the activation, not the matrix pipe, was the codec's largest single cost).  Only
// the SQUARE is needed, and sin^2 has period pi and no sign: n = rint(x / pi), r = x - n pi in two fused steps (pi = PI_A + PI_B; the
// fma keeps n PI_A exact), sin r on [-pi/2, pi/2] as r + r t p(t), t = r^2 (degree-4 fit of sin(r)/r - 1: 7e-9 before rounding).
// Measured against fp64 over |x| <= 3e4: max abs error 2.5e-7, rms 3.7e-8 — the fp32 sinf squared gives 1.2e-7 / 2.9e-8; both are
// rounding noise of the final multiply.  The error stays at that level while n is exact (|x| < ~1e6; the fp16-plane activation
// format itself ends at 65504) and grows smoothly beyond.  tests/test_gpu_ops.py::test_snake_sine.
__device__ __forceinline__ float snake_sin2(float x) {
  const float n = __builtin_rintf(x * 0.318309886183790672f);
  float r = fmaf(-n, 3.14159274101257324f, x);
  r = fmaf(-n, -8.74227765734758577e-8f, r);
  const float t = r * r;
  float p = fmaf(t, 2.6051661734527443e-06f, -0.00019809046352747828f);
  p = fmaf(t, p, 0.008333050645887852f);
  p = fmaf(t, p, -0.16666658222675323f);
  const float sn = fmaf(r, t * p, r);
  return sn * sn;
}
// inv = 1.0f / (al + 1e-9f): one IEEE division per CHANNEL where a thread keeps its channels (the tile store, the fused unit), not per element
__device__ __forceinline__ float snake_fi(float v, float al, float inv) {
  if constexpr ((VA_CONV_ABL & 64) != 0) { const float sn = sinf(al * v); return v + inv * (sn * sn); }      // round 3's sine, for the A/B
  const float s2 = (VA_CONV_ABL & 16) ? al * v : snake_sin2(al * v);
  return v + inv * s2;
}
__device__ __forceinline__ float snake_f(float v, float al) { return snake_fi(v, al, 1.0f / (al + 1e-9f)); }

__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs a) {
  __shared__ f32x4 Ws[2][BK / 4][BN + 1];
  __shared__ f32x4 Xs[2][BK / 4][BM + 1];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wn = wv & 1, wm = wv >> 1;
  const int j0 = blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int tiles_m = gridDim.x;
  (void)tiles_m;
  const int phases = a.ostride;              // transposed conv: one phase per blockIdx.z % ostride
  const int b = blockIdx.z / phases, ph = blockIdx.z % phases;
  const float* in = a.in + (size_t)b * a.Lin * a.Cin;
  const float* wbase = a.w + (size_t)ph * a.NT * a.Cout * a.Cin;
  const int kc = a.Cin / BK;
  const int nk = a.NT * kc;

  f32x4 wreg[3], xreg[4];
  auto load_tile = [&](int kt) {
    const int t = kt / kc, ci0 = (kt % kc) * BK;
    const float* wt = wbase + (size_t)t * a.Cout * a.Cin;
    const int off = a.off_base + t * a.off_step;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int qd = tid + 256 * i, row = qd >> 3, kq = qd & 7;
      wreg[i] = *reinterpret_cast<const f32x4*>(wt + (size_t)(n0 + row) * a.Cin + ci0 + 4 * kq);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int qd = tid + 256 * i, row = qd >> 3, kq = qd & 7;
      const int jr = j0 + row + off;
      xreg[i] = (jr >= 0 && jr < a.Lin) ? *reinterpret_cast<const f32x4*>(in + (size_t)jr * a.Cin + ci0 + 4 * kq)
                                        : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 3; ++i) { const int qd = tid + 256 * i; Ws[buf][qd & 7][qd >> 3] = wreg[i]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int qd = tid + 256 * i; Xs[buf][qd & 7][qd >> 3] = xreg[i]; }
  };

  f32x4 acc[3][4];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      f32x4 af[3], bf[4];
#pragma unroll
      for (int i = 0; i < 3; ++i) af[i] = Ws[buf][4 * c + (lane >> 4)][wn * 48 + i * 16 + (lane & 15)];
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j] = Xs[buf][4 * c + (lane >> 4)][wm * 64 + j * 16 + (lane & 15)];
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }

  // epilogue: lane holds channels co..co+3 of one position
  const size_t obase = (size_t)b * a.Lout * a.Cout;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int jr = j0 + wm * 64 + j * 16 + (lane & 15);
    if (jr >= a.jcount) continue;
    const int orow = jr * a.ostride + a.oshift0 + ph;
    if (orow < 0 || orow >= a.Lout) continue;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int co = n0 + wn * 48 + i * 16 + 4 * (lane >> 4);
      const size_t o = obase + (size_t)orow * a.Cout + co;
      f32x4 v = acc[i][j] + *reinterpret_cast<const f32x4*>(a.bias + co);
      if (a.res) v = *reinterpret_cast<const f32x4*>(a.res + o) + v;
      if (a.out_raw) *reinterpret_cast<f32x4*>(a.out_raw + o) = v;
      if (a.out_act) {
        const f32x4 al = *reinterpret_cast<const f32x4*>(a.alpha + co);
        f32x4 s;
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r] = snake_f(v[r], al[r]);
        *reinterpret_cast<f32x4*>(a.out_act + o) = s;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// fp16-pair variant (codec precision 1): activations and weights travel as (hi, lo) fp16 pairs,
//   x = fp16(x) + fp16(x - fp16(x))   (22 significand bits; |rel err| <= 2^-22),
// and a product is three v_mfma_f32_16x16x32_f16: lo_w*hi_x + hi_w*lo_x + hi_w*hi_x (fp16 x fp16 products
// are exact in fp32; the dropped lo*lo term is < 2^-22).  5.3x fewer matrix-pipe cycles than the exact
// fp32 MFMA above at an error far inside the codec's 1e-4 RMS budget (measured).
// "pair layout" of an activated (L x C) buffer / a weight tap (Cout x Cin): per row, C/8 octets of
// [plane hi|lo][8] halves = 32 B -> one row of 32 channels is still one 128-B line, same bytes as fp32.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct ConvPArgs {
  const uint16_t* in;    // pair layout (B, Lin, Cin)
  const uint16_t* w;     // pair layout [phase][tap][Cout][Cin]
  const float* bias;
  const float* res;      // fp32 (B, Lout, Cout)
  const float* alpha;
  float* out_raw;        // fp32
  uint16_t* out_act;     // pair layout, Snake applied
  int Lin, Lout, Cin, Cout, NT;
  int off_base, off_step, ostride, oshift0, jcount;
  int act;               // activation written to out_act: 0 Snake(alpha) (the codec), 1 exact GELU, 2 identity (row f2's linears)
  // --- block-scaled fp8 ("mx8") activations (codec precision 3, conv_mx8_kernel below)
  int act_fmt;           // out_act format: 0 (hi, lo) fp16 pairs, 1 mx8 = e4m3 bytes (B, Lout, Cout) + out_scale
  uint8_t* out_scale;    // (B, Lout, ceil(Cout/128)) words of four E8M0 bytes: one power-of-two scale per 32 channels of a row
  const uint32_t* in_scale;   // the same for an mx8 input
  const float* wscale;   // (Cout) power-of-two scale of each output channel's e4m3 weights
  // --- residual unit in one launch (conv_pair_kernel<.., FUSE = true>): this conv (7 taps) -> Snake(alpha_mid) -> 1 x 1 conv (w2, bias2)
  //     -> + res -> out_raw / Snake(alpha) -> out_act.  bias / alpha_mid belong to the first conv, res / out_* / alpha to the second.
  const uint16_t* w2;    // pair layout [Cout][Cout]
  const float* bias2;
  const float* alpha_mid;
};

// Exact (erf) GELU, 0.5 x (1 + erf(x / sqrt 2)), as row f2's fc1 epilogue applies it 1.85 G times per 8 clips.  The library erff (inlined,
// branches) cost 1.2 ms of the extractor's 43.7 (experiment build -DVA_CONV_ABL=128).  Own form: with t = |x| / sqrt 2,
//   1 + erf(x / sqrt 2) = erfc(t) for x < 0, 2 - erfc(t) for x >= 0,   erfc(t) = exp(-t^2) k P(k),  k = 1 / (1 + 0.35 t),
// P a degree-8 fit of erfcx(t) / k on t in [0, 6] (3e-8 relative before rounding).  No cancellation on the negative side (the library
// form rounds 1 + erf to 0 below x = -5.5).  Against fp64: |error| <= 1.3e-7 max(1, |x|) — the rounding of the final products; the fp32
// 0.5 x (1 + erff(..)) form is at 4.5e-7.  tests/test_gpu_avclip.py keeps the features within 1e-4 of the reference's (observed 5e-6).
__device__ __forceinline__ float gelu_erf_f(float x) {
  if constexpr ((VA_CONV_ABL & 128) != 0) return 0.5f * x * (1.0f + x * 0.70710678118654752440f);      // timing ablation: no erf
  if constexpr ((VA_CONV_ABL & 256) != 0) return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); // round 3's form, for the A/B
  const float t = fabsf(x) * 0.70710678118654752440f;
  const float k = __builtin_amdgcn_rcpf(fmaf(t, 0.35f, 1.0f));
  float p = 0.06797561794519424f;
  p = fmaf(p, k, -0.4044332206249237f);
  p = fmaf(p, k, 0.8392713069915771f);
  p = fmaf(p, k, -0.7225478887557983f);
  p = fmaf(p, k, 0.6403822898864746f);
  p = fmaf(p, k, -0.046077974140644073f);
  p = fmaf(p, k, 0.23748238384723663f);
  p = fmaf(p, k, 0.1900186985731125f);
  p = fmaf(p, k, 0.1979287713766098f);
  const float e = (k * p) * __builtin_amdgcn_exp2f(-(t * t) * 1.4426950408889634f);
  return (0.5f * x) * (x < 0.f ? e : 2.0f - e);
}

__device__ __forceinline__ void store_pair4(uint16_t* base, size_t row, int c0, int C, const f32x4 v) {
  // 4 consecutive channels c0..c0+3 (c0 % 4 == 0) of one row
  _Float16 h[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float x = v[i];
    h[i] = (_Float16)x;
    l[i] = (_Float16)(x - (float)h[i]);
  }
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  f16x4* p = reinterpret_cast<f16x4*>(base + ((row * (size_t)(C >> 3) + (size_t)(c0 >> 3)) * 2) * 8 + (c0 & 7));
  p[0] = f16x4{h[0], h[1], h[2], h[3]};
  p[2] = f16x4{l[0], l[1], l[2], l[3]};   // +8 halves = the lo plane of the same octet
}

// E8M0 byte of the smallest power of two s with amax <= 448 * s (448 = 1.75 * 2^8 = the e4m3 maximum); same rule as
// vaura_amd/quant.py::fp8_row_scales, clamped to the bytes whose reciprocal is a normal float.
__device__ __forceinline__ int mx8_scale_byte(float amax) {
  const uint32_t bits = __builtin_bit_cast(uint32_t, amax);
  const int ea = (int)((bits >> 23) & 0xffu);
  const int e8 = ea - ((bits & 0x7fffffu) <= 0x600000u ? 8 : 7);
  return e8 < 1 ? 1 : (e8 > 253 ? 253 : e8);
}

// One octet (8 consecutive channels co..co+7 of global row `grow`) of an activated tensor in the next layer's input format.
// mx8: the 4 threads of a 32-channel block must be consecutive lanes, all active.
__device__ __forceinline__ void store_act_octet(void* out_act, uint8_t* out_scale, int fmt, size_t grow, int co, int C,
                                                const float (&sv)[8]) {
  if (fmt == 0) {
    f16x8 hi, lo;
#pragma unroll
    for (int r = 0; r < 8; ++r) { hi[r] = (_Float16)sv[r]; lo[r] = (_Float16)(sv[r] - (float)hi[r]); }
    f16x8* dst = reinterpret_cast<f16x8*>(reinterpret_cast<uint16_t*>(out_act) + ((grow * (size_t)(C >> 3) + (size_t)(co >> 3)) * 2) * 8);
    dst[0] = hi;
    dst[1] = lo;
  } else {
    float amax = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) amax = fmaxf(amax, fabsf(sv[r]));
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
    const int e8 = mx8_scale_byte(amax);
    const float inv = __builtin_bit_cast(float, (uint32_t)(254 - e8) << 23);
    int q0 = 0, q1 = 0;
    q0 = __builtin_amdgcn_cvt_pk_fp8_f32(sv[0] * inv, sv[1] * inv, q0, false);
    q0 = __builtin_amdgcn_cvt_pk_fp8_f32(sv[2] * inv, sv[3] * inv, q0, true);
    q1 = __builtin_amdgcn_cvt_pk_fp8_f32(sv[4] * inv, sv[5] * inv, q1, false);
    q1 = __builtin_amdgcn_cvt_pk_fp8_f32(sv[6] * inv, sv[7] * inv, q1, true);
    typedef int i32x2 __attribute__((ext_vector_type(2)));
    *reinterpret_cast<i32x2*>(reinterpret_cast<uint8_t*>(out_act) + grow * (size_t)C + co) = i32x2{q0, q1};
    if (((co >> 3) & 3) == 0) out_scale[(grow * (size_t)((C + 127) >> 7) + (co >> 7)) * 4 + ((co >> 5) & 3)] = (uint8_t)e8;
  }
}

// Second half of the conv epilogue, shared by the pair and the mx8 kernel: `stage` holds the BM x BN_ fp32 tile (bias
// added), every thread owns whole octets of a row: + residual, raw store, activation, and the activated copy in the next
// layer's input format.
// Snake for an fp8 consumer: hardware sine (v_sin_f32, argument in revolutions) and reciprocal — absolute error ~1e-6 of a
// value that is about to be rounded to 3 mantissa bits; the library sinf (argument reduction + polynomial, ~10x the
// instructions) costs more than the layer's matrix instructions once those run at the fp8 rate.
__device__ __forceinline__ float snake_fast(float v, float al) {
  const float sn = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(al * v * 0.15915494309189535f));   // v_sin_f32's domain is +-256 revolutions
  return v + __builtin_amdgcn_rcpf(al + 1e-9f) * (sn * sn);
}

template <int BN_, int SP, bool FAST = false, int ROWS = BM>
__device__ __forceinline__ void conv_tile_store(const ConvPArgs& a, const float* stage, int b, int ph, int j0, int n0, int tid) {
  constexpr int OCT = BN_ / 8;              // octets per tile row
  static_assert(OCT % 4 == 0, "a 32-channel scale block = 4 consecutive threads of one row");
  // a thread keeps ONE octet column and walks the rows (256 / OCT rows per sweep: 21 for 96 columns — threads 252..255 sit out, a whole
  // 4-thread scale group — 32 for 64): its eight Snake alphas and their inverses are loaded / divided once, not per element
  constexpr int RSTEP = 256 / OCT;
  if (tid >= RSTEP * OCT) return;
  const int oc = tid % OCT, row0 = tid / OCT;
  const int co = n0 + oc * 8;
  const size_t obase = (size_t)b * a.Lout;
  f32x4 al0 = f32x4{0.f, 0.f, 0.f, 0.f}, al1 = al0, iv0 = al0, iv1 = al0;
  if (a.out_act && a.act == 0) {
    al0 = *reinterpret_cast<const f32x4*>(a.alpha + co);
    al1 = *reinterpret_cast<const f32x4*>(a.alpha + co + 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) { iv0[r] = 1.0f / (al0[r] + 1e-9f); iv1[r] = 1.0f / (al1[r] + 1e-9f); }
  }
  for (int row = row0; row < ROWS; row += RSTEP) {
    const int jr = j0 + row;
    if (jr >= a.jcount) continue;
    const int orow = jr * a.ostride + a.oshift0 + ph;
    if (orow < 0 || orow >= a.Lout) continue;
    const size_t o = (obase + (size_t)orow) * a.Cout + co;
    f32x4 v0 = *reinterpret_cast<const f32x4*>(stage + row * SP + oc * 8);
    f32x4 v1 = *reinterpret_cast<const f32x4*>(stage + row * SP + oc * 8 + 4);
    if (a.res) {
      v0 = *reinterpret_cast<const f32x4*>(a.res + o) + v0;
      v1 = *reinterpret_cast<const f32x4*>(a.res + o + 4) + v1;
    }
    if (a.out_raw) {
      *reinterpret_cast<f32x4*>(a.out_raw + o) = v0;
      *reinterpret_cast<f32x4*>(a.out_raw + o + 4) = v1;
    }
    if (a.out_act) {
      float sv[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (FAST && a.act == 0) {
          sv[r] = snake_fast(v0[r], al0[r]);
          sv[r + 4] = snake_fast(v1[r], al1[r]);
        } else {
          sv[r] = a.act == 0 ? snake_fi(v0[r], al0[r], iv0[r]) : (a.act == 1 ? gelu_erf_f(v0[r]) : v0[r]);
          sv[r + 4] = a.act == 0 ? snake_fi(v1[r], al1[r], iv1[r]) : (a.act == 1 ? gelu_erf_f(v1[r]) : v1[r]);
        }
      }
      store_act_octet(a.out_act, a.out_scale, a.act_fmt, obase + (size_t)orow, co, a.Cout, sv);
    }
  }
}

// Loop order: input-channel chunk (32 channels) outermost, taps inside.  The activation block of a chunk is staged in
// LDS ONCE with its halo — rows [j0 + min offset, j0 + BM + max offset), at most BM + 6*9 — and every tap reads it at
// its own row shift; only the weight tile (BN x 32) is re-staged per tap.  The previous order (one (tap, chunk) tile of
// both operands per step) pulled 28 KB through L2 per 2.4 MFLOP and sat at the L2 rate, a third of the matrix-pipe
// rate; for a 7-tap layer this one moves 15 KB per step.
#define XHALO 56                      // >= (taps - 1) * dilation = 54
#define XROWS (BM + XHALO)
// NI = 16-column tiles per wave: output-channel tile BN_ = 32 * NI (96 for the decoder's widths, 64 for the
// encoder's powers of two)
// WS: weights have a single fp16 plane (fp8-quantised weights are exact in fp16: the lo plane is zero and its MFMA is
// skipped — two instead of three matrix instructions per product)
// XS (codec precision 4, "f16"): ONE matrix instruction per product, hi(w) x hi(x) — plain fp16 operands with fp32 accumulate,
// the arithmetic class the reference itself runs DAC in (models/vaura_model.py:92 casts the codec to fp16).  The buffers keep the
// pair layout (the lo planes are simply not read from LDS), so every producer / consumer kernel is shared with precision 1.
// MJ = 16-row blocks per wave: 4 (128 rows per workgroup) or 8 (256 rows, round 3).  The kernel is bound by what comes through L2 into
// LDS, and most of that is the weight tile a workgroup re-stages for every (tap, chunk) — C = 192, 7 taps: 516 KB of weights next
// to 141 KB of activations per 128 x 96 outputs (the plain-fp16 mode, a third of the matrix work on the same bytes, is only 10 %
// faster).  With MJ = 8 one staged weight tile serves twice the rows: 756 KB per 256 x 96 outputs against 1 314.  The activation
// block (256 + halo rows) then has ONE LDS buffer — it is restaged between two barriers once per chunk, every NT steps — so two
// workgroups still share a CU (70 KB each); the output tile goes out in two 128-row passes.
// VA_CONV_ABL: timing ablations for tools/experiment.sh codec-abl (WRONG RESULTS; product build = 0): 1 no weight-tile loads / stores
// after the first, 2 no matrix instructions, 4 no per-step barrier, 16 Snake without its sine, 64 Snake on the library sinf (round 3)
#ifndef VA_CONV_ABL
#define VA_CONV_ABL 0
#endif
template <int NI, bool WS, bool XS = false, int MJ = 4, bool FUSE = false>
__global__ __launch_bounds__(256, 2) void conv_pair_kernel(ConvPArgs a) {
  constexpr int BN_ = 32 * NI;
  constexpr int BMT = 32 * MJ;              // rows per workgroup (two waves along the rows)
  constexpr int XROWS_T = BMT + XHALO;
  constexpr int NXB = MJ == 4 ? 2 : 1;      // LDS buffers of the activation block
  static_assert(MJ == 4 || MJ == 8, "128 or 256 rows");
  // one raw LDS block: weight tiles | activation blocks during the main loop, the fp32 output tile afterwards.
  // LDS image [buf][kq][row + c(kq)], kq = 16-byte quad of the 32-channel chunk (octet g: kq = 2g hi plane, 2g + 1 lo plane), plane
  // stride a multiple of 16 rows, c = {0,1,0,1,2,3,2,3}.  Why: a wave's ds_read_b128 of a fragment (lane -> (row r16, octet g)) is
  // served in four lane groups, {0-3,12-15,20-27}, ... (MI355X_MICROARCH.md §LDS): every group mixes 8 rows of octet g with the
  // OTHER 8 rows of octet g + 1, so the two planes read together (kq = 0 and 2, 1 and 3, 4 and 6, 5 and 7) must start on the same
  // bank for the 16 rows to cover all 64 banks once — for any halo shift.  Round 2 padded every plane by one row (stride BN_ + 1 /
  // XROWS + 1): the two halves of every group then overlapped on 2 of 16 rows and every fragment read took 2 x the cycles
  // (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.37 for this kernel, profiles/r03_codec_mfma.json).  The staging writes (8 lanes =
  // the 8 quads of one row) now meet pairwise (2-way), which the store's own issue time hides.
  constexpr int WSTR = (BN_ + 3 + 15) / 16 * 16, XSTR = (XROWS_T + 3 + 15) / 16 * 16;
  constexpr int WS_ELEMS = 2 * (BK / 4) * WSTR, XS_ELEMS = NXB * (BK / 4) * XSTR;
  constexpr int SP = BN_ + 4;               // padded row stride (floats) of the staged output tile
  static_assert((WS_ELEMS + XS_ELEMS) * 16 >= BM * SP * 4, "the output tile must fit in the main loop's LDS");
  // FUSE: the epilogue holds the 1 x 1 conv's whole weight matrix (3 chunks) and the activated tile of 64 rows as fragment images
  static_assert(!FUSE || (NI == 3 && MJ == 8), "the fused residual unit: all 96 channels in one 256-row workgroup");
  constexpr int W1STR = 112, YSTR = 80;     // plane strides (rows) of those images: multiples of 16, see above
  constexpr int W1_ELEMS = FUSE ? 3 * (BK / 4) * W1STR : 0, Y_ELEMS = FUSE ? 3 * (BK / 4) * YSTR : 0;
  static_assert(!FUSE || Y_ELEMS * 16 >= 64 * SP * 4, "the 64-row output tile is staged over the activated tile");
  constexpr int SMEM_ELEMS = (WS_ELEMS + XS_ELEMS) > (W1_ELEMS + Y_ELEMS) ? (WS_ELEMS + XS_ELEMS) : (W1_ELEMS + Y_ELEMS);
  constexpr int TAB_ELEMS = FUSE ? 2 * BN_ / 4 : 0;      // FUSE: alpha_mid and 1 / (alpha_mid + 1e-9) per channel, behind everything else
  __shared__ u32x4 smem[SMEM_ELEMS + TAB_ELEMS];
  auto coff = [](int kq) { return (kq & 1) + ((kq >> 2) << 1); };     // {0,1,0,1,2,3,2,3}
  auto Ws = [&](int buf, int kq, int row) -> u32x4& { return smem[(buf * (BK / 4) + kq) * WSTR + row + coff(kq)]; };
  auto Xs = [&](int buf, int kq, int row) -> u32x4& { return smem[WS_ELEMS + (buf * (BK / 4) + kq) * XSTR + row + coff(kq)]; };
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wn = wv & 1, wm = wv >> 1;
  const int j0 = blockIdx.x * BMT;
  const int n0 = blockIdx.y * BN_;
  const int phases = a.ostride;
  const int b = blockIdx.z / phases, ph = blockIdx.z % phases;
  const int cq = a.Cin / 4;                 // 16-B quads per row (C/8 octets x 2 planes)
  const u32x4* in = reinterpret_cast<const u32x4*>(a.in) + (size_t)b * a.Lin * cq;
  const u32x4* wbase = reinterpret_cast<const u32x4*>(a.w) + (size_t)ph * a.NT * a.Cout * cq;
  const int kc = a.Cin / BK;
  const int NT = a.NT;
  const int nk = NT * kc;
  const int span = (NT - 1) * (a.off_step < 0 ? -a.off_step : a.off_step);
  const int lo_off = a.off_base + (a.off_step < 0 ? (NT - 1) * a.off_step : 0);   // smallest row offset of any tap
  const int xrows = BMT + span;
  constexpr int XL = (XROWS_T * (BK / 4) + 255) / 256;   // activation quads per thread per chunk (6, or 10 for 256 rows)

  u32x4 wreg[NI], xreg[XL];
  auto load_w = [&](int kt) {
    const int c = kt / NT, t = kt - c * NT;
    const u32x4* wt = wbase + (size_t)t * a.Cout * cq;
    const int q0 = c * (BK / 4);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int qd = tid + 256 * i, row = qd >> 3, kq = qd & 7;
      wreg[i] = wt[(size_t)(n0 + row) * cq + q0 + kq];
    }
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NI; ++i) { const int qd = tid + 256 * i; Ws(buf, qd & 7, qd >> 3) = wreg[i]; }
  };
  auto load_x = [&](int c) {
    const int q0 = c * (BK / 4);
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int qd = tid + 256 * i, row = qd >> 3, kq = qd & 7;
      const int jr = j0 + lo_off + row;
      xreg[i] = (row < xrows && jr >= 0 && jr < a.Lin) ? in[(size_t)jr * cq + q0 + kq] : u32x4{0u, 0u, 0u, 0u};
    }
  };
  auto store_x = [&](int buf) {
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int qd = tid + 256 * i;
      if ((qd >> 3) < XROWS_T) Xs(buf, qd & 7, qd >> 3) = xreg[i];
    }
  };

  f32x4 acc[NI][MJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  VA_STAMP_DECL(stamps);
  VA_STAMP(stamps, 0);                       // (diagnostic build, tools/mfma_driver MFMA_STAMPS) wave start
  load_w(0);
  load_x(0);
  store_w(0);
  store_x(0);
  if constexpr (FUSE) {
    if (tid < BN_) {
      float* tab = reinterpret_cast<float*>(smem + SMEM_ELEMS);
      const float al = a.alpha_mid[tid];
      tab[tid] = al;
      tab[BN_ + tid] = 1.0f / (al + 1e-9f);
    }
  }
  __syncthreads();
  VA_STAMP(stamps, 1);                       // first tiles staged
  const int g = lane >> 4, r16 = lane & 15;
  int kt = 0;
  for (int c = 0; c < kc; ++c) {
    const int xb = NXB == 2 ? (c & 1) : 0;
    for (int t = 0; t < NT; ++t, ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < nk && !(VA_CONV_ABL & 1)) load_w(kt + 1);
      // (256 rows: the next chunk's block is requested at the LAST tap — its registers are free of the step's fragments only there —
      // and stored between two barriers below)
      if (t == (NXB == 2 ? 0 : NT - 1) && c + 1 < kc) load_x(c + 1);
      const int shift = (VA_CONV_ABL & 8) ? 0 : a.off_base + t * a.off_step - lo_off;
      f16x8 wh[NI], wl[NI];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        wh[i] = __builtin_bit_cast(f16x8, Ws(buf, 2 * g, wn * (NI * 16) + i * 16 + r16));
        if constexpr (!WS && !XS) wl[i] = __builtin_bit_cast(f16x8, Ws(buf, 2 * g + 1, wn * (NI * 16) + i * 16 + r16));
      }
      constexpr int JG = MJ == 4 ? 4 : 2;       // row blocks whose fragments are live together (256 rows: the register budget)
#pragma unroll
      for (int jh = 0; jh < MJ; jh += JG) {
        f16x8 xh[JG], xl[JG];
#pragma unroll
        for (int j = 0; j < JG; ++j) {
          xh[j] = __builtin_bit_cast(f16x8, Xs(xb, 2 * g, shift + wm * (16 * MJ) + (jh + j) * 16 + r16));
          if constexpr (!XS) xl[j] = __builtin_bit_cast(f16x8, Xs(xb, 2 * g + 1, shift + wm * (16 * MJ) + (jh + j) * 16 + r16));
        }
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < JG; ++j) {
            if constexpr (VA_CONV_ABL & 2) {
              acc[i][jh + j][0] += (float)wh[i][0] * (float)xh[j][0];
              if constexpr (!XS) acc[i][jh + j][1] += (float)xl[j][0];
              if constexpr (!WS && !XS) acc[i][jh + j][2] += (float)wl[i][0];
              continue;
            }
            if constexpr (!WS && !XS) acc[i][jh + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[i], xh[j], acc[i][jh + j], 0, 0, 0);
            if constexpr (!XS) acc[i][jh + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], xl[j], acc[i][jh + j], 0, 0, 0);
            acc[i][jh + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], xh[j], acc[i][jh + j], 0, 0, 0);
          }
      }
      if (kt + 1 < nk && !(VA_CONV_ABL & 1)) store_w(buf ^ 1);
      if (t == NT - 1 && c + 1 < kc) {
        if constexpr (NXB == 1) __syncthreads();        // every wave is done with the block before it is overwritten
        store_x(NXB == 2 ? (xb ^ 1) : 0);
      }
      if constexpr (!(VA_CONV_ABL & 4)) __syncthreads();
    }
  }

  VA_STAMP(stamps, 2);                       // main loop done
  // ---- epilogue through LDS: the MFMA layout gives a lane 4 channels of one row (16-byte pieces, 64-byte runs per
  // row: half cache lines for the fp32 stream, 8-byte pieces for the pair stream); staging the tile lets every thread
  // own a whole octet of a row, so residual reads and both output streams move 32 contiguous bytes per thread and
  // whole 128-byte lines per 4 threads.  (The main loop ended with a barrier: its LDS is free.)
  if constexpr (FUSE) {
    // ---- the residual unit's second half without leaving the CU.  Per 64-row pass: the two waves that own those rows turn their
    // accumulators into the 1 x 1 conv's input — + bias, Snake(alpha_mid), (hi, lo) split: exactly what conv_tile_store writes to
    // memory in the two-launch form — as a fragment image [chunk][kq][row] in LDS; all four waves multiply it by the 1 x 1 weights
    // (three chunks, same three matrix instructions per product in the same order as the stand-alone launch: bit-identical sums);
    // the 64 x 96 result is staged as fp32 over the image and leaves through the shared tile store (+ residual, raw, Snake, pairs).
    u32x4* W1 = smem;
    u32x4* Yi = smem + W1_ELEMS;
    auto W1s = [&](int c2, int kq, int row) -> u32x4& { return W1[(c2 * (BK / 4) + kq) * W1STR + row + coff(kq)]; };
    auto Ys = [&](int c2, int kq, int row) -> u32x4& { return Yi[(c2 * (BK / 4) + kq) * YSTR + row + coff(kq)]; };
    // the 1 x 1 conv's weights (96 x 96 pairs = 2 304 quads, 9 per thread; L2 hits): requested here, parked in LDS behind the first
    // pass's Snake arithmetic
    u32x4 w1reg[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) w1reg[i] = reinterpret_cast<const u32x4*>(a.w2)[tid + 256 * i];
    float* stage = reinterpret_cast<float*>(Yi);
    const float* tab = reinterpret_cast<const float*>(smem + SMEM_ELEMS);
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    auto pass = [&](auto pc) {                  // (a lambda per compile-time pass: the accumulator indices must be static)
      constexpr int p = decltype(pc)::value;
      if (p > 0) __syncthreads();               // the previous pass's tile store is done with `stage`
      if (wm == (p >> 1)) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int i = 0; i < NI; ++i) {
            const int col = wn * (NI * 16) + i * 16 + 4 * g;
            const f32x4 v = acc[i][4 * (p & 1) + jj] + *reinterpret_cast<const f32x4*>(a.bias + col);
            const f32x4 al = *reinterpret_cast<const f32x4*>(tab + col), iv = *reinterpret_cast<const f32x4*>(tab + BN_ + col);
            f16x4 h4, l4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float sv = snake_fi(v[e], al[e], iv[e]);
              h4[e] = (_Float16)sv;
              l4[e] = (_Float16)(sv - (float)h4[e]);
            }
            const int c2 = col >> 5, o = (col & 31) >> 3, half = (col & 7) >> 2, row = jj * 16 + r16;
            reinterpret_cast<f16x4*>(&Ys(c2, 2 * o, row))[half] = h4;
            reinterpret_cast<f16x4*>(&Ys(c2, 2 * o + 1, row))[half] = l4;
          }
      }
      if (p == 0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) {
          const int qd = tid + 256 * i, n = qd / 24, q = qd - n * 24;
          W1s(q >> 3, q & 7, n) = w1reg[i];
        }
      }
      __syncthreads();
      if (p == 0) VA_STAMP(stamps, 3);         // pass 0: Snake image + 1 x 1 weights in LDS
      f32x4 acc2[NI][2];
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c2 = 0; c2 < 3; ++c2) {
        f16x8 wh[NI], wl[NI], xh[2], xl[2];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          wh[i] = __builtin_bit_cast(f16x8, W1s(c2, 2 * g, wn * (NI * 16) + i * 16 + r16));
          if constexpr (!WS && !XS) wl[i] = __builtin_bit_cast(f16x8, W1s(c2, 2 * g + 1, wn * (NI * 16) + i * 16 + r16));
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          xh[j] = __builtin_bit_cast(f16x8, Ys(c2, 2 * g, (wm * 2 + j) * 16 + r16));
          if constexpr (!XS) xl[j] = __builtin_bit_cast(f16x8, Ys(c2, 2 * g + 1, (wm * 2 + j) * 16 + r16));
        }
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if constexpr (!WS && !XS) acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[i], xh[j], acc2[i][j], 0, 0, 0);
            if constexpr (!XS) acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], xl[j], acc2[i][j], 0, 0, 0);
            acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], xh[j], acc2[i][j], 0, 0, 0);
          }
      }
      __syncthreads();                          // every wave has read the image: the fp32 tile may overwrite it
      if (p == 0) VA_STAMP(stamps, 4);         // pass 0: 1 x 1 products done
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int row = (wm * 2 + j) * 16 + r16, col = wn * (NI * 16) + i * 16 + 4 * g;
          *reinterpret_cast<f32x4*>(stage + row * SP + col) = acc2[i][j] + *reinterpret_cast<const f32x4*>(a.bias2 + col);
        }
      __syncthreads();
      conv_tile_store<BN_, SP, false, 64>(a, stage, b, ph, j0 + p * 64, n0, tid);
      if (p == 0) VA_STAMP(stamps, 5);         // pass 0: tile stored
    };
    pass(std::integral_constant<int, 0>{});
    pass(std::integral_constant<int, 1>{});
    pass(std::integral_constant<int, 2>{});
    pass(std::integral_constant<int, 3>{});
    VA_STAMP(stamps, 6);
    VA_STAMP_FLUSH(stamps, 21);
    return;
  }
  float* stage = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int h = 0; h < MJ / 4; ++h) {          // 128 rows per pass: row block wm * MJ + j of the workgroup, eight per pass
    if (h > 0) __syncthreads();
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
      const int rbk = wm * MJ + j;
      if (rbk / 8 != h) continue;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int row = (rbk % 8) * 16 + r16, col = wn * (NI * 16) + i * 16 + 4 * g;
        *reinterpret_cast<f32x4*>(stage + row * SP + col) = acc[i][j] + *reinterpret_cast<const f32x4*>(a.bias + n0 + col);
      }
    }
    __syncthreads();
    if (h == 0) VA_STAMP(stamps, 4);          // first 128-row pass staged
    conv_tile_store<BN_, SP>(a, stage, b, ph, j0 + h * BM, n0, tid);
    if (h == 0) VA_STAMP(stamps, 5);
  }
  VA_STAMP(stamps, 6);
  VA_STAMP_FLUSH(stamps, 20);
}

// ---------------------------------------------------------------------------------------------
// A whole residual unit — 7-tap dilated conv -> Snake -> 1 x 1 conv -> + residual -> raw / Snake / pairs — in one launch, for the
// widths where ONE workgroup can hold every channel (C = 96: 256 rows per workgroup, C = 192: 128 rows).  The codec is bound by its
// activation streams (profiles/r04_codec_ablations.txt); here the intermediate activation never exists in memory, not even in LDS:
//  * the four waves split the ROWS (wave w owns 16 RB rows x all C columns: accumulators acc[C / 16][RB], 96 registers as before), so a
//    wave ends the first conv holding every channel of its rows;
//  * the first conv's weight tile is stored into LDS with its output channels permuted inside each group of 32 (row 16 i + 4 q + e of
//    the tile image <- channel 32 (i / 2) + 8 q + 4 (i % 2) + e): a permutation of the M rows of the matrix instruction changes no
//    sum, but now accumulator tiles 2 c and 2 c + 1 of lane group q hold channels 32 c + 8 q .. + 7 — which IS the B-operand fragment
//    (row r16, k-octet q of chunk c) the second conv needs.  + bias, Snake(alpha_mid), (hi, lo) split happen in registers and the
//    1 x 1 conv multiplies straight out of them;
//  * the 1 x 1 conv is the main loop again with one tap and a register-resident activation: its weight tiles (96 output channels x 32,
//    like every tile of this file) pass through the same two LDS buffers, C / 96 column tiles one after the other;
//  * same matrix instructions on the same fragments in the same order as the two stand-alone launches: bit-identical results.
template <int N, typename F>
__device__ __forceinline__ void va_static_for(F&& f) {
  if constexpr (N > 0) {
    va_static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

template <int NCT, int RB, bool WS, bool XS>
__global__ __launch_bounds__(256, 2) void conv_unit_kernel(ConvPArgs a) {
  constexpr int C_ = 16 * NCT;               // channels = columns of the workgroup
  constexpr int BMT = 64 * RB;               // rows of the workgroup
  constexpr int KC = C_ / BK;                // 32-channel chunks (of both convs)
  constexpr int XROWS_T = BMT + XHALO;
  constexpr int WSTR = (C_ + 3 + 15) / 16 * 16, XSTR = (XROWS_T + 3 + 15) / 16 * 16, W2STR = (BN + 3 + 15) / 16 * 16;
  constexpr int WS_ELEMS = 2 * (BK / 4) * WSTR, XS_ELEMS = (BK / 4) * XSTR;
  constexpr int SP = BN + 4;                 // fp32 output tile: 64 rows x 96 columns
  constexpr int W2_ELEMS = 2 * (BK / 4) * W2STR, ST_ELEMS = 64 * SP / 4;
  constexpr int MAIN_ELEMS = WS_ELEMS + XS_ELEMS > W2_ELEMS + ST_ELEMS ? WS_ELEMS + XS_ELEMS : W2_ELEMS + ST_ELEMS;
  constexpr int TAB_ELEMS = 2 * C_ / 4;      // alpha_mid and 1 / (alpha_mid + 1e-9) per channel
  static_assert(NCT % 6 == 0 && (RB == 2 || RB == 4) && NCT * RB == 24, "96 channels x 256 rows or 192 x 128: 96 accumulator registers");
  __shared__ u32x4 smem[MAIN_ELEMS + TAB_ELEMS];
  auto coff = [](int kq) { return (kq & 1) + ((kq >> 2) << 1); };
  auto Ws = [&](int buf, int kq, int row) -> u32x4& { return smem[(buf * (BK / 4) + kq) * WSTR + row + coff(kq)]; };
  auto Xs = [&](int kq, int row) -> u32x4& { return smem[WS_ELEMS + kq * XSTR + row + coff(kq)]; };
  auto W2s = [&](int buf, int kq, int row) -> u32x4& { return smem[(buf * (BK / 4) + kq) * W2STR + row + coff(kq)]; };
  float* tab = reinterpret_cast<float*>(smem + MAIN_ELEMS);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r16 = lane & 15;
  const int j0 = blockIdx.x * BMT, b = blockIdx.z;
  constexpr int cq = C_ / 4;                 // 16-B quads per row
  const u32x4* in = reinterpret_cast<const u32x4*>(a.in) + (size_t)b * a.Lin * cq;
  const u32x4* wbase = reinterpret_cast<const u32x4*>(a.w);
  const int NT = a.NT, nk = NT * KC;
  const int span = (NT - 1) * (a.off_step < 0 ? -a.off_step : a.off_step);
  const int lo_off = a.off_base + (a.off_step < 0 ? (NT - 1) * a.off_step : 0);
  const int xrows = BMT + span;
  constexpr int WL = C_ * (BK / 4) / 256;                        // weight quads per thread per step (3 or 6)
  constexpr int XL = (XROWS_T * (BK / 4) + 255) / 256;           // activation quads per thread per chunk

  u32x4 wreg[WL], xreg[XL];
  auto load_w = [&](int kt) {
    const int c = kt / NT, t = kt - c * NT;
    const u32x4* wt = wbase + (size_t)t * C_ * cq + c * (BK / 4);
#pragma unroll
    for (int i = 0; i < WL; ++i) { const int qd = tid + 256 * i; wreg[i] = wt[(size_t)(qd >> 3) * cq + (qd & 7)]; }
  };
  auto store_w = [&](int buf) {          // output channel ch -> tile-image row 32 (ch / 32) + 16 ((ch / 4) % 2) + 4 ((ch % 32) / 8) + ch % 4
#pragma unroll
    for (int i = 0; i < WL; ++i) {
      const int qd = tid + 256 * i, ch = qd >> 3, w5 = ch & 31;
      Ws(buf, qd & 7, (ch & ~31) + (((w5 >> 2) & 1) << 4) + ((w5 >> 3) << 2) + (w5 & 3)) = wreg[i];
    }
  };
  auto load_x = [&](int c) {
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int qd = tid + 256 * i, row = qd >> 3, kq = qd & 7;
      const int jr = j0 + lo_off + row;
      xreg[i] = (row < xrows && jr >= 0 && jr < a.Lin) ? in[(size_t)jr * cq + c * (BK / 4) + kq] : u32x4{0u, 0u, 0u, 0u};
    }
  };
  auto store_x = [&]() {
#pragma unroll
    for (int i = 0; i < XL; ++i) { const int qd = tid + 256 * i; if ((qd >> 3) < XROWS_T) Xs(qd & 7, qd >> 3) = xreg[i]; }
  };

  f32x4 acc[NCT][RB];
#pragma unroll
  for (int i = 0; i < NCT; ++i)
#pragma unroll
    for (int j = 0; j < RB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  VA_STAMP_DECL(stamps);
  VA_STAMP(stamps, 0);                       // (diagnostic build, tools/mfma_driver MFMA_STAMPS) wave start
  load_w(0);
  load_x(0);
  store_w(0);
  store_x();
  for (int ch = tid; ch < C_; ch += 256) {
    const float al = a.alpha_mid[ch];
    tab[ch] = al;
    tab[C_ + ch] = 1.0f / (al + 1e-9f);
  }
  __syncthreads();
  VA_STAMP(stamps, 1);                       // first tiles staged
  int kt = 0;
  for (int c = 0; c < KC; ++c) {
    for (int t = 0; t < NT; ++t, ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < nk) load_w(kt + 1);
      if (t == NT - 1 && c + 1 < KC) load_x(c + 1);
      const int shift = a.off_base + t * a.off_step - lo_off;
      f16x8 xh[RB], xl[RB];
#pragma unroll
      for (int j = 0; j < RB; ++j) {
        xh[j] = __builtin_bit_cast(f16x8, Xs(2 * g, shift + (wv * RB + j) * 16 + r16));
        if constexpr (!XS) xl[j] = __builtin_bit_cast(f16x8, Xs(2 * g + 1, shift + (wv * RB + j) * 16 + r16));
      }
#pragma unroll
      for (int ig = 0; ig < NCT; ig += 3) {
        f16x8 wh[3], wl[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          wh[i] = __builtin_bit_cast(f16x8, Ws(buf, 2 * g, (ig + i) * 16 + r16));
          if constexpr (!WS && !XS) wl[i] = __builtin_bit_cast(f16x8, Ws(buf, 2 * g + 1, (ig + i) * 16 + r16));
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < RB; ++j) {
            if constexpr (!WS && !XS) acc[ig + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[i], xh[j], acc[ig + i][j], 0, 0, 0);
            if constexpr (!XS) acc[ig + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], xl[j], acc[ig + i][j], 0, 0, 0);
            acc[ig + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], xh[j], acc[ig + i][j], 0, 0, 0);
          }
      }
      if (kt + 1 < nk) store_w(buf ^ 1);
      if (t == NT - 1 && c + 1 < KC) {
        __syncthreads();        // every wave is done with the block before it is overwritten
        store_x();
      }
      __syncthreads();
    }
  }

  // ---- the 1 x 1 conv's weight tiles: 96 output channels (column tile nt) x chunk c2, through the first two LDS buffers
  const u32x4* w2q = reinterpret_cast<const u32x4*>(a.w2);
  u32x4 w2reg[3];
  auto load_w2 = [&](int nt, int c2) {
#pragma unroll
    for (int i = 0; i < 3; ++i) { const int qd = tid + 256 * i; w2reg[i] = w2q[(size_t)(nt * BN + (qd >> 3)) * cq + c2 * (BK / 4) + (qd & 7)]; }
  };
  auto store_w2 = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 3; ++i) { const int qd = tid + 256 * i; W2s(buf, qd & 7, qd >> 3) = w2reg[i]; }
  };
  VA_STAMP(stamps, 2);                       // main loop done
  load_w2(0, 0);      // (in flight under the Snake arithmetic below; the main loop ended with a barrier: its LDS is free)

  // ---- first conv's epilogue in registers: + bias, Snake(alpha_mid), (hi, lo) split -> the second conv's B fragments
  f16x8 yh[KC][RB], yl[KC][RB];
  va_static_for<KC>([&](auto cc) {
    constexpr int c2 = decltype(cc)::value;
    const int ch = 32 * c2 + 8 * g;           // this lane's eight channels of the chunk: ch .. ch + 3 in tile 2 c2, ch + 4 .. ch + 7 in tile 2 c2 + 1
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.bias + ch), b1 = *reinterpret_cast<const f32x4*>(a.bias + ch + 4);
    const f32x4 al0 = *reinterpret_cast<const f32x4*>(tab + ch), al1 = *reinterpret_cast<const f32x4*>(tab + ch + 4);
    const f32x4 iv0 = *reinterpret_cast<const f32x4*>(tab + C_ + ch), iv1 = *reinterpret_cast<const f32x4*>(tab + C_ + ch + 4);
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      const f32x4 v0 = acc[2 * c2][j] + b0, v1 = acc[2 * c2 + 1][j] + b1;
      f16x8 h, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float s0 = snake_fi(v0[e], al0[e], iv0[e]), s1 = snake_fi(v1[e], al1[e], iv1[e]);
        h[e] = (_Float16)s0; l[e] = (_Float16)(s0 - (float)h[e]);
        h[e + 4] = (_Float16)s1; l[e + 4] = (_Float16)(s1 - (float)h[e + 4]);
      }
      yh[c2][j] = h;
      yl[c2][j] = l;
    }
  });

  VA_STAMP(stamps, 3);                       // Snake + split in registers done
  // ---- second conv: column tiles of 96 output channels, k = the C channels held in registers
  float* stage = reinterpret_cast<float*>(smem + W2_ELEMS);
  constexpr int WPP = 4 / RB;                 // waves that own a 64-row pass (1 or 2)
#pragma unroll 1
  for (int nt = 0; nt < C_ / BN; ++nt) {
    f32x4 acc2[BN / 16][RB];
#pragma unroll
    for (int i = 0; i < BN / 16; ++i)
#pragma unroll
      for (int j = 0; j < RB; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (nt > 0) load_w2(nt, 0);
    store_w2(0);
    __syncthreads();
    va_static_for<KC>([&](auto cc) {
      constexpr int c2 = decltype(cc)::value;
      constexpr int buf = c2 & 1;
      if (c2 + 1 < KC) load_w2(nt, c2 + 1);
#pragma unroll
      for (int ig = 0; ig < BN / 16; ig += 3) {
        f16x8 wh[3], wl[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          wh[i] = __builtin_bit_cast(f16x8, W2s(buf, 2 * g, (ig + i) * 16 + r16));
          if constexpr (!WS && !XS) wl[i] = __builtin_bit_cast(f16x8, W2s(buf, 2 * g + 1, (ig + i) * 16 + r16));
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < RB; ++j) {
            if constexpr (!WS && !XS) acc2[ig + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[i], yh[c2][j], acc2[ig + i][j], 0, 0, 0);
            if constexpr (!XS) acc2[ig + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], yl[c2][j], acc2[ig + i][j], 0, 0, 0);
            acc2[ig + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], yh[c2][j], acc2[ig + i][j], 0, 0, 0);
          }
      }
      if (c2 + 1 < KC) store_w2(buf ^ 1);
      __syncthreads();
    });
    if (nt == 0) VA_STAMP(stamps, 4);        // first column tile: 1 x 1 products done
    // 64 rows at a time through the fp32 tile (owned by wave p with 256 rows, by waves 2 p and 2 p + 1 with 128)
#pragma unroll
    for (int p = 0; p < BMT / 64; ++p) {
      if (p > 0) __syncthreads();               // the previous pass's store is done with the tile
      if (wv / WPP == p) {
#pragma unroll
        for (int i = 0; i < BN / 16; ++i)
#pragma unroll
          for (int j = 0; j < RB; ++j) {
            const int row = ((wv % WPP) * RB + j) * 16 + r16, col = i * 16 + 4 * g;
            *reinterpret_cast<f32x4*>(stage + row * SP + col) = acc2[i][j] + *reinterpret_cast<const f32x4*>(a.bias2 + nt * BN + col);
          }
      }
      __syncthreads();
      conv_tile_store<BN, SP, false, 64>(a, stage, b, 0, j0 + p * 64, nt * BN, tid);
    }
    if (nt == 0) VA_STAMP(stamps, 5);        // first column tile stored
    if (nt + 1 < C_ / BN) __syncthreads();      // ... before the next column tile's passes reuse it
  }
  VA_STAMP(stamps, 6);
  VA_STAMP_FLUSH(stamps, 22);
}

// ---------------------------------------------------------------------------------------------
// Block-scaled fp8 variant (codec precision 3; BASELINE configs[4] "fp8 MFMA ... codec conv"): activations are e4m3 bytes
// with one power-of-two (E8M0) scale per 32 channels of a row — written by the producer's epilogue above — weights are
// e4m3 with one power-of-two scale per output channel, and a product is ONE v_mfma_scale_f32_16x16x128_f8f6f4 per 128 k
// (2x the bf16 MFMA rate; the activation's block scale rides on the instruction, the weight's is applied to the
// accumulator).  Staging geometry is the pair kernel's — 128 B per row per step, there 32 channels x (hi, lo) x 2 B, here
// 128 channels x 1 B — so a step covers 4x the channels in 2/3 of the matrix-pipe time.
//   k order.  A "super-chunk" is 128 input channels (the last one of a layer may hold 32/64/96: C = 96, 192); its k-blocks
//   (32 channels of one tap) are numbered kb = tap * nch + ch and MFMA step s takes kb = 4s .. 4s+3 (one scale block of
//   the instruction each): with nch = 4 a step is one tap (every lane the same row shift, like the pair kernel), with
//   nch < 4 the blocks of a step come from different taps.  The weight stream is packed by the caller in the lane order
//   of the instruction (32 bytes per lane group, see the main loop), zero where kb >= nch * taps (those lanes re-read the
//   last valid activation block so that no NaN byte meets the zero).
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int NI>
__global__ __launch_bounds__(256, 2) void conv_mx8_kernel(ConvPArgs a) {
  constexpr int BN_ = 32 * NI;
  constexpr int WS_ELEMS = 2 * 8 * (BN_ + 1), XS_ELEMS = 2 * 8 * (XROWS + 1);
  constexpr int SP = BN_ + 4;
  static_assert((WS_ELEMS + XS_ELEMS) * 16 >= BM * SP * 4, "the output tile must fit in the main loop's LDS");
  __shared__ u32x4 smem[WS_ELEMS + XS_ELEMS];
  __shared__ uint32_t xsc[2][XROWS];
  auto Ws = [&](int buf, int kq, int row) -> u32x4& { return smem[(buf * 8 + kq) * (BN_ + 1) + row]; };
  auto Xs = [&](int buf, int kq, int row) -> u32x4& { return smem[WS_ELEMS + (buf * 8 + kq) * (XROWS + 1) + row]; };
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wn = wv & 1, wm = wv >> 1;
  const int j0 = blockIdx.x * BM;
  const int n0 = blockIdx.y * BN_;
  const int phases = a.ostride;
  const int b = blockIdx.z / phases, ph = blockIdx.z % phases;
  const int cq = a.Cin / 16;                // 16-B quads per row
  const int NT = a.NT;
  const int nsc = (a.Cin + 127) >> 7;
  const int nch_last = (a.Cin - 128 * (nsc - 1)) >> 5;
  const int steps_last = (nch_last * NT + 3) >> 2;
  const int nk = (nsc - 1) * NT + steps_last;
  const u32x4* in = reinterpret_cast<const u32x4*>(a.in) + (size_t)b * a.Lin * cq;
  const uint32_t* insc = a.in_scale + (size_t)b * a.Lin * nsc;
  const u32x4* wbase = reinterpret_cast<const u32x4*>(a.w) + (size_t)ph * nk * a.Cout * 8;
  const int span = (NT - 1) * (a.off_step < 0 ? -a.off_step : a.off_step);
  const int lo_off = a.off_base + (a.off_step < 0 ? (NT - 1) * a.off_step : 0);
  const int xrows = BM + span;
  constexpr int XL = (XROWS * 8 + 255) / 256;

  u32x4 wreg[NI], xreg[XL];
  uint32_t screg = 0;
  auto load_w = [&](int kt) {
    const u32x4* wt = wbase + (size_t)kt * a.Cout * 8;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int qd = tid + 256 * i;
      wreg[i] = wt[(size_t)(n0 + (qd >> 3)) * 8 + (qd & 7)];
    }
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NI; ++i) { const int qd = tid + 256 * i; Ws(buf, qd & 7, qd >> 3) = wreg[i]; }
  };
  auto load_x = [&](int sc) {
    const int q0 = sc * 8, nq = (sc == nsc - 1 ? nch_last : 4) * 2;
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int qd = tid + 256 * i, row = qd >> 3, kq = qd & 7;
      const int jr = j0 + lo_off + row;
      xreg[i] = (row < xrows && kq < nq && jr >= 0 && jr < a.Lin) ? in[(size_t)jr * cq + q0 + kq] : u32x4{0u, 0u, 0u, 0u};
    }
    const int jr = j0 + lo_off + tid;
    screg = (tid < xrows && jr >= 0 && jr < a.Lin) ? insc[(size_t)jr * nsc + sc] : 0u;
  };
  auto store_x = [&](int buf) {
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int qd = tid + 256 * i;
      if ((qd >> 3) < XROWS) Xs(buf, qd & 7, qd >> 3) = xreg[i];
    }
    if (tid < XROWS) xsc[buf][tid] = screg;
  };

  f32x4 acc[NI][4];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  load_w(0);
  load_x(0);
  store_w(0);
  store_x(0);
  __syncthreads();
  const int g = lane >> 4, r16 = lane & 15;
  int kt = 0;
  for (int sc = 0; sc < nsc; ++sc) {
    const int xb = sc & 1;
    const int nch = sc == nsc - 1 ? nch_last : 4;
    const int nsteps = sc == nsc - 1 ? steps_last : NT;
    const int nkb = nch * NT;
    const float rnch = 1.0f / (float)nch;
    for (int st = 0; st < nsteps; ++st, ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < nk) load_w(kt + 1);
      if (st == 0 && sc + 1 < nsc) load_x(sc + 1);
      // operand layout of the instruction (tools/microbench/mfma_mx8_probe.hip): bytes 0..15 of lane group G are
      // k = 16G .. 16G+15, bytes 16..31 are k = 64 + 16G ..; the scale of lane group G covers k = 32G .. 32G+31.  So the
      // lane's first quad is half (G & 1) of k-block G >> 1, its second quad the same half of k-block 2 + (G >> 1), and
      // its scale is k-block G's.
      const int h = g & 1;
      int chA = g >> 1, chB = 2 + (g >> 1), chS = g;
      int shA = a.off_base + st * a.off_step - lo_off, shB = shA, shS = shA;
      if (nch != 4) {
        auto split = [&](int kb, int& ch) {
          kb = kb < nkb ? kb : nkb - 1;
          const int t = (int)(((float)kb + 0.5f) * rnch);     // kb / nch, exact for these small integers
          ch = kb - t * nch;
          return a.off_base + t * a.off_step - lo_off;
        };
        shA = split(4 * st + (g >> 1), chA);
        shB = split(4 * st + 2 + (g >> 1), chB);
        shS = split(4 * st + g, chS);
      }
      i32x8 wf[NI], xf[4];
      int xs[4];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const u32x4 q0 = Ws(buf, 2 * g, wn * (NI * 16) + i * 16 + r16), q1 = Ws(buf, 2 * g + 1, wn * (NI * 16) + i * 16 + r16);
        wf[i] = i32x8{(int)q0.x, (int)q0.y, (int)q0.z, (int)q0.w, (int)q1.x, (int)q1.y, (int)q1.z, (int)q1.w};
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = wm * 64 + j * 16 + r16;
        const u32x4 q0 = Xs(xb, 2 * chA + h, shA + row), q1 = Xs(xb, 2 * chB + h, shB + row);
        xf[j] = i32x8{(int)q0.x, (int)q0.y, (int)q0.z, (int)q0.w, (int)q1.x, (int)q1.y, (int)q1.z, (int)q1.w};
        xs[j] = (int)((xsc[xb][shS + row] >> (8 * chS)) & 0xffu);
      }
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[i], xf[j], acc[i][j], 0, 0, 0, 127, 0, xs[j]);
      if (kt + 1 < nk) store_w(buf ^ 1);
      if (st == nsteps - 1 && sc + 1 < nsc) store_x(xb ^ 1);
      __syncthreads();
    }
  }

  float* stage = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int row = wm * 64 + j * 16 + r16, col = wn * (NI * 16) + i * 16 + 4 * g;
      *reinterpret_cast<f32x4*>(stage + row * SP + col) =
          acc[i][j] * *reinterpret_cast<const f32x4*>(a.wscale + n0 + col) + *reinterpret_cast<const f32x4*>(a.bias + n0 + col);
    }
  __syncthreads();
  conv_tile_store<BN_, SP, true>(a, stage, b, ph, j0, n0, tid);
}

// z[b][t][c] = sum_k ( W_k[c][:] . codebook_k[code] + b_k[c] )   — quantizer.from_codes
// A workgroup takes FC_NT frames of one clip: a thread keeps the 9 x 8 projection weights of a channel in registers and
// reuses them for every frame (one workgroup per frame re-read the 295 KB of projection weights 1 760 times: 219 us).
#define FC_NT 8
__global__ __launch_bounds__(256) void from_codes_kernel(const int32_t* __restrict__ codes, const float* __restrict__ cb,
                                                          const float* __restrict__ pw, const float* __restrict__ pb,
                                                          float* __restrict__ z, int K, int T, int size, int dim, int latent,
                                                          int pairs) {
  __shared__ float e[FC_NT][16][8];
  const int t0 = blockIdx.x * FC_NT, b = blockIdx.y;
  if (dim < 8) {     // slots i >= dim meet zero weights below: they must hold finite numbers
    for (int u = threadIdx.x; u < FC_NT * 16 * 8; u += blockDim.x) (&e[0][0][0])[u] = 0.f;
    __syncthreads();
  }
  for (int u = threadIdx.x; u < FC_NT * K * dim; u += blockDim.x) {
    const int f = u / (K * dim), k = (u / dim) % K, i = u % dim;
    const int t = t0 + f < T ? t0 + f : T - 1;
    const int code = codes[((size_t)b * K + k) * T + t];
    e[f][k][i] = cb[((size_t)k * size + code) * dim + i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < latent; c += blockDim.x) {
    float w[16][8], bias[16];       // fully unrolled with predicates: the arrays stay in registers (K <= 16, dim <= 8)
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float* wr = pw + ((size_t)(k < K ? k : 0) * latent + c) * dim;
#pragma unroll
      for (int i = 0; i < 8; ++i) w[k][i] = (k < K && i < dim) ? wr[i] : 0.f;      // a zero weight leaves the fmaf chain unchanged
      bias[k] = k < K ? pb[(size_t)k * latent + c] : 0.f;
    }
    for (int f = 0; f < FC_NT && t0 + f < T; ++f) {
      float o = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if (k < K) {
          float zk = 0.f;
#pragma unroll
          for (int i = 0; i < 8; ++i) zk = fmaf(w[k][i], e[f][k][i], zk);
          o += zk + bias[k];
        }
      }
      const int t = t0 + f;
      if (pairs) {
        const _Float16 h = (_Float16)o, l = (_Float16)(o - (float)h);
        _Float16* zp = reinterpret_cast<_Float16*>(z) + ((((size_t)b * T + t) * (latent >> 3) + (c >> 3)) * 2) * 8 + (c & 7);
        zp[0] = h;
        zp[8] = l;
      } else {
        z[((size_t)b * T + t) * latent + c] = o;
      }
    }
  }
}

// wav[b][l] = tanh( bias + sum_t sum_c act[l + t - 3][c] * w[t][c] )   — last conv (C -> 1, k = 7)
__global__ __launch_bounds__(256) void conv_out_kernel(const float* __restrict__ act, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ wav, int L, int C,
                                                        int pairs) {
  const int b = blockIdx.y;
  const int sub = threadIdx.x & 7;
  const int l = blockIdx.x * 32 + (threadIdx.x >> 3);
  const float* in = act + (size_t)b * L * C;
  float d = 0.f;
  for (int t = 0; t < 7; ++t) {
    const int r = l + t - 3;
    if (r < 0 || r >= L || l >= L) continue;
    for (int cq = sub; cq < C / 4; cq += 8) {
      f32x4 x;
      if (pairs == 3) {          // mx8: e4m3 bytes + one E8M0 scale byte per 32 channels, scales behind the bytes
        const uint8_t* q = reinterpret_cast<const uint8_t*>(act) + (size_t)b * L * C;
        const uint8_t* sc = reinterpret_cast<const uint8_t*>(act) + mx8_scale_offset((size_t)gridDim.y * L * C) +
                            (size_t)b * L * ((C + 127) >> 7) * 4;
        const int w4 = *reinterpret_cast<const int*>(q + (size_t)r * C + 4 * cq);
        const int e8 = sc[((size_t)r * ((C + 127) >> 7) + (cq >> 5)) * 4 + ((cq >> 3) & 3)];
        const float scl = __builtin_bit_cast(float, (uint32_t)e8 << 23);
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 lo2 = __builtin_amdgcn_cvt_pk_f32_fp8(w4, false), hi2 = __builtin_amdgcn_cvt_pk_f32_fp8(w4, true);
        x = f32x4{lo2.x * scl, lo2.y * scl, hi2.x * scl, hi2.y * scl};
      } else if (pairs) {
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        const f16x4* pp = reinterpret_cast<const f16x4*>(reinterpret_cast<const _Float16*>(in) +
                                                          (((size_t)r * (C >> 3) + (cq >> 1)) * 2) * 8 + (cq & 1) * 4);
        const f16x4 h = pp[0], l = pp[2];
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = (float)h[e] + (float)l[e];
      } else {
        x = *reinterpret_cast<const f32x4*>(in + (size_t)r * C + 4 * cq);
      }
      const f32x4 ww = *reinterpret_cast<const f32x4*>(w + (size_t)t * C + 4 * cq);
      d = fmaf(x[0], ww[0], d); d = fmaf(x[1], ww[1], d); d = fmaf(x[2], ww[2], d); d = fmaf(x[3], ww[3], d);
    }
  }
  d += __shfl_xor(d, 1, 64);
  d += __shfl_xor(d, 2, 64);
  d += __shfl_xor(d, 4, 64);
  if (sub == 0 && l < L) wav[(size_t)b * L + l] = tanhf(d + bias[0]);
}

// The same last conv with the 7-row window of every output staged ONCE: a workgroup takes 128 outputs, stages rows
// l0-3 .. l0+130 of the activation (dequantised to fp32, row stride C + 4 floats = an odd number of 16-byte quads: the 16-byte
// reads of consecutive outputs hit different banks) and two waves per channel half sum 7 x C/2 products per output in four
// independent chains, with wave-uniform weight addresses (scalar loads).
// The kernel above fetched every activation row seven times through L1 (275 us for 346 MB); C <= 128, C % 8 == 0.
#define CO_TL 128
__global__ __launch_bounds__(256) void conv_out_tiled_kernel(const float* __restrict__ act, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ wav, int L, int C,
                                                              int pairs) {
  extern __shared__ __attribute__((aligned(16))) float rows[];      // (CO_TL + 6) x (C + 1) | CO_TL partial sums
  const int b = blockIdx.y, l0 = blockIdx.x * CO_TL, tid = threadIdx.x;
  const int SR = C + 4, oct = C >> 3;
  for (int u = tid; u < (CO_TL + 6) * oct; u += 256) {
    const int rr = u / oct, oc = u - rr * oct;
    const int r = l0 - 3 + rr;
    float x[8];
    if (r < 0 || r >= L) {
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = 0.f;
    } else if (pairs == 3) {
      const uint8_t* q = reinterpret_cast<const uint8_t*>(act) + (size_t)b * L * C;
      const int nsc = (C + 127) >> 7;
      const uint8_t* sc = reinterpret_cast<const uint8_t*>(act) + mx8_scale_offset((size_t)gridDim.y * L * C) + (size_t)b * L * nsc * 4;
      typedef int i32x2 __attribute__((ext_vector_type(2)));
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      const i32x2 w8 = *reinterpret_cast<const i32x2*>(q + (size_t)r * C + 8 * oc);
      const int e8 = sc[((size_t)r * nsc + (oc >> 4)) * 4 + ((oc >> 2) & 3)];
      const float scl = __builtin_bit_cast(float, (uint32_t)e8 << 23);
      const f32x2 a0 = __builtin_amdgcn_cvt_pk_f32_fp8(w8.x, false), a1 = __builtin_amdgcn_cvt_pk_f32_fp8(w8.x, true);
      const f32x2 a2 = __builtin_amdgcn_cvt_pk_f32_fp8(w8.y, false), a3 = __builtin_amdgcn_cvt_pk_f32_fp8(w8.y, true);
      x[0] = a0.x * scl; x[1] = a0.y * scl; x[2] = a1.x * scl; x[3] = a1.y * scl;
      x[4] = a2.x * scl; x[5] = a2.y * scl; x[6] = a3.x * scl; x[7] = a3.y * scl;
    } else if (pairs) {
      const f16x8* pp = reinterpret_cast<const f16x8*>(reinterpret_cast<const _Float16*>(act) + (((size_t)b * L + r) * oct + oc) * 16);
      const f16x8 h = pp[0], lo = pp[1];
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = (float)h[e] + (float)lo[e];
    } else {
      const f32x4* pp = reinterpret_cast<const f32x4*>(act + ((size_t)b * L + r) * C + 8 * oc);
      const f32x4 a0 = pp[0], a1 = pp[1];
      x[0] = a0[0]; x[1] = a0[1]; x[2] = a0[2]; x[3] = a0[3]; x[4] = a1[0]; x[5] = a1[1]; x[6] = a1[2]; x[7] = a1[3];
    }
    *reinterpret_cast<f32x4*>(rows + rr * SR + 8 * oc) = f32x4{x[0], x[1], x[2], x[3]};
    *reinterpret_cast<f32x4*>(rows + rr * SR + 8 * oc + 4) = f32x4{x[4], x[5], x[6], x[7]};
  }
  __syncthreads();
  const int half = __builtin_amdgcn_readfirstlane(tid >> 7), lo_ = tid & 127;      // waves 0, 1: channels [0, C/2); waves 2, 3: the rest
  const int c0 = half * (C >> 1), c1 = c0 + (C >> 1);
  f32x4 d4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    const f32x4* xr = reinterpret_cast<const f32x4*>(rows + (lo_ + t) * SR + c0);
    const f32x4* wr = reinterpret_cast<const f32x4*>(w + (size_t)t * C + c0);
#pragma unroll 4
    for (int c = 0; c < (c1 - c0) / 4; ++c) {
      const f32x4 x = xr[c], ww = wr[c];
      d4[0] = fmaf(x[0], ww[0], d4[0]); d4[1] = fmaf(x[1], ww[1], d4[1]); d4[2] = fmaf(x[2], ww[2], d4[2]); d4[3] = fmaf(x[3], ww[3], d4[3]);
    }
  }
  const float d = (d4[0] + d4[1]) + (d4[2] + d4[3]);
  float* part = rows + (CO_TL + 6) * SR;
  if (half) part[lo_] = d;
  __syncthreads();
  if (!half && l0 + lo_ < L) wav[(size_t)b * L + l0 + lo_] = tanhf((d + part[lo_]) + bias[0]);
}

// fp32 (rows, C) -> the activation format of a codec precision (op-level entry vaura_dac_conv below)
__global__ __launch_bounds__(256) void act_convert_kernel(const float* __restrict__ in, void* out, uint8_t* out_scale, size_t rows,
                                                          int C, int fmt) {
  const size_t u = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int oct = C >> 3;
  const size_t row = u / oct;
  if (row >= rows) return;                      // C % 32 == 0: the 4 threads of a block leave together
  const int co = (int)(u - row * oct) * 8;
  const f32x4 v0 = *reinterpret_cast<const f32x4*>(in + row * C + co), v1 = *reinterpret_cast<const f32x4*>(in + row * C + co + 4);
  const float sv[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
  store_act_octet(out, out_scale, fmt, row, co, C, sv);
}

// host-side count of conv launches that took the 256-row workgroup instances (conv_pair_kernel<..., 8>): a test that claims to
// check those instances against the oracle asserts that they were the ones launched (vaura_debug_counter(0); read-and-clear)
static long long va_conv256_launches = 0;

static int launch_conv(const vaura_conv& cv, const float* in, const float* res, const float* alpha, float* out_raw,
                       float* out_act, int B, int Lin, int pairs, hipStream_t s) {
  if (!cv.w || !cv.bias || (cv.cin % BK)) return VAURA_ERR_SHAPE;
  if (out_act && !alpha) return VAURA_ERR_ARG;
  if (pairs) {
    if ((cv.cout % BN) && (cv.cout % 64)) return VAURA_ERR_SHAPE;
    ConvPArgs p;
    p.in = reinterpret_cast<const uint16_t*>(in); p.w = reinterpret_cast<const uint16_t*>(cv.w); p.bias = cv.bias; p.res = res;
    p.alpha = alpha; p.out_raw = out_raw; p.out_act = reinterpret_cast<uint16_t*>(out_act);
    p.Lin = Lin; p.Cin = cv.cin; p.Cout = cv.cout; p.act = 0;
    p.act_fmt = 0; p.out_scale = nullptr; p.in_scale = nullptr; p.wscale = nullptr;
    p.w2 = nullptr; p.bias2 = nullptr; p.alpha_mid = nullptr;
    int ph = 1;
    if (cv.stride > 1) {
      if (cv.stride % 2) return VAURA_ERR_SHAPE;
      p.NT = 2; p.off_base = 0; p.off_step = -1; p.ostride = cv.stride; p.oshift0 = -(cv.stride / 2);
      p.Lout = Lin * cv.stride; p.jcount = Lin + 1; ph = cv.stride;
    } else {
      p.NT = cv.taps; p.off_base = -((cv.taps - 1) / 2) * cv.dilation; p.off_step = cv.dilation;
      p.ostride = 1; p.oshift0 = 0; p.Lout = Lin; p.jcount = Lin;
      if ((cv.taps - 1) * cv.dilation > XHALO) return VAURA_ERR_SHAPE;
    }
    if (pairs == 3) {   // mx8 activations out; in as well when the layer's weights are mx8 (everything but conv_in)
      if (out_act) {
        p.act_fmt = 1;
        p.out_scale = reinterpret_cast<uint8_t*>(out_act) + mx8_scale_offset((size_t)B * p.Lout * cv.cout);
      }
      if (cv.wscale) {
        if (cv.cout % BN) return VAURA_ERR_SHAPE;
        p.wscale = cv.wscale;
        p.in_scale = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(in) + mx8_scale_offset((size_t)B * Lin * cv.cin));
        VA_LAUNCH((conv_mx8_kernel<3>), dim3((p.jcount + BM - 1) / BM, cv.cout / BN, B * ph), dim3(256), 0, s, p);
        return 0;
      }
      pairs = 2;        // conv_in: fp8-valued weights in one fp16 plane, pair input from the quantizer
    }
    if (pairs == 4) {   // "f16": one matrix instruction per product (hi planes only)
      const int g256 = (p.jcount + 2 * BM - 1) / (2 * BM);
      if (cv.cout % BN == 0 && !(va_debug_flags_get() & 0x100000u) && (int64_t)g256 * (cv.cout / BN) * B * ph >= 384) {
        ++va_conv256_launches;
        VA_LAUNCH((conv_pair_kernel<3, true, true, 8>), dim3(g256, cv.cout / BN, B * ph), dim3(256), 0, s, p);
      } else if (cv.cout % BN == 0) VA_LAUNCH((conv_pair_kernel<3, true, true>), dim3((p.jcount + BM - 1) / BM, cv.cout / BN, B * ph), dim3(256), 0, s, p);
      else VA_LAUNCH((conv_pair_kernel<2, true, true>), dim3((p.jcount + BM - 1) / BM, cv.cout / 64, B * ph), dim3(256), 0, s, p);
      return 0;
    }
    if (cv.cout % BN == 0) {
      // 256-row workgroups (half the weight bytes through L2 per output) where two per CU still fill the chip; debug flag bit 20: never
      const int g256 = (p.jcount + 2 * BM - 1) / (2 * BM);
      const bool big = !(va_debug_flags_get() & 0x100000u) && (int64_t)g256 * (cv.cout / BN) * B * ph >= 384;
      if (big) {
        ++va_conv256_launches;
        if (pairs == 2) VA_LAUNCH((conv_pair_kernel<3, true, false, 8>), dim3(g256, cv.cout / BN, B * ph), dim3(256), 0, s, p);
        else VA_LAUNCH((conv_pair_kernel<3, false, false, 8>), dim3(g256, cv.cout / BN, B * ph), dim3(256), 0, s, p);
      } else if (pairs == 2) VA_LAUNCH((conv_pair_kernel<3, true>), dim3((p.jcount + BM - 1) / BM, cv.cout / BN, B * ph), dim3(256), 0, s, p);
      else VA_LAUNCH((conv_pair_kernel<3, false>), dim3((p.jcount + BM - 1) / BM, cv.cout / BN, B * ph), dim3(256), 0, s, p);
    } else {
      const int g256 = (p.jcount + 2 * BM - 1) / (2 * BM);
      const bool big = !(va_debug_flags_get() & 0x100000u) && (int64_t)g256 * (cv.cout / 64) * B * ph >= 384;
      if (big) {
        ++va_conv256_launches;
        if (pairs == 2) VA_LAUNCH((conv_pair_kernel<2, true, false, 8>), dim3(g256, cv.cout / 64, B * ph), dim3(256), 0, s, p);
        else VA_LAUNCH((conv_pair_kernel<2, false, false, 8>), dim3(g256, cv.cout / 64, B * ph), dim3(256), 0, s, p);
      } else if (pairs == 2) VA_LAUNCH((conv_pair_kernel<2, true>), dim3((p.jcount + BM - 1) / BM, cv.cout / 64, B * ph), dim3(256), 0, s, p);
      else VA_LAUNCH((conv_pair_kernel<2, false>), dim3((p.jcount + BM - 1) / BM, cv.cout / 64, B * ph), dim3(256), 0, s, p);
    }
    return 0;
  }
  if (cv.cout % BN) return VAURA_ERR_SHAPE;
  ConvArgs a;
  a.in = in; a.w = cv.w; a.bias = cv.bias; a.res = res; a.alpha = alpha; a.out_raw = out_raw; a.out_act = out_act;
  a.Lin = Lin; a.Cin = cv.cin; a.Cout = cv.cout;
  int phases = 1;
  if (cv.stride > 1) {   // transposed: k = 2r, pad = r/2
    if (cv.stride % 2) return VAURA_ERR_SHAPE;
    a.NT = 2; a.off_base = 0; a.off_step = -1; a.ostride = cv.stride; a.oshift0 = -(cv.stride / 2);
    a.Lout = Lin * cv.stride; a.jcount = Lin + 1; phases = cv.stride;
  } else {
    a.NT = cv.taps; a.off_base = -((cv.taps - 1) / 2) * cv.dilation; a.off_step = cv.dilation;
    a.ostride = 1; a.oshift0 = 0; a.Lout = Lin; a.jcount = Lin;
  }
  dim3 grid((a.jcount + BM - 1) / BM, cv.cout / BN, B * phases);
  VA_LAUNCH(conv_mfma_kernel, grid, dim3(256), 0, s, a);
  return 0;
}


// A whole residual unit (7-tap dilated conv -> Snake -> 1 x 1 conv -> + residual) as ONE launch where one workgroup holds every channel
// (C = 96: the codec's last block, a third of its activation bytes).  The codec is bound by its activation streams, not by the matrix
// pipe (profiles/r04_codec_ablations.txt): fused, the intermediate activation (4 B per element written and read back) never leaves the
// CU.  Same sums in the same order as the two launches: bit-identical (tests/test_gpu_generate.py).  Returns VA_UNIT_NOT_ELIGIBLE when
// the unit is not of that shape (the caller then launches the two convs), 0 when launched, anything else = an ERROR to propagate (a
// negative VAURA_ERR_* or the positive hipError_t of a failed launch — VA_LAUNCH returns those, and hipErrorInvalidValue IS 1: "not
// eligible" must not share a value with them).  out_act must not alias `in` (neighbours read halo rows).
#define VA_UNIT_NOT_ELIGIBLE (-0x7fffff00)
static long long va_conv_units_fused = 0;
static int launch_conv_unit(const vaura_conv& c7, const vaura_conv& c1, const float* in, const float* res, const float* alpha_mid,
                            const float* alpha_next, float* out_raw, float* out_act, int B, int L, int pairs, hipStream_t s) {
  if (pairs != 1 && pairs != 2 && pairs != 4) return VA_UNIT_NOT_ELIGIBLE;
  if (va_debug_flags_get() & (0x200000u | 0x100000u)) return VA_UNIT_NOT_ELIGIBLE;       // debug flag bit 21: the two-launch form (bit 20: no 256-row instances at all)
  const int C = c7.cin;
  if ((C != BN && C != 2 * BN) || c7.cout != C || c1.cin != C || c1.cout != C || c7.stride != 1 || c1.stride != 1 || c1.taps != 1 || c7.taps < 1 ||
      (c7.taps - 1) * c7.dilation > XHALO || !c7.w || !c7.bias || !c1.w || !c1.bias || !alpha_mid || !alpha_next || !res || !out_act || (const float*)out_act == in)
    return VA_UNIT_NOT_ELIGIBLE;
  const int rows = C == BN ? 256 : 128;
  const int gx = (L + rows - 1) / rows;
  if ((int64_t)gx * B < 384) return VA_UNIT_NOT_ELIGIBLE;
  ConvPArgs p;
  p.in = reinterpret_cast<const uint16_t*>(in); p.w = reinterpret_cast<const uint16_t*>(c7.w); p.bias = c7.bias; p.res = res;
  p.alpha = alpha_next; p.out_raw = out_raw; p.out_act = reinterpret_cast<uint16_t*>(out_act);
  p.Lin = L; p.Lout = L; p.Cin = C; p.Cout = C; p.act = 0;
  p.act_fmt = 0; p.out_scale = nullptr; p.in_scale = nullptr; p.wscale = nullptr;
  p.NT = c7.taps; p.off_base = -((c7.taps - 1) / 2) * c7.dilation; p.off_step = c7.dilation;
  p.ostride = 1; p.oshift0 = 0; p.jcount = L;
  p.w2 = reinterpret_cast<const uint16_t*>(c1.w); p.bias2 = c1.bias; p.alpha_mid = alpha_mid;
  ++va_conv_units_fused;
  const dim3 grid(gx, 1, B);
  if (C == BN) {      // 96 channels: conv_pair_kernel's own fused epilogue (the row-split kernel's 256-row instance does not fit the register budget)
    if (pairs == 4) VA_LAUNCH((conv_pair_kernel<3, true, true, 8, true>), grid, dim3(256), 0, s, p);
    else if (pairs == 2) VA_LAUNCH((conv_pair_kernel<3, true, false, 8, true>), grid, dim3(256), 0, s, p);
    else VA_LAUNCH((conv_pair_kernel<3, false, false, 8, true>), grid, dim3(256), 0, s, p);
  } else {
    if (pairs == 4) VA_LAUNCH((conv_unit_kernel<12, 2, true, true>), grid, dim3(256), 0, s, p);
    else if (pairs == 2) VA_LAUNCH((conv_unit_kernel<12, 2, true, false>), grid, dim3(256), 0, s, p);
    else VA_LAUNCH((conv_unit_kernel<12, 2, false, false>), grid, dim3(256), 0, s, p);
  }
  return 0;
}


// ---------------------------------------------------------------------------------------------
// Row f2's linear layers (vit.hip) are one-tap problems: no halo to reuse, so per k-step both operand tiles come through
// L2 and the 128 x 96 conv tile above moves 28 KB per 2.4 MFLOP — it sat at the L2 rate (130 TFLOP/s-equivalent).  This
// kernel is the same pair arithmetic on a 128 x 192 tile (wave tile 64 x 96: 20 fragment reads per 72 MFMAs instead of
// 14 per 36; 40 KB per 4.7 MFLOP), ONE LDS stage of 40 KB (global -> registers runs ahead by one step, two barriers per
// step) so that two workgroups share a CU and one's epilogue / barriers sit under the other's matrix instructions, and
// an XCD-aware tile order: workgroup id % 8 is the XCD, and an XCD walks "its" row tiles with the column tiles fastest,
// so an activation tile is fetched into one L2 only and reused there by every column tile.
#define LBN 192
// VA_LIN_ABL: timing ablations for tools/experiment.sh linear-abl (WRONG RESULTS; product build = 0): 1 no global loads inside the
// k loop, 2 no fragment reads / matrix instructions, 4 no LDS staging stores, 8 no global traffic in the epilogue
#ifndef VA_LIN_ABL
#define VA_LIN_ABL 0
#endif
// MW = wave rows: 2 -> 128 x 192 tile, 256 threads, two workgroups per CU; 4 -> 256 x 192 tile, 512 threads, one per CU
template <int MW>
__global__ __launch_bounds__(128 * MW, MW == 2 ? 2 : 1) void linear_pair_kernel(ConvPArgs a, int mtiles, int ntiles) {
  constexpr int NTH = 128 * MW, LBM_ = 64 * MW, NWQ = LBN * 8 / NTH, NXQ = LBM_ * 8 / NTH;
  __shared__ u32x4 smem[8 * (LBN + LBM_)];
  // [kq][row ^ kq]: fragment reads (16 consecutive rows, one kq) and staging writes (one row, 8 kq) are both conflict-free
  auto Ws = [&](int kq, int row) -> u32x4& { return smem[kq * LBN + (row ^ kq)]; };
  auto Xs = [&](int kq, int row) -> u32x4& { return smem[8 * LBN + kq * LBM_ + (row ^ kq)]; };
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wn = wv & 1, wm = wv >> 1;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int nt = local % ntiles, mt = (local / ntiles) * 8 + xcd;
  if (mt >= mtiles) return;
  const int b = blockIdx.y;
  const int j0 = mt * LBM_, n0 = nt * LBN;
  const int cq = a.Cin / 4;                 // 16-B quads per row (C/8 octets x 2 planes)
  const u32x4* in = reinterpret_cast<const u32x4*>(a.in) + (size_t)b * a.Lin * cq;
  const u32x4* wb = reinterpret_cast<const u32x4*>(a.w);
  const int nk = a.Cin / BK;

  u32x4 wreg[NWQ], xreg[NXQ];
  auto load_regs = [&](int kt) {
    const int q0 = kt * 8;
#pragma unroll
    for (int i = 0; i < NWQ; ++i) {
      const int qd = tid + NTH * i;
      wreg[i] = wb[(size_t)(n0 + (qd >> 3)) * cq + q0 + (qd & 7)];
    }
#pragma unroll
    for (int i = 0; i < NXQ; ++i) {
      const int qd = tid + NTH * i, jr = j0 + (qd >> 3);
      xreg[i] = jr < a.Lin ? in[(size_t)jr * cq + q0 + (qd & 7)] : u32x4{0u, 0u, 0u, 0u};
    }
  };
  auto store_lds = [&]() {
#pragma unroll
    for (int i = 0; i < NWQ; ++i) { const int qd = tid + NTH * i; Ws(qd & 7, qd >> 3) = wreg[i]; }
#pragma unroll
    for (int i = 0; i < NXQ; ++i) { const int qd = tid + NTH * i; Xs(qd & 7, qd >> 3) = xreg[i]; }
  };

  f32x4 acc[6][4];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int g = lane >> 4, r16 = lane & 15;
  load_regs(0);
  for (int kt = 0; kt < nk; ++kt) {
    if constexpr (VA_LIN_ABL & 4) { if (a.Cin == 12345) store_lds(); } else store_lds();
    __syncthreads();
    if (kt + 1 < nk && !(VA_LIN_ABL & 1)) load_regs(kt + 1);
    f16x8 xh[4], xl[4];
    if constexpr (!(VA_LIN_ABL & 2)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      xh[j] = __builtin_bit_cast(f16x8, Xs(2 * g, wm * 64 + j * 16 + r16));
      xl[j] = __builtin_bit_cast(f16x8, Xs(2 * g + 1, wm * 64 + j * 16 + r16));
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const f16x8 wh = __builtin_bit_cast(f16x8, Ws(2 * g, wn * 96 + i * 16 + r16));
      const f16x8 wl = __builtin_bit_cast(f16x8, Ws(2 * g + 1, wn * 96 + i * 16 + r16));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[j], acc[i][j], 0, 0, 0);
      }
    }
    }
    __syncthreads();
  }

  // ---- epilogue, one 16-row slab of every wave per pass (16 MW rows x 192 columns staged in LDS): every thread then owns whole
  // octets of a row (32 contiguous bytes per stream), like conv_tile_store
  constexpr int SP = LBN + 4, OCT = LBN / 8;
  static_assert(8 * (LBN + LBM_) * 16 >= 16 * MW * SP * 4, "the staged slab must fit in the main loop's LDS");
  float* stage = reinterpret_cast<float*>(smem);
  const size_t obase = (size_t)b * a.Lout;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int col = wn * 96 + i * 16 + 4 * g;
      *reinterpret_cast<f32x4*>(stage + (wm * 16 + r16) * SP + col) = acc[i][j] + *reinterpret_cast<const f32x4*>(a.bias + n0 + col);
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 16 * MW * OCT / NTH; ++it) {
      const int u = tid + NTH * it, lr = u / OCT, oc = u - lr * OCT;
      const int jr = j0 + (lr >> 4) * 64 + j * 16 + (lr & 15);
      if (jr < a.Lin && (!(VA_LIN_ABL & 8) || stage[lr * SP + oc * 8] == 12345.678f)) {
        const int co = n0 + oc * 8;
        const size_t orow = obase + (size_t)(jr + a.oshift0);
        const size_t o = orow * a.Cout + co;
        f32x4 v0 = *reinterpret_cast<const f32x4*>(stage + lr * SP + oc * 8);
        f32x4 v1 = *reinterpret_cast<const f32x4*>(stage + lr * SP + oc * 8 + 4);
        if (a.res) {
          v0 = *reinterpret_cast<const f32x4*>(a.res + o) + v0;
          v1 = *reinterpret_cast<const f32x4*>(a.res + o + 4) + v1;
        }
        if (a.out_raw) {
          *reinterpret_cast<f32x4*>(a.out_raw + o) = v0;
          *reinterpret_cast<f32x4*>(a.out_raw + o + 4) = v1;
        }
        if (a.out_act) {
          float sv[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            sv[r] = a.act == 1 ? gelu_erf_f(v0[r]) : v0[r];
            sv[r + 4] = a.act == 1 ? gelu_erf_f(v1[r]) : v1[r];
          }
          store_act_octet(a.out_act, nullptr, 0, orow, co, a.Cout, sv);
        }
      }
    }
    __syncthreads();
  }
}

// The same tile, wave layout, products and summation order (bit-identical outputs) with the operands brought in by LDS-DMA: no
// staging registers, no ds_write, NST stages of 40 KB in flight (2: two workgroups share a CU, the next chunk flies during this one's
// matrix instructions; 3: one workgroup, two chunks ahead), ONE barrier per chunk.
// A DMA piece (one global_load_lds_dwordx4: lane l's 16 bytes land at slot l of 1 KB) is 8 ROWS x the whole 128-byte chunk of each
// (8 lanes = one cache line; the first cut fetched a fragment per piece — 16 rows x every other quad, 64 requests per instruction —
// and was bound by the address path: 52 ms against 45).  What makes the fragment reads conflict-free on such an image: a block of 32
// rows (two 16-row tiles) is four pieces; piece p holds rows p, p + 4, p + 8, p + 12 of each tile at k = 4 tile + 2 ((r >> 2) & 1) +
// ((r >> 3) & 1), and stores a row's 8 quads ROTATED by s(p) = {0, 1, 4, 5}: quad q at position (q + s(p)) mod 8.  A ds_read_b128
// is served in groups of 16 lanes holding 16 different rows with two neighbouring octets g (MI355X_MICROARCH.md, LDS): their slots
// mod 16 are 8 (r >> 3 & 1) + (2 g + plane + s(r & 3)) mod 8 — sixteen different values.  The epilogue writes from the accumulators: a lane holds 4 consecutive output columns of one row (16 bytes
// of the fp32 tensor); for the pair-layout copy the lanes g, g ^ 1 of a row swap halves (v_permlane16_swap) and write the octet's hi
// and lo quad — no LDS staging, no barriers.  Measured by ablation on round 4's kernel (tools/experiment.sh linear-abl, 35 ms of
// linears): global loads through registers 7 ms, the staging stores 8 ms, the staged epilogue 9 ms, matrix instructions 12 ms —
// nothing overlapped well.
// EARLY (two stages, the product): a chunk's fragments are all read into registers at the head of its step and a second barrier
// frees its stage at once for chunk kt + 2 — a DMA piece then has two steps of matrix instructions to land instead of one, in the
// same 80 KB.  Whole extractor forwards, 8 clips (round 4's kernel 44.2 ms): late reads 43.2, EARLY 42.0.
// MW = wave rows, NWN = waves along the 192 columns; only <2, 2> (128 x 192, four waves of 64 x 96, two workgroups per CU) is
// instantiated.  Measured and dropped (DESIGN_HISTORY.md, round 5): <NST 3, 2, 2> one workgroup of four waves 50.3 ms; <2, MW 4, 2>
// 256 x 192 on eight waves (43 % less traffic from beyond L2) 43.6; <3 | 4, 2, NWN 4> eight waves of 64 x 48 with three / four
// stages 45.2 / 45.4; the accumulators stored without the LDS turn-around 41.7 (no better: what the epilogue costs is its bytes).
template <int NST, int MW = 2, int NWN = 2, bool EARLY = false>
__global__ __launch_bounds__(64 * MW * NWN, (NST == 2 && MW * NWN == 4) ? 2 : 1) void linear_dma_kernel(ConvPArgs a, int mtiles, int ntiles, int panel) {
  static_assert(!EARLY || NST == 2, "EARLY is the two-stage schedule");
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  constexpr int LBM_ = 64 * MW, NWV = NWN * MW, NI = 12 / NWN, WC = NI * 16, XF = (LBM_ / 16) * 2, WF = (LBN / 16) * 2, PPW = (XF + WF) / NWV, STB = (XF + WF) * 1024;
  static_assert((XF + WF) % NWV == 0, "the same number of pieces per wave");
  extern __shared__ __attribute__((aligned(16))) unsigned char lind_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wv % NWN, wm = wv / NWN;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  // Tile order inside an XCD: column tiles in PANELS of `panel` (a divisor of ntiles), all of the XCD's row tiles per panel, the
  // panel's column tiles fastest.  With panel = ntiles (round 4's order) every group of workgroups in flight streams the WHOLE weight
  // matrix through a 4 MB L2 (7 - 9 MB for 768 -> 2304 / 3072): measured 1.45 / 2.0 GB fetched from beyond L2 per launch for 0.16 GB
  // of operands.  A panel of 4 column tiles (2.4 MB at K = 768) stays resident; the activations are then fetched once per panel.
  const int rows_x = (mtiles + 7) >> 3, per_panel = rows_x * panel;
  const int pn = local / per_panel, lp = local - pn * per_panel;
  const int nt = pn * panel + lp % panel, mt = (lp / panel) * 8 + xcd;
  if (mt >= mtiles) return;
  const int b = blockIdx.y;
  const int j0 = mt * LBM_, n0 = nt * LBN;
  const int cq = a.Cin / 4;                 // 16-B quads per row (C/8 octets x 2 planes)
  const int nk = a.Cin / BK;
  const int g = lane >> 4, r16 = lane & 15;

  // this wave's PPW pieces of a stage: piece P < XF = activation rows (32-row block P / 4, piece P % 4 of it), else weight rows.
  // Rows past the end re-read the last one (their outputs are not stored).
  const unsigned char* src[PPW];
  {
    const int k = lane >> 3, kk = k & 3;
    const int rr = (k >> 2) * 16 + 4 * (kk >> 1) + 8 * (kk & 1);        // row of the block, before + p
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int P = wv * PPW + j, pp = P & 3;
      const int q = ((lane & 7) - (pp + (pp >> 1) * 2)) & 7;            // s(p) = {0, 1, 4, 5}
      if (P < XF) {
        const int row = min(j0 + (P >> 2) * 32 + rr + pp, a.Lin - 1);
        src[j] = reinterpret_cast<const unsigned char*>(a.in) + (((size_t)b * a.Lin + row) * cq + q) * 16;
      } else {
        src[j] = reinterpret_cast<const unsigned char*>(a.w) + ((size_t)(n0 + ((P - XF) >> 2) * 32 + rr + pp) * cq + q) * 16;
      }
    }
  }
  auto issue = [&](int stage) {
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int p = wv * PPW + j;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src[j],
                                       (__attribute__((address_space(3))) void*)(lind_lds + stage * STB + p * 1024), 16, 0, 0);
      src[j] += 128;                        // the next 32-channel chunk of the same rows
    }
  };
  // fragment (16-row tile m of the stage's rows, plane pl) as lane (g, r16) reads it
  int foff[2];
  {
    const int pp = r16 & 3, sp = pp + (pp >> 1) * 2, kq = (((r16 >> 3) & 1) + 2 * ((r16 >> 2) & 1)) * 8;
    foff[0] = pp * 1024 + (kq + ((2 * g + sp) & 7)) * 16;
    foff[1] = pp * 1024 + (kq + ((2 * g + 1 + sp) & 7)) * 16;
  }
  auto xfrag = [&](int stage, int m, int pl) -> f16x8 {
    return __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(lind_lds + stage * STB + (m >> 1) * 4096 + (m & 1) * 512 + foff[pl]));
  };
  auto wfrag = [&](int stage, int n, int pl) -> f16x8 {
    return __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(lind_lds + stage * STB + XF * 1024 + (n >> 1) * 4096 + (n & 1) * 512 + foff[pl]));
  };

  f32x4 acc[NI][4];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if constexpr (EARLY) {
    issue(0);
    if (nk > 1) issue(1);
    for (int kt = 0; kt < nk; ++kt) {
      const int st = kt & 1;
      if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");      // chunk kt + 1 may still fly
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      f16x8 xh[4], xl[4], wh[NI], wl[NI];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xh[j] = xfrag(st, wm * 4 + j, 0);
        xl[j] = xfrag(st, wm * 4 + j, 1);
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        wh[i] = wfrag(st, wn * NI + i, 0);
        wl[i] = wfrag(st, wn * NI + i, 1);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();          // every wave holds its fragments: the stage is free
      asm volatile("" ::: "memory");
      if (kt + 2 < nk && !(VA_LIN_ABL & 1)) issue(st);
      if constexpr (!(VA_LIN_ABL & 2)) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[i], xh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], xl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], xh[j], acc[i][j], 0, 0, 0);
          }
      }
    }
  } else {
#pragma unroll
  for (int st = 0; st < NST - 1; ++st)
    if (st < nk) issue(st);
  int st0 = 0;                              // stage of chunk kt
  for (int kt = 0; kt < nk; ++kt) {
    // this wave's pieces of chunk kt have landed (chunks issued after it may still fly: none with two stages); the barrier publishes
    // every wave's pieces and says that every wave is done reading chunk kt - 1, whose stage takes chunk kt + NST - 1
    {
      const int ahead = min(NST - 2, nk - 1 - kt);      // chunks issued after chunk kt
      if (NST >= 4 && ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
      else if (NST >= 3 && ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + NST - 1 < nk && !(VA_LIN_ABL & 1)) issue(st0 == 0 ? NST - 1 : st0 - 1);
    if constexpr (!(VA_LIN_ABL & 2)) {
      f16x8 xh[4], xl[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xh[j] = xfrag(st0, wm * 4 + j, 0);
        xl[j] = xfrag(st0, wm * 4 + j, 1);
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const f16x8 wh = wfrag(st0, wn * NI + i, 0);
        const f16x8 wl = wfrag(st0, wn * NI + i, 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[j], acc[i][j], 0, 0, 0);
        }
      }
    }
    st0 = st0 == NST - 1 ? 0 : st0 + 1;
  }
  }

  // ---- epilogue: acc[i][j][r] = out[row j0 + wm 64 + 16 j + r16][column n0 + wn WC + 16 i + 4 g + r].  Stored straight from the
  // accumulators a wave instruction covers 16 rows x 64 bytes with NEIGHBOURING LANES IN DIFFERENT ROWS — 64 separate 16-byte
  // requests; that cut (no LDS, no barriers) cost as much as round 4's staged one (10 of 34 ms: tools/experiment.sh linear-abl).
  // Here every wave turns its own 32 x WC slab around through a private piece of the (now idle) stages — no workgroup barrier —
  // and reads it back with consecutive lanes on consecutive 16 bytes of a row: residual loads and both stores in whole lines.
  constexpr int SPW = WC + 4, QPR = WC / 4;      // padded slab row (floats; stride = 4 mod 8 banks: conflict-free b128 stores), quads per row
  static_assert(NWV * 32 * SPW * 4 <= NST * STB && (32 * QPR) % 64 == 0, "a 32-row slab per wave inside the stages");
  float* slab = reinterpret_cast<float*>(lind_lds) + wv * (32 * SPW);
  if constexpr (!EARLY) __syncthreads();         // the last chunk's fragment reads of the slower waves (EARLY: behind its second barrier)
  const size_t obase = (size_t)b * a.Lout;
  f32x4 bias4[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) bias4[i] = *reinterpret_cast<const f32x4*>(a.bias + n0 + wn * WC + i * 16 + 4 * g);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int i = 0; i < NI; ++i)
        *reinterpret_cast<f32x4*>(slab + (jj * 16 + r16) * SPW + i * 16 + 4 * g) = acc[i][2 * half + jj] + bias4[i];
    // the residual quads of the whole half, requested before the first store: `res` may be `out_raw` (x += ...), so the compiler must
    // keep every load behind the stores in front of it — one memory round trip per iteration if they are written where they are used
    constexpr int NIT = 32 * QPR / 64;
    f32x4 rq[NIT];
    if (a.res && !(VA_LIN_ABL & 8)) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int Q = it * 64 + lane, row = Q / QPR, cq = Q - row * QPR;
        const int jr = min(j0 + wm * 64 + half * 32 + row, a.Lin - 1);
        rq[it] = *reinterpret_cast<const f32x4*>(a.res + (obase + (size_t)(jr + a.oshift0)) * a.Cout + n0 + wn * WC + cq * 4);
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int Q = it * 64 + lane, row = Q / QPR, cq = Q - row * QPR;
      f32x4 v = *reinterpret_cast<const f32x4*>(slab + row * SPW + cq * 4);
      const int jr = j0 + wm * 64 + half * 32 + row;
      const bool live = jr < a.Lin && (!(VA_LIN_ABL & 8) || v[0] == 12345.678f);
      const size_t orow = obase + (size_t)(min(jr, a.Lin - 1) + a.oshift0);
      const int co = n0 + wn * WC + cq * 4;
      if (a.res && !(VA_LIN_ABL & 8)) v = rq[it] + v;
      if (a.out_raw && live) *reinterpret_cast<f32x4*>(a.out_raw + orow * a.Cout + co) = v;
      if (a.out_act) {
        f16x4 hi, lo;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float x = a.act == 1 ? gelu_erf_f(v[r]) : v[r];
          hi[r] = (_Float16)x;
          lo[r] = (_Float16)(x - (float)hi[r]);
        }
        // lanes l (even: columns 0-3 of the octet) and l ^ 1 (columns 4-7) trade halves: the even one writes the hi quad, the odd one the lo
        const u32x2 h2 = __builtin_bit_cast(u32x2, hi), l2 = __builtin_bit_cast(u32x2, lo);
        const bool odd = lane & 1;
        const uint32_t s0 = odd ? h2[0] : l2[0], s1 = odd ? h2[1] : l2[1];
        const uint32_t t0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s0, VA_DPP_XOR1, 0xf, 0xf, true);
        const uint32_t t1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s1, VA_DPP_XOR1, 0xf, 0xf, true);
        if (live) {
          u32x4* dst = reinterpret_cast<u32x4*>(a.out_act) + (orow * (size_t)(a.Cout >> 3) + (size_t)(co >> 3)) * 2 + (odd ? 1 : 0);
          *dst = odd ? u32x4{t0, t1, l2[0], l2[1]} : u32x4{h2[0], h2[1], t0, t1};
        }
      }
    }
  }
}

VA_STAMP_SETTER(vaura_stamps_set_dac)

// Plain linear layer on the pair GEMM (row f2, vit.hip): out[b][row + oshift][:] = act( in[b][row][:] . W^T + bias (+ res) ).
// in: pair layout (B, Lin, Cin); w: pair layout (Cout, Cin); out rows live in sequences of Lout rows per b.
int va_launch_linear_pair(const uint16_t* in, const uint16_t* w, const float* bias, const float* res, float* out_raw,
                          uint16_t* out_act, int act, int B, int Lin, int Lout, int oshift, int Cin, int Cout, hipStream_t s) {
  if (!in || !w || !bias || (Cin % BK) || (Cout % BN) || B <= 0 || Lin <= 0) return VAURA_ERR_SHAPE;
  ConvPArgs p;
  p.in = in; p.w = w; p.bias = bias; p.res = res; p.alpha = nullptr; p.out_raw = out_raw; p.out_act = out_act;
  p.Lin = Lin; p.Lout = Lout; p.Cin = Cin; p.Cout = Cout; p.NT = 1; p.off_base = 0; p.off_step = 1; p.ostride = 1;
  p.oshift0 = oshift; p.jcount = Lin; p.act = act;
  p.act_fmt = 0; p.out_scale = nullptr; p.in_scale = nullptr; p.wscale = nullptr;
  p.w2 = nullptr; p.bias2 = nullptr; p.alpha_mid = nullptr;
  if (Cout % LBN == 0 && act != 0 && !(va_debug_flags_get() & 64)) {     // debug flag bit 6: the 128 x 96 conv tile instead
    const int ntiles = Cout / LBN;
    // debug flag bit 8: 256 x 192 tiles (512 threads, one workgroup per CU) — 30 % fewer operand bytes per flop, and measured
    // SLOWER (48.8 vs 46.2 ms per forward): eight waves behind one barrier lose more than the L2 traffic saved
    if (Lin >= 256 * 64 && (va_debug_flags_get() & 256)) {
      const int mtiles = (Lin + 255) / 256;
      VA_LAUNCH(linear_pair_kernel<4>, dim3((unsigned)(((mtiles + 7) / 8) * 8 * ntiles), B), dim3(512), 0, s, p, mtiles, ntiles);
    } else {
      const int mtiles = (Lin + 127) / 128;
#ifndef VA_LIN_F2
#define VA_LIN_F2 0u       // experiment builds: instance bits forced on (tools/experiment.sh linear-abl drives a C++ program without flag control)
#endif
      const unsigned f2 = va_debug_flags2_get() | VA_LIN_F2;
      // column-tile panel (see linear_dma_kernel): 4 tiles while a tile of the weights is <= 1 MB (K <= 1280), else the whole width;
      // second flag word bits 12..15 override (1 .. 15 tiles; must divide ntiles)
      int panel = (Cin <= 1280 && ntiles % 4 == 0) ? 4 : ntiles;
      if (((f2 >> 12) & 15u) && ntiles % (int)((f2 >> 12) & 15u) == 0) panel = (int)((f2 >> 12) & 15u);
      if (((f2 >> 12) & 15u) == 15u) panel = ntiles;
      const dim3 grid((unsigned)(((mtiles + 7) / 8) * 8 * ntiles), B);
      if (f2 & 32u) {                      // second flag word, bit 5: round 4's register-staged kernel
        VA_LAUNCH(linear_pair_kernel<2>, grid, dim3(256), 0, s, p, mtiles, ntiles);
      } else if (f2 & 64u) {               // bit 6: LDS-DMA with the fragments read next to their matrix instructions (one barrier per chunk)
        static unsigned long long big2 = 0;
        if (va_big_lds_once(reinterpret_cast<const void*>(linear_dma_kernel<2, 2, 2, false>), 2 * 40 * 1024, &big2)) return VAURA_ERR_STATE;
        VA_LAUNCH((linear_dma_kernel<2, 2, 2, false>), grid, dim3(256), 2 * 40 * 1024, s, p, mtiles, ntiles, panel);
      } else {
        static unsigned long long bige = 0;
        if (va_big_lds_once(reinterpret_cast<const void*>(linear_dma_kernel<2, 2, 2, true>), 2 * 40 * 1024, &bige)) return VAURA_ERR_STATE;
        VA_LAUNCH((linear_dma_kernel<2, 2, 2, true>), grid, dim3(256), 2 * 40 * 1024, s, p, mtiles, ntiles, panel);
      }
    }
    return 0;
  }
  VA_LAUNCH((conv_pair_kernel<3, false>), dim3((Lin + BM - 1) / BM, Cout / BN, B), dim3(256), 0, s, p);
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Encode side (DacModelWrapper.encode, models/modules/dac/model.py:30-39).
// First conv of the encoder: 1 -> C channels, k = 7, pad 3, from the raw waveform; emits the raw fp32 rows (residual
// stream) and Snake(alpha) of them in pair layout (the input of the first residual unit).
__global__ __launch_bounds__(256) void enc_conv_in_kernel(const float* __restrict__ wav, const float* __restrict__ w /* [7][C] */,
                                                          const float* __restrict__ bias, const float* __restrict__ alpha,
                                                          float* __restrict__ out_raw, uint16_t* __restrict__ out_act, int64_t L,
                                                          int C) {
  const int b = blockIdx.y;
  const int qpr = C / 4;                                   // quads per row
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t l = gid / qpr;
  const int cq = (int)(gid % qpr);
  if (l >= L) return;
  const float* x = wav + (size_t)b * L;
  f32x4 v = *reinterpret_cast<const f32x4*>(bias + cq * 4);
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    const int64_t r = l + t - 3;
    const float xv = (r >= 0 && r < L) ? x[r] : 0.f;
    v += *reinterpret_cast<const f32x4*>(w + t * C + cq * 4) * xv;
  }
  const size_t row = (size_t)b * L + (size_t)l;
  *reinterpret_cast<f32x4*>(out_raw + row * C + cq * 4) = v;
  const f32x4 al = *reinterpret_cast<const f32x4*>(alpha + cq * 4);
  f32x4 sn;
#pragma unroll
  for (int r = 0; r < 4; ++r) sn[r] = snake_f(v[r], al[r]);
  store_pair4(out_act, row, cq * 4, C, sn);
}

// One residual-VQ stage for every latent row (dac/nn/quantize.py VectorQuantize.forward + the residual update of
// ResidualVectorQuantize.forward):  z_e = in_proj(residual); code = argmax_j -(|e|^2 - 2 e.c_j + |c_j|^2) over the
// L2-normalised encodings / codewords (first index wins a tie, like torch.max); residual -= out_proj(z_e + (c - z_e)).
// One 256-thread workgroup per row; latent <= 2048, codebook_dim <= 8, codebook_size <= 1024.
__global__ __launch_bounds__(256) void rvq_stage_kernel(float* __restrict__ residual /* (rows, latent) */,
                                                        const float* __restrict__ in_w /* (dim, latent) */,
                                                        const float* __restrict__ in_b, const float* __restrict__ cb /* (size, dim) */,
                                                        const float* __restrict__ out_w /* (latent, dim) */,
                                                        const float* __restrict__ out_b, int32_t* __restrict__ codes, int latent,
                                                        int dim, int size, int T, int K, int k) {
  __shared__ float part[4][8];
  __shared__ float ze[8];
  __shared__ float bestv[4];
  __shared__ int besti[4];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float* res = residual + (size_t)row * latent;
  // ---- z_e = in_proj(residual)
  float acc[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) acc[d] = 0.f;
  for (int c = tid; c < latent; c += 256) {
    const float x = res[c];
#pragma unroll
    for (int d = 0; d < 8; ++d)
      if (d < dim) acc[d] = fmaf(in_w[(size_t)d * latent + c], x, acc[d]);
  }
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    const float s = wave_sum(acc[d]);
    if (lane == 0) part[wv][d] = s;
  }
  __syncthreads();
  if (tid < 8) ze[tid] = tid < dim ? ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid])) + in_b[tid] : 0.f;
  __syncthreads();
  // ---- nearest codeword on the unit sphere
  float e[8], en2 = 0.f;
#pragma unroll
  for (int d = 0; d < 8; ++d) en2 = fmaf(ze[d], ze[d], en2);
  const float einv = 1.0f / fmaxf(sqrtf(en2), 1e-12f);     // F.normalize: x / max(|x|, eps)
  float e2 = 0.f;
#pragma unroll
  for (int d = 0; d < 8; ++d) { e[d] = ze[d] * einv; e2 = fmaf(e[d], e[d], e2); }
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int j = tid; j < size; j += 256) {
    float c[8], cn2 = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) { c[d] = d < dim ? cb[(size_t)j * dim + d] : 0.f; cn2 = fmaf(c[d], c[d], cn2); }
    const float cinv = 1.0f / fmaxf(sqrtf(cn2), 1e-12f);
    float dot = 0.f, c2 = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) { const float cn = c[d] * cinv; dot = fmaf(e[d], cn, dot); c2 = fmaf(cn, cn, c2); }
    const float score = -((e2 - 2.0f * dot) + c2);
    if (score > bv) { bv = score; bi = j; }                // ascending j per thread: first index kept on ties
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(bv, off, 64);
    const int oi = __shfl_xor(bi, off, 64);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if (lane == 0) { bestv[wv] = bv; besti[wv] = bi; }
  __syncthreads();
  float fv = bestv[0];
  int fi = besti[0];
#pragma unroll
  for (int w = 1; w < 4; ++w)
    if (bestv[w] > fv || (bestv[w] == fv && besti[w] < fi)) { fv = bestv[w]; fi = besti[w]; }
  // ---- residual -= out_proj(z_e + (codeword - z_e))
  float zq[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) zq[d] = d < dim ? ze[d] + (cb[(size_t)fi * dim + d] - ze[d]) : 0.f;
  for (int c = tid; c < latent; c += 256) {
    float o = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d)
      if (d < dim) o = fmaf(out_w[(size_t)c * dim + d], zq[d], o);
    res[c] -= o + out_b[c];
  }
  if (tid == 0) codes[((size_t)(row / T) * K + k) * T + (row % T)] = fi;
}

// the codec's activation on its own (op-level entry vaura_snake): the same snake_f every conv epilogue applies
__global__ __launch_bounds__(256) void snake_kernel(const float* __restrict__ x, const float* __restrict__ alpha, float* __restrict__ y,
                                                    int64_t n, int C) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = snake_f(x[i], alpha[i % C]);
}

extern "C" {

long long vaura_debug_counter(int which) {
  if (which == 1) {       // residual units launched as one kernel (launch_conv_unit)
    const long long v = va_conv_units_fused;
    va_conv_units_fused = 0;
    return v;
  }
  if (which != 0) return -1;
  const long long v = va_conv256_launches;
  va_conv256_launches = 0;
  return v;
}

size_t vaura_dac_workspace_elems(const vaura_codec* c, int B, int T) {
  if (!c || B <= 0 || T <= 0) return 0;
  size_t L = (size_t)T, best = (size_t)T * (size_t)c->latent_dim;
  int ch = c->conv_in.cout;
  if ((size_t)T * ch > best) best = (size_t)T * ch;
  for (int b = 0; b < c->n_blocks; ++b) {
    L *= (size_t)c->rates[b];
    ch = c->up[b].cout;
    if (L * ch > best) best = L * ch;
  }
  return best * (size_t)B;
}

int vaura_dac_decode(const vaura_codec* c, const int32_t* codes, int B, int T, float* wav, vaura_stream_t s_) {
  if (!c || !codes || !wav || B <= 0 || T <= 0) return VAURA_ERR_ARG;
  if (c->n_blocks < 1 || c->n_blocks > 4 || c->n_units != 3 || c->n_codebooks > 16 || c->codebook_dim > 8)
    return VAURA_ERR_SHAPE;
  if (c->ws_elems < vaura_dac_workspace_elems(c, B, T)) return VAURA_ERR_ARG;
  for (int i = 0; i < 4; ++i) if (!c->ws[i]) return VAURA_ERR_ARG;
  hipStream_t s = as_stream(s_);
  const int pr = c->precision;
  if (pr < 0 || pr > 4) return VAURA_ERR_DTYPE;
  float* R = c->ws[0]; float* A = c->ws[1]; float* Y = c->ws[2]; float* Z = c->ws[3];

  VA_LAUNCH(from_codes_kernel, dim3((T + FC_NT - 1) / FC_NT, B), dim3(256), 0, s, codes, c->codebooks, c->out_proj_w, c->out_proj_b, Y,
                     c->n_codebooks, T, c->codebook_size, c->codebook_dim, c->latent_dim, pr ? 1 : 0);
  // conv_in: only the activated output is consumed (by the first transposed conv)
  int rc = launch_conv(c->conv_in, Y, nullptr, c->alpha_up[0], nullptr, A, B, T, pr, s);
  if (rc) return rc;
  int L = T;
  for (int b = 0; b < c->n_blocks; ++b) {
    // Snake (already applied by the producer) -> transposed conv; raw kept for the first residual
    rc = launch_conv(c->up[b], A, nullptr, c->alpha_res[b][0][0], Y, Z, B, L, pr, s);
    if (rc) return rc;
    L *= c->rates[b];
    { float* t = R; R = Y; Y = t; t = A; A = Z; Z = t; }
    for (int u = 0; u < 3; ++u) {
      const float* next_alpha = (u < 2) ? c->alpha_res[b][u + 1][0] : (b + 1 < c->n_blocks ? c->alpha_up[b + 1] : c->alpha_out);
      // the whole unit in one launch where one workgroup holds every channel (the activated result goes to Y: neighbours still read A's halo rows)
      rc = launch_conv_unit(c->res[b][u][0], c->res[b][u][1], A, R, c->alpha_res[b][u][1], next_alpha, (u < 2) ? R : nullptr, Y, B, L, pr, s);
      if (rc == 0) { float* t = A; A = Y; Y = t; continue; }
      if (rc != VA_UNIT_NOT_ELIGIBLE) return rc;          // a real failure of the one-launch unit (argument error or hipError_t): not a fallback
      // y = Snake2(conv7(Snake1(x)))  (Snake1 applied by the producer)
      rc = launch_conv(c->res[b][u][0], A, nullptr, c->alpha_res[b][u][1], nullptr, Y, B, L, pr, s);
      if (rc) return rc;
      // x = x + conv1(y); emit Snake_next(x)
      rc = launch_conv(c->res[b][u][1], Y, R, next_alpha, (u < 2) ? R : nullptr, A, B, L, pr, s);
      if (rc) return rc;
    }
  }
  const int C = c->conv_out.cin;
  if (c->conv_out.cout != 1 || c->conv_out.taps != 7 || (C % 4)) return VAURA_ERR_SHAPE;
  if (C <= 128 && C % 8 == 0 && !(va_debug_flags_get() & 8192)) {     // debug flag bit 13: the untiled kernel
    const size_t sm = sizeof(float) * ((size_t)(CO_TL + 6) * (C + 4) + CO_TL);
    static unsigned long long big = 0;
    if (va_big_lds_once(reinterpret_cast<const void*>(conv_out_tiled_kernel), 80 * 1024, &big)) return VAURA_ERR_STATE;
    VA_LAUNCH(conv_out_tiled_kernel, dim3((L + CO_TL - 1) / CO_TL, B), dim3(256), sm, s, A, c->conv_out.w, c->conv_out.bias, wav, L, C, pr);
  } else {
    VA_LAUNCH(conv_out_kernel, dim3((L + 31) / 32, B), dim3(256), 0, s, A, c->conv_out.w, c->conv_out.bias, wav, L, C, pr);
  }
  return 0;
}

int vaura_snake(const float* x, const float* alpha, float* y, int64_t rows, int C, vaura_stream_t s_) {
  if (!x || !alpha || !y || rows <= 0 || C <= 0) return VAURA_ERR_ARG;
  const int64_t n = rows * C;
  VA_LAUNCH(snake_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(s_), x, alpha, y, n, C);
  return 0;
}

int vaura_dac_conv(const vaura_conv* cv, int precision, const float* in, float* out, float* scratch, int B, int Lin,
                   vaura_stream_t s_) {
  if (!cv || !in || !out || !scratch || B <= 0 || Lin <= 0 || precision < 0 || precision > 4) return VAURA_ERR_ARG;
  if (cv->cin % 32) return VAURA_ERR_SHAPE;
  hipStream_t s = as_stream(s_);
  const float* x = in;
  if (precision) {
    const bool mx = precision == 3 && cv->wscale;
    const size_t rows = (size_t)B * Lin, n = rows * (cv->cin / 8);
    uint8_t* sc = reinterpret_cast<uint8_t*>(scratch) + mx8_scale_offset(rows * cv->cin);
    VA_LAUNCH(act_convert_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, scratch, sc, rows, cv->cin, mx ? 1 : 0);
    x = scratch;
  }
  return launch_conv(*cv, x, nullptr, nullptr, out, nullptr, B, Lin, precision, s);
}

size_t vaura_dac_encode_workspace_elems(const vaura_codec_encoder* c, int B, int64_t n_samples) {
  if (!c || B <= 0 || n_samples <= 0) return 0;
  size_t L = (size_t)n_samples, best = L * (size_t)c->enc_dim;
  int ch = c->enc_dim;
  for (int b = 0; b < c->n_blocks; ++b) {
    L /= (size_t)c->rates[b];
    ch *= 2;
    if (L * ch > best) best = L * ch;
  }
  if (L * (size_t)c->latent_dim > best) best = L * (size_t)c->latent_dim;
  return best * (size_t)B;
}

int vaura_dac_encode(const vaura_codec_encoder* c, const float* wav, int B, int64_t n_samples, int32_t* codes, vaura_stream_t s_) {
  if (!c || !wav || !codes || B <= 0 || n_samples <= 0) return VAURA_ERR_ARG;
  if (c->n_blocks < 1 || c->n_blocks > 4 || c->n_units != 3 || c->n_codebooks > 16 || c->codebook_dim > 8 ||
      c->codebook_size > 1024 || c->latent_dim > 2048 || (c->enc_dim % 32))
    return VAURA_ERR_SHAPE;
  int64_t hop = 1;
  for (int b = 0; b < c->n_blocks; ++b) hop *= c->rates[b];
  if (n_samples % hop) return VAURA_ERR_SHAPE;                 // DAC.preprocess pads; the caller hands in the padded clip
  if (c->ws_elems < vaura_dac_encode_workspace_elems(c, B, n_samples)) return VAURA_ERR_ARG;
  for (int i = 0; i < 4; ++i) if (!c->ws[i]) return VAURA_ERR_ARG;
  hipStream_t s = as_stream(s_);
  float* R = c->ws[0]; float* A = c->ws[1]; float* Y = c->ws[2]; float* Z = c->ws[3];

  int64_t L = n_samples;
  int C = c->enc_dim;
  {
    const int64_t total = L * (C / 4);
    VA_LAUNCH(enc_conv_in_kernel, dim3((unsigned)((total + 255) / 256), B), dim3(256), 0, s, wav, c->conv_in_w, c->conv_in_b,
              c->alpha_res[0][0][0], R, reinterpret_cast<uint16_t*>(A), L, C);
  }
  int rc = 0;
  for (int b = 0; b < c->n_blocks; ++b) {
    for (int u = 0; u < 3; ++u) {
      // y = Snake2(conv7(Snake1(x)))   (Snake1 applied by the producer)
      rc = launch_conv(c->res[b][u][0], A, nullptr, c->alpha_res[b][u][1], nullptr, Y, B, (int)L, 1, s);
      if (rc) return rc;
      // x = x + conv1(y); emit Snake_next(x)
      const float* next_alpha = (u < 2) ? c->alpha_res[b][u + 1][0] : c->alpha_down[b];
      rc = launch_conv(c->res[b][u][1], Y, R, next_alpha, (u < 2) ? R : nullptr, A, B, (int)L, 1, s);
      if (rc) return rc;
    }
    // strided conv (k = 2r, stride r, pad r/2) == 3-tap conv over rows of r*C channels (see vaura_codec_encoder.down)
    const int r = c->rates[b];
    const bool last = b + 1 == c->n_blocks;
    rc = launch_conv(c->down[b], A, nullptr, last ? c->alpha_out : c->alpha_res[b + 1][0][0], last ? nullptr : Y, Z, B, (int)(L / r), 1, s);
    if (rc) return rc;
    L /= r;
    C *= 2;
    { float* t = R; R = Y; Y = t; t = A; A = Z; Z = t; }
  }
  // Snake (applied by the producer) -> conv k3 -> latent z, fp32 rows (B*T, latent)
  rc = launch_conv(c->conv_out, A, nullptr, nullptr, Y, nullptr, B, (int)L, 1, s);
  if (rc) return rc;
  const int T = (int)L;
  for (int k = 0; k < c->n_codebooks; ++k) {
    VA_LAUNCH(rvq_stage_kernel, dim3((unsigned)(B * T)), dim3(256), 0, s, Y,
              c->in_proj_w + (size_t)k * c->codebook_dim * c->latent_dim, c->in_proj_b + (size_t)k * c->codebook_dim,
              c->codebooks + (size_t)k * c->codebook_size * c->codebook_dim,
              c->out_proj_w + (size_t)k * c->latent_dim * c->codebook_dim, c->out_proj_b + (size_t)k * c->latent_dim, codes,
              c->latent_dim, c->codebook_dim, c->codebook_size, T, c->n_codebooks, k);
  }
  return 0;
}

}  // extern "C"
