// Weight-streaming skinny GEMM on the fp16 matrix cores with (hi, lo) fp16 PAIR operands: fp32 semantics to 22 significand bits.
//
// v_mfma_f32_16x16x4_f32 (gemv_kernel.h) is exact but runs at 1/16 of the 16-bit MFMA rate: at 16 rows its pipe time per
// decode GEMV is as long as the HBM stream it should hide under.  Here the fp32 activation is split ONCE, by the kernel that
// produces it, into two fp16 planes
//        x ~ hi + lo      hi = fp16(x) (round to nearest), lo = fp16(x - hi)            |x - hi - lo| <= 2^-23 |x|
// and the consumer issues one v_mfma_f32_16x16x32_f16 per plane and 32-deep k-group (fp16 x fp16 products are exact in
// fp32), each plane in its own fp32 accumulator (lo products never align against hi ones), added once at the end.
// Weights are fp16 planes too, pre-split at pack time with a power-of-two scale per output row (so that neither plane sits
// in fp16's denormal range; the scale multiplies the fp32 sum in the epilogue, exactly):
//   one plane  ("h1")  a checkpoint whose weights fit 11 significand bits (bf16-representable ones do): w * x = w*hi + w*lo
//   two planes ("h2")  fp32 checkpoints: w ~ whi + wlo (22 bits, same 4 bytes per weight as fp32):
//                      w * x = whi*hi + (whi*lo + wlo*hi); the wlo*lo term (2^-22 relative) is below the operands' own rounding
//   fp8 e4m3 tile pairs widen to fp16 exactly in registers.
// Round 2 carried activations as THREE bf16 planes (exact 24-bit split, 6 bytes per element) and split fp32 weights into
// three bf16 planes in registers (9 MFMAs + ~44 VALU per k-group).  The in-kernel stamps of round 3
// (profiles/r03_stage_stamps_*.json) showed what that cost: every workgroup pulls the whole activation through its CU's
// memory pipeline (147 KB per GEMV at K = 1536: more than its weight slice), and the fp32-weight kernels spent 4.5-6 us per
// launch in the register split.  Pairs move 2/3 of the activation bytes, need no VALU on the weights, and keep the logits
// within ~1e-5 of the fp32 reference (tokens of every reference golden unchanged; tests/test_gpu_generate.py).
//
// "split rows" layout of a (rows x C) activation:  [row_block][plane hi, lo][C/8][16 rows][8] fp16
//   -> one wave load = the B operand (16 rows x 32 k) of one plane, contiguous 1 KiB.
// RMSNorm: the producer multiplies by the NEXT norm's gain before splitting and writes per-tile partial
// sums of squares; the consumer adds the partials in a fixed order and applies rsqrt(mean+eps) in its
// epilogue (W.(g*x*rinv) == rinv * W.(g*x)).
#pragma once
#include <type_traits>
#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#define VA_NPL 2   // activation planes
// WT = storage tag of the kernels below: 0 one fp16 weight plane, 1 fp8 tile pairs, 2 (hi, lo) fp16 weight planes — all against BOTH
// activation planes — and 3 (round 6, BASELINE configs[4]): fp8 tile pairs against the HI activation plane ONLY.  With 4-bit-significand
// weights the lo plane (bits 12..22 of an activation) is far below the weights' own rounding, but it was half of what every CU takes
// in: at 32 rows a phase-2 workgroup of the one-launch MLP read 524 KB of planes next to 65 KB of weights.  The planes keep their
// layout (producers still write both); a WT = 3 consumer skips the lo plane's loads and products.
template <int WT>
struct WTag {
  static constexpr bool FP8 = WT == 1 || WT == 3, F32 = WT == 2;
  static constexpr int XPL = WT == 3 ? 1 : VA_NPL;        // activation planes a consumer loads and multiplies
};

enum { E3_STORE = 0, E3_RESID = 1, E3_SWIGLU = 2, E3_LOGITS = 4 };

struct Gemv3Args {
  const void* W;          // one fp16 plane of MFMA tiles (wq = 0), fp8 tile pairs (wq = 1) or (hi, lo) fp16 planes (wq = 2); + row scales
  int wq;                 // 0: one fp16 plane, 1: fp8 e4m3, 2: two fp16 planes — all with a power-of-two scale per output row
  const float* wscale;    // (weight rows) power-of-two scales (set by the launcher: they follow the tiles)
  const uint16_t* XP;     // split rows (rows x K)
  const float* ss_in;     // NORM: (R, n_ss_in, 16) partial sums of squares of the raw input rows
  int n_ss_in;
  const float* res;       // E3_RESID: fp32 packed rows (rows x N)
  float* out;             // fp32 packed rows (rows x N) / row-major logits; may be null for E3_SWIGLU
  float* out2;            // K-split instances (KS = 2): the partial sum over the second half of K goes here, `out`
                          // holds the first half's; the consumer adds the two (E3_STORE only)
  uint16_t* outp;         // optional split rows of (out * gain_out)
  const float* gain_out;  // optional (N) gain applied before splitting (next RMSNorm)
  float* ss_out;          // optional (R, N/16, 16) partial sums of squares of `out`
  int rows, R, N;
  float eps;
  int k_total;            // K of the normalised vector (mean denominator)
  float out_scale;        // E3_SWIGLU: the planes hold silu(w1 x) * (w3 x) * out_scale (a power of two, 1 by default: vaura_decoder.plane_shift)
};

__device__ __forceinline__ size_t split_index16(int rb, int plane, int octet, int row16, int C) {
  // index in 16-byte units
  return (((size_t)rb * VA_NPL + (size_t)plane) * (size_t)(C >> 3) + (size_t)octet) * 16 + (size_t)row16;
}

// NB: take scalars by value — __builtin_bit_cast applied directly to an ext-vector ELEMENT expression
// (v[i], u.y) reads element 0 for every index on hipcc 7.2.
__device__ __forceinline__ uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t pack_h2(_Float16 a, _Float16 b) {
  const f16x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, v);
}

// (hi, lo) fp16 split of 4 fp32 values -> two 8-byte fp16 quads.  |x| must stay below fp16's 65504 (activations of the
// decoder are O(1..100)); values below 2^-14 lose their lo plane to fp16's denormal spacing (absolute error 2^-25).
__device__ __forceinline__ void split2(const f32x4 v, uint2& hi, uint2& lo) {
  _Float16 h[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float x = v[i];
    h[i] = (_Float16)x;
    l[i] = (_Float16)(x - (float)h[i]);
  }
  hi = uint2{pack_h2(h[0], h[1]), pack_h2(h[2], h[3])};
  lo = uint2{pack_h2(l[0], l[1]), pack_h2(l[2], l[3])};
}

// store 4 consecutive columns (c0 % 4 == 0) of one row as the two planes
__device__ __forceinline__ void store_split4(uint16_t* base, int row, int c0, int C, const f32x4 v) {
  uint2 hi, lo;
  split2(v, hi, lo);
  const int rb = row >> 4, m = row & 15, oct = c0 >> 3, half = (c0 >> 2) & 1;
  uint2* p = reinterpret_cast<uint2*>(base);
  va_st8(p + split_index16(rb, 0, oct, m, C) * 2 + half, hi);
  va_st8(p + split_index16(rb, 1, oct, m, C) * 2 + half, lo);
}

// fp8 weights ("fp8 tile pairs"): [N/16][K/64][64 lanes][16 bytes]; bytes 0..7 of a lane are its 8 e4m3
// values of the even 32-deep k-group, bytes 8..15 those of the odd one (same lane->(n, k) map as the fp16
// tiles), followed by float scale[N].  e4m3 -> fp16 is exact (3 significand bits, exponents inside fp16's range), the row
// scale is a power of two applied to the fp32 sum, so the kernel computes exactly what the one-plane path computes on the
// dequantised matrix  W_eff = fp8 * scale.
__device__ __forceinline__ f16x8 fp8x8_to_f16(uint32_t a, uint32_t b) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 p0 = __builtin_amdgcn_cvt_pk_f32_fp8((int)a, false), p1 = __builtin_amdgcn_cvt_pk_f32_fp8((int)a, true);
  const f32x2 p2 = __builtin_amdgcn_cvt_pk_f32_fp8((int)b, false), p3 = __builtin_amdgcn_cvt_pk_f32_fp8((int)b, true);
  const float e0 = p0[0], e1 = p0[1], e2 = p1[0], e3 = p1[1], e4 = p2[0], e5 = p2[1], e6 = p3[0], e7 = p3[1];
  const f16x8 r = {(_Float16)e0, (_Float16)e1, (_Float16)e2, (_Float16)e3, (_Float16)e4, (_Float16)e5, (_Float16)e6, (_Float16)e7};
  return r;
}

// One 32-deep k-group of one 16-column tile.  One weight plane (WT 0 / 1): a product per activation plane, acc[p] for plane p
// of x.  Two weight planes (WT 2): acc[0] = whi.xhi, acc[1] = whi.xlo + wlo.xhi (both 2^-11 terms share an accumulator; the
// 2^-22 term wlo.xlo is dropped).  The accumulators are added smallest first, once, in the epilogue.
template <int WT>
__device__ __forceinline__ void mfma_group(const f16x8* wf /* 1 or 2 planes */, const u32x4* x2, f32x4* acc) {
  const f16x8 x0 = __builtin_bit_cast(f16x8, x2[0]);
  acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0], x0, acc[0], 0, 0, 0);
  if constexpr (WT != 3) {      // WT = 3: the hi plane only (x2 has ONE element)
    const f16x8 x1 = __builtin_bit_cast(f16x8, x2[1]);
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0], x1, acc[1], 0, 0, 0);
  }
  if constexpr (WT == 2) acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[1], x0, acc[1], 0, 0, 0);
}
template <int WT>
__device__ __forceinline__ f32x4 acc_sum(const f32x4* acc) { return acc[1] + acc[0]; }

__device__ __forceinline__ float silu3_f(float a) { return a / (1.0f + expf(-a)); }

// Epilogue shared by the decode GEMV and the prefill GEMM: lane (m = lane & 15, q = lane >> 4) holds columns
// 4q..4q+3 of row m of each of the T consecutive 16-column tiles starting at tile0 (already scaled by rinv).
// Operands of the residual epilogue that do not depend on this kernel's own sums: requested at kernel start by the wave
// that will need them (the residual rows were written by the PREVIOUS kernel: a cold read of ~1 us if it only starts
// after the reduction barrier).
struct EpiPre {
  f32x4 res, gain;
  bool have;
};
template <int EPI>
__device__ __forceinline__ EpiPre gemv3_epilogue_prefetch(const Gemv3Args& a, int rb, int tile, int lane) {
  EpiPre p;
  p.have = false;
  p.res = p.gain = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (EPI == E3_RESID) {
    const int q = lane >> 4;
    p.res = reinterpret_cast<const f32x4*>(a.res)[((size_t)rb * (a.N / 4) + (size_t)tile * 4) * 16 + lane];
    if (a.outp && a.gain_out) p.gain = *reinterpret_cast<const f32x4*>(a.gain_out + tile * 16 + 4 * q);
    p.have = true;
  }
  return p;
}

template <int T, int EPI>
__device__ __forceinline__ void gemv3_epilogue(const Gemv3Args& a, int rb, int tile0, int lane, const f32x4* v,
                                               const EpiPre* pre = nullptr, f32x4* keep = nullptr) {
  const int m = lane & 15, q = lane >> 4;
  const int row = rb * 16 + m;
  if constexpr (EPI == E3_SWIGLU) {
    static_assert(T % 2 == 0 || EPI != E3_SWIGLU, "SwiGLU needs (w1, w3) tile pairs");
#pragma unroll
    for (int pr = 0; pr < T / 2; ++pr) {
      f32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = silu3_f(v[2 * pr][r]) * v[2 * pr + 1][r];
      const int tile = tile0 / 2 + pr;   // tile of the ffn dimension
      if (a.out) va_st16(reinterpret_cast<f32x4*>(a.out) + ((size_t)rb * (a.N / 4) + (size_t)tile * 4) * 16 + lane, o);
      if (a.outp) store_split4(a.outp, row, tile * 16 + 4 * q, a.N, o * a.out_scale);
    }
  } else {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int tile = tile0 + t;
      const int c0 = tile * 16 + 4 * q;
      if constexpr (EPI == E3_LOGITS) {
        if (row < a.rows) va_st16(a.out + (size_t)row * a.N + c0, v[t]);
      } else {
        const size_t idx = ((size_t)rb * (a.N / 4) + (size_t)tile * 4) * 16 + lane;
        f32x4 o = v[t];
        if constexpr (EPI == E3_RESID) o += (pre && pre->have && T == 1) ? pre->res : reinterpret_cast<const f32x4*>(a.res)[idx];
        if (keep) keep[t] = o;      // the caller keeps the new residual rows in registers (layer-tail engine: wo's output is w2's residual)
        if (a.out) va_st16(reinterpret_cast<f32x4*>(a.out) + idx, o);
        if (a.ss_out) {
          float s = ((o[0] * o[0] + o[1] * o[1]) + o[2] * o[2]) + o[3] * o[3];
          s += va_xor16(s);
          s += va_xor32(s);
          if (q == 0) va_st4(a.ss_out + ((size_t)rb * (a.N / 16) + tile) * 16 + m, s);
        }
        if (a.outp) {
          f32x4 u = o;
          if (a.gain_out) u *= (pre && pre->have && T == 1) ? pre->gain : *reinterpret_cast<const f32x4*>(a.gain_out + c0);
          store_split4(a.outp, row, c0, a.N, u);
        }
      }
    }
  }
}

// ABL: ablation bits for tools/microbench only (0 in the product): 1 = no MFMA, 2 = no x loads, 4 = no weight
// loads, 8 = same k-slice order in every workgroup.
// XB = number of x batches (2: the second half of the k-groups is fetched after the first half has
// been consumed, for depths whose planes do not fit the register budget at once)
// WT = storage of the weights: 0 one fp16 plane of MFMA tiles, 1 fp8 tile pairs, 2 (hi, lo) fp16 planes (two 16-byte halves
// per lane and k-group: the A operands as loaded) — each followed by float scale[N].  With WT = 2 and XB > 1 the weights
// travel in the same batches as the activation planes (two batches in flight) instead of all up front: 8 bytes x 16 k-groups
// do not fit.
// KS = 2: two workgroups per tile group, each over one half of K, each writing its own partial output (out / out2)
// which the CONSUMER adds (attention gathers q, k, v anyway: one more 16-byte load).  For the qkv GEMV this turns
// 144 workgroups into 192 with half the activation bytes each: more CUs, fewer bytes through each CU's memory pipeline, no
// in-kernel seam.
// RBK = row blocks per WEIGHT PASS (round 5): with 17..32 decoder rows (the reference's default batch 16 under CFG, BASELINE configs[4])
// a workgroup multiplies each weight fragment against the planes of RBK = 2 row blocks — second accumulator set, both blocks' planes
// requested up front, ONE reduction barrier, the two blocks' epilogues on different waves — instead of walking the blocks one after
// the other (which re-fetched the weights where they travel in batches, WT = 2 / XB > 1, and paid a reduction + epilogue round per
// block everywhere: w1||w3 18.7 us against 10.0 at 16 rows).  Per row block the same products in the same order: bit-identical.
template <int G, int NW, int T, int EPI, bool NORM, int XB = 1, int ABL = 0, int WT = 0, int KS = 1, int RBK = 1>
__global__ __launch_bounds__(NW * 64) void gemv3_kernel(const void* __restrict__ Wq, const uint16_t* __restrict__ XPq, Gemv3Args a) {
  // Wq / XPq duplicate a.W / a.XP as explicit scalar arguments: with -amdgpu-kernarg-preload-count they arrive in SGPRs
  // at wave launch, so the address arithmetic of the first (weight) loads does not wait for a kernarg s_load
  VA_STAMP_DECL(stamps);
  VA_STAMP(stamps, 0);                       // wave start
  a.W = Wq;
  a.XP = XPq;
  constexpr bool FP8 = WTag<WT>::FP8, F32 = WTag<WT>::F32;
  constexpr int XPL = WTag<WT>::XPL;
  static_assert(!FP8 || (G % 2 == 0 && (G / XB) % 2 == 0), "fp8 tile pairs hold two k-groups per lane");
  static_assert(KS == 1 || EPI == E3_STORE, "K-split partials are summed by the consumer: plain stores only");
  static_assert(!FP8 || (G * NW) % 2 == 0, "fp8 tile pairs: a K part must start on an even k-group");
  constexpr int K = 32 * G * NW * KS;
  constexpr int KG = K / 32;
  constexpr int GB = G / XB;
  static_assert(G % XB == 0, "x batches must divide the groups");
  constexpr bool WBATCH = F32 && XB > 1;      // weights fetched batch by batch, with the planes
  constexpr int WH = F32 ? 2 : 1;             // 16-byte loads per lane, tile and k-group
  constexpr int GW = FP8 ? G / 2 : (WBATCH ? 2 * GB : G);   // weight register groups per tile
  constexpr int NACC = 2;
  __shared__ f32x4 red[RBK][NW][T][64];
  constexpr int NSS = K / 64;            // partial sums of squares per lane: n_ss_in = K / 16 tiles, 4 lane groups
  constexpr int EWN = (EPI == E3_SWIGLU) ? 1 : T;   // waves that run the epilogue of ONE row block (see below)
  // epilogue groups: the row blocks of a pass finish side by side on different waves where the workgroup has enough of them
  // (group e = waves [e EWN, (e + 1) EWN) takes row block rb0 + e), else one group walks them
  constexpr int EG = (EWN * RBK <= NW) ? RBK : 1;
  constexpr int EPW = RBK / EG;          // row blocks per epilogue wave
  // the epilogue waves fetch them straight into registers (24 VGPRs next to the weight slice); instances with more than 8
  // waves would park them in LDS instead (whole workgroup, one load each)
  constexpr bool SS_DIRECT = NORM && NW <= 8;
  constexpr int SSL = (NORM && !SS_DIRECT) ? K : 4;   // n_ss_in * 16 <= K floats
  __shared__ float ssl[SSL];

  const int lane = threadIdx.x & 63;
  // the wave index is uniform, but only readfirstlane proves it to the compiler: weight / plane bases then live in SGPRs
  // (global_load with an SGPR base + one VGPR lane offset) instead of one 64-bit VGPR address per 4 KB of stream
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // de-phase the k-slices across workgroups: every workgroup reads the SAME activation planes, and with
  // identical slice order all CUs of an XCD hit the same L2 channel at the same time
  const int w = (ABL & 8) ? wid : (int)((wid + blockIdx.x) % NW);
  const int m = lane & 15;
  const int q = lane >> 4;
  const int ks = KS > 1 ? (int)(blockIdx.x % KS) : 0;            // which part of K
  const int kgo = ks * G * NW;                                   // its first k-group
  const int tile0 = (int)(blockIdx.x / KS) * T;
  if (KS > 1 && ks > 0) a.out = a.out2;
  // Every stream load is a buffer load: descriptor + uniform byte offset in SGPRs, ONE 32-bit VGPR (lane * 16) as the only
  // per-lane address part (the split-rows index of lane (m, q) is uniform + 16 * q + m = uniform + lane).  With flat 64-bit
  // addresses hipcc kept a VGPR pair per 4 KB of stream alive (spills in the fp32-weight instances) and spent VALU on them.
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.W), 0, -16, 0x00020000);
  // the activation descriptor ends with the last row block: lanes of batch rows that do not exist (a decode step of fewer
  // than 16 rows: configs[3] has 4, a single clip 1 or 2) are sent out of range — they read zeros, as the padding rows of the
  // planes would give them, WITHOUT moving the bytes: every workgroup pulls rows/16 of the 147 KB instead of all of it
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.XP), 0, a.R * VA_NPL * (K / 8) * 256, 0x00020000);
  const int lane16 = lane * 16;

  // the weight slice of this wave lives in registers for the whole kernel: the decode step has one row
  // block; a prefill pass loops row blocks (one per prompt position) over the same registers
  u32x4 wb[T][GW][WH];
  // weights of k-groups [g0, g0 + n) of this wave's slice -> register groups [slot0, slot0 + n)
  auto load_w = [&](int g0, int n, int slot0) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (g < g0 || g >= g0 + n) continue;
      if (FP8 && (g & 1)) continue;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const size_t kg = FP8 ? (size_t)(tile0 + t) * (KG / 2) + (size_t)((kgo + w * G + g) >> 1)
                              : (size_t)(tile0 + t) * KG + (size_t)(kgo + w * G + g);
        const int slot = slot0 + (FP8 ? (g - g0) / 2 : g - g0);
#pragma unroll
        for (int hh = 0; hh < WH; ++hh)
          wb[t][slot][hh] = (ABL & 4) ? u32x4{(uint32_t)lane, 1u, 2u, 3u}
                                      : __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16, (int)((kg * WH + hh) * 1024), 2 /* nt */);
      }
    }
  };
  // activation planes of the current batch of k-groups; the first batch of the NEXT row block is requested as soon as
  // the last MFMA of this one has been issued, so its round trip runs under the reduction / barrier / epilogue
  constexpr int NXB = XB > 1 ? 2 : 1;   // with several batches two are in flight (double buffer)
  u32x4 xb[RBK][NXB][GB][XPL];
  auto load_x1 = [&](int r, int rb, int b) {
    const int xl16 = rb * 16 + m < a.rows ? lane16 : 0x7ffffff0;      // (a row block past the last one: every lane out of range -> zeros)
#pragma unroll
    for (int g = 0; g < GB; ++g)
#pragma unroll
      for (int p = 0; p < XPL; ++p)
        xb[r][b % NXB][g][p] = (ABL & 2) ? u32x4{(uint32_t)lane, 1u, 2u, 3u}
                                      : __builtin_amdgcn_raw_buffer_load_b128(
                                            xrs, xl16, (int)(((rb * VA_NPL + p) * (K / 8) * 16 + (kgo + w * G + b * GB + g) * 64) * 16), 0);
  };
  auto load_x = [&](int rb0, int b) {
#pragma unroll
    for (int r = 0; r < RBK; ++r) load_x1(r, rb0 + r, b);
  };

  // the power-of-two row scales of the epilogue waves' tiles: requested with the first row block's operands (round 4: they were a
  // dependent L2 round trip BEHIND the reduction barrier of every GEMV: -0.2 .. -0.9 us per launch)
  constexpr int ET_ = (EPI == E3_SWIGLU) ? T : 1;
  f32x4 wsc[ET_];
  const int eg = wid / EWN, ewi = wid - eg * EWN;   // this wave's epilogue group and its index inside it (waves past the groups: no epilogue)
  const bool epi_wave = wid < EWN * EG;
  auto row_block = [&](const int rb, const bool first) {     // rb = first row block of the pass
    if (first || WBATCH) {
      // all weight tiles first (HBM misses), then the planes (L2 hits)
      if constexpr (WBATCH) load_w(0, GB, 0);
      else load_w(0, G, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (first) load_x(rb, 0);
    }
    // rinv inputs: the producer's per-tile partial sums of squares.  The waves that will run the epilogue request their
    // lane's partials (row m, tiles q, q + 4, ...: every instruction a contiguous 256 bytes) right behind the stream loads,
    // add them in tile order after their last MFMA and have rinv in a register BEFORE the reduction barrier.  (Round 2 parked
    // the partials in LDS and added them after the barrier in a load -> wait -> add loop of 24 dependent LDS round trips:
    // 1.0-1.3 us on the critical path of every normed GEMV, profiles/r03_stage_stamps.json.  Same sums, same order.)
    static_assert(RBK == 1 || !NORM || (NORM && NW <= 8), "several row blocks per pass: the partial sums travel in registers");
    float ssv[EPW][SS_DIRECT ? NSS : 1];
    constexpr int SSN = (NORM && !SS_DIRECT) ? (K + NW * 64 - 1) / (NW * 64) : 1;
    float ssr[SSN];
    if constexpr (SS_DIRECT) {
      if (epi_wave) {
#pragma unroll
        for (int e = 0; e < EPW; ++e) {
          const int rbe = min(rb + eg * EPW + e, a.R - 1);       // (a pass's row block past the last: any valid address, never used)
          const float* sp = a.ss_in + (size_t)rbe * a.n_ss_in * 16 + m;
#pragma unroll
          for (int j = 0; j < NSS; ++j) {
            const int i = q + 4 * j;
            ssv[e][j] = (i < a.n_ss_in) ? sp[i * 16] : 0.f;
          }
        }
      }
    } else if constexpr (NORM) {
      const float* sp = a.ss_in + (size_t)rb * a.n_ss_in * 16;
#pragma unroll
      for (int j = 0; j < SSN; ++j) {
        const int i = threadIdx.x + j * NW * 64;
        ssr[j] = (i < a.n_ss_in * 16) ? sp[i] : 0.f;
      }
    }

    // residual / gain of the epilogue wave's tile: behind the stream loads in issue order, landed long before the barrier
    EpiPre pre;
    pre.have = false;
    if constexpr (EPI == E3_RESID && T == 1 && EPW == 1) {
      if (epi_wave && rb + eg < a.R) pre = gemv3_epilogue_prefetch<EPI>(a, rb + eg, tile0, lane);
    }
    if (first && epi_wave) {
#pragma unroll
      for (int e = 0; e < ET_; ++e)
        wsc[e] = *reinterpret_cast<const f32x4*>(a.wscale + (size_t)(tile0 + ((EPI == E3_SWIGLU) ? e : ewi)) * 16 + 4 * q);
    }

    f32x4 acc[RBK][T][NACC];
#pragma unroll
    for (int r = 0; r < RBK; ++r)
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int p = 0; p < NACC; ++p) acc[r][t][p] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (first) {
      VA_STAMP(stamps, 1);                   // every request of the first batch issued
      VA_WAIT_VM(GB * XPL + (SS_DIRECT ? NSS : (NORM ? SSN : 0)));
      VA_STAMP(stamps, 2);                   // the weight tiles (HBM) have landed
      VA_WAIT_VM(0);
      VA_STAMP(stamps, 3);                   // the activation planes / partial sums (written by the previous kernel) have landed
    }

#pragma unroll
    for (int b = 0; b < XB; ++b) {
      if (b + 1 < XB) {   // into the buffers batch b-1 has just released
        if constexpr (WBATCH) load_w((b + 1) * GB, GB, ((b + 1) & 1) * GB);
        load_x(rb, b + 1);
      }
#pragma unroll
      for (int g = 0; g < GB; ++g) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
          f16x8 wf[F32 ? 2 : 1];
          if constexpr (FP8) {
            const u32x4 pr = wb[t][(b * GB + g) / 2][0];
            wf[0] = (g & 1) ? fp8x8_to_f16(pr.z, pr.w) : fp8x8_to_f16(pr.x, pr.y);
          } else if constexpr (F32) {
            const int slot = WBATCH ? (b & 1) * GB + g : b * GB + g;
            wf[0] = __builtin_bit_cast(f16x8, wb[t][slot][0]);          // hi plane
            wf[1] = __builtin_bit_cast(f16x8, wb[t][slot][WH - 1]);     // lo plane
          } else {
            wf[0] = __builtin_bit_cast(f16x8, wb[t][b * GB + g][0]);
          }
          if constexpr (ABL & 1) {
            asm volatile("" ::"v"(wf[0]), "v"(xb[0][b % NXB][g][0]), "v"(xb[0][b % NXB][g][1]));
          } else {
#pragma unroll
            for (int r = 0; r < RBK; ++r) mfma_group<WT>(wf, xb[r][b % NXB][g], acc[r][t]);
          }
        }
        if (first) __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (rb + RBK < a.R) load_x(rb + RBK, 0);

#pragma unroll
    for (int r = 0; r < RBK; ++r)
#pragma unroll
      for (int t = 0; t < T; ++t) red[r][wid][t][lane] = acc_sum<WT>(acc[r][t]);
    float rinv[EPW];
#pragma unroll
    for (int e = 0; e < EPW; ++e) rinv[e] = 1.f;
    if constexpr (SS_DIRECT) {
      if (epi_wave) {
#pragma unroll
        for (int e = 0; e < EPW; ++e) {
          float ssp = 0.f;
#pragma unroll
          for (int j = 0; j < NSS; ++j) ssp += ssv[e][j];      // tile order q, q + 4, ... (slots past n_ss_in hold 0)
          ssp += va_xor16(ssp);
          ssp += va_xor32(ssp);
          rinv[e] = 1.0f / sqrtf(ssp * (1.0f / (float)a.k_total) + a.eps);
        }
      }
    } else if constexpr (NORM) {
#pragma unroll
      for (int j = 0; j < SSN; ++j) {
        const int i = threadIdx.x + j * NW * 64;
        if (i < SSL) ssl[i] = ssr[j];
      }
    }
    if (first) VA_STAMP(stamps, 4);          // MFMAs done, partial tiles written to LDS
    __syncthreads();
    if (first) VA_STAMP(stamps, 5);          // all 8 waves have arrived
    // Epilogue: one wave per output tile where the tiles are independent (store / residual / logits), so T waves
    // finish the T tiles side by side; SwiGLU needs both tiles of a (w1, w3) pair in the same lane: one wave does all.
    constexpr int EW = (EPI == E3_SWIGLU) ? 1 : T;     // waves taking part
    constexpr int ET = (EPI == E3_SWIGLU) ? T : 1;     // tiles per such wave
    static_assert(EW == EWN, "the waves that fetched the partial sums run the epilogue");
    if (epi_wave) {
      if constexpr (NORM && !SS_DIRECT) {
        // every LDS read issued before the first add (a fixed trip count: the round-2 loop over a.n_ss_in waited for each read)
        float pv[NSS];
#pragma unroll
        for (int j = 0; j < NSS; ++j) pv[j] = (q + 4 * j < a.n_ss_in) ? ssl[(q + 4 * j) * 16 + m] : 0.f;
        float ssp = 0.f;
#pragma unroll
        for (int j = 0; j < NSS; ++j) ssp += pv[j];
        ssp += va_xor16(ssp);
        ssp += va_xor32(ssp);
        rinv[0] = 1.0f / sqrtf(ssp * (1.0f / (float)a.k_total) + a.eps);
      }
#pragma unroll
      for (int er = 0; er < EPW; ++er) {
        const int r = eg * EPW + er;                      // row block of the pass this wave finishes
        if (rb + r >= a.R) break;
        f32x4 v[ET];
#pragma unroll
        for (int e = 0; e < ET; ++e) {
          const int t = (EPI == E3_SWIGLU) ? e : ewi;
          f32x4 sacc = red[r][0][t][lane];
#pragma unroll
          for (int i = 1; i < NW; ++i) sacc += red[r][i][t][lane];
          sacc *= wsc[e];                                   // power-of-two row scales: exact
          v[e] = sacc * rinv[er];
        }
        gemv3_epilogue<ET, EPI>(a, rb + r, (EPI == E3_SWIGLU) ? tile0 : tile0 + ewi, lane, v, &pre);
      }
    }
  };

  row_block(0, true);
#ifdef VAURA_STAMPS
  VA_WAIT_VM(0);
  VA_STAMP(stamps, 6);                       // wave 0's epilogue stores acknowledged
  VA_STAMP_FLUSH(stamps, EPI == E3_SWIGLU ? 4 : (EPI == E3_LOGITS ? 6 : ((EPI == E3_STORE && NORM) ? 1 : 9)));
#endif
  for (int rb = RBK; rb < a.R; rb += RBK) {
    __syncthreads();
    row_block(rb, false);
  }
}

// ---------------------------------------------------------------------------------------------------
// Row-split GEMV for the narrow outputs (wo, w2: N = 1536 -> only 96 column tiles).  With one workgroup per tile every
// workgroup streams the activation planes of ALL 16 rows (147 KB at K = 1536, 393 KB at K = 4096) next to 49 / 131 KB of
// weights, on 96 of 256 CUs: the planes, not the weights, set the time (DESIGN.md "per-CU bytes").  Here TWO workgroups share a
// tile, 8 rows each, so a workgroup reads half the planes — without half-empty load instructions: one MFMA covers a PAIR of
// 32-deep k-groups,
//      A row  i = (n8, s)  = weight row n8 (+ 8 nh) of the tile, k-group 2 gp + s          (16 rows = 8 n x 2 k-groups)
//      B col  j = (r8, s') = batch row r8 (+ 8 h) of the block, k-group 2 gp + s'          (16 cols = 8 rows x 2 k-groups)
//      D[i][j] = sum_k A[i][k] B[k][j]  is a wanted partial product where s == s' (half of the tile; the matrix pipe is idle anyway)
// and the two k-groups of a pair are added with one cross-lane exchange (lane ^ 40) before the usual LDS reduction over the 8
// waves.  Every load instruction still moves 64 x 16 B in 128-byte runs.  Weight bytes per workgroup are unchanged (the two
// workgroups of a tile read the same tiles: ids b and b + 8, i.e. the same XCD's L2 under round-robin placement — speed only).
// G2 = k-group pairs per wave, XB = batches (weights and planes together, two in flight), WT = 0 bf16 | 2 fp32 weights.
// RBK = row blocks per weight pass (see gemv3_kernel): the workgroup of row half h takes rows 8 h .. 8 h + 7 of RBK row blocks at once.
template <int G2, int NW, int EPI, int XB = 1, int WT = 0, int NBF = 2, int RBK = 1>
__global__ __launch_bounds__(NW * 64) void gemv3h_kernel(const void* __restrict__ Wq, const uint16_t* __restrict__ XPq, Gemv3Args a, int halves) {
  VA_STAMP_DECL(stamps);
  VA_STAMP(stamps, 0);
  a.W = Wq;
  a.XP = XPq;
  // WT = 1 (round 5): fp8 tile pairs.  One 1-KiB fragment holds BOTH k-groups of a pair (lane (n, k-octet): 8 bytes of the even
  // k-group, 8 of the odd one), so the A operand of lane (la, s, q) is an 8-byte load of half s of that fragment, widened to fp16 in
  // registers: the same products in the same order as the one-plane instance on the dequantised matrix.
  static_assert(WT >= 0 && WT <= 3, "one or two fp16 weight planes, or fp8 tile pairs (3: against the hi activation plane only)");
  static_assert(EPI == E3_RESID || EPI == E3_STORE, "independent output tiles only");
  constexpr bool F32 = WTag<WT>::F32, FP8 = WTag<WT>::FP8;
  constexpr int XPL = WTag<WT>::XPL;
  constexpr int WH = F32 ? 2 : 1;
  constexpr int NACC = 2;
  constexpr int K = 64 * G2 * NW;
  constexpr int KG = K / 32;
  constexpr int GB = G2 / XB;
  static_assert(G2 % XB == 0, "batches must divide the pairs");
  constexpr int NB = XB > 1 ? NBF : 1;        // batches in flight (buffers)
  static_assert(NBF >= 2 && NBF <= XB + 1, "two or three batches in flight");
  constexpr int BS = 1024 * WH;               // bytes of one (tile, k-group) block
  __shared__ f32x4 red[RBK][NW][2][64];
  static_assert(RBK <= NW, "one epilogue wave per row block of a pass");

  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // blocks b and b + 8 share a tile (same XCD under round-robin dispatch): h = bit 3 of the block id
  const int bid = blockIdx.x;
  const int h = halves == 2 ? (bid >> 3) & 1 : 0;
  const int tile = halves == 2 ? (bid & 7) + 8 * (bid >> 4) : bid;
  const int w = (wid + tile) % NW;            // de-phase the k-slices across workgroups
  const int la = lane & 7, sb = (lane >> 3) & 1, q = lane >> 4;
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.W), 0, -16, 0x00020000);
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.XP), 0, a.R * VA_NPL * (K / 8) * 256, 0x00020000);
  const int voffw0 = FP8 ? (la + 16 * q) * 16 + sb * 8      // fp8: half s of the pair's fragment
                         : (la + 16 * q) * 16 + sb * BS;    // weight rows 0..7 of the tile; + 128 bytes for rows 8..15
  const int voffx = (sb * 64 + q * 16 + la + 8 * h) * 16;   // batch row la + 8 h of the block (out of range below when it does not exist)

  u32x4 wb[NB][GB][2][WH];
  u32x4 xb[RBK][NB][GB][XPL];
  f32x4 wsc = f32x4{1.f, 1.f, 1.f, 1.f};
  auto load_w = [&](int b) {
#pragma unroll
    for (int g = 0; g < GB; ++g) {
      if constexpr (FP8) {
        const int soff = (tile * (KG / 2) + (w * G2 + b * GB + g)) * 1024;
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
          const auto v2 = __builtin_amdgcn_raw_buffer_load_b64(wrs, voffw0 + nh * 128, soff, 2 /* nt */);
          wb[b % NB][g][nh][0] = u32x4{v2[0], v2[1], 0u, 0u};
        }
      } else {
        const int soff = (tile * KG + 2 * (w * G2 + b * GB + g)) * BS;
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
          for (int hh = 0; hh < WH; ++hh)
            wb[b % NB][g][nh][hh] = __builtin_amdgcn_raw_buffer_load_b128(wrs, voffw0 + nh * 128, soff + hh * 1024, 2 /* nt */);
      }
    }
  };
  auto load_x = [&](int rb0, int b) {
#pragma unroll
    for (int r = 0; r < RBK; ++r) {
      const int rb = rb0 + r;
#pragma unroll
      for (int g = 0; g < GB; ++g)
#pragma unroll
        for (int p = 0; p < XPL; ++p)
          xb[r][b % NB][g][p] = __builtin_amdgcn_raw_buffer_load_b128(xrs, rb * 16 + la + 8 * h < a.rows ? voffx : 0x7ffffff0,
                                                                      ((rb * VA_NPL + p) * (K / 8) * 16 + (w * G2 + b * GB + g) * 128) * 16, 0);
    }
  };

  for (int rb = 0; rb < a.R; rb += RBK) {          // rb = first row block of the pass
    if (rb > 0) __syncthreads();
    if (rb == 0 || XB > 1) load_w(0);
    __builtin_amdgcn_sched_barrier(0);
    load_x(rb, 0);
    if constexpr (XB > 1 && NB > 2) {       // a third batch in flight from the start
#pragma unroll
      for (int b = 1; b < NB - 1; ++b) { load_w(b); load_x(rb, b); }
    }
    EpiPre pre;
    pre.have = false;
    // epilogue waves: wave r finishes row block rb + r of the pass
    if (wid < RBK && rb + wid < a.R && ((lane >> 3) & 1) == h) pre = gemv3_epilogue_prefetch<EPI>(a, rb + wid, tile, lane);
    if (wid < RBK && rb == 0) wsc = *reinterpret_cast<const f32x4*>(a.wscale + (size_t)tile * 16 + 4 * q);   // with the operands, not behind the barrier

    f32x4 acc[RBK][2][NACC];
#pragma unroll
    for (int r = 0; r < RBK; ++r)
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int p = 0; p < NACC; ++p) acc[r][nh][p] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (rb == 0) {
      VA_STAMP(stamps, 1);
      VA_WAIT_VM(GB * XPL + 2);           // wave 0 also holds the residual / gain requests
      VA_STAMP(stamps, 2);
      VA_WAIT_VM(0);
      VA_STAMP(stamps, 3);
    }
#pragma unroll
    for (int b = 0; b < XB; ++b) {
      if (XB > 1 && b + NB - 1 < XB) { load_w(b + NB - 1); load_x(rb, b + NB - 1); }
#pragma unroll
      for (int g = 0; g < GB; ++g) {
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
          f16x8 wf[F32 ? 2 : 1];
          if constexpr (FP8) wf[0] = fp8x8_to_f16(wb[b % NB][g][nh][0].x, wb[b % NB][g][nh][0].y);
          else wf[0] = __builtin_bit_cast(f16x8, wb[b % NB][g][nh][0]);
          if constexpr (F32) wf[1] = __builtin_bit_cast(f16x8, wb[b % NB][g][nh][WH - 1]);
#pragma unroll
          for (int r = 0; r < RBK; ++r) mfma_group<WT>(wf, xb[r][b % NB][g], acc[r][nh]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // the two k-groups of a pair sit in lanes (s' = 0, q < 2) and (s' = 1, q >= 2) of the same (n, row): lane ^ 40
#pragma unroll
    for (int r = 0; r < RBK; ++r)
#pragma unroll
      for (int nh = 0; nh < 2; ++nh) {
        f32x4 v = acc_sum<WT>(acc[r][nh]);
        f32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float x = v[c];
          o[c] = x + va_dpp<VA_DPP_ROR8>(va_xor32(x));      // lane ^ 40
        }
        red[r][wid][nh][lane] = o;
      }
    if (rb == 0) VA_STAMP(stamps, 4);
    __syncthreads();
    if (rb == 0) VA_STAMP(stamps, 5);
    if (wid < RBK && rb + wid < a.R) {
      // epilogue lane (m = lane & 15, q' = lane >> 4) = row m, columns 4 q' .. 4 q' + 3 of the tile: weight rows n = 4 q' + r
      // -> nh = q' >> 1, source lane (s' = 0 copy) = (m & 7) + 16 (q' & 1); rows of the other half are not this workgroup's
      const int m = lane & 15;
      const bool mine = (m >> 3) == h;
      const int src = (m & 7) + 16 * (q & 1);
      f32x4 v = red[wid][0][q >> 1][src];
#pragma unroll
      for (int i = 1; i < NW; ++i) v += red[wid][i][q >> 1][src];
      v *= wsc;                                         // power-of-two row scales: exact
      if (mine) gemv3_epilogue<1, EPI>(a, rb + wid, tile, lane, &v, &pre);
    }
#ifdef VAURA_STAMPS
    if (rb == 0) {
      VA_WAIT_VM(0);
      VA_STAMP(stamps, 6);
      VA_STAMP_FLUSH(stamps, K == 1536 ? 3 : 5);
    }
#endif
  }
}

// ---------------------------------------------------------------------------------------------------
// Prefill GEMM (the prompt of the sliding-window caller, scripts/generate.py:327-365): many row blocks at once.
// The decode kernel above keeps a workgroup's weight slice in registers and loops the row blocks, so EVERY workgroup
// streams the activation planes of ALL positions (166 positions x 147 KB through each CU's L2 port: 85 us per GEMV
// pass of 32 positions).  Here a workgroup owns 64 rows (4 row blocks) x 256 columns: the planes of a 32-deep k-group
// (12 KB) are staged once in LDS and shared by 8 waves, each wave streams the weight tiles of its own 2 column tiles
// straight into MFMA A operands (MFMA-tile layout: 1 KB contiguous per tile and k-group).  Same exact 3-plane
// arithmetic, same epilogues; the K-sum runs in one wave in k order.
#define G3M_RB 4
#define G3M_T 2
#define G3M_NW 8
// RBT = row blocks per workgroup: 4 (64 rows) or 8 (128 rows: every weight tile is re-read by half as many workgroups and a
// k-group carries twice the matrix work per barrier; bf16 / fp8 weights — the fp32-weight accumulators do not fit)
// PF = k-groups of weights in flight ahead of the one being multiplied (register sets).  gx, gy: the tile grid (column tiles of
// 256, row tiles of 16 RBT rows), launched as gx * gy blocks in one dimension; `remap`: blocks that share an XCD (ids equal
// mod 8 under round-robin dispatch) take a contiguous range of the tile order, and that order walks panels of 4 row tiles
// column by column — the 32 workgroups an XCD runs together then cover ~4 row tiles x 8 column tiles, whose current k-slices
// its L2 serves 8 and 4 times over.
template <int EPI, bool NORM, int WT = 0, int RBT = G3M_RB, int PF = 1>
__global__ __launch_bounds__(G3M_NW * 64) void gemm3_kernel(Gemv3Args a, int K, int gx, int gy, int remap) {
  constexpr int RB = RBT, T = G3M_T, NW = G3M_NW;
  static_assert(PF == 1 || (PF == 2 && WT != 1), "two register sets: fp16-plane weights");
  constexpr bool FP8 = WT == 1, F32 = WT == 2;
  constexpr int WH = F32 ? 2 : 1;
  constexpr int NACC = 2;
  __shared__ u32x4 xs[2][RB * VA_NPL * 64];
  __shared__ float rinv_s[RB * 16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bx, by;
  if (remap) {
    const int total = gx * gy, l = blockIdx.x, xcd = l & 7, per = total >> 3, extra = total & 7;
    const int lp = xcd * per + (xcd < extra ? xcd : extra) + (l >> 3);
    const int panel = lp / (4 * gx), rem = lp - panel * 4 * gx, pr = min(4, gy - panel * 4);
    bx = rem / pr;
    by = panel * 4 + rem - bx * pr;
  } else {
    bx = blockIdx.x % gx;
    by = blockIdx.x / gx;
  }
  const int rb0 = by * RB;
  const int tile0 = (bx * NW + wid) * T;
  const int KG = K / 32;
  constexpr int XL = (RB * VA_NPL * 64 + NW * 64 - 1) / (NW * 64);   // activation quads per thread per k-group
  // buffer loads (see gemv3_kernel): SGPR descriptor + uniform offset, one VGPR offset per load; the activation descriptor
  // ends with the last row block, so row blocks past a.R read zeros without a branch
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.W), 0, -16, 0x00020000);
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.XP), 0, a.R * VA_NPL * (K / 8) * 256, 0x00020000);
  int xvoff[XL];
#pragma unroll
  for (int i = 0; i < XL; ++i) {
    const int idx = tid + i * NW * 64;                 // (rbi, plane, lane') with lane' = q * 16 + m
    const int rbi = idx / (64 * VA_NPL), p = (idx / 64) % VA_NPL, l = idx & 63;
    xvoff[i] = idx < RB * 64 * VA_NPL ? ((rbi * VA_NPL + p) * (K / 8) * 16 + l) * 16 : 0x7ffffff0;     // past the tile: out of range -> 0
  }
  const int xsoff0 = rb0 * VA_NPL * (K / 8) * 256;

  u32x4 xr[PF][XL], wr[PF][T][WH];
  auto load_x = [&](int kg, auto slot) {
    constexpr int S = decltype(slot)::value;
#pragma unroll
    for (int i = 0; i < XL; ++i) xr[S][i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, xvoff[i], xsoff0 + kg * 1024, 0);
  };
  auto store_x = [&](int buf, auto slot) {
    constexpr int S = decltype(slot)::value;
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int idx = tid + i * NW * 64;
      if (idx < RB * 64 * VA_NPL) xs[buf][idx] = xr[S][i];
    }
  };
  auto load_w = [&](int kg, auto slot) {   // fp8 tile pairs: one 16-byte load carries k-groups kg and kg + 1
    constexpr int S = decltype(slot)::value;
    if (FP8 && (kg & 1)) return;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int hh = 0; hh < WH; ++hh)
        wr[S][t][hh] = __builtin_amdgcn_raw_buffer_load_b128(
            wrs, lane * 16, (int)(((FP8 ? (size_t)(tile0 + t) * (KG / 2) + (kg >> 1) : (size_t)(tile0 + t) * KG + kg) * WH + hh) * 1024), 2);
  };

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, PF - 1>;
  load_x(0, S0{});
  load_w(0, S0{});
  if constexpr (PF == 2) {
    load_x(1, S1{});
    load_w(1, S1{});
  }
  if constexpr (NORM) {   // rinv of the 64 rows: ordered sum of the producer's per-tile partial sums of squares
    if (tid < RB * 16) {
      const int rbi = tid >> 4, mm = tid & 15;
      float ssp = 0.f;
      if (rb0 + rbi < a.R) {
        const float* sp = a.ss_in + (size_t)(rb0 + rbi) * a.n_ss_in * 16 + mm;
        for (int i0 = 0; i0 < a.n_ss_in; i0 += 8) {      // eight loads in flight, summed in order
          float pv[8];
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) {
            const float x = sp[min(i0 + jj, a.n_ss_in - 1) * 16];
            pv[jj] = i0 + jj < a.n_ss_in ? x : 0.f;
          }
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) ssp += pv[jj];
        }
      }
      rinv_s[tid] = 1.0f / sqrtf(ssp * (1.0f / (float)a.k_total) + a.eps);
    }
  }
  store_x(0, S0{});
  __syncthreads();

  f32x4 acc[RB][T][NACC];
#pragma unroll
  for (int r = 0; r < RB; ++r)
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int p = 0; p < NACC; ++p) acc[r][t][p] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto kstep = [&](int kg, auto slot) {
    constexpr int S = decltype(slot)::value;
    const int buf = kg & 1;
    f16x8 wf[T][F32 ? 2 : 1];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      if constexpr (FP8) wf[t][0] = (kg & 1) ? fp8x8_to_f16(wr[S][t][0].z, wr[S][t][0].w) : fp8x8_to_f16(wr[S][t][0].x, wr[S][t][0].y);
      else wf[t][0] = __builtin_bit_cast(f16x8, wr[S][t][0]);
      if constexpr (F32) wf[t][1] = __builtin_bit_cast(f16x8, wr[S][t][WH - 1]);
    }
    // PF = 2: the planes and weights of k-group kg + 2 go into the register sets k-group kg just left (planes: stored to LDS at
    // the end of the previous step); a load then has a whole step to land before anything waits for it
    if (kg + PF < KG) {
      load_w(kg + PF, slot);
      load_x(kg + PF, slot);
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      u32x4 x3[VA_NPL];
#pragma unroll
      for (int p = 0; p < VA_NPL; ++p) x3[p] = xs[buf][(r * VA_NPL + p) * 64 + lane];
#pragma unroll
      for (int t = 0; t < T; ++t) mfma_group<WT>(wf[t], x3, acc[r][t]);
    }
    if (kg + 1 < KG) store_x(buf ^ 1, std::integral_constant<int, (PF == 2 ? S ^ 1 : 0)>{});
    __syncthreads();
  };
  if constexpr (PF == 2) {
    for (int kg = 0; kg < KG; kg += 2) {     // K / 32 is even for every K the launcher admits
      kstep(kg, S0{});
      kstep(kg + 1, S1{});
    }
  } else {
    for (int kg = 0; kg < KG; ++kg) kstep(kg, S0{});
  }

#pragma unroll
  for (int r = 0; r < RB; ++r) {
    if (rb0 + r >= a.R) break;
    const float rinv = NORM ? rinv_s[r * 16 + (lane & 15)] : 1.f;
    f32x4 v[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      f32x4 sacc = acc_sum<WT>(acc[r][t]);
      sacc *= *reinterpret_cast<const f32x4*>(a.wscale + (size_t)(tile0 + t) * 16 + 4 * (lane >> 4));
      v[t] = sacc * rinv;
    }
    gemv3_epilogue<T, EPI>(a, rb0 + r, tile0, lane, v);
  }
}

// ---------------------------------------------------------------------------------------------------
// Prefill GEMM, round 3: both operands through an LDS ring filled by LDS-DMA (`global_load_lds_dwordx4`: no register
// destination, so nothing the compiler has to wait for before a barrier), three stages of one 32-deep k-group each, one raw
// barrier per k-group with a COUNTED vmcnt in front of it: the pieces of k-group kg + 1 stay in flight across the barrier and
// every piece has two k-steps to land.  (gemm3_kernel above stages the planes through registers and streams each wave's own
// weight tiles into registers; hipcc waits vmcnt(0) for both before every barrier, so a load has less than one k-step —
// the asm shows it — and the MFMA pipe idles ~75 % of the time.)  Both operands are already MFMA fragments in memory (1 KB
// lane-linear per (row block | column tile, plane, k-group)), so a DMA piece is one fragment, the LDS image needs no swizzle
// (every ds_read_b128 reads 64 consecutive 16-byte slots) and fragments are shared: a weight fragment by the WM waves along the
// rows, a plane fragment by the WN waves along the columns.
//   workgroup = 8 waves = RBW row blocks (16 RBW rows) x 16 column tiles (256 columns); waves WM x WN = (RBW / 4) x (8 / WM);
//   wave tile = 4 row blocks x T = 16 / WN column tiles; accumulators 4 x T x 2 x 4 registers.
//   stage = RBW x 2 plane fragments + 16 x WH weight fragments (RBW 8: 32 / 48 KB with one / two weight planes)
// Same arithmetic and epilogues as gemm3_kernel: the accumulator of an output element sums the same products in the same
// (k-group) order, so the results are bit-identical to it.
#define G4_NW 8
#define G4_CT 16
#ifndef G4_SCHED
#define G4_SCHED 1
#endif
// CT = column tiles per workgroup: 16 (256 columns) or 8 (128 columns: with two weight planes a 128 x 128 workgroup stages 32 KB per
// k-group where 64 x 256 stages 40 KB, and four stages fit — the instances for wo / w2, whose 6 x 256 columns give too few workgroups
// for 128 rows)
template <int WT, int RBW, int CT = G4_CT>
struct G4Shape {
  static constexpr int WH = WT == 2 ? 2 : 1;
  static constexpr int XP = RBW * VA_NPL, WP = CT * WH;             // DMA pieces (1 KB) per stage
  // per wave: every wave issues the same number (the counted vmcnt waits are compile-time), so with 96 rows (28 / 44 pieces) the
  // last slots repeat the stage's first pieces — the same bytes to the same LDS address a second time
  static constexpr int PPW = (XP + WP + G4_NW - 1) / G4_NW;
  static constexpr int STB = (XP + WP) * 1024;                      // bytes per stage
  // stages: a k-group's stage is read during its own step and the one before, so NST stages leave NST - 2 k-steps between a
  // piece's issue and the barrier that needs it.  Four where they fit in 160 KB without costing a co-resident workgroup
  // (one weight plane, 128 rows: 128 KB); else three (two planes x 128 rows: 144 KB; one plane x 64 rows: 72 KB, two per CU)
  static constexpr int NST = (CT == 8 || (WT == 0 && RBW >= 6)) ? 4 : 3;
  static constexpr int LDS = NST * STB + 4 * RBW * 16 * 4;          // + four partial sums of squares per row of the workgroup
};

template <int N>
__device__ __forceinline__ void va_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// rb_off: first row block of this launch (a GEMM may be cut into two launches of different workgroup heights, gemv3.hip)
template <int EPI, bool NORM, int WT, int RBW, int CT = G4_CT>
__global__ __launch_bounds__(G4_NW * 64) void gemm4_kernel(Gemv3Args a, int K, int gx, int gy, int remap, int rb_off) {
  using SH = G4Shape<WT, RBW, CT>;
  constexpr int WH = SH::WH, PPW = SH::PPW, STB = SH::STB, NST = SH::NST;
  // waves WM x WN, RPW row blocks and T column tiles per wave: 64 rows = 1 x 8 waves of 4 x 2, 96 rows = 2 x 4 of 3 x 4, 128 rows = 2 x 4 of 4 x 4
  constexpr int WM = RBW == 4 ? 1 : 2, RPW = RBW / WM, WN = G4_NW / WM, T = CT / WN, NACC = 2;
  static_assert(CT % WN == 0 && (T * WH) % 2 == 0, "whole column tiles per wave, weight fragments in two halves");
  static_assert(WT == 0 || WT == 2, "fp16-plane weights (fp8 tile pairs keep gemm3_kernel)");
  static_assert(RBW == 4 || RBW == 6 || RBW == 8, "64, 96 or 128 rows");
  extern __shared__ __attribute__((aligned(16))) unsigned char g4_lds[];      // the ONE LDS object of this kernel (ring + rinv)
  float* ssq_s = reinterpret_cast<float*>(g4_lds + NST * STB);      // [4 parts][RBW * 16 rows]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid % WM, wn = wid / WM;
  int bx, by;
  if (remap & 1) {     // see gemm3_kernel
    const int total = gx * gy, l = blockIdx.x, xcd = l & 7, per = total >> 3, extra = total & 7;
    const int lp = xcd * per + (xcd < extra ? xcd : extra) + (l >> 3);
    const int panel = lp / (4 * gx), rem = lp - panel * 4 * gx, pr = min(4, gy - panel * 4);
    bx = rem / pr;
    by = panel * 4 + rem - bx * pr;
  } else {
    bx = blockIdx.x % gx;
    by = blockIdx.x / gx;
  }
  const int rb0 = rb_off + by * RBW, ct0 = bx * CT;
  const int KG = K / 32;

  // this wave's PPW pieces of a stage: piece p < XP = plane fragment (row block p / 2, plane p % 2), else weight fragment
  // (column tile (p - XP) / WH, plane (p - XP) % WH).  Row blocks past the last one re-read it (their outputs are not stored).
  const unsigned char* src[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int p = (wid * PPW + j) % (SH::XP + SH::WP);
    if (p < SH::XP) {
      const int rb = min(rb0 + p / VA_NPL, a.R - 1);
      src[j] = reinterpret_cast<const unsigned char*>(a.XP) + ((size_t)(rb * VA_NPL + p % VA_NPL) * KG) * 1024 + lane * 16;
    } else {
      const int pw = p - SH::XP;
      src[j] = static_cast<const unsigned char*>(a.W) + ((size_t)(ct0 + pw / WH) * KG * WH + pw % WH) * 1024 + lane * 16;
    }
  }
  const int xstep = 1024, wstep = 1024 * WH;      // bytes from one k-group's fragment to the next
  auto issue = [&](int stage) {                  // the next k-group's pieces of this wave -> LDS stage
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int p = (wid * PPW + j) % (SH::XP + SH::WP);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src[j],
                                       (__attribute__((address_space(3))) void*)(g4_lds + stage * STB + p * 1024), 16, 0, 0);
      src[j] += (p < SH::XP) ? xstep : wstep;
    }
  };
#pragma unroll
  for (int st = 0; st < NST - 1; ++st)
    if (st < KG) issue(st);

  if constexpr (NORM) {
    // sum of squares of the workgroup's rows from the producer's per-tile partials, in the decode GEMV's order (part q = tiles
    // q, q + 4, ... in turn, then (p0 + p1) + (p2 + p3)): the same rinv, bit for bit, as when the position is decoded alone.
    // Eight loads in flight per thread (round 2's loop over n_ss_in waited for each of its 96 loads: ~20 us per GEMM).
    if (tid < RBW * 64) {
      const int row = tid % (RBW * 16), q = tid / (RBW * 16);
      float ssp = 0.f;
      if (rb0 + (row >> 4) < a.R) {
        const float* sp = a.ss_in + (size_t)(rb0 + (row >> 4)) * a.n_ss_in * 16 + (row & 15);
        for (int j0 = 0; 4 * j0 < a.n_ss_in; j0 += 8) {
          float pv[8];
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) {
            const int i = q + 4 * (j0 + jj);
            const float x = sp[min(i, a.n_ss_in - 1) * 16];      // unconditional load (a select on the load itself makes hipcc branch around it)
            pv[jj] = i < a.n_ss_in ? x : 0.f;
          }
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) ssp += pv[jj];
        }
      }
      ssq_s[q * (RBW * 16) + row] = ssp;
    }
  }

  f32x4 acc[RPW][T][NACC];
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int pp = 0; pp < NACC; ++pp) acc[r][t][pp] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Fragment addresses: plane fragment (row block r of this wave, plane pl) and weight fragment (tile t, plane hh) of a stage
  auto xfrag = [&](int stage, int r, int pl) -> u32x4 {
    return (reinterpret_cast<const u32x4*>(g4_lds + stage * STB) + lane)[((wm * RPW + r) * VA_NPL + pl) * 64];
  };
  auto wfrag = [&](int stage, int t, int hh) -> f16x8 {
    return __builtin_bit_cast(f16x8, (reinterpret_cast<const u32x4*>(g4_lds + stage * STB) + lane)[(SH::XP + (wn * T + t) * WH + hh) * 64]);
  };
  // One k-group.  On entry the weight fragments of k-group kg (register set S) and the planes of its row block 0 are already
  // requested; during its four row-block steps the wave requests, ahead of their use, the planes of the next row block and — from
  // stage kg + 1, landed and published by this step's barrier — the weight fragments of k-group kg + 1 (set S ^ 1) and the planes of
  // ITS row block 0: LDS reads and MFMAs of one wave overlap, and the waves of the workgroup no longer read LDS in one burst
  // behind the barrier with the matrix pipe idle (compute alone, DMA disabled, ran at 37 % of the MFMA peak that way).
  // The stage k-group kg - 1 lived in is free at this barrier: it takes k-group kg + NST - 1.
  f16x8 wf[2][T][WH];
  u32x4 x3[2][VA_NPL];
  int st0 = 0;            // stage of k-group kg
  auto kstep = [&](int kg, auto slot) {
    constexpr int S = decltype(slot)::value;
    const int st1 = st0 == NST - 1 ? 0 : st0 + 1;           // stage of k-group kg + 1
    const int stf = st0 == 0 ? NST - 1 : st0 - 1;           // the free one
    // this wave's pieces of k-group kg + 1 have landed: the stages issued after it may still fly (none near the end)
    if (kg + NST - 2 < KG) va_wait_vmcnt<PPW * (NST - 3)>(); else va_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kg + NST - 1 < KG && !(remap & 2)) issue(stf);       // remap bits 1, 2: ablations for tools/time_prefill_gemm.py (no DMA / no products)
    if (remap & 4) { st0 = st1; return; }
    // (the last k-group reads a stale stage here and drops it: no branch in the step)
    constexpr int PB = (S * RPW) & 1;       // the two plane-fragment sets alternate per row block, across k-groups too (RPW may be odd)
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      if (r + 1 < RPW) {
#pragma unroll
        for (int pl = 0; pl < VA_NPL; ++pl) x3[(PB + r + 1) & 1][pl] = xfrag(st0, r + 1, pl);
      } else {
#pragma unroll
        for (int pl = 0; pl < VA_NPL; ++pl) x3[(PB + RPW) & 1][pl] = xfrag(st1, 0, pl);
      }
      if (r == 1 || r == 2) {      // next k-group's weight fragments: half with row block 1, half with row block 2
#pragma unroll
        for (int i = (r - 1) * (T * WH / 2); i < r * (T * WH / 2); ++i) wf[S ^ 1][i / WH][i % WH] = wfrag(st1, i / WH, i % WH);
      }
#pragma unroll
      for (int t = 0; t < T; ++t) mfma_group<WT>(wf[S][t], x3[(PB + r) & 1], acc[r][t]);
    }
    if constexpr (G4_SCHED) {
      // the order above, pinned: per row block {the reads written with it, its MFMAs}
      constexpr int MPR = T * (WT == 2 ? 3 : 2);
      __builtin_amdgcn_sched_group_barrier(0x100, VA_NPL, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, MPR, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, VA_NPL + T * WH / 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, MPR, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, VA_NPL + T * WH / 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, MPR, 0);
      if constexpr (RPW == 4) {
        __builtin_amdgcn_sched_group_barrier(0x100, VA_NPL, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, MPR, 0);
      }
    }
    st0 = st1;
  };
  // k-group 0's fragments: its stage has landed (NST - 2 younger stages may fly) and is published by a barrier
  if (KG >= NST - 1) va_wait_vmcnt<PPW * (NST - 2)>(); else va_wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int hh = 0; hh < WH; ++hh) wf[0][t][hh] = wfrag(0, t, hh);
#pragma unroll
  for (int pl = 0; pl < VA_NPL; ++pl) x3[0][pl] = xfrag(0, 0, pl);
  for (int kg = 0; kg < KG; kg += 2) {       // K / 32 is even for every K the launcher admits
    kstep(kg, std::integral_constant<int, 0>{});
    kstep(kg + 1, std::integral_constant<int, 1>{});
  }

  if constexpr (NORM) __syncthreads();      // ssq_s (written before the loop; KG >= 1 barriers passed, but say so)
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int rb = rb0 + wm * RPW + r;
    if (rb >= a.R) break;
    float rinv = 1.f;
    if constexpr (NORM) {
      const float* pq = ssq_s + (wm * RPW + r) * 16 + (lane & 15);
      const float ssp = (pq[0] + pq[RBW * 16]) + (pq[2 * RBW * 16] + pq[3 * RBW * 16]);
      rinv = 1.0f / sqrtf(ssp * (1.0f / (float)a.k_total) + a.eps);
    }
    f32x4 v[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      f32x4 sacc = acc_sum<WT>(acc[r][t]);
      sacc *= *reinterpret_cast<const f32x4*>(a.wscale + (size_t)(ct0 + wn * T + t) * 16 + 4 * (lane >> 4));
      v[t] = sacc * rinv;
    }
    gemv3_epilogue<T, EPI>(a, rb, ct0 + wn * T, lane, v);
  }
}
