// Decode-step causal self-attention for ONE new position per sequence, with the K/V cache the
// reference lacks.  Replaces, for the last row of the causal product,
//   apply_rotary_emb(q), apply_rotary_emb(k)       models/modules/sampler/llama.py:234-235, 633-650
//   F.scaled_dot_product_attention(is_causal=True) llama.py:246-255   (scale = 1/sqrt(head_dim))
//   merge heads                                    llama.py:257
//
// Bound: HBM (every cached K and V element of the (row, head) is read once; fp32 cache).
// One 256-thread workgroup per (head, row).  8 lanes cover one cached position (3 x 16 B each,
// 128-B segments per load instruction), 32 positions per pass; scores -> LDS, block softmax,
// then the same mapping accumulates P.V, reduced through shuffles + LDS.  The rotated key and the
// value of the new position are appended to the cache by this kernel.
#include "common.h"
#include "gemv3_kernel.h"
#include "mlp_engine.h"

#define ATT_THREADS 256

template <int HD>
__global__ __launch_bounds__(ATT_THREADS) void attention_step_kernel(
    const float* __restrict__ qkv,   // packed rows (rows x 3*D)
    const float* __restrict__ qkv2,  // optional second K-half partial of qkv (decode), added on load
    const float* __restrict__ rope,  // (max_len, HD/2, 2)
    float* __restrict__ kcache,      // (rows, H, max_len, HD)
    float* __restrict__ vcache,
    float* __restrict__ out,         // packed rows (rows x D)
    uint16_t* __restrict__ outp,     // optional split rows (rows x D)
    int n_head, int max_len, const int32_t* __restrict__ pos_dev, int pos_host, int prefill_rows16,
    float pscale) {                  // the planes hold out * pscale (a power of two; 1 = the product default: engine plane_shift)
  constexpr int QUADS = HD / 4;          // 24
  constexpr int QPL = QUADS / 8;         // float4 per lane per position = 3
  static_assert(QUADS % 8 == 0, "head_dim must be a multiple of 32");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sq = smem;                      // HD   rotated query (pre-scaled by nothing; scale applied to score)
  float* sk = sq + HD;                   // HD   rotated key of the new position
  float* sv = sk + HD;                   // HD   value of the new position
  float* red = sv + HD;                  // 4*HD cross-wave P.V partials
  float* rbuf = red + 4 * HD;            // 8    block reductions
  float* sc = rbuf + 8;                  // pos+1 scores / probabilities

  const int h = blockIdx.x, row = blockIdx.y;
  const int tid = threadIdx.x;
  const int D = n_head * HD;
  // decode (prefill_rows16 == 0): one new position `pos`, rotated/appended here, cache holds [0, pos).
  // prefill (prefill_rows16 > 0): workgroup z handles position pos_host + z of a teacher-forced chunk whose
  // rotated q and cache rows [0, pos] were written by rope_append_kernel; row block z holds its q / output.
  const bool prefill = prefill_rows16 > 0;
  const int pos = prefill ? pos_host + (int)blockIdx.z : (pos_dev ? pos_dev[0] : pos_host);
  const int vrow = prefill ? (int)blockIdx.z * prefill_rows16 + row : row;
  const int ncache = prefill ? pos + 1 : pos;   // positions read from the cache
  const int L = pos + 1;
  const float scale = 1.0f / sqrtf((float)HD);

  float* kc = kcache + ((size_t)row * n_head + h) * (size_t)max_len * HD;
  float* vc = vcache + ((size_t)row * n_head + h) * (size_t)max_len * HD;

  // ---- 1. gather q, k, v of this head; rotate q and k; append k, v
  if (tid < (prefill ? QUADS : 3 * QUADS)) {
    const int which = tid / QUADS, cq = tid % QUADS;
    const int col = which * D + h * HD + cq * 4;
    f32x4 x = reinterpret_cast<const f32x4*>(qkv)[packed_quad(vrow, col >> 2, 3 * D)];
    if (qkv2) x += reinterpret_cast<const f32x4*>(qkv2)[packed_quad(vrow, col >> 2, 3 * D)];
    if (which < 2 && !prefill) {
      const f32x4 cs = *reinterpret_cast<const f32x4*>(rope + ((size_t)pos * (HD / 2) + cq * 2) * 2);  // c0 s0 c1 s1
      f32x4 y;
      y[0] = x[0] * cs[0] - x[1] * cs[1];
      y[1] = x[1] * cs[0] + x[0] * cs[1];
      y[2] = x[2] * cs[2] - x[3] * cs[3];
      y[3] = x[3] * cs[2] + x[2] * cs[3];
      x = y;
    }
    reinterpret_cast<f32x4*>(which == 0 ? sq : (which == 1 ? sk : sv))[cq] = x;
    if (which == 1 && !prefill) reinterpret_cast<f32x4*>(kc + (size_t)pos * HD)[cq] = x;
    if (which == 2 && !prefill) reinterpret_cast<f32x4*>(vc + (size_t)pos * HD)[cq] = x;
  }
  __syncthreads();

  // ---- 2. scores over cached positions [0, pos) + the new one from LDS
  const int sub = tid & 7;       // which 16-B column group (x3) of the position
  const int prow = tid >> 3;     // position inside a pass of 32
  f32x4 qf[QPL];
#pragma unroll
  for (int i = 0; i < QPL; ++i) qf[i] = reinterpret_cast<const f32x4*>(sq)[sub + 8 * i];

  for (int p0 = 0; p0 < ncache; p0 += 32 * 4) {
    f32x4 kf[4][QPL];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = p0 + u * 32 + prow;
#pragma unroll
      for (int i = 0; i < QPL; ++i)
        kf[u][i] = (p < ncache) ? reinterpret_cast<const f32x4*>(kc + (size_t)p * HD)[sub + 8 * i] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float d = 0.f;
#pragma unroll
      for (int i = 0; i < QPL; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) d = fmaf(qf[i][c], kf[u][i][c], d);
      d += va_dpp<VA_DPP_XOR1>(d);
      d += va_dpp<VA_DPP_XOR2>(d);
      d += va_dpp<VA_DPP_HALF_MIRROR>(d);
      const int p = p0 + u * 32 + prow;
      if (sub == 0 && p < ncache) sc[p] = d * scale;
    }
  }
  if (tid < 64 && !prefill) {  // new position: one wave, lanes 0..23 hold a quad each
    float d = 0.f;
    if (tid < QUADS) {
      const f32x4 a = reinterpret_cast<const f32x4*>(sq)[tid], b = reinterpret_cast<const f32x4*>(sk)[tid];
      d = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
    }
    d = wave_sum(d);
    if (tid == 0) sc[pos] = d * scale;
  }
  __syncthreads();

  // ---- 3. softmax over sc[0..L)
  float mx = -INFINITY;
  for (int i = tid; i < L; i += ATT_THREADS) mx = fmaxf(mx, sc[i]);
  mx = wave_max(mx);
  if ((tid & 63) == 0) rbuf[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(rbuf[0], rbuf[1]), fmaxf(rbuf[2], rbuf[3]));
  float sm = 0.f;
  for (int i = tid; i < L; i += ATT_THREADS) {
    const float e = expf(sc[i] - mx);
    sc[i] = e;
    sm += e;
  }
  sm = wave_sum(sm);
  if ((tid & 63) == 0) rbuf[4 + (tid >> 6)] = sm;
  __syncthreads();
  const float inv = 1.0f / (((rbuf[4] + rbuf[5]) + rbuf[6]) + rbuf[7]);

  // ---- 4. P.V with the same lane mapping
  f32x4 av[QPL];
#pragma unroll
  for (int i = 0; i < QPL; ++i) av[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int p0 = 0; p0 < ncache; p0 += 32 * 4) {
    f32x4 vf[4][QPL];
    float pw[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = p0 + u * 32 + prow;
      pw[u] = (p < ncache) ? sc[p] * inv : 0.f;
#pragma unroll
      for (int i = 0; i < QPL; ++i)
        vf[u][i] = (p < ncache) ? reinterpret_cast<const f32x4*>(vc + (size_t)p * HD)[sub + 8 * i] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < QPL; ++i) av[i] += vf[u][i] * pw[u];
  }
  // reduce over the 8 position-rows of the wave (lane bits 3..5), then over the 4 waves
#pragma unroll
  for (int i = 0; i < QPL; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v = av[i][c];
      v += va_dpp<VA_DPP_ROR8>(v);
      v += va_xor16(v);
      v += va_xor32(v);
      av[i][c] = v;
    }
  const int wv = tid >> 6, lane = tid & 63;
  if (lane < 8) {
#pragma unroll
    for (int i = 0; i < QPL; ++i) reinterpret_cast<f32x4*>(red + wv * HD)[lane + 8 * i] = av[i];
  }
  __syncthreads();
  if (tid < QUADS) {
    f32x4 o = reinterpret_cast<const f32x4*>(red)[tid];
#pragma unroll
    for (int i = 1; i < 4; ++i) o += reinterpret_cast<const f32x4*>(red + i * HD)[tid];
    if (!prefill) o += reinterpret_cast<const f32x4*>(sv)[tid] * (sc[pos] * inv);
    va_st16(reinterpret_cast<f32x4*>(out) + packed_quad(vrow, (h * HD) / 4 + tid, D), o);
    if (outp) store_split4(outp, vrow, h * HD + 4 * tid, D, o * pscale);
  }
}

// ---------------------------------------------------------------------------------------------------
// Decode step for caches of at most 256 positions (every 2.56 s configuration: S = 229): ONE memory round
// trip.  The generic kernel above pays four dependent HBM round trips at L > 128 (K chunk, K chunk, V chunk,
// V chunk: 4.4 us at L = 1 -> 13.1 us at L = 228).  Here every cached K and V row of the (row, head) is
// requested before anything else — the addresses need only `pos` — and lands in registers (512 threads x
// 96 VGPRs) while the new q/k/v are gathered, rotated and appended; scores, a per-wave softmax (running max
// per wave, combined once across the 8 waves, flash-decoding style) and P.V then run out of registers with two
// workgroup barriers in total.
#define ATT1_THREADS 512

// fp16 K/V cache (round 6, vaura_decoder.kv_dtype = 1: the low-precision serving configuration, BASELINE configs[4]): the cache holds
// fp16(rotated k) / fp16(v), the same [layer][row][head][max_len][96] layout at two bytes per element; everything else (q, scores, softmax,
// P.V accumulation) stays fp32.  At 32 rows the attention launch IS its K / V stream (44.8 MB of fp32 per layer at the loop's mean length):
// half the bytes.  A quad of four channels is 8 bytes.
__device__ __forceinline__ f32x4 kv_quad_to_f32(const uint2 q) {
  const f16x2 a = __builtin_bit_cast(f16x2, q.x), b = __builtin_bit_cast(f16x2, q.y);
  return f32x4{(float)a[0], (float)a[1], (float)b[0], (float)b[1]};
}
__device__ __forceinline__ uint2 kv_quad_to_f16(const f32x4 v) {
  return uint2{pack_h2((_Float16)v[0], (_Float16)v[1]), pack_h2((_Float16)v[2], (_Float16)v[3])};
}
// kv_dtype = 2: OCP e4m3, unscaled (rotated keys and values of this model are O(1..100); values beyond +-448 saturate): a quad is 4 bytes
__device__ __forceinline__ f32x4 kv_quad8_to_f32(const uint32_t q) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8((int)q, false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)q, true);
  const float a0 = a[0], a1 = a[1], b0 = b[0], b1 = b[1];
  return f32x4{a0, a1, b0, b1};
}
__device__ __forceinline__ uint32_t kv_quad_to_f8(const f32x4 v) {
  auto sat = [](float x) { return fminf(fmaxf(x, -448.f), 448.f); };
  int q = __builtin_amdgcn_cvt_pk_fp8_f32(sat(v[0]), sat(v[1]), 0, false);
  q = __builtin_amdgcn_cvt_pk_fp8_f32(sat(v[2]), sat(v[3]), q, true);
  return (uint32_t)q;
}
// K / V storage of a kernel instance: KVT = vaura_decoder.kv_dtype (0 fp32, 1 fp16, 2 fp8 e4m3); Q = one quad (4 channels) as stored
template <int KVT> struct KvT;
template <> struct KvT<0> {
  using Q = f32x4; using E = float;
  static __device__ __forceinline__ f32x4 widen(const Q& q) { return q; }
  static __device__ __forceinline__ Q narrow(const f32x4& v) { return v; }
};
template <> struct KvT<1> {
  using Q = uint2; using E = _Float16;
  static __device__ __forceinline__ f32x4 widen(const Q& q) { return kv_quad_to_f32(q); }
  static __device__ __forceinline__ Q narrow(const f32x4& v) { return kv_quad_to_f16(v); }
};
template <> struct KvT<2> {
  using Q = uint32_t; using E = uint8_t;
  static __device__ __forceinline__ f32x4 widen(const Q& q) { return kv_quad8_to_f32(q); }
  static __device__ __forceinline__ Q narrow(const f32x4& v) { return kv_quad_to_f8(v); }
};

// NU = number of 64-position passes that hold cached rows (0..4).  One straight-line body per NU: the compiler
// then waits with exact vmcnt values (vmcnt retires in order), and short caches issue no dead loads.
struct NoHook {
  __device__ __forceinline__ void operator()() const {}
};
// `after_requests`: called once every request of the attention itself has been issued (the fused attention + wo launch queues
// its wo weight requests there: behind the K/V rows, which the dependent chain needs first)
template <int HD, int NU, typename HOOK = NoHook, int KVT = 0>
__device__ __forceinline__ void attention256_body(const float* __restrict__ qkv, const float* __restrict__ qkv2,
                                                  const float* __restrict__ rope,
                                                  float* __restrict__ kc, float* __restrict__ vc, float* __restrict__ out,
                                                  uint16_t* __restrict__ outp, int n_head, int pos, f32x4* sqkv,
                                                  f32x4 (*wacc)[HD / 4], float* wm, float* wl, float pscale = 1.f,
                                                  HOOK after_requests = HOOK()) {
  constexpr int QUADS = HD / 4;   // 24
  constexpr int QPL = QUADS / 8;  // 3
  constexpr int NW = ATT1_THREADS / 64;
  constexpr int NUA = NU > 0 ? NU : 1;
  const f32x4* sq4 = sqkv;
  const f32x4* sk4 = sqkv + QUADS;
  const f32x4* sv4 = sqkv + 2 * QUADS;
  const int h = blockIdx.x, row = blockIdx.y, tid = threadIdx.x;
  const int D = n_head * HD;
  const float scale = 1.0f / sqrtf((float)HD);
  const int sub = tid & 7;    // 16-B column group (x3) of a position
  const int prow = tid >> 3;  // 0..63: position inside a pass of 64

  // ---- 1. requests first: the new q/k/v quad + its rope entry, then the whole cache of this (row, head), K
  //         before V.  Slots past the end of the last pass re-read the last cached row and are masked below.
  VA_STAMP_DECL(stamps);
  VA_STAMP(stamps, 0);
  const int gt = min(tid, 3 * QUADS - 1);
  const int which = gt / QUADS, cq = gt % QUADS;
  f32x4 gx = reinterpret_cast<const f32x4*>(qkv)[packed_quad(row, (which * D + h * HD + cq * 4) >> 2, 3 * D)];
  // the second k-half of a split qkv GEMV (the one-launch MLP's third phase): loaded UNCONDITIONALLY (from qkv again when there is
  // none) and selected after the cache requests — `if (qkv2) gx += load` compiled to a branch with its own s_waitcnt vmcnt(0), i.e. one
  // whole memory round trip before the first K row was even requested
  const f32x4 gx2 = reinterpret_cast<const f32x4*>(qkv2 ? qkv2 : qkv)[packed_quad(row, (which * D + h * HD + cq * 4) >> 2, 3 * D)];
  const f32x4 gcs = *reinterpret_cast<const f32x4*>(rope + ((size_t)pos * (HD / 2) + cq * 2) * 2);  // c0 s0 c1 s1
  __builtin_amdgcn_sched_barrier(0);   // keep these first in program order (the scheduler sinks them otherwise)
  // KVT != 0: kc / vc point at fp16 / fp8 rows (the kernel offsets them in ELEMENTS of that type); a quad is 8 / 4 bytes and is widened
  // where it is used
  using KV = KvT<KVT>;
  using KVQ = typename KV::Q;
  KVQ kf[NUA][QPL], vf[NUA][QPL];
  auto kvrow = [&](const float* base, int p) {
    return reinterpret_cast<const KVQ*>(reinterpret_cast<const typename KV::E*>(base) + (size_t)p * HD);
  };
  auto widen = [&](const KVQ& x) -> f32x4 { return KV::widen(x); };
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int p = min(u * 64 + prow, pos - 1);
#pragma unroll
    for (int i = 0; i < QPL; ++i) kf[u][i] = kvrow(kc, p)[sub + 8 * i];
  }
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int p = min(u * 64 + prow, pos - 1);
#pragma unroll
    for (int i = 0; i < QPL; ++i) vf[u][i] = kvrow(vc, p)[sub + 8 * i];
  }

  after_requests();
  VA_STAMP(stamps, 1);                       // every request issued
  VA_WAIT_VM(2 * NU * QPL);
  VA_STAMP(stamps, 2);                       // the new q / k / v quads (written by the previous kernel) and the rope entry have landed
  // ---- 2. rotate q and k (v passes through), park them in LDS, append k and v to the cache.  Every thread
  //         rotates and writes LDS (threads past the 72 real quads hit a scratch slot): an unconditional use keeps
  //         the compiler from sinking the two loads into a branch behind the cache loads.
  {
    const f32x4 both = gx + gx2;
    gx = qkv2 ? both : gx;
  }
  f32x4 y;
  y[0] = gx[0] * gcs[0] - gx[1] * gcs[1];
  y[1] = gx[1] * gcs[0] + gx[0] * gcs[1];
  y[2] = gx[2] * gcs[2] - gx[3] * gcs[3];
  y[3] = gx[3] * gcs[2] + gx[2] * gcs[3];
  if (which == 2) y = gx;   // v is not rotated
  if constexpr (KVT != 0) {      // the new position's k / v are what later steps will read back: the stored (fp16 / fp8) values, for this step too
    if (which >= 1) y = KV::widen(KV::narrow(y));
  }
  sqkv[tid < 3 * QUADS ? tid : 3 * QUADS + (tid & 63)] = y;
  if (tid >= QUADS && tid < 3 * QUADS) {
    if constexpr (KVT == 1) va_st8(reinterpret_cast<uint2*>(reinterpret_cast<_Float16*>(which == 1 ? kc : vc) + (size_t)pos * HD) + cq, kv_quad_to_f16(y));
    else if constexpr (KVT == 2) va_st4(reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(which == 1 ? kc : vc) + (size_t)pos * HD) + cq, __builtin_bit_cast(float, kv_quad_to_f8(y)));
    else va_st16(reinterpret_cast<f32x4*>((which == 1 ? kc : vc) + (size_t)pos * HD) + cq, y);
  }
  // Only the LDS writes have to be visible behind this barrier.  __syncthreads() would also drain the vector-memory counter, i.e. wait
  // for EVERY cached K and V row before the first score; with a raw barrier the rows stay in flight and the compiler's own counted
  // waits let pass u's scores start when pass u's rows have landed.  (Measured neutral on the 228-step loop, like the unconditional
  // qkv2 load above: the launch is bound by the K / V stream itself — profiles/r04_ab_mlp_engine.txt, attention section.)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#ifdef VAURA_STAMPS
  VA_STAMP(stamps, 3);                       // rotated q / k / v parked in LDS (first barrier)
  VA_WAIT_VM(0);
  VA_STAMP(stamps, 4);                       // every cached K and V row has landed
#endif

  // ---- 3. scores: 8 lanes per position; the new position's score is computed by every 8-lane group
  f32x4 qf[QPL];
#pragma unroll
  for (int i = 0; i < QPL; ++i) qf[i] = sq4[sub + 8 * i];
  auto dot8 = [&](const f32x4* kv) {
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < QPL; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c) d = fmaf(qf[i][c], kv[i][c], d);
    d += va_dpp<VA_DPP_XOR1>(d);
    d += va_dpp<VA_DPP_XOR2>(d);
    d += va_dpp<VA_DPP_HALF_MIRROR>(d);
    return d;
  };
  f32x4 knew[QPL];
#pragma unroll
  for (int i = 0; i < QPL; ++i) knew[i] = sk4[sub + 8 * i];
  const float snew = dot8(knew) * scale;
  float sc[NUA];
  float m = snew;   // every wave's running max includes the new position, so it is finite
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    f32x4 kw[QPL];
#pragma unroll
    for (int i = 0; i < QPL; ++i) kw[i] = widen(kf[u][i]);
    const float d = dot8(kw) * scale;
    sc[u] = (u * 64 + prow < pos) ? d : -INFINITY;
    m = fmaxf(m, sc[u]);
  }
  m = wave_max(m);
  // ---- 4. per-wave softmax numerators and P.V partial (relative to the wave's max)
  float l = 0.f;
  f32x4 av[QPL];
#pragma unroll
  for (int i = 0; i < QPL; ++i) av[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const float e = expf(sc[u] - m);   // exp(-inf) = 0 for the masked slots (they hold a finite, valid row)
    if (sub == 0) l += e;
#pragma unroll
    for (int i = 0; i < QPL; ++i) av[i] += widen(vf[u][i]) * e;
  }
  l = wave_sum(l);
#pragma unroll
  for (int i = 0; i < QPL; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v = av[i][c];
      v += va_dpp<VA_DPP_ROR8>(v);
      v += va_xor16(v);
      v += va_xor32(v);
      av[i][c] = v;
    }
  const int wv = tid >> 6, lane = tid & 63;
  if (lane < 8) {
#pragma unroll
    for (int i = 0; i < QPL; ++i) wacc[wv][lane + 8 * i] = av[i];
  }
  if (lane == 0) { wm[wv] = m; wl[wv] = l; }
  __syncthreads();
  VA_STAMP(stamps, 5);                       // scores, softmax, P.V of every wave done (second barrier)

  // ---- 5. combine the 8 waves and the new position
  if (tid < QUADS) {
    float M = wm[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) M = fmaxf(M, wm[w]);
    const float en = expf(snew - M);
    float denom = en;
    f32x4 o = sv4[tid] * en;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float f = expf(wm[w] - M);
      denom += f * wl[w];
      o += wacc[w][tid] * f;
    }
    o *= 1.0f / denom;
    va_st16(reinterpret_cast<f32x4*>(out) + packed_quad(row, (h * HD) / 4 + tid, D), o);
    if (outp) store_split4(outp, row, h * HD + 4 * tid, D, o * pscale);
  }
#ifdef VAURA_STAMPS
  VA_WAIT_VM(0);
  VA_STAMP(stamps, 6);                       // output stores acknowledged
  VA_STAMP_FLUSH(stamps, 2);
#endif
}

template <int HD, int KVT = 0>
__global__ __launch_bounds__(ATT1_THREADS) void attention_step256_kernel(
    // argument order = what the dependent chain needs first (the leading 14 dwords are preloaded into SGPRs)
    const int32_t* __restrict__ pos_dev, float* __restrict__ kcache, float* __restrict__ vcache, const float* __restrict__ qkv,
    const float* __restrict__ qkv2, const float* __restrict__ rope, int n_head, int max_len, int pos_host,
    float* __restrict__ out, uint16_t* __restrict__ outp, float pscale) {
  constexpr int QUADS = HD / 4;
  __shared__ f32x4 sqkv[3 * QUADS + 64];   // rotated q | rotated k | v of the new position | scratch
  __shared__ f32x4 wacc[ATT1_THREADS / 64][QUADS];
  __shared__ float wm[ATT1_THREADS / 64], wl[ATT1_THREADS / 64];
  const int pos = pos_dev ? pos_dev[0] : pos_host;   // cache holds [0, pos), pos <= 255
  // KVT != 0: the cache is fp16 / fp8 — the (row, head) offset in ELEMENTS of that type (the pointer stays float* in the signature: same kernarg layout)
  const size_t off = ((size_t)blockIdx.y * n_head + blockIdx.x) * (size_t)max_len * HD;
  float* kc = reinterpret_cast<float*>(reinterpret_cast<typename KvT<KVT>::E*>(kcache) + off);
  float* vc = reinterpret_cast<float*>(reinterpret_cast<typename KvT<KVT>::E*>(vcache) + off);
  switch ((pos + 63) >> 6) {
    case 0: attention256_body<HD, 0, NoHook, KVT>(qkv, qkv2, rope, kc, vc, out, outp, n_head, pos, sqkv, wacc, wm, wl, pscale); break;
    case 1: attention256_body<HD, 1, NoHook, KVT>(qkv, qkv2, rope, kc, vc, out, outp, n_head, pos, sqkv, wacc, wm, wl, pscale); break;
    case 2: attention256_body<HD, 2, NoHook, KVT>(qkv, qkv2, rope, kc, vc, out, outp, n_head, pos, sqkv, wacc, wm, wl, pscale); break;
    case 3: attention256_body<HD, 3, NoHook, KVT>(qkv, qkv2, rope, kc, vc, out, outp, n_head, pos, sqkv, wacc, wm, wl, pscale); break;
    default: attention256_body<HD, 4, NoHook, KVT>(qkv, qkv2, rope, kc, vc, out, outp, n_head, pos, sqkv, wacc, wm, wl, pscale); break;
  }
}

// ---------------------------------------------------------------------------------------------------
// Long caches with few (row, head) pairs (BASELINE configs[3]: 10.24 s single pass, 4 rows x 16 heads = 64
// workgroups for 256 CUs; also small batches): split the cache range of one (row, head) over n_split workgroups
// (flash-decoding).  Split z streams positions [z*per, (z+1)*per) in blocks of 256 with the same request-first
// register scheme as above and an online softmax across blocks; the last split also owns the new position
// (rotation of k, cache append, its score).  Every split writes (o relative to its max M, M, denominator) to a
// small workspace; attention_combine_kernel merges the n_split partials, normalises and emits the packed /
// split-row output.  part layout: [((row * H + h) * n_split + z)][HD + 8] floats, [HD] = M, [HD + 1] = denom.
#define ATT_PART_STRIDE(HD) ((HD) + 8)

// One block of up to NU*64 cached positions [b0, hi): request every K and V row first (load), then scores, online
// softmax update of the wave's running (m, l, av) and P.V (consume).
template <int HD, int NU>
struct AttentionBlock {
  static constexpr int QPL = HD / 32;
  f32x4 kf[NU][QPL], vf[NU][QPL];
  __device__ __forceinline__ void load(const float* __restrict__ kc, const float* __restrict__ vc, int b0, int hi, int sub,
                                       int prow) {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int p = min(b0 + u * 64 + prow, hi - 1);
#pragma unroll
      for (int i = 0; i < QPL; ++i) kf[u][i] = reinterpret_cast<const f32x4*>(kc + (size_t)p * HD)[sub + 8 * i];
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int p = min(b0 + u * 64 + prow, hi - 1);
#pragma unroll
      for (int i = 0; i < QPL; ++i) vf[u][i] = reinterpret_cast<const f32x4*>(vc + (size_t)p * HD)[sub + 8 * i];
    }
  }
  __device__ __forceinline__ void consume(int b0, int hi, const f32x4* qf, float scale, int sub, int prow, float& m_run,
                                          float& l_run, f32x4* av) const {
    float sc[NU];
    float m = m_run;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      float d = 0.f;
#pragma unroll
      for (int i = 0; i < QPL; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) d = fmaf(qf[i][c], kf[u][i][c], d);
      d += va_dpp<VA_DPP_XOR1>(d);
      d += va_dpp<VA_DPP_XOR2>(d);
      d += va_dpp<VA_DPP_HALF_MIRROR>(d);
      sc[u] = (b0 + u * 64 + prow < hi) ? d * scale : -INFINITY;
      m = fmaxf(m, sc[u]);
    }
    m = wave_max(m);                                   // new running max of this wave
    const float mref = (m == -INFINITY) ? 0.f : m;     // nothing valid yet: every exp below is exp(-inf) = 0
    const float f = expf(m_run - mref);                // rescale of what the wave has accumulated so far
    l_run *= f;
#pragma unroll
    for (int i = 0; i < QPL; ++i) av[i] *= f;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const float e = expf(sc[u] - mref);
      if (sub == 0) l_run += e;
#pragma unroll
      for (int i = 0; i < QPL; ++i) av[i] += vf[u][i] * e;
    }
    m_run = m;
  }
};

// NU0 = passes of the FIRST block of this split's range (0 = empty range): its loads are requested before the
// new q/k/v quad is waited for, so the gather / rotation / barrier runs under the cache round trip.
template <int HD, int NU0>
__device__ __forceinline__ void attention_split_body(const float* __restrict__ qkv, const float* __restrict__ qkv2,
                                                     const float* __restrict__ rope,
                                                     float* __restrict__ kc, float* __restrict__ vc, float* __restrict__ pp,
                                                     int n_head, int pos, int lo, int hi, bool last, f32x4* sqkv,
                                                     f32x4 (*wacc)[HD / 4], float* wm, float* wl) {
  constexpr int QUADS = HD / 4;
  constexpr int QPL = QUADS / 8;
  constexpr int NW = ATT1_THREADS / 64;
  const int h = blockIdx.x, row = blockIdx.y, tid = threadIdx.x;
  const int D = n_head * HD;
  const float scale = 1.0f / sqrtf((float)HD);
  const int sub = tid & 7, prow = tid >> 3;

  // new q/k/v quad of this head + rope entry, then the first block of the cache range
  VA_STAMP_DECL(stamps);
  VA_STAMP(stamps, 0);
  const int gt = min(tid, 3 * QUADS - 1);
  const int which = gt / QUADS, cq = gt % QUADS;
  f32x4 gx = reinterpret_cast<const f32x4*>(qkv)[packed_quad(row, (which * D + h * HD + cq * 4) >> 2, 3 * D)];
  const f32x4 gx2 = reinterpret_cast<const f32x4*>(qkv2 ? qkv2 : qkv)[packed_quad(row, (which * D + h * HD + cq * 4) >> 2, 3 * D)];  // see attention256_body
  const f32x4 gcs = *reinterpret_cast<const f32x4*>(rope + ((size_t)pos * (HD / 2) + cq * 2) * 2);
  __builtin_amdgcn_sched_barrier(0);
  AttentionBlock<HD, NU0 ? NU0 : 1> first;
  if constexpr (NU0 > 0) first.load(kc, vc, lo, hi, sub, prow);

  // rotate q (every split) and k (stored by the last split only), park in LDS; unconditional use of the loads
  {
    const f32x4 both = gx + gx2;
    gx = qkv2 ? both : gx;
  }
  f32x4 y;
  y[0] = gx[0] * gcs[0] - gx[1] * gcs[1];
  y[1] = gx[1] * gcs[0] + gx[0] * gcs[1];
  y[2] = gx[2] * gcs[2] - gx[3] * gcs[3];
  y[3] = gx[3] * gcs[2] + gx[2] * gcs[3];
  if (which == 2) y = gx;
  sqkv[tid < 3 * QUADS ? tid : 3 * QUADS + (tid & 63)] = y;
  if (last && tid >= QUADS && tid < 3 * QUADS)
    va_st16(reinterpret_cast<f32x4*>((which == 1 ? kc : vc) + (size_t)pos * HD) + cq, y);
  __syncthreads();
  f32x4 qf[QPL];
#pragma unroll
  for (int i = 0; i < QPL; ++i) qf[i] = sqkv[sub + 8 * i];

  float m_run = -INFINITY, l_run = 0.f;
  f32x4 av[QPL];
#pragma unroll
  for (int i = 0; i < QPL; ++i) av[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float snew = -INFINITY;
  if (last) {   // score of the new position, computed by every 8-lane group; seeds the running max
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < QPL; ++i) {
      const f32x4 kn = sqkv[QUADS + sub + 8 * i];
#pragma unroll
      for (int c = 0; c < 4; ++c) d = fmaf(qf[i][c], kn[c], d);
    }
    d += va_dpp<VA_DPP_XOR1>(d);
    d += va_dpp<VA_DPP_XOR2>(d);
    d += va_dpp<VA_DPP_HALF_MIRROR>(d);
    snew = d * scale;
    m_run = snew;
  }
  if constexpr (NU0 > 0) first.consume(lo, hi, qf, scale, sub, prow, m_run, l_run, av);
  if constexpr (NU0 == 4) {   // ranges longer than one block (cache > 256 * n_split positions)
    for (int b0 = lo + 256; b0 < hi; b0 += 256) {
      switch (min(4, (hi - b0 + 63) >> 6)) {
        case 1: { AttentionBlock<HD, 1> blk; blk.load(kc, vc, b0, hi, sub, prow); blk.consume(b0, hi, qf, scale, sub, prow, m_run, l_run, av); break; }
        case 2: { AttentionBlock<HD, 2> blk; blk.load(kc, vc, b0, hi, sub, prow); blk.consume(b0, hi, qf, scale, sub, prow, m_run, l_run, av); break; }
        case 3: { AttentionBlock<HD, 3> blk; blk.load(kc, vc, b0, hi, sub, prow); blk.consume(b0, hi, qf, scale, sub, prow, m_run, l_run, av); break; }
        default: { AttentionBlock<HD, 4> blk; blk.load(kc, vc, b0, hi, sub, prow); blk.consume(b0, hi, qf, scale, sub, prow, m_run, l_run, av); break; }
      }
    }
  }
  l_run = wave_sum(l_run);
#pragma unroll
  for (int i = 0; i < QPL; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v = av[i][c];
      v += va_dpp<VA_DPP_ROR8>(v);
      v += va_xor16(v);
      v += va_xor32(v);
      av[i][c] = v;
    }
  const int wv = tid >> 6, lane = tid & 63;
  if (lane < 8) {
#pragma unroll
    for (int i = 0; i < QPL; ++i) wacc[wv][lane + 8 * i] = av[i];
  }
  if (lane == 0) { wm[wv] = m_run; wl[wv] = l_run; }
  __syncthreads();
  if (tid < QUADS) {
    float M = wm[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) M = fmaxf(M, wm[w]);
    const float mref = (M == -INFINITY) ? 0.f : M;
    const float en = last ? expf(snew - mref) : 0.f;
    float denom = en;
    f32x4 o = last ? sqkv[2 * QUADS + tid] * en : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float f = expf(wm[w] - mref);
      denom += f * wl[w];
      o += wacc[w][tid] * f;
    }
    va_st16(reinterpret_cast<f32x4*>(pp) + tid, o);      // write-through: the last split to arrive may merge them inside this launch
    if (tid == 0) { va_st4(pp + HD, M); va_st4(pp + HD + 1, denom); }
  }
}

// merge of the n_split (<= 8) partials of one (row, head) by thread tid < HD / 4.  FRESH: the partials were written by other workgroups
// of the SAME launch (write-through): read them past the caches (sc1) — all of them requested before the one wait.
template <int HD, bool FRESH>
__device__ __forceinline__ void attention_combine(float pscale, const float* __restrict__ pp, float* __restrict__ out, uint16_t* __restrict__ outp,
                                                  int n_head, int n_split, int h, int row, int tid) {
  float Mz[8], dz[8];
  f32x4 oz[8];
#pragma unroll
  for (int z = 0; z < 8; ++z) {
    const float* pz = pp + min(z, n_split - 1) * ATT_PART_STRIDE(HD);      // slots past the last split re-read it and are not used
    if constexpr (FRESH) {
      asm volatile("global_load_dword %0, %1, off sc1" : "=&v"(Mz[z]) : "v"(pz + HD) : "memory");
      asm volatile("global_load_dword %0, %1, off sc1" : "=&v"(dz[z]) : "v"(pz + HD + 1) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(oz[z]) : "v"(pz + 4 * tid) : "memory");
    } else {
      Mz[z] = pz[HD];
      dz[z] = pz[HD + 1];
      oz[z] = *reinterpret_cast<const f32x4*>(pz + 4 * tid);
    }
  }
  if constexpr (FRESH) {
    // (the outputs above are early-clobber asm results the compiler cannot track in flight: one explicit wait, tied to every register)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int z = 0; z < 8; ++z) asm volatile("" : "+v"(Mz[z]), "+v"(dz[z]), "+v"(oz[z]));
  }
  float M = -INFINITY;
#pragma unroll
  for (int z = 0; z < 8; ++z)
    if (z < n_split) M = fmaxf(M, Mz[z]);                                  // finite: the last split holds the new position
  float denom = 0.f;
  f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int z = 0; z < 8; ++z)
    if (z < n_split) {
      const float f = expf(Mz[z] - M);
      denom += f * dz[z];
      o += oz[z] * f;
    }
  o *= 1.0f / denom;
  const int D = n_head * HD;
  va_st16(reinterpret_cast<f32x4*>(out) + packed_quad(row, (h * HD) / 4 + tid, D), o);
  if (outp) store_split4(outp, row, h * HD + 4 * tid, D, o * pscale);
}

template <int HD>
__global__ __launch_bounds__(ATT1_THREADS) void attention_split_kernel(
    const int32_t* __restrict__ pos_dev, float* __restrict__ kcache, float* __restrict__ vcache, const float* __restrict__ qkv,
    const float* __restrict__ qkv2, const float* __restrict__ rope, int n_head, int max_len, int pos_host,
    float* __restrict__ part, uint32_t* __restrict__ arrivals, float* __restrict__ out, uint16_t* __restrict__ outp, float pscale) {
  constexpr int QUADS = HD / 4;
  __shared__ f32x4 sqkv[3 * QUADS + 64];
  __shared__ f32x4 wacc[ATT1_THREADS / 64][QUADS];
  __shared__ float wm[ATT1_THREADS / 64], wl[ATT1_THREADS / 64];
  __shared__ int s_last;
  const int h = blockIdx.x, row = blockIdx.y, z = blockIdx.z, n_split = gridDim.z;
  const int pos = pos_dev ? pos_dev[0] : pos_host;   // cache holds [0, pos)
  const bool last = z == n_split - 1;                // owner of the new position
  const int per = (pos + n_split * 64 - 1) / (n_split * 64) * 64;
  const int lo = z * per, hi = min(lo + per, pos);
  float* kc = kcache + ((size_t)row * n_head + h) * (size_t)max_len * HD;
  float* vc = vcache + ((size_t)row * n_head + h) * (size_t)max_len * HD;
  float* pp = part + (((size_t)row * n_head + h) * n_split + z) * ATT_PART_STRIDE(HD);
  switch (hi > lo ? min(4, (hi - lo + 63) >> 6) : 0) {
    case 0: attention_split_body<HD, 0>(qkv, qkv2, rope, kc, vc, pp, n_head, pos, lo, hi, last, sqkv, wacc, wm, wl); break;
    case 1: attention_split_body<HD, 1>(qkv, qkv2, rope, kc, vc, pp, n_head, pos, lo, hi, last, sqkv, wacc, wm, wl); break;
    case 2: attention_split_body<HD, 2>(qkv, qkv2, rope, kc, vc, pp, n_head, pos, lo, hi, last, sqkv, wacc, wm, wl); break;
    case 3: attention_split_body<HD, 3>(qkv, qkv2, rope, kc, vc, pp, n_head, pos, lo, hi, last, sqkv, wacc, wm, wl); break;
    default: attention_split_body<HD, 4>(qkv, qkv2, rope, kc, vc, pp, n_head, pos, lo, hi, last, sqkv, wacc, wm, wl); break;
  }
  // The merge inside this launch (arrivals != nullptr: one zero-initialised word per (row, head), left at zero): the partial of this
  // split is out write-through and drained, the (row, head)'s arrival count moves (device-scope atomic), and the LAST of its n_split
  // workgroups to arrive reads all partials past the caches and does what attention_combine_kernel does — same sums, same order.
  // One launch and one kernel boundary less per layer wherever the range is split (long caches with few rows: configs[3]).
  if (arrivals) {
    if (threadIdx.x < 64) {      // wave 0 holds the threads that stored the partial
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (threadIdx.x == 0) {
        uint32_t* cnt = arrivals + (size_t)row * n_head + h;
        const uint32_t n = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = n == (uint32_t)n_split - 1u;
        if (s_last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();
    if (s_last && threadIdx.x < QUADS)
      attention_combine<HD, true>(pscale, part + ((size_t)row * n_head + h) * n_split * ATT_PART_STRIDE(HD), out, outp, n_head, n_split, h, row, threadIdx.x);
  }
}

template <int HD>
__global__ __launch_bounds__(64) void attention_combine_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                               uint16_t* __restrict__ outp, int n_head, int n_split, float pscale) {
  constexpr int QUADS = HD / 4;
  const int h = blockIdx.x, row = blockIdx.y, tid = threadIdx.x;
  if (tid >= QUADS) return;
  attention_combine<HD, false>(pscale, part + ((size_t)row * n_head + h) * n_split * ATT_PART_STRIDE(HD), out, outp, n_head, n_split, h, row, tid);
}

// rope(q, k) + K/V append for every (row, head, position) of a teacher-forced chunk; q is rotated in place
template <int HD, int KVT = 0>
__global__ __launch_bounds__(128) void rope_append_kernel(float* __restrict__ qkv, const float* __restrict__ rope,
                                                          float* __restrict__ kcache, float* __restrict__ vcache, int n_head,
                                                          int max_len, int p0, int rows16) {
  constexpr int QUADS = HD / 4;
  const int h = blockIdx.x, row = blockIdx.y, pos = p0 + (int)blockIdx.z, tid = threadIdx.x;
  if (tid >= 3 * QUADS) return;
  const int D = n_head * HD;
  const int vrow = (int)blockIdx.z * rows16 + row;
  const int which = tid / QUADS, cq = tid % QUADS;
  const int col = which * D + h * HD + cq * 4;
  f32x4* src = reinterpret_cast<f32x4*>(qkv) + packed_quad(vrow, col >> 2, 3 * D);
  f32x4 x = *src;
  if (which < 2) {
    const f32x4 cs = *reinterpret_cast<const f32x4*>(rope + ((size_t)pos * (HD / 2) + cq * 2) * 2);
    f32x4 y;
    y[0] = x[0] * cs[0] - x[1] * cs[1];
    y[1] = x[1] * cs[0] + x[0] * cs[1];
    y[2] = x[2] * cs[2] - x[3] * cs[3];
    y[3] = x[3] * cs[2] + x[2] * cs[3];
    x = y;
  }
  const size_t cbase = (((size_t)row * n_head + h) * (size_t)max_len + pos) * HD;
  if (which == 0) *src = x;
  if constexpr (KVT != 0) {
    using KV = KvT<KVT>;
    if (which == 1) reinterpret_cast<typename KV::Q*>(reinterpret_cast<typename KV::E*>(kcache) + cbase)[cq] = KV::narrow(x);
    if (which == 2) reinterpret_cast<typename KV::Q*>(reinterpret_cast<typename KV::E*>(vcache) + cbase)[cq] = KV::narrow(x);
  } else {
    if (which == 1) reinterpret_cast<f32x4*>(kcache + cbase)[cq] = x;
    if (which == 2) reinterpret_cast<f32x4*>(vcache + cbase)[cq] = x;
  }
}

// ---------------------------------------------------------------------------------------------------
// Prefill attention on the matrix cores (round 2): a teacher-forced chunk of n positions (the prompt of the sliding-window
// caller) attends causally over the cache.  The per-(row, head, position) workgroups of attention_step_kernel's prefill mode
// re-read the K / V rows of a (row, head) once per query (191 us per layer for 166 positions x 16 rows); here a workgroup owns 64
// consecutive queries of one (row, head), stages 64-key blocks of K and V in LDS once and every wave runs 16 queries through
//     S^T (16 keys x 16 queries) = K_tile . Q^T     24 x v_mfma_f32_16x16x4_f32   (exact fp32 FMA chains, like the reference's fp32)
//     online softmax per query (running max / sum; 4 lanes per query, two shuffles)
//     O (16 queries x 96) += P . V_tile             24 x v_mfma_f32_16x16x4_f32
// Operand maps are chosen so that nothing is transposed: the k index of the first product is (24 g + s) for MFMA step s and lane
// group g (a lane reads 24 CONSECUTIVE floats of its K row: 6 ds_read_b128), and key (4 g + s) of the second product is the
// accumulator register s the lane already holds.  q was rotated in place and k / v appended by rope_append_kernel.
#define APF_Q 64
#define APF_STRIDE 100      // floats per staged K / V row: 16 keys x 4-bank reads land on 64 distinct banks (36 k mod 64)
template <int HD, int KVT = 0>
__global__ __launch_bounds__(256) void attention_prefill_kernel(const float* __restrict__ qkv, const float* __restrict__ kcache,
                                                                const float* __restrict__ vcache, float* __restrict__ out,
                                                                uint16_t* __restrict__ outp, int n_head, int max_len, int p0, int n_pos,
                                                                int rows16, float pscale) {
  static_assert(HD == 96, "24 k-steps of 4");
  constexpr int KS = HD / 4;        // 24 MFMA steps per S^T tile
  constexpr int DT = HD / 16;       // 6 output column tiles
  __shared__ __attribute__((aligned(16))) float Ks[64 * APF_STRIDE];
  __shared__ __attribute__((aligned(16))) float Vs[64 * APF_STRIDE];
  const int h = blockIdx.x, row = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int D = n_head * HD;
  const int q0 = (int)blockIdx.z * APF_Q + wv * 16;            // first query (index in the chunk) of this wave
  const int ql = lane & 15, g = lane >> 4;
  const int qi = q0 + ql;                                       // this lane's query as B-operand column / softmax owner
  const int qpos = p0 + qi;
  const float scale = 1.0f / sqrtf((float)HD);
  const size_t kvoff = ((size_t)row * n_head + h) * (size_t)max_len * HD;
  const float* kc = reinterpret_cast<const float*>(reinterpret_cast<const typename KvT<KVT>::E*>(kcache) + kvoff);
  const float* vc = reinterpret_cast<const float*>(reinterpret_cast<const typename KvT<KVT>::E*>(vcache) + kvoff);
  // Q^T operand: lane (q, g) holds Q[q][24 g + s], s = 0..23 (queries past the chunk read the last valid row; never stored)
  float qreg[KS];
  {
    const int qc = min(qi, n_pos - 1);
    const int vrow = qc * rows16 + row;
#pragma unroll
    for (int c = 0; c < KS / 4; ++c) {
      const f32x4 t = reinterpret_cast<const f32x4*>(qkv)[packed_quad(vrow, (h * HD + 24 * g) / 4 + c, 3 * D)];
      qreg[4 * c] = t[0]; qreg[4 * c + 1] = t[1]; qreg[4 * c + 2] = t[2]; qreg[4 * c + 3] = t[3];
    }
  }
  f32x4 o[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t) o[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, l = 0.f;
  const int last_q = p0 + min((int)blockIdx.z * APF_Q + APF_Q - 1, n_pos - 1);     // last key position any query of the block sees
  const int wave_last = p0 + min(q0 + 15, n_pos - 1);
  for (int kb = 0; kb <= last_q; kb += 64) {
    __syncthreads();                                                              // previous block fully consumed
    for (int u = tid; u < 64 * (HD / 4); u += 256) {
      const int j = u / (HD / 4), c = u % (HD / 4);
      const int kp = min(kb + j, last_q);
      if constexpr (KVT != 0) {      // fp16 / fp8 cache: widened on the way into the fp32 staging (the products stay exact-fp32 MFMAs on the stored values)
        using KV = KvT<KVT>;
        *reinterpret_cast<f32x4*>(Ks + j * APF_STRIDE + 4 * c) = KV::widen(reinterpret_cast<const typename KV::Q*>(reinterpret_cast<const typename KV::E*>(kc) + (size_t)kp * HD)[c]);
        *reinterpret_cast<f32x4*>(Vs + j * APF_STRIDE + 4 * c) = KV::widen(reinterpret_cast<const typename KV::Q*>(reinterpret_cast<const typename KV::E*>(vc) + (size_t)kp * HD)[c]);
      } else {
        *reinterpret_cast<f32x4*>(Ks + j * APF_STRIDE + 4 * c) = reinterpret_cast<const f32x4*>(kc + (size_t)kp * HD)[c];
        *reinterpret_cast<f32x4*>(Vs + j * APF_STRIDE + 4 * c) = reinterpret_cast<const f32x4*>(vc + (size_t)kp * HD)[c];
      }
    }
    __syncthreads();
    if (q0 < n_pos) {
#pragma unroll 1
      for (int t = 0; t < 4; ++t) {
        const int key0 = kb + 16 * t;
        if (key0 > wave_last) break;                                              // causal: no query of this wave sees the tile
        // ---- S^T tile
        float ak[KS];
#pragma unroll
        for (int c = 0; c < KS / 4; ++c) {
          const f32x4 v4 = *reinterpret_cast<const f32x4*>(Ks + (16 * t + ql) * APF_STRIDE + 24 * g + 4 * c);
          ak[4 * c] = v4[0]; ak[4 * c + 1] = v4[1]; ak[4 * c + 2] = v4[2]; ak[4 * c + 3] = v4[3];
        }
        f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) st = __builtin_amdgcn_mfma_f32_16x16x4f32(ak[s], qreg[s], st, 0, 0, 0);
        // lane (q = ql, g) holds S[q][key0 + 4 g + r]
        float sc[4];
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float x = st[r];
          sc[r] = (key0 + 4 * g + r <= qpos) ? x * scale : -INFINITY;
          tmax = fmaxf(tmax, sc[r]);
        }
        tmax = fmaxf(tmax, va_xor16(tmax));
        tmax = fmaxf(tmax, va_xor32(tmax));
        const float mn = fmaxf(m, tmax);                 // finite from the first tile on: key 0 is visible to every query
        const float f = expf(m - mn);
        float p[4], rs = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { p[r] = expf(sc[r] - mn); rs += p[r]; }
        rs += va_xor16(rs);
        rs += va_xor32(rs);
        l = l * f + rs;
        m = mn;
        // ---- rescale O (lane holds O[q' = 4 g + r][d]) by the factor of query q', then O += P . V
        float fr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) fr[r] = __shfl(f, 4 * g + r, 64);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r) o[dt][r] *= fr[r];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const float* vr = Vs + (16 * t + 4 * g + s) * APF_STRIDE + ql;
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(p[s], vr[16 * dt], o[dt], 0, 0, 0);
        }
      }
    }
  }
  __syncthreads();                                                                 // K / V staging is free: reuse Ks as the output stage
  if (q0 >= n_pos) return;
  float li[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) li[r] = 1.0f / __shfl(l, 4 * g + r, 64);
  float* os = Ks + (size_t)wv * 16 * APF_STRIDE;
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 4; ++r) os[(4 * g + r) * APF_STRIDE + 16 * dt + ql] = o[dt][r] * li[r];
  // wave-local exchange through LDS (each wave reads only what it wrote): 16 queries x 24 quads -> packed + split rows
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int it = 0; it < (16 * (HD / 4)) / 64; ++it) {
    const int idx = lane + 64 * it, qq = idx / (HD / 4), cq = idx % (HD / 4);
    if (q0 + qq >= n_pos) continue;
    const f32x4 v4 = *reinterpret_cast<const f32x4*>(os + qq * APF_STRIDE + 4 * cq);
    const int vrow = (q0 + qq) * rows16 + row;
    reinterpret_cast<f32x4*>(out)[packed_quad(vrow, (h * HD) / 4 + cq, D)] = v4;
    if (outp) store_split4(outp, vrow, h * HD + 4 * cq, D, v4 * pscale);
  }
}

VA_STAMP_SETTER(vaura_stamps_set_attention)

int va_attention_splits(int rows, int n_head, int max_len) {
  if (max_len <= 256) return 1;
  const int pairs = rows * n_head;
  return pairs >= 256 ? 1 : max(1, min(8, 256 / pairs));
}

int va_launch_attention(const float* qkv, const float* qkv2, const float* rope, float* kc, float* vc, float* out,
                        uint16_t* outp, int rows, int n_head, int head_dim, int max_len, const int32_t* pos_dev, int pos_host,
                        float* part, int n_split, hipStream_t s, uint32_t* arrivals, float pscale, int kv_half) {
  if (!qkv || !rope || !kc || !vc || !out || rows <= 0 || n_head <= 0) return VAURA_ERR_ARG;
  if (head_dim != 96) return VAURA_ERR_SHAPE;
  // fp16 K / V cache: the single-round-trip kernel only (every 2.56 s configuration); long caches / range splits keep fp32
  if (kv_half && (max_len > 256 || (part && n_split > 1))) return VAURA_ERR_SHAPE;
  if (part && n_split > 1) {   // few (row, head) pairs over a long cache: split the range, then combine
    if (n_split > 8) return VAURA_ERR_ARG;
    // arrivals (rows * n_head zeroed words, e.g. the decoder's ws_sync + 512): the last split to arrive merges; debug flag bit 19: own launch
    if (va_debug_flags_get() & 0x80000u) arrivals = nullptr;
    VA_LAUNCH(attention_split_kernel<96>, dim3(n_head, rows, n_split), dim3(ATT1_THREADS), 0, s, pos_dev, kc, vc, qkv, qkv2,
              rope, n_head, max_len, pos_host, part, arrivals, out, outp, pscale);
    if (!arrivals) VA_LAUNCH(attention_combine_kernel<96>, dim3(n_head, rows), dim3(64), 0, s, (const float*)part, out, outp, n_head, n_split, pscale);
    return 0;
  }
  if (max_len <= 256) {   // static per descriptor (the step graph is captured once): single-round-trip kernel
    if (kv_half == 1)
      VA_LAUNCH((attention_step256_kernel<96, 1>), dim3(n_head, rows), dim3(ATT1_THREADS), 0, s, pos_dev, kc, vc, qkv, qkv2, rope,
                n_head, max_len, pos_host, out, outp, pscale);
    else if (kv_half == 2)
      VA_LAUNCH((attention_step256_kernel<96, 2>), dim3(n_head, rows), dim3(ATT1_THREADS), 0, s, pos_dev, kc, vc, qkv, qkv2, rope,
                n_head, max_len, pos_host, out, outp, pscale);
    else
      VA_LAUNCH(attention_step256_kernel<96>, dim3(n_head, rows), dim3(ATT1_THREADS), 0, s, pos_dev, kc, vc, qkv, qkv2, rope,
                n_head, max_len, pos_host, out, outp, pscale);
    return 0;
  }
  const size_t smem = sizeof(float) * (size_t)(3 * 96 + 4 * 96 + 8 + max_len + 4);
  VA_LAUNCH(attention_step_kernel<96>, dim3(n_head, rows), dim3(ATT_THREADS), smem, s, qkv, qkv2, rope, kc, vc, out, outp,
            n_head, max_len, pos_dev, pos_host, 0, pscale);
  return 0;
}

#ifdef VAURA_EXPERIMENT_ENGINES
// measured-negative engine (attention + wo as one launch; DESIGN_HISTORY.md round 4): experiment builds only
#include "experiments/attn_wo.h"
#endif

int va_launch_rope_append(const vaura_decoder* d, int layer, int p0, int n_pos, hipStream_t s) {
  const int H = d->dims.n_head, hd = d->dims.d_model / H;
  if (hd != 96) return VAURA_ERR_SHAPE;
  const size_t kv_layer = (size_t)d->rows * H * (size_t)d->max_len * hd;
  if (d->kv_dtype == 1 || d->kv_dtype == 2) {      // fp16 / fp8 cache: the layer offset in elements of that type
    if (d->kv_dtype == 1)
      VA_LAUNCH((rope_append_kernel<96, 1>), dim3(H, d->rows, n_pos), dim3(128), 0, s, d->ws_qkv, d->rope, va_kv_layer(d, d->kcache, layer),
                va_kv_layer(d, d->vcache, layer), H, d->max_len, p0, (d->rows + 15) / 16 * 16);
    else
      VA_LAUNCH((rope_append_kernel<96, 2>), dim3(H, d->rows, n_pos), dim3(128), 0, s, d->ws_qkv, d->rope, va_kv_layer(d, d->kcache, layer),
                va_kv_layer(d, d->vcache, layer), H, d->max_len, p0, (d->rows + 15) / 16 * 16);
    return 0;
  }
  VA_LAUNCH(rope_append_kernel<96>, dim3(H, d->rows, n_pos), dim3(128), 0, s, d->ws_qkv, d->rope, d->kcache + layer * kv_layer,
            d->vcache + layer * kv_layer, H, d->max_len, p0, (d->rows + 15) / 16 * 16);
  return 0;
}

extern unsigned va_debug_flags;   // gemv3.hip; bit 4: the per-position prefill attention (A/B of the MFMA kernel)
int va_launch_attention_prefill(const vaura_decoder* d, int layer, int p0, int n_pos, hipStream_t s) {
  const int H = d->dims.n_head, hd = d->dims.d_model / H;
  if (hd != 96) return VAURA_ERR_SHAPE;
  const size_t kv_layer = (size_t)d->rows * H * (size_t)d->max_len * hd;
  if (d->kv_dtype == 1 || d->kv_dtype == 2) {
    const dim3 grid(H, d->rows, (n_pos + APF_Q - 1) / APF_Q);
    if (d->kv_dtype == 1)
      VA_LAUNCH((attention_prefill_kernel<96, 1>), grid, dim3(256), 0, s, (const float*)d->ws_qkv, (const float*)va_kv_layer(d, d->kcache, layer),
                (const float*)va_kv_layer(d, d->vcache, layer), d->ws_attn, d->ws_attn_split, H, d->max_len, p0, n_pos, (d->rows + 15) / 16 * 16,
                ldexpf(1.f, -d->plane_shift));
    else
      VA_LAUNCH((attention_prefill_kernel<96, 2>), grid, dim3(256), 0, s, (const float*)d->ws_qkv, (const float*)va_kv_layer(d, d->kcache, layer),
                (const float*)va_kv_layer(d, d->vcache, layer), d->ws_attn, d->ws_attn_split, H, d->max_len, p0, n_pos, (d->rows + 15) / 16 * 16,
                ldexpf(1.f, -d->plane_shift));
    return 0;
  }
  if (!(va_debug_flags & 16u)) {
    VA_LAUNCH(attention_prefill_kernel<96>, dim3(H, d->rows, (n_pos + APF_Q - 1) / APF_Q), dim3(256), 0, s, (const float*)d->ws_qkv,
              (const float*)(d->kcache + layer * kv_layer), (const float*)(d->vcache + layer * kv_layer), d->ws_attn, d->ws_attn_split, H,
              d->max_len, p0, n_pos, (d->rows + 15) / 16 * 16, ldexpf(1.f, -d->plane_shift));
    return 0;
  }
  const size_t smem = sizeof(float) * (size_t)(3 * 96 + 4 * 96 + 8 + d->max_len + 4);
  VA_LAUNCH(attention_step_kernel<96>, dim3(H, d->rows, n_pos), dim3(ATT_THREADS), smem, s, d->ws_qkv, (const float*)nullptr, d->rope,
            d->kcache + layer * kv_layer, d->vcache + layer * kv_layer, d->ws_attn, d->ws_attn_split, H, d->max_len, nullptr,
            p0, (d->rows + 15) / 16 * 16, ldexpf(1.f, -d->plane_shift));
  return 0;
}

extern "C" int vaura_attention_step(const float* qkv, const float* rope, float* kcache, float* vcache, float* out,
                                    int rows, int n_head, int head_dim, int max_len, int pos, vaura_stream_t s) {
  if (pos < 0 || pos >= max_len) return VAURA_ERR_ARG;
  return va_launch_attention(qkv, nullptr, rope, kcache, vcache, out, nullptr, rows, n_head, head_dim, max_len, nullptr, pos,
                             nullptr, 1, as_stream(s), nullptr, 1.f, 0);
}

extern "C" int vaura_attention_splits(int rows, int n_head, int max_len) { return va_attention_splits(rows, n_head, max_len); }

extern "C" int vaura_attention_step_split(const float* qkv, const float* rope, float* kcache, float* vcache, float* out,
                                          float* part, int rows, int n_head, int head_dim, int max_len, int pos, int n_split,
                                          vaura_stream_t s) {
  if (pos < 0 || pos >= max_len || !part || n_split < 2 || n_split > 8) return VAURA_ERR_ARG;
  return va_launch_attention(qkv, nullptr, rope, kcache, vcache, out, nullptr, rows, n_head, head_dim, max_len, nullptr, pos,
                             part, n_split, as_stream(s), nullptr, 1.f, 0);
}
