// 2-D (N x K) decomposed decode GEMV with an in-launch split-K reduce-scatter.  bf16 weights, exact
// hi/mid/lo activation planes (gemv3_kernel.h), rows <= 16.
//
// Why: every workgroup of an N-only split needs the whole activation (16 rows x K x 6 B = 147 KB at
// K = 1536, 393 KB at K = 4096).  Measured on MI355X the per-XCD L2s deliver ~7 TB/s in aggregate for
// that pattern — no more than HBM — so re-reading the activation 288x per GEMV costs 3x the weight
// stream itself.  Here a workgroup owns an (8*T tiles) x (KS k-groups) rectangle: it stages only its
// K slice of the planes in LDS (18-24 KB, shared by its 8 waves), streams its weight rectangle once,
// and publishes one 1-KiB partial tile per 16 output columns.  The S workgroups of a column block then
// reduce-scatter: each sums (fixed k order => deterministic) the S partials of ITS share of the tiles
// and runs the epilogue for them.
//
// Hand-off protocol (MI355X_MICROARCH.md "Valid forms", cdna_hip_programming.md Guideline 16 R1):
//   producer: partial tiles stored write-through (sc1), every storing wave drains vmcnt(0), workgroup
//             barrier, ONE lane does an agent-scope relaxed fetch_add on the block's arrival counter;
//   consumer: ONE lane of the reducing wave polls that counter (relaxed agent load, s_sleep), the same
//             wave then loads the partials with sc1 loads (L1 bypass).  No placement assumption.
//   epochs:   counters are monotonic; target = (value returned by the own fetch_add / S + 1) * S, so no
//             per-call reset and no host-side epoch argument (graph-replay safe).
// Residency: grids are <= 288 workgroups of 512 threads at >= 2 workgroups per CU (launch bounds), and
// the S partners of a block have adjacent block ids, so every waited-for workgroup is resident or
// becomes resident independently of the waiters.  Spins are bounded; a timeout raises *timeout.
#pragma once
#include "gemv3_kernel.h"

struct Gemv4Args {
  const void* W;
  const uint16_t* XP;
  const float* ss_in;
  int n_ss_in;
  const float* res;
  float* out;
  uint16_t* outp;
  const float* gain_out;
  float* ss_out;
  float* scratch;       // [n_tiles][S][64] f32x4 partial tiles
  uint32_t* counters;   // [n_blocks] arrival counters of this launch slot
  uint32_t* timeout;    // set to 1 if a bounded spin gave up
  uint32_t scratch_bytes;
  int rows, N;
  float eps;
  int k_total;
};

#define G4_WAVES 8
#define G4_SC1 16  // aux bit: sc1 (system-coherent level 1: write-through store / L1-bypassing load)

template <int KS, int S, int T, int EPI, bool NORM>
__global__ __launch_bounds__(G4_WAVES * 64, 4) void gemv4_kernel(Gemv4Args a) {
  constexpr int KG = KS * S;          // k-groups of the whole K
  constexpr int K = 32 * KG;
  constexpr int TB = G4_WAVES * T;    // tiles per column block
  constexpr int UNIT = (EPI == E3_SWIGLU) ? 2 : 1;   // tiles reduced together by one wave
  constexpr int UNITS = TB / UNIT;
  static_assert(T % UNIT == 0, "SwiGLU pairs must not straddle waves");
  __shared__ u32x4 xs[3][KS][64];
  __shared__ uint32_t s_target;

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int m = lane & 15, q = lane >> 4;
  const int nb = blockIdx.x / S, ks = blockIdx.x % S;
  const u32x4* Wp = reinterpret_cast<const u32x4*>(a.W);

  // ---- 1. this wave's weight rectangle: T tiles x KS k-groups, streamed once (non-temporal)
  u32x4 wb[T][KS];
#pragma unroll
  for (int g = 0; g < KS; ++g)
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const size_t kg = (size_t)(nb * TB + wv * T + t) * KG + (size_t)(ks * KS + g);
      wb[t][g] = __builtin_nontemporal_load(Wp + kg * 64 + lane);
    }
  // ---- 2. the K slice of the activation planes -> LDS (each plane's slice is KS contiguous KiB)
  {
    const u32x4* Xp = reinterpret_cast<const u32x4*>(a.XP);
    constexpr int CH = 3 * KS * 64;
#pragma unroll
    for (int i = 0; i < (CH + G4_WAVES * 64 - 1) / (G4_WAVES * 64); ++i) {
      const int c = tid + i * G4_WAVES * 64;
      if (c < CH) {
        const int p = c / (KS * 64), r = c % (KS * 64);
        (&xs[0][0][0])[c] = Xp[split_index16(0, p, ks * KS * 4, 0, K) + r];
      }
    }
  }
  __syncthreads();

  // ---- 3. three bf16 MFMAs per (tile, k-group); one accumulator per plane
  f32x4 acc[T][3];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int p = 0; p < 3; ++p) acc[t][p] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < KS; ++g) {
    bf16x8 xf[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) xf[p] = __builtin_bit_cast(bf16x8, xs[p][g][lane]);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const bf16x8 wf = __builtin_bit_cast(bf16x8, wb[t][g]);
#pragma unroll
      for (int p = 0; p < 3; ++p) acc[t][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[p], acc[t][p], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- 4. publish the partial tiles (write-through), arrive
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.scratch, 0, (int)a.scratch_bytes, 0x00020000);
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const f32x4 part = (acc[t][2] + acc[t][1]) + acc[t][0];
    const uint32_t off = (uint32_t)((((nb * TB + wv * T + t) * S + ks) * 64 + lane) * 16);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, part), rs, off, 0, G4_SC1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const uint32_t prev = __hip_atomic_fetch_add(a.counters + nb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_target = (prev / S + 1u) * S;
  }
  __syncthreads();

  // ---- 5. reduce-scatter: unit u of the block is reduced by workgroup (u % S), wave (u / S)
  const int unit = wv * S + ks;
  if (unit >= UNITS) return;
  float ssp = 0.f;
  if constexpr (NORM) {   // rsqrt(mean(x^2)+eps) of the input rows from the producer's per-tile partial sums
    for (int i = q; i < a.n_ss_in; i += 4) ssp += a.ss_in[i * 16 + m];
  }
  {
    const uint32_t target = s_target;
    uint32_t spins = 0;
    while (__hip_atomic_load(a.counters + nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 24)) { if (lane == 0) *a.timeout = 1u; break; }
    }
  }
  f32x4 v[UNIT];
#pragma unroll
  for (int j = 0; j < UNIT; ++j) {
    const int tile = nb * TB + unit * UNIT + j;
    f32x4 sacc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s2 = 0; s2 < S; ++s2) {
      const uint32_t off = (uint32_t)(((tile * S + s2) * 64 + lane) * 16);
      const u32x4 pv = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, G4_SC1);
      sacc = (s2 == 0) ? __builtin_bit_cast(f32x4, pv) : sacc + __builtin_bit_cast(f32x4, pv);
    }
    v[j] = sacc;
  }
  float rinv = 1.f;
  if constexpr (NORM) {
    ssp += __shfl_xor(ssp, 16, 64);
    ssp += __shfl_xor(ssp, 32, 64);
    rinv = 1.0f / sqrtf(ssp * (1.0f / (float)a.k_total) + a.eps);
  }
#pragma unroll
  for (int j = 0; j < UNIT; ++j) v[j] *= rinv;

  // ---- 6. epilogue: lane holds out[row m][16*tile + 4q .. +3]
  const int row = m;
  if constexpr (EPI == E3_SWIGLU) {
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = silu3_f(v[0][r]) * v[1][r];
    const int ft = nb * (TB / 2) + unit;   // tile of the ffn dimension
    if (a.out) reinterpret_cast<f32x4*>(a.out)[(size_t)ft * 64 + lane] = o;
    if (a.outp) store_split4(a.outp, row, ft * 16 + 4 * q, a.N, o);
  } else {
    const int tile = nb * TB + unit;
    const int c0 = tile * 16 + 4 * q;
    if constexpr (EPI == E3_LOGITS) {
      if (row < a.rows) *reinterpret_cast<f32x4*>(a.out + (size_t)row * a.N + c0) = v[0];
    } else {
      const size_t idx = (size_t)tile * 64 + lane;
      f32x4 o = v[0];
      if constexpr (EPI == E3_RESID) o += reinterpret_cast<const f32x4*>(a.res)[idx];
      if (a.out) reinterpret_cast<f32x4*>(a.out)[idx] = o;
      if (a.ss_out) {
        float s = ((o[0] * o[0] + o[1] * o[1]) + o[2] * o[2]) + o[3] * o[3];
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (q == 0) a.ss_out[(size_t)tile * 16 + m] = s;
      }
      if (a.outp) {
        f32x4 u = o;
        if (a.gain_out) u *= *reinterpret_cast<const f32x4*>(a.gain_out + c0);
        store_split4(a.outp, row, c0, a.N, u);
      }
    }
  }
}
