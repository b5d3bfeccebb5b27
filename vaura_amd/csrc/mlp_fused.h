// SwiGLU MLP of one decoder layer as ONE launch:  ffn = silu(W1 x) * (W3 x)  ->  h += W2 ffn      (llama.py:176-177, 282)
//
// Two weight-streaming GEMVs with an all-to-all hand-off in between (every w2 workgroup needs the ffn vector of all 16 rows
// produced by all 256 w1|w3 workgroups).  As two launches the pair costs  7.9 + 0.6 (gap) + 6.6 us: each stage = a fixed ~3.5 us
// (kernel boundary, first-byte latency, reduction, drain) + its HBM stream at ~6 TB/s, and no weight byte of w2 moves while
// w1|w3 drains.  Here the 256 workgroups (one per CU, all resident) stay: after its last w1|w3 MFMA a workgroup requests its w2
// weight tiles, finishes the SwiGLU epilogue, publishes its ffn planes and waits for everybody else's; the w2 stream runs under
// that hand-off.
//
// Hand-off protocol (cdna_hip_programming.md Guideline 16 R1, MI355X_MICROARCH.md "Valid forms", row 1 of the measured table):
//   producer   every byte of the ffn planes is stored write-through (8-byte `sc1` stores, whole 128-byte lines per store
//              instruction); the ONE storing wave drains them (`s_waitcnt vmcnt(0)`), then ONE lane stores the workgroup's
//              flag word = epoch (`sc1`).
//   consumer   ONE wave polls all flag words (`sc1` loads, 1 KB per sweep) until every producer shows the epoch, then issues ONE
//              agent-scope acquire (L1 invalidate) and waits for it; the other waves wait at a workgroup barrier that wave then
//              joins; the ffn planes are then read with plain loads (nothing in this launch read them before, and they were
//              stored write-through).  One workgroup per CU (512 threads at > 128 VGPRs: a second one does not fit).
//   epoch      (sequence id << 20) | (position * 32 + layer + 1) from the device-side state: strictly increasing within a
//              sequence, different for every launch that shares the flag words, valid under graph replay; no memset node.
//   safety     results never depend on placement or timing; the spin is bounded (timeout word set, the launch completes with
//              wrong numbers instead of hanging — the host checks the word); the grid must equal the CU count, which the
//              launcher verifies (else the two-launch path is used).
#pragma once
#include "gemv3_kernel.h"

struct MlpFusedArgs {
  const void* W13;        // (2F x D) MFMA tiles, w1 / w3 interleaved per 16-row tile
  const uint16_t* XP;     // split rows of gain * h (rows x D)
  const float* ss_in;     // (1, n_ss_in, 16) partial sums of squares of h
  int n_ss_in;
  uint16_t* ffnp;         // split rows (rows x F): written by phase 1 (sc1), read by phase 2 (sc1)
  const void* W2;         // (D x F) MFMA tiles
  const float* res;       // packed rows (rows x D): h
  float* out;             // packed rows: h + W2 ffn
  uint16_t* outp;         // split rows of out * gain_out
  const float* gain_out;  // next RMSNorm gain
  float* ss_out;          // (1, D/16, 16) partial sums of squares of out
  int rows, halves;
  float eps;
  uint32_t* flags;        // [gridDim.x]
  uint32_t* tmo;          // timeout word
  const int32_t* state;   // {position, arrivals, step, sequence id}
  int layer;
};

__device__ __forceinline__ void store_split4_sc1(uint16_t* base, int row, int c0, int C, const f32x4 v) {
  uint2 hi, mid, lo;
  split3(v, hi, mid, lo);
  const int rb = row >> 4, m = row & 15, oct = c0 >> 3, half = (c0 >> 2) & 1;
  unsigned long long* p = reinterpret_cast<unsigned long long*>(base);
  auto st = [&](int plane, uint2 x) {
    __hip_atomic_store(p + split_index16(rb, plane, oct, m, C) * 2 + half, ((unsigned long long)x.y << 32) | x.x, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  };
  st(0, hi); st(1, mid); st(2, lo);
}

template <int WT>
__global__ __launch_bounds__(512) void mlp_fused_kernel(const void* __restrict__ W13q, const uint16_t* __restrict__ XPq, MlpFusedArgs a) {
  static_assert(WT == 0, "bf16 weights");
  constexpr int NW = 8, D = 1536, F = 4096;
  constexpr int G1 = 6, T1 = 2, KG1 = D / 32;          // phase 1: 6 k-groups per wave, a (w1, w3) tile pair per workgroup
  constexpr int G2 = 8, KG2 = F / 32;                  // phase 2: 8 k-group pairs per wave (row-split pair kernel)
  a.W13 = W13q;
  a.XP = XPq;
  __shared__ f32x4 red[NW][2][64];
  __shared__ float ssl[2048];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bid = blockIdx.x;
  const int lane16 = lane * 16;
  const int m = lane & 15, q = lane >> 4;
  const uint32_t epoch = ((uint32_t)a.state[3] << 20) | (uint32_t)(a.state[0] * 32 + a.layer + 1);

  // ------------------------------------------------------------------ phase 1: ffn tile `bid` = silu(w1 x) * (w3 x)
  {
    const int w = (wid + bid) % NW;
    const int tile0 = bid * T1;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.W13), 0, -16, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.XP), 0, -16, 0x00020000);
    u32x4 wb[T1][G1], xb[G1][3];
#pragma unroll
    for (int g = 0; g < G1; ++g)
#pragma unroll
      for (int t = 0; t < T1; ++t)
        wb[t][g] = __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16, (int)(((size_t)(tile0 + t) * KG1 + (size_t)(w * G1 + g)) * 1024), 2);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < G1; ++g)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        xb[g][p] = __builtin_amdgcn_raw_buffer_load_b128(xrs, lane16, (p * (D / 8) * 16 + (w * G1 + g) * 64) * 16, 0);
    float ssr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = threadIdx.x + j * 512;
      ssr[j] = (i < a.n_ss_in * 16) ? a.ss_in[i] : 0.f;
    }
    f32x4 acc[T1][3];
#pragma unroll
    for (int t = 0; t < T1; ++t)
#pragma unroll
      for (int p = 0; p < 3; ++p) acc[t][p] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < G1; ++g) {
#pragma unroll
      for (int t = 0; t < T1; ++t) {
        const bf16x8 wf = __builtin_bit_cast(bf16x8, wb[t][g]);
        mfma_group<0>(&wf, xb[g], acc[t]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < T1; ++t) red[wid][t][lane] = acc_sum<0>(acc[t]);
#pragma unroll
    for (int j = 0; j < 4; ++j) ssl[threadIdx.x + j * 512] = ssr[j];
  }

  // ------------------------------------------------------------------ phase 2 operands that depend on nothing: w2 weight tiles
  const int halves = a.halves & 3;
  const int n2 = 96 * halves;                           // workgroups with a w2 work item
  const bool has2 = bid < n2;
  const int h2 = halves == 2 ? (bid >> 3) & 1 : 0;
  const int tile2 = halves == 2 ? (bid & 7) + 8 * (bid >> 4) : bid;
  const int w2i = (wid + tile2) % NW;
  const int la = lane & 7, sb = (lane >> 3) & 1;
  u32x4 wb2[G2][2];
  auto load_w2 = [&]() {
    const __amdgpu_buffer_rsrc_t wrs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.W2), 0, -16, 0x00020000);
    const int voffw0 = (la + 16 * q) * 16 + sb * 1024;
#pragma unroll
    for (int g = 0; g < G2; ++g)
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
        wb2[g][nh] = __builtin_amdgcn_raw_buffer_load_b128(wrs2, voffw0 + nh * 128, (tile2 * KG2 + 2 * (w2i * G2 + g)) * 1024, 2);
  };
  // waves 1..7 request their w2 slice now (under the reduction, the epilogue and the hand-off); wave 0 first publishes — its
  // `vmcnt(0)` drain before the flag store would otherwise wait for these HBM loads too
  if (has2 && wid != 0) load_w2();
  __syncthreads();

  // ------------------------------------------------------------------ phase 1 epilogue (wave 0) + publish
  if (wid == 0) {
    float ssp = 0.f;
    for (int i = q; i < a.n_ss_in; i += 4) ssp += ssl[i * 16 + m];
    ssp += __shfl_xor(ssp, 16, 64);
    ssp += __shfl_xor(ssp, 32, 64);
    const float rinv = 1.0f / sqrtf(ssp * (1.0f / (float)D) + a.eps);
    f32x4 v[T1];
#pragma unroll
    for (int t = 0; t < T1; ++t) {
      f32x4 sacc = red[0][t][lane];
#pragma unroll
      for (int i = 1; i < NW; ++i) sacc += red[i][t][lane];
      v[t] = sacc * rinv;
    }
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = silu3_f(v[0][r]) * v[1][r];
    store_split4_sc1(a.ffnp, m, bid * 16 + 4 * q, F, o);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(a.flags + bid, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (has2) load_w2();
  }
  if (!has2) return;

  // ------------------------------------------------------------------ hand-off: wait for every producer's flag
  if (wid == 0) {
    const int nprod = (int)gridDim.x;
    bool ok = false;
    for (int it = 0; it < 400000 && !ok; ++it) {
      bool mine = true;
      for (int j = lane; j < nprod; j += 64)
        mine &= __hip_atomic_load(a.flags + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch;
      ok = __all(mine);
      if (!ok) __builtin_amdgcn_s_sleep(1);
    }
    if (!ok && lane == 0) atomicOr(a.tmo, 1u);
    // ONE agent-scope acquire per workgroup (buffer_inv sc1: drops this CU's L1 lines), completed before the barrier releases
    // the other waves: the plane loads below are then PLAIN loads, served and shared by the XCD's L2.  (With `sc1` loads every
    // one of the 192 consumers pulled its 196 KB through the fabric: 37 MB per hand-off, measured slower than two launches.)
    if (!(a.halves & 4)) {      // experiment switch (halves bit 2): skip the acquire
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // compiler-only: the plane loads below stay below the barrier

  // ------------------------------------------------------------------ phase 2: out tile `tile2`, rows 8 h2 .. 8 h2 + 7
  {
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(a.ffnp, 0, -16, 0x00020000);
    const int voffx = (sb * 64 + q * 16 + la + 8 * h2) * 16;
    u32x4 xb[G2][3];
#pragma unroll
    for (int g = 0; g < G2; ++g)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        xb[g][p] = __builtin_amdgcn_raw_buffer_load_b128(xrs, voffx, (p * (F / 8) * 16 + (w2i * G2 + g) * 128) * 16, 0);
    Gemv3Args e;                          // the shared residual epilogue works on a Gemv3Args view
    e.res = a.res; e.out = a.out; e.outp = a.outp; e.gain_out = a.gain_out; e.ss_out = a.ss_out; e.N = D; e.rows = a.rows; e.out2 = nullptr;
    EpiPre pre;
    pre.have = false;
    if (wid == 0 && ((lane >> 3) & 1) == h2) pre = gemv3_epilogue_prefetch<E3_RESID>(e, 0, tile2, lane);
    f32x4 acc[2][3];
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
      for (int p = 0; p < 3; ++p) acc[nh][p] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < G2; ++g) {
#pragma unroll
      for (int nh = 0; nh < 2; ++nh) {
        const bf16x8 wf = __builtin_bit_cast(bf16x8, wb2[g][nh]);
        mfma_group<0>(&wf, xb[g], acc[nh]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
      const f32x4 v = acc_sum<0>(acc[nh]);
      f32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float x = v[r];
        o[r] = x + __shfl_xor(x, 40, 64);
      }
      red[wid][nh][lane] = o;              // phase 1's use of `red` ended before the hand-off barrier
    }
    __syncthreads();
    if (wid == 0) {
      const bool mine = (m >> 3) == h2;
      const int src = (m & 7) + 16 * (q & 1);
      f32x4 v = red[0][q >> 1][src];
#pragma unroll
      for (int i = 1; i < NW; ++i) v += red[i][q >> 1][src];
      if (mine) gemv3_epilogue<1, E3_RESID>(e, 0, tile2, lane, &v, &pre);
    }
  }
}
