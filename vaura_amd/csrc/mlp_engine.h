// The MLP half of a decoder layer (llama.py:161-177, 282: h += w2( silu(w1 x) * (w3 x) ), x = rmsnorm(h)) as ONE persistent launch:
//   phase 1   w1||w3 GEMV + SwiGLU  (the arithmetic of gemv3_kernel<6, 8, 2, E3_SWIGLU, NORM>: same products, same order)
//   run-ahead every wave, as soon as its own phase-1 products are issued, requests its slice of w2's weights — half of it by
//             LDS-DMA into a ring of 1-KiB fragments (no register destination: the registers still hold phase-1 state), the rest
//             straight into the registers phase 1 has just released — so w2's weight stream runs under the SwiGLU epilogue, the
//             hand-off and the planes' round trip instead of after a kernel boundary
//   hand-off  the ffn planes (16 rows x 4096 columns x (hi, lo) fp16 = 262 KB) go from the 256 producing workgroups to the 192
//             consuming ones inside the launch: write-through (sc0 sc1) stores, drained, ONE sc1 flag word per producer carrying
//             an epoch (sequence id, position, layer: no reset between replays of a captured graph); one wave per consumer polls
//             the 256 flags (bounded, with a give-up status bit), a workgroup barrier releases the others, and every load of the
//             planes is an sc1 buffer load (MI355X_MICROARCH.md "Hand-offs measured with sc1 loads in place of the acquire", row 1)
//   phase 2   w2 GEMV + residual + next norm's gain / partial sums of squares / planes (gemv3h_kernel<8, 8, E3_RESID>: same
//             products, same order) on workgroups 0..191 = (tile, row half); workgroups 192..255 leave after publishing
// Both phases sum exactly what the two-launch path sums, in its order: results are BIT-IDENTICAL to it (tests run every golden on
// both).  One workgroup per CU (144 KB of LDS), 256 workgroups: the launcher refuses devices with fewer than 256 CUs, and the
// hand-off wait is bounded — a consumer that gives up raises VAURA_STATUS_HANDOFF_TIMEOUT and later waits return at once.
#pragma once
#include "gemv3_kernel.h"

// Run-ahead throttle (round 4, second half).  A CU serves its vector-memory requests in order: the seven waves' run-ahead requests,
// issued the moment their products are done, sit in front of wave 0's epilogue stores, and a hand-off passes only when the SLOWEST
// producer has published (stamps: publish 5.3 us behind the products where the 64 workgroups without a run-ahead need 2.9).  Holding
// the whole run-ahead until the stores are issued lost (the weights then land late); holding PART of it wins: waves 1..7 request
// Q2 quarters of their w2 slice (phase 2) / PRE3 of their three qkv k-groups (phase 3) at once and the rest when wave 0 has issued its
// stores (an LDS word).  Measured on whole loops (tools/experiment.sh lib-ab, profiles/r04_ab_mlp_engine.txt): two planes best at
// (2, 1): 211.3 -> 202.0 ms; one plane at (1, 2): 167.0 -> 161.6 ms; neighbours are 1-4 % worse, no hold at all is ablation bit 3.
// -DMLPE_Q2 / -DMLPE_PRE3 override both storages (experiment builds).
#ifndef MLPE_REL
#define MLPE_REL 0        // 1: the held part is released behind the publish (drained stores + flag) instead of behind the stores' issue
#endif
template <typename F>
__device__ __forceinline__ void va_static_for9(F&& f) {
  f(std::integral_constant<int, 0>{}); f(std::integral_constant<int, 1>{}); f(std::integral_constant<int, 2>{});
  f(std::integral_constant<int, 3>{}); f(std::integral_constant<int, 4>{}); f(std::integral_constant<int, 5>{});
  f(std::integral_constant<int, 6>{}); f(std::integral_constant<int, 7>{}); f(std::integral_constant<int, 8>{});
}
template <int WT, int RBK = 1>
struct MlpeThrottle {
#ifdef MLPE_Q2
  static constexpr int Q2_1 = MLPE_Q2;
#else
  static constexpr int Q2_1 = WT == 2 ? 2 : 1;
#endif
#ifdef MLPE_PRE3
  static constexpr int PRE3 = MLPE_PRE3;
#else
  static constexpr int PRE3 = WT == 2 ? 1 : 2;
#endif
#ifdef MLPE_PRE3U
  static constexpr int PRE3U_1 = MLPE_PRE3U;      // the same in ninths (finer experiment builds)
#else
  static constexpr int PRE3U_1 = 3 * PRE3;
#endif
  // two row blocks per pass (RBK = 2): their own optimum (-DMLPE_Q2R / -DMLPE_PRE3UR in experiment builds)
#ifdef MLPE_Q2R
  static constexpr int Q2_2 = MLPE_Q2R;
#else
  static constexpr int Q2_2 = Q2_1;
#endif
#ifdef MLPE_PRE3UR
  static constexpr int PRE3U_2 = MLPE_PRE3UR;
#else
  static constexpr int PRE3U_2 = PRE3U_1;
#endif
  static constexpr int Q2 = RBK == 2 ? Q2_2 : Q2_1;
  static constexpr int PRE3U = RBK == 2 ? PRE3U_2 : PRE3U_1;
};
struct MlpEngineArgs {
  Gemv3Args p1;            // w1||w3: W, XP (h planes), ss_in, outp (ffn planes), N = ffn_dim, rows, R = 1, eps, k_total, wscale
  Gemv3Args p2;            // w2: W, XP (= p1.outp), res / out (h), outp (h planes), gain_out, ss_out, N = d_model, wscale
  Gemv3Args p3;            // QKV instances: the NEXT layer's wqkv: W, XP (= p2.outp), ss_in (= p2.ss_out), out / out2 (K-half partials), wscale
  uint32_t* flags;         // [512]: [0, 256) phase-1 producers, [256, 448) phase-2 producers (QKV instances)
  const int32_t* state;    // device state: [0] position, [3] sequence id, [4] status bits
  int32_t* state_rw;
  int layer;
  int abl;                 // timing ablations (tools only; 1 gives wrong results): 1 = no flag wait, 4 = no run-ahead (w2's weights requested
                           // behind the hand-off barrier)
  int qlocal;              // experiment builds only (no effect, round 6): the qkv phase's work items of head h on XCD h % 8 + plain stores
  int pollwave;            // experiment builds only (measured negative, round 6): 1 = every wave polls the 32 producers of its own K slice
  // EXPERIMENT builds only (-DVAURA_EXPERIMENT_ENGINES; measured negative: the launch grows by 6.8 .. 10.7 us, profiles/r06_ab_mall_warm.txt).
  // Infinity-Cache warm-up by the 64 workgroups that have no phase-2 / phase-3 duty (round 6): once the first hand-off has passed (the
  // HBM pipe then runs far below its rate until the launch ends) they touch one dword per 128-byte line of pf_lines lines of what a
  // LATER launch will stream — the next layer's w1||w3 — so that stream is served from the 256 MiB memory-side cache instead of HBM.
  const unsigned char* pf_ptr;
  int pf_lines;            // 0 = off
  const unsigned char* pf_ptr0;      // a first region touched before it (the next layer's wo)
  int pf_lines0;
  int pf_early;            // 1: start right behind the workgroup's own publish instead of behind the first hand-off
  // ATT instances (round 5): the NEXT layer's attention as a fourth phase — K / V cache of that layer (this (row, head)'s rows are
  // requested while the qkv phase still runs), rope table, outputs (fp32 packed rows + planes for wo); flags + 512 .. 703: qkv producers
  const float* att_rope;
  float* att_kc;           // (rows, n_head, max_len, 96) of layer + 1
  float* att_vc;
  float* att_out;
  uint16_t* att_outp;
  int att_max_len;
};

#define MLPE_NW 8
#define MLPE_SPIN_LIMIT 20000
// TIMING-ONLY ablation (experiment builds: -DMLPE_ABL_X8, WRONG RESULTS): the activation planes fetched 8 bytes per lane instead of 16 — what an
// fp8 activation format would move — to price that format before building it (round 6)
#ifdef MLPE_ABL_X8
#define MLPE_XLOAD(rs, voff, soff, aux) [&] { const auto v2_ = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, aux); return u32x4{v2_[0], v2_[1], 0x3c003c00u, 0x3c003c00u}; }()
#else
#define MLPE_XLOAD(rs, voff, soff, aux) __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, aux)
#endif

// RBK = row blocks per weight pass (round 5; gemv3_kernel.h): 2 for 17..32 decoder rows.  The reduction tiles double (48 KB), so the
// ring gives up a quarter (12 KB per wave) and one more k-group pair per plane travels in registers.
template <int WT, int RBK = 1>
struct MlpEngineShape {
  static constexpr int WH = WT == 2 ? 2 : 1;
  static constexpr int G2 = 8;                         // k-group pairs per wave in phase 2 (K = 4096)
  // WT = 1 (fp8 tile pairs, round 5; two row blocks only): ONE 1-KiB fragment holds both k-groups of a pair, so the whole w2 slice of
  // a wave (8 fragments) waits in LDS
  static constexpr int PL = (WT == 1 || WT == 3) ? 8 : (RBK == 2 ? (WT == 2 ? 3 : 6) : (WT == 2 ? 4 : 8));   // pairs per wave whose weights wait in LDS (the rest in registers)
  static constexpr int NF = (WT == 1 || WT == 3) ? PL : 2 * PL * WH;   // 1-KiB fragments of ring per wave
  static constexpr int WAVE_RING = NF * 1024;          // bytes of ring per wave: 16 KB (12 KB with two row blocks, 8 KB fp8)
  static constexpr int RED = RBK * MLPE_NW * 3 * 64 * 16;    // reduction tiles (every phase; the qkv phase has three tiles per wave)
  static constexpr int LDS = MLPE_NW * WAVE_RING + RED + 128;     // + the arrival / hand-shake words
};

// spin until the MLPE_NW words at `w` all hold `epoch` (LDS; each written by a compute wave behind its reduction tiles)
__device__ __forceinline__ void mlpe_wait_words(const unsigned* w, uint32_t epoch) {
  for (;;) {
    bool all = true;
#pragma unroll
    for (int i = 0; i < 8; ++i) all &= __hip_atomic_load(w + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == epoch;
    if (all) break;
    __builtin_amdgcn_s_sleep(1);
  }
}

// wave 0: poll 4 * n_lanes producer flags (lane i: flags 4 i .. 4 i + 3) with sc1 loads until every one holds `epoch`; bounded — a
// give-up raises VAURA_STATUS_HANDOFF_TIMEOUT, after which (and when it is already set) waits return at once.  Returns "broken".
template <typename ARGS>
__device__ __forceinline__ bool mlpe_poll_flags(const uint32_t* flags, int n_lanes, uint32_t epoch, const ARGS& e, int wid, int lane,
                                                bool broken = false) {
  if (wid != 0) return broken;
  uint32_t st4;
  asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(st4) : "v"(e.state + 4) : "memory");
  broken = broken || (st4 & VAURA_STATUS_HANDOFF_TIMEOUT) != 0 || (e.abl & 1);
  const uint32_t* p = flags + 4 * (lane < n_lanes ? lane : 0);
  int spin = 0;
  for (;;) {
    u32x4 f;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(f) : "v"(p) : "memory");
    const bool ok = f.x == epoch && f.y == epoch && f.z == epoch && f.w == epoch;
    if (__builtin_amdgcn_ballot_w64(ok) == ~0ull || broken) break;
    if (++spin >= MLPE_SPIN_LIMIT) {
      if (lane == 0) __hip_atomic_fetch_or(e.state_rw + 4, VAURA_STATUS_HANDOFF_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      broken = true;
      break;
    }
    __builtin_amdgcn_s_sleep(2);
  }
  return broken;
}

__device__ __forceinline__ uint32_t mlpe_ld_sc1(const uint32_t* p) {
  uint32_t v;
  asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}

// RBK = 2 (17..32 decoder rows): every phase multiplies each weight fragment against the planes of BOTH row blocks (second accumulator
// set, the planes of both requested together); wave r finishes row block r of a phase (waves 3 r .. 3 r + 2 in the qkv phase) and wave
// 0 publishes once wave 1 has drained its stores too (an LDS word).  Per row block the same products in the same order as the separate
// two-row-block launches (gemv3_kernel / gemv3h_kernel RBK = 2): bit-identical to them.
// ATT (QKV instances, one row block, 16 heads, cache <= 256 positions: the headline shape): the NEXT layer's attention as a FOURTH
// phase.  All 256 workgroups stay; workgroup b is (head b & 15, row b >> 4) of attention_step256_kernel's grid and runs its arithmetic
// (csrc/attention.hip attention256_body: same sums in the same order -> bit-identical).  What the fusion buys over the separate launch:
// the cached K / V rows of that (row, head) depend on nothing this launch computes, so they are requested as soon as a wave's qkv
// products are issued (workgroups 192..255, which have no w2 / qkv tile: right after phase 1) and stream under the qkv epilogue and its
// hand-off — and the kernel boundary in front of the attention (~2.7 us of ramp + the dispatch gap) is gone.  The hand-off itself is
// small: (row, head) needs the q, k and v quads of ITS head = 18 column tiles = 12 of the 192 qkv producers (three aligned groups of
// four flags), not all of them.
template <int WT, bool QKV, int RBK = 1, bool ATT = false>
__global__ __launch_bounds__(MLPE_NW * 64) void mlp_engine_kernel(const void* __restrict__ W13q, const uint16_t* __restrict__ XPq,
                                                                  const void* __restrict__ W2q, MlpEngineArgs e) {
  using SH = MlpEngineShape<WT, RBK>;
  constexpr bool F32 = WTag<WT>::F32, FP8 = WTag<WT>::FP8;
  constexpr int XPL = WTag<WT>::XPL;
  static_assert(!FP8 || (RBK == 2 && !ATT), "fp8 tile pairs: the two-row-block instances only (configs[4]'s shape)");
  constexpr int WH = SH::WH, NW = MLPE_NW, NACC = 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char mlpe_lds[];
  unsigned char* ring = mlpe_lds;
  f32x4* red = reinterpret_cast<f32x4*>(mlpe_lds + NW * SH::WAVE_RING);            // [NW][2][64]
  unsigned* arrive = reinterpret_cast<unsigned*>(mlpe_lds + NW * SH::WAVE_RING + SH::RED);
  unsigned* arrive2 = arrive + NW;             // phase-2 tiles written (QKV instances: their phase-2 reduction has no barrier either)

  VA_STAMP_DECL(stamps);
  VA_STAMP(stamps, 0);                       // wave start
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bid = blockIdx.x;
  // epoch of this hand-off: unique per (sequence, position, layer); the flags hold the previous hand-off's epoch until rewritten
  const uint32_t epoch = va_handoff_epoch(e.state, e.layer);
  // LDS arrival words carry the epoch mixed with THIS workgroup's id: on a busy chip the 256 workgroups of a launch do not all start at
  // once, and a late one can land on a CU another workgroup of the SAME launch has just left — whose arrival words hold this very
  // epoch (round 4: found by the concurrent-load test; on an idle chip every workgroup has a CU of its own and it never shows)
  const uint32_t ltag = epoch ^ ((uint32_t)(blockIdx.x + 1) * 0x9E3779B1u);

  EpiPre pre;
  pre.have = false;
  f32x4 ws1[2] = {f32x4{1.f, 1.f, 1.f, 1.f}, f32x4{1.f, 1.f, 1.f, 1.f}}, ws2 = f32x4{1.f, 1.f, 1.f, 1.f};
  // LDS words of the two-row-block instances: arrive[18] / arrive[19] <- wave 1 has drained its phase-1 / phase-2 epilogue stores
  const bool epw = wid < RBK;                  // this wave finishes a row block (wave r: row block r) in phases 1 and 2
  static_assert(!ATT || (QKV && RBK == 1), "the attention phase follows the qkv phase of a one-row-block launch");

  // ---- attention phase (ATT): the cached K / V rows of this workgroup's (row, head), 8 lanes per position, 64 positions per pass,
  //      up to four passes (cache <= 256); requested by att_request() wherever the wave has registers and nothing left to request
  constexpr int AQ = 3;                        // 16-byte quads of a 96-wide row per lane (8 lanes per position)
  const int att_h = bid & 15, att_row = bid >> 4;
  const int att_pos = ATT ? e.state[0] : 0;    // the cache holds positions [0, pos)
  const int att_nu = (att_pos + 63) >> 6;
  const bool att_live = ATT && att_row < e.p1.rows;
  f32x4 kf[ATT ? 4 : 1][AQ], vf[ATT ? 4 : 1][AQ];
  auto att_request = [&]() {
    if constexpr (ATT) {
      if (!att_live) return;
      const int sub = threadIdx.x & 7, prow = threadIdx.x >> 3;
      const float* kc = e.att_kc + ((size_t)att_row * 16 + att_h) * (size_t)e.att_max_len * 96;
      const float* vc = e.att_vc + ((size_t)att_row * 16 + att_h) * (size_t)e.att_max_len * 96;
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (u < att_nu) {                      // uniform: whole passes only (slots past the end of the last pass re-read the last row)
          const int p = min(u * 64 + prow, att_pos - 1);
#pragma unroll
          for (int i = 0; i < AQ; ++i) kf[u][i] = reinterpret_cast<const f32x4*>(kc + (size_t)p * 96)[sub + 8 * i];
        }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (u < att_nu) {
          const int p = min(u * 64 + prow, att_pos - 1);
#pragma unroll
          for (int i = 0; i < AQ; ++i) vf[u][i] = reinterpret_cast<const f32x4*>(vc + (size_t)p * 96)[sub + 8 * i];
        }
    }
  };

  // ----============================================================ phase 4 (ATT): the next layer's attention for (head att_h, row att_row)
  auto att_phase = [&]() {
    if (!att_live) return;
    constexpr int HD = 96, QUADS = HD / 4, DM = 16 * HD;
    f32x4* sqkv = red;                                   // rotated q | rotated k | v of the new position | scratch   (3 QUADS + 64 quads)
    f32x4(*wacc)[QUADS] = reinterpret_cast<f32x4(*)[QUADS]>(red + 3 * QUADS + 64);     // [NW][QUADS]
    float* wm = reinterpret_cast<float*>(red + 3 * QUADS + 64 + NW * QUADS);
    float* wl = wm + NW;
    // ---- hand-off: the q, k and v quads of head h are column tiles 6h .. 6h + 5 of each section = qkv producers 4h .. 4h + 3,
    //      64 + 4h .. and 128 + 4h .. (workgroup = 2 (tile / 3) + K half): lanes 0 .. 2 of wave 0 poll one aligned group of four
    if (wid == 0) {
      const bool broken = (mlpe_ld_sc1(reinterpret_cast<const uint32_t*>(e.state + 4)) & VAURA_STATUS_HANDOFF_TIMEOUT) != 0;
      const uint32_t* fp = e.flags + 512 + 64 * (lane < 3 ? lane : 0) + 4 * att_h;
      int spin = 0;
      for (;;) {
        u32x4 f;
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(f) : "v"(fp) : "memory");
        const bool ok = f.x == epoch && f.y == epoch && f.z == epoch && f.w == epoch;
        if (__builtin_amdgcn_ballot_w64(ok) == ~0ull || broken || (e.abl & 1)) break;
        if (++spin >= MLPE_SPIN_LIMIT) {
          if (lane == 0) __hip_atomic_fetch_or(e.state_rw + 4, VAURA_STATUS_HANDOFF_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // ---- from here on: attention256_body (csrc/attention.hip), steps 2 .. 5, on the rows requested above
    const int tid = threadIdx.x;
    const int sub = tid & 7, prow = tid >> 3;
    const float scale = 1.0f / sqrtf((float)HD);
    const int gt = min(tid, 3 * QUADS - 1);
    const int which = gt / QUADS, cq = gt % QUADS;
    const Gemv3Args& aq = e.p3;
    // the new position's q / k / v quad: both K-half partials, written in THIS launch -> sc1
    const __amdgpu_buffer_rsrc_t q1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(aq.out), 0, 16 * 3 * DM * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t q2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(aq.out2), 0, 16 * 3 * DM * 4, 0x00020000);
    const int qoff = (int)(packed_quad(att_row, (which * DM + att_h * HD + cq * 4) >> 2, 3 * DM) * 16);
    constexpr int QAUX = 16;     // sc1: past L1, served by L2 (MLPE_ATT_LOCAL: the producers' plain stores left the lines in THIS XCD's L2)
    f32x4 gx = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(q1, qoff, 0, QAUX));
    const f32x4 gx2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(q2, qoff, 0, QAUX));
    const f32x4 gcs = *reinterpret_cast<const f32x4*>(e.att_rope + ((size_t)att_pos * (HD / 2) + cq * 2) * 2);  // c0 s0 c1 s1
    float* kc = e.att_kc + ((size_t)att_row * 16 + att_h) * (size_t)e.att_max_len * HD;
    float* vc = e.att_vc + ((size_t)att_row * 16 + att_h) * (size_t)e.att_max_len * HD;
    gx = gx + gx2;
    f32x4 y;
    y[0] = gx[0] * gcs[0] - gx[1] * gcs[1];
    y[1] = gx[1] * gcs[0] + gx[0] * gcs[1];
    y[2] = gx[2] * gcs[2] - gx[3] * gcs[3];
    y[3] = gx[3] * gcs[2] + gx[2] * gcs[3];
    if (which == 2) y = gx;   // v is not rotated
    sqkv[tid < 3 * QUADS ? tid : 3 * QUADS + (tid & 63)] = y;
    if (tid >= QUADS && tid < 3 * QUADS)
      va_st16(reinterpret_cast<f32x4*>((which == 1 ? kc : vc) + (size_t)att_pos * HD) + cq, y);
    __syncthreads();                                       // (also drains this wave's K / V requests: needed next anyway)
    const f32x4* sq4 = sqkv;
    const f32x4* sk4 = sqkv + QUADS;
    const f32x4* sv4 = sqkv + 2 * QUADS;
    f32x4 qf[AQ];
#pragma unroll
    for (int i = 0; i < AQ; ++i) qf[i] = sq4[sub + 8 * i];
    auto dot8 = [&](const f32x4* kv) {
      float d = 0.f;
#pragma unroll
      for (int i = 0; i < AQ; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) d = fmaf(qf[i][c], kv[i][c], d);
      d += va_dpp<VA_DPP_XOR1>(d);
      d += va_dpp<VA_DPP_XOR2>(d);
      d += va_dpp<VA_DPP_HALF_MIRROR>(d);
      return d;
    };
    f32x4 knew[AQ];
#pragma unroll
    for (int i = 0; i < AQ; ++i) knew[i] = sk4[sub + 8 * i];
    const float snew = dot8(knew) * scale;
    float sc[4];
    float m = snew;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (u < att_nu) {
        const float d = dot8(kf[u]) * scale;
        sc[u] = (u * 64 + prow < att_pos) ? d : -INFINITY;
        m = fmaxf(m, sc[u]);
      }
    m = wave_max(m);
    float l = 0.f;
    f32x4 av[AQ];
#pragma unroll
    for (int i = 0; i < AQ; ++i) av[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (u < att_nu) {
        const float ex = expf(sc[u] - m);
        if (sub == 0) l += ex;
#pragma unroll
        for (int i = 0; i < AQ; ++i) av[i] += vf[u][i] * ex;
      }
    l = wave_sum(l);
#pragma unroll
    for (int i = 0; i < AQ; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float v = av[i][c];
        v += va_dpp<VA_DPP_ROR8>(v);
        v += va_xor16(v);
        v += va_xor32(v);
        av[i][c] = v;
      }
    if (lane < 8) {
#pragma unroll
      for (int i = 0; i < AQ; ++i) wacc[wid][lane + 8 * i] = av[i];
    }
    if (lane == 0) { wm[wid] = m; wl[wid] = l; }
    __syncthreads();
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
      va_st16(reinterpret_cast<f32x4*>(e.att_out) + packed_quad(att_row, (att_h * HD) / 4 + tid, DM), o);
      if (e.att_outp) store_split4(e.att_outp, att_row, att_h * HD + 4 * tid, DM, o);
    }
    VA_WAIT_VM(0);
    VA_STAMP(stamps, 6);
    VA_STAMP_FLUSH(stamps, 11);
  };

  // ================================================================ phase 1: w1||w3 + SwiGLU (gemv3_kernel<6, 8, 2, E3_SWIGLU, true>)
  {
    Gemv3Args a = e.p1;
    a.W = W13q;
    a.XP = XPq;
    constexpr int G = 6, T = 2, XB = F32 ? 3 : 1;
    constexpr int K = 32 * G * NW, KG = K / 32, GB = G / XB;
    constexpr bool WBATCH = F32 && XB > 1;
    constexpr int GW = FP8 ? G / 2 : (WBATCH ? 2 * GB : G);      // fp8: one 16-byte load carries k-groups g and g + 1
    constexpr int NSS = K / 64;
    const int w = (wid + bid) % NW;
    const int m = lane & 15, q = lane >> 4;
    const int tile0 = bid * T;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.W), 0, -16, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.XP), 0, a.R * VA_NPL * (K / 8) * 256, 0x00020000);
    const int lane16 = lane * 16;
    u32x4 wb[T][GW][WH];
    auto load_w = [&](int g0, int n, int slot0) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (g < g0 || g >= g0 + n) continue;
        if (FP8 && (g & 1)) continue;
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const size_t kg = FP8 ? (size_t)(tile0 + t) * (KG / 2) + (size_t)((w * G + g) >> 1) : (size_t)(tile0 + t) * KG + (size_t)(w * G + g);
          const int slot = slot0 + (FP8 ? (g - g0) / 2 : g - g0);
#pragma unroll
          for (int hh = 0; hh < WH; ++hh)
            wb[t][slot][hh] = __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16, (int)((kg * WH + hh) * 1024), 2 /* nt */);
        }
      }
    };
    constexpr int NXB = XB > 1 ? 2 : 1;
    u32x4 xb[RBK][NXB][GB][XPL];
    auto load_x = [&](int b) {
#pragma unroll
      for (int r = 0; r < RBK; ++r) {
        const int xl16 = r * 16 + m < a.rows ? lane16 : 0x7ffffff0;
#pragma unroll
        for (int g = 0; g < GB; ++g)
#pragma unroll
          for (int p = 0; p < XPL; ++p)
            xb[r][b % NXB][g][p] = MLPE_XLOAD(xrs, xl16, (int)(((r * VA_NPL + p) * (K / 8) * 16 + (w * G + b * GB + g) * 64) * 16), 0);
      }
    };
    if constexpr (WBATCH) load_w(0, GB, 0);
    else load_w(0, G, 0);
    __builtin_amdgcn_sched_barrier(0);
    load_x(0);
    float ssv[NSS];                                    // n_ss_in == K / 16 == 4 NSS (checked by the launcher): no bounds to test
    if (epw) {
      const float* sp = a.ss_in + (size_t)wid * a.n_ss_in * 16 + m;         // row block wid
#pragma unroll
      for (int j = 0; j < NSS; ++j) ssv[j] = sp[(q + 4 * j) * 16];
    }
    // wave 0, behind its stream requests (HBM misses first): what the two epilogues need and earlier KERNELS wrote — the residual
    // tile and next norm's gain of phase 2, the power-of-two row scales of both (a dependent L2 round trip behind the reduction
    // otherwise)
    if (epw) {
      if (bid < 192) {
        const int h2_ = (bid >> 3) & 1, tile2_ = (bid & 7) + 8 * (bid >> 4);
        if (((lane >> 3) & 1) == h2_) pre = gemv3_epilogue_prefetch<E3_RESID>(e.p2, wid, tile2_, lane);
        ws2 = *reinterpret_cast<const f32x4*>(e.p2.wscale + (size_t)tile2_ * 16 + 4 * q);
      }
      ws1[0] = *reinterpret_cast<const f32x4*>(e.p1.wscale + (size_t)(bid * 2) * 16 + 4 * q);
      ws1[1] = *reinterpret_cast<const f32x4*>(e.p1.wscale + (size_t)(bid * 2 + 1) * 16 + 4 * q);
    }
    __builtin_amdgcn_sched_barrier(0);                 // every request of the first batch is out before anything is waited for
#ifndef MLPE_DIAG2
    VA_STAMP(stamps, 1);                               // phase 1: first batch requested
#endif
    f32x4 acc[RBK][T][NACC];
#pragma unroll
    for (int r = 0; r < RBK; ++r)
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int p = 0; p < NACC; ++p) acc[r][t][p] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < XB; ++b) {
      if (b + 1 < XB) {
        if constexpr (WBATCH) load_w((b + 1) * GB, GB, ((b + 1) & 1) * GB);
        load_x(b + 1);
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
            wf[0] = __builtin_bit_cast(f16x8, wb[t][slot][0]);
            wf[1] = __builtin_bit_cast(f16x8, wb[t][slot][WH - 1]);
          } else {
            wf[0] = __builtin_bit_cast(f16x8, wb[t][b * GB + g][0]);
          }
#pragma unroll
          for (int r = 0; r < RBK; ++r) mfma_group<WT>(wf, xb[r][b % NXB][g], acc[r][t]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int r = 0; r < RBK; ++r)
#pragma unroll
      for (int t = 0; t < T; ++t) red[((r * NW + wid) * T + t) * 64 + lane] = acc_sum<WT>(acc[r][t]);
    // arrival words instead of a workgroup barrier: waves 1..7 go on to request their w2 slice at once; wave 0 alone waits for
    // the tiles.  A wave's word carries this launch's epoch mixed with the workgroup id (LDS keeps what the previous workgroup on
    // this CU left: another epoch or another workgroup's tag, never this one), so nothing has to be initialised and no barrier
    // opens the kernel.  LDS operations of a wave execute in order:
    // the word lands behind the tiles (release: the compiler keeps that order too).
    if (lane == 0) __hip_atomic_store(arrive + wid, ltag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifndef MLPE_DIAG2
    VA_STAMP(stamps, 2);                               // phase 1: products done, tiles in LDS
#endif
    if (epw) {
      float ssp = 0.f;
#pragma unroll
      for (int j = 0; j < NSS; ++j) ssp += ssv[j];
      ssp += va_xor16(ssp);
      ssp += va_xor32(ssp);
      const float rinv = 1.0f / sqrtf(ssp * (1.0f / (float)a.k_total) + a.eps);
      for (;;) {
        bool all = true;
#pragma unroll
        for (int i = 0; i < NW; ++i) all &= __hip_atomic_load(arrive + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == ltag;
        if (all) break;
        __builtin_amdgcn_s_sleep(1);
      }
#ifdef MLPE_DIAG       // diagnostic stamps build with -DMLPE_DIAG: slot 1 <- wave 0 has seen all eight waves' phase-1 tiles, slot 5 <- its stores are issued
      VA_STAMP(stamps, 1);
#endif
      f32x4 v[T];
#pragma unroll
      for (int t = 0; t < T; ++t) {
        f32x4 sacc = red[((wid * NW + 0) * T + t) * 64 + lane];
#pragma unroll
        for (int i = 1; i < NW; ++i) sacc += red[((wid * NW + i) * T + t) * 64 + lane];
        sacc *= ws1[t];
        v[t] = sacc * rinv;
      }
      if (wid * 16 < a.rows) gemv3_epilogue<T, E3_SWIGLU>(a, wid, tile0, lane, v, nullptr);
#ifdef MLPE_DIAG
      VA_STAMP(stamps, 5);
#endif
      if constexpr (RBK == 2) {
        if (wid == 1) {      // the second row block's tile: stored (write-through), drained, then the word wave 0 waits for before it publishes
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (lane == 0) __hip_atomic_store(arrive + 18, ltag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    }
    if (wid == 0) {
      if (!MLPE_REL && lane == 0) __hip_atomic_store(arrive + 16, ltag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);   // releases the held part of the other waves' run-ahead (phase 2)
      // (Measured and rejected, round 4: holding the other waves' run-ahead requests back until these stores are in the CU's memory
      // pipeline.  A CU serves its vector-memory requests in order, so the publish waits behind the seven waves' 224 KB of requests
      // — 2.6 us median in the stamps — and with the hold it comes 2.8 us earlier; but the run-ahead then starts 2.3 us later, wave
      // 0's poll comes back behind its own later requests, and the loop is 1 % (two planes) / 3 % (one) SLOWER: what bounds the
      // second phase is the 393 KB each CU has to take in, in whatever order.)
      // publish: this wave stored the workgroup's whole ffn tile (write-through); drained, then the flag
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if constexpr (RBK == 2) {
        while (__hip_atomic_load(arrive + 18, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != ltag) __builtin_amdgcn_s_sleep(1);
      }
      if (lane == 0) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(e.flags + bid), "v"(epoch) : "memory");
      if (MLPE_REL && lane == 0) __hip_atomic_store(arrive + 16, ltag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifndef MLPE_DIAG2
      VA_STAMP(stamps, 3);                             // wave 0: published
#endif
    }
  }
  if (bid >= 192) {
    if constexpr (!ATT) {
#ifdef VAURA_EXPERIMENT_ENGINES      // measured negative (round 6, profiles/r06_ab_mall_warm.txt): experiment builds only
      if (e.pf_lines + e.pf_lines0 > 0) {
        // helper: wait (bounded, like a consumer) until every phase-1 producer has published, then touch
        if (!e.pf_early) {
          (void)mlpe_poll_flags(e.flags, 64, epoch, e, wid, lane);
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
        }
        const int t0 = (bid - 192) * (MLPE_NW * 64) + (int)threadIdx.x;       // 0 .. 32767
        // at most 12 lines per thread (w1||w3 on two planes: 393 216 lines over 32 768 threads); every load is issued before anything
        // waits, and the destination registers stay live up to the wait (the compiler does not know these asm statements are loads).
        // Two regions: [0, pf_lines0) of pf_ptr0 (the next layer's wo) first, then [0, pf_lines) of pf_ptr (its w1||w3).
        uint32_t v[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const int total = e.pf_lines0 + e.pf_lines;
        const int npt = (total + 64 * MLPE_NW * 64 - 1) / (64 * MLPE_NW * 64);        // uniform
#pragma unroll
        for (int j = 0; j < 12; ++j) {
          if (j < npt) {
            int i = t0 + j * (64 * MLPE_NW * 64);
            if (i >= total) i = t0 % total;                                            // past the end: a line already requested
            const unsigned char* ptr = i < e.pf_lines0 ? e.pf_ptr0 + (size_t)i * 128 : e.pf_ptr + (size_t)(i - e.pf_lines0) * 128;
            asm volatile("global_load_dword %0, %1, off nt" : "=v"(v[j]) : "v"(ptr) : "memory");
          }
        }
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]),
                       "+v"(v[10]), "+v"(v[11])
                     :
                     : "memory");
      }
#endif
      VA_STAMP_FLUSH(stamps, 11);
    } else {
      att_request();                           // no w2 / qkv tile here: this (row, head)'s K / V rows at once,
      att_phase();                             // then straight to the attention (its own copy of the code: the rows' registers are not
    }                                          // live through the phases these workgroups skip)
    return;
  }

  // ================================================================ phase 2: w2 + residual (gemv3h_kernel<8, 8, E3_RESID>)
  {
    Gemv3Args a = e.p2;
    a.W = W2q;
    constexpr int G2 = SH::G2, PL = SH::PL;
    constexpr int K = 64 * G2 * NW, KG = K / 32, BS = 1024 * WH;
    const int h = (bid >> 3) & 1;
    const int tile = (bid & 7) + 8 * (bid >> 4);
    const int w = (wid + tile) % NW;
    const int la = lane & 7, sb = (lane >> 3) & 1, q = lane >> 4;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.W), 0, -16, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.XP), 0, a.R * VA_NPL * (K / 8) * 256, 0x00020000);
    const int voffw0 = (la + 16 * q) * 16 + sb * BS;
    const int voffx = (sb * 64 + q * 16 + la + 8 * h) * 16;
    unsigned char* myring = ring + wid * SH::WAVE_RING;

    // ---- run-ahead: this wave's w2 slice.  Pairs [0, PL): LDS-DMA, one 1-KiB fragment (k-group c, plane hh) per instruction;
    //      pairs [PL, G2): straight into the registers phase 1 has just released.  Waves 1..7 come here while wave 0 still reduces
    //      and publishes phase 1; wave 0 follows as soon as it has published — BEFORE it polls: its first poll then comes back
    //      behind its own weight requests (a wave's vector-memory counter retires in order), which is about when the slowest
    //      producer has published anyway, and nothing of the weight stream is left for after the hand-off.
    u32x4 wreg[G2 - PL > 0 ? G2 - PL : 1][2][WH];
    // the slice in four quarters: 0, 1 = the LDS-DMA half (k-group pairs [0, PL)), 2, 3 = the register half
    auto prefetch_q = [&](auto qc) {
      constexpr int Q = decltype(qc)::value;
      if constexpr (Q < 2) {
        // the wave's slice = consecutive 1-KiB fragments ((k-group, plane) blocks; fp8: one per k-group PAIR); quarter Q = half of those
        // that wait in LDS
        const unsigned char* src = static_cast<const unsigned char*>(a.W) +
                                   (FP8 ? ((size_t)tile * (KG / 2) + (size_t)(w * G2)) * 1024 : ((size_t)tile * KG + 2 * (size_t)(w * G2)) * BS) + lane * 16;
        constexpr int NFQ = SH::NF / 2;
#pragma unroll
        for (int f = Q * NFQ; f < (Q + 1) * NFQ; ++f)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)f * 1024),
                                           (__attribute__((address_space(3))) void*)(myring + f * 1024), 16, 0, 2 /* nt */);
      } else {
        constexpr int HALF = (G2 - PL) / 2;
#pragma unroll
        for (int j = PL + (Q - 2) * HALF; j < (Q == 2 ? PL + HALF : G2); ++j) {
          const int soff = (tile * KG + 2 * (w * G2 + j)) * BS;
#pragma unroll
          for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int hh = 0; hh < WH; ++hh)
              wreg[j - PL][nh][hh] = __builtin_amdgcn_raw_buffer_load_b128(wrs, voffw0 + nh * 128, soff + hh * 1024, 2 /* nt */);
        }
      }
    };
    auto prefetch_w2 = [&]() {
      prefetch_q(std::integral_constant<int, 0>{}); prefetch_q(std::integral_constant<int, 1>{});
      prefetch_q(std::integral_constant<int, 2>{}); prefetch_q(std::integral_constant<int, 3>{});
    };
    if (!(e.abl & 8) && !(e.abl & 4)) {
      // partial hold: waves 1..7 request MLPE_Q2 quarters of their slice at once and the rest only when wave 0's epilogue stores are in
      // the CU's memory pipeline (ablation bit 3: no hold, round 4's first form); wave 0 comes here behind its publish and holds nothing
      constexpr int Q2 = MlpeThrottle<WT, RBK>::Q2;
      const bool hold = wid >= RBK;                    // (the waves that finish a row block come here behind their own stores and hold nothing)
      if (!hold || Q2 > 0) prefetch_q(std::integral_constant<int, 0>{});
      if (!hold || Q2 > 1) prefetch_q(std::integral_constant<int, 1>{});
      if (!hold || Q2 > 2) prefetch_q(std::integral_constant<int, 2>{});
      if (!hold || Q2 > 3) prefetch_q(std::integral_constant<int, 3>{});
      if (hold && Q2 < 4) {
        while (__hip_atomic_load(arrive + 16, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != ltag) __builtin_amdgcn_s_sleep(1);
        if (Q2 < 1) prefetch_q(std::integral_constant<int, 0>{});
        if (Q2 < 2) prefetch_q(std::integral_constant<int, 1>{});
        if (Q2 < 3) prefetch_q(std::integral_constant<int, 2>{});
        if (Q2 < 4) prefetch_q(std::integral_constant<int, 3>{});
      }
    } else if (!(e.abl & 4)) prefetch_w2();
#ifndef MLPE_DIAG2
    if (wid != 0) VA_STAMP(stamps, 3);                 // waves 1..7: w2 slice requested
#endif

    // ---- hand-off 1: wave 0 polls the 256 producer flags (lane i: flags 4i .. 4i + 3), bounded; the barrier releases the rest.
    //      (Round 6, measured NEGATIVE and kept in experiment builds only — -DVAURA_EXPERIMENT_ENGINES, second flag word bit 7: every wave
    //      polling just the 32 producers of its own K slice, no workgroup barrier.  Bit-identical, and slower: two planes + 0.6 %, one plane
    //      + 3.3 %, fp8h + 2.1 % on whole loops (profiles/r06_ab_pollwave.txt) — eight pollers per consumer instead of one hammer the flag
    //      lines the producers are still storing to, and a wave's poll waits for its own w2 prefetch.)
#ifdef VAURA_EXPERIMENT_ENGINES
    if (e.pollwave) {
      const bool broken = (mlpe_ld_sc1(reinterpret_cast<const uint32_t*>(e.state + 4)) & VAURA_STATUS_HANDOFF_TIMEOUT) != 0;
      const uint32_t* fp = e.flags + 32 * w + 4 * (lane & 7);
      int spin = 0;
      for (;;) {
        u32x4 f;
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(f) : "v"(fp) : "memory");
        const bool ok = f.x == epoch && f.y == epoch && f.z == epoch && f.w == epoch;
        if (__builtin_amdgcn_ballot_w64(ok) == ~0ull || broken || (e.abl & 1)) break;
        if (++spin >= MLPE_SPIN_LIMIT) {
          if (lane == 0) __hip_atomic_fetch_or(e.state_rw + 4, VAURA_STATUS_HANDOFF_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      asm volatile("" ::: "memory");
    } else
#endif
    {
    if (wid == 0) {
      const bool broken = (mlpe_ld_sc1(reinterpret_cast<const uint32_t*>(e.state + 4)) & VAURA_STATUS_HANDOFF_TIMEOUT) != 0;
      int spin = 0;
      for (;;) {
        u32x4 f;
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(f) : "v"(e.flags + 4 * lane) : "memory");
        const bool ok = f.x == epoch && f.y == epoch && f.z == epoch && f.w == epoch;
        if (__builtin_amdgcn_ballot_w64(ok) == ~0ull || broken || (e.abl & 1)) break;
        if (++spin >= MLPE_SPIN_LIMIT) {
          if (lane == 0) __hip_atomic_fetch_or(e.state_rw + 4, VAURA_STATUS_HANDOFF_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    // raw barrier: __syncthreads() would first drain every wave's vector-memory counter — i.e. wait for the whole w2 prefetch —
    // before anybody may request the planes.  Only control has to pass here: the planes are requested behind it in program order.
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    }
    VA_STAMP(stamps, 4);                               // hand-off barrier passed (wave 0: its poll matched just before)
    if (e.abl & 4) prefetch_w2();

    // ---- the planes of this workgroup's 8 rows (of every row block) over its wave's K slice: every load sc1 (the producers stored
    //      write-through).  Two row blocks: the second block's last four pairs are requested into the registers the first block's
    //      first four release (all 32 fragments at once do not fit next to the weights).
    u32x4 xb[RBK][G2][XPL];
    auto load_xr = [&](auto rc, auto j0c, auto j1c) {
      constexpr int r = decltype(rc)::value, j0 = decltype(j0c)::value, j1 = decltype(j1c)::value;
      const int vx = r * 16 + la + 8 * h < a.rows ? voffx : 0x7ffffff0;
#pragma unroll
      for (int j = j0; j < j1; ++j)
#pragma unroll
        for (int p = 0; p < XPL; ++p)
          xb[r][j][p] = MLPE_XLOAD(xrs, vx, ((r * VA_NPL + p) * (K / 8) * 16 + (w * G2 + j) * 128) * 16, 16 /* sc1 */);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using IH = std::integral_constant<int, G2 / 2>;
    using IG = std::integral_constant<int, G2>;
    load_xr(I0{}, I0{}, IG{});
    if constexpr (RBK == 2) load_xr(I1{}, I0{}, IH{});
    __builtin_amdgcn_sched_barrier(0);                 // ONE round trip for the planes: all 16 requests before the first wait
    // The ring was filled by LDS-DMA: the compiler does not know those instructions write LDS and would read fragments while they
    // are still in flight (it issued the first ds_reads AHEAD of its own wait for the planes).  On a quiet chip the DMA has landed
    // microseconds earlier; with another stream loading the chip it has not (test_one_launch_mlp_under_concurrent_load_and_repeats
    // found tokens differing).  A wave reads only fragments it requested itself, so its own counter is the whole synchronisation: every
    // request of this wave — DMA, register weights, planes (needed next anyway) — has landed before the first fragment is read.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    VA_WAIT_VM(0);
#if !defined(MLPE_DIAG) && !defined(MLPE_DIAG2)
    VA_STAMP(stamps, 5);                               // (diagnostic build) weights and planes landed
#endif
    f32x4 acc[RBK][2][NACC];
#pragma unroll
    for (int r = 0; r < RBK; ++r)
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int p = 0; p < NACC; ++p) acc[r][nh][p] = f32x4{0.f, 0.f, 0.f, 0.f};
    // products of row block r over pairs [j0, j1): k order inside a row block is that of gemv3h_kernel, whatever the interleaving
    // the A operands of pair j, weight-row half nh: from the LDS ring (pairs [0, PL); fp8: all of them, widened) or from registers
    constexpr int WFN = F32 ? 2 : 1;
    auto wfetch = [&](int j, int nh, f16x8* wf) {
      if constexpr (FP8) {      // half sb of the pair's fragment: 8 bytes of lane (la + 8 nh, q)
        const uint2 pr = *reinterpret_cast<const uint2*>(myring + j * 1024 + (la + 8 * nh + 16 * q) * 16 + sb * 8);
        wf[0] = fp8x8_to_f16(pr.x, pr.y);
      } else if (j < PL) {
        const u32x4* fr = reinterpret_cast<const u32x4*>(myring + ((2 * j + sb) * WH) * 1024) + (la + 8 * nh + 16 * q);
        wf[0] = __builtin_bit_cast(f16x8, fr[0]);
        if constexpr (F32) wf[1] = __builtin_bit_cast(f16x8, fr[64]);
      } else {
        wf[0] = __builtin_bit_cast(f16x8, wreg[j - PL][nh][0]);
        if constexpr (F32) wf[1] = __builtin_bit_cast(f16x8, wreg[j - PL][nh][WH - 1]);
      }
    };
    // Round 6: the fragments of pair j + 1 are fetched (LDS read, fp8 widening) BEFORE the products of pair j are issued — the stamps
    // (profiles/r06_engine_stamps_diag2.txt) show 1.0 .. 1.7 us between "planes landed" and "products done" for 0.2 .. 0.4 us of matrix
    // instructions: every pair paid its LDS round trip (and the fp8 -> fp16 conversion) in front of its own products.  Same products into
    // the same accumulators in the same order: bit-identical.  (-DMLPE_NO_WPIPE: the previous form, for the A/B.)
    auto products = [&](auto rc, auto j0c, auto j1c) {
      constexpr int r = decltype(rc)::value, j0 = decltype(j0c)::value, j1 = decltype(j1c)::value;
#ifdef MLPE_NO_WPIPE
#pragma unroll
      for (int j = j0; j < j1; ++j) {
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
          f16x8 wf[WFN];
          wfetch(j, nh, wf);
          mfma_group<WT>(wf, xb[r][j], acc[r][nh]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#else
      f16x8 cur[2][WFN], nxt[2][WFN];
      wfetch(j0, 0, cur[0]);
      wfetch(j0, 1, cur[1]);
#pragma unroll
      for (int j = j0; j < j1; ++j) {
        if (j + 1 < j1) {
          wfetch(j + 1, 0, nxt[0]);
          wfetch(j + 1, 1, nxt[1]);
        }
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) mfma_group<WT>(cur[nh], xb[r][j], acc[r][nh]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
          for (int i = 0; i < WFN; ++i) cur[nh][i] = nxt[nh][i];
      }
#endif
    };
    if constexpr (RBK == 1) {
      products(I0{}, I0{}, IG{});
    } else {
      products(I0{}, I0{}, IH{});
      load_xr(I1{}, IH{}, IG{});                       // into the registers block 0's first half has just released
      __builtin_amdgcn_sched_barrier(0);
      products(I0{}, IH{}, IG{});
      products(I1{}, I0{}, IH{});
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      products(I1{}, IH{}, IG{});
    }
#pragma unroll
    for (int r = 0; r < RBK; ++r)
#pragma unroll
      for (int nh = 0; nh < 2; ++nh) {
        const f32x4 v = acc_sum<WT>(acc[r][nh]);
        f32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float x = v[c];
          o[c] = x + va_dpp<VA_DPP_ROR8>(va_xor32(x));
        }
        red[((r * NW + wid) * 2 + nh) * 64 + lane] = o;
      }
    if constexpr (!QKV) {
      __syncthreads();
      if (epw && wid * 16 < a.rows) {
        const int m = lane & 15;
        const bool mine = (m >> 3) == h;
        const int src = (m & 7) + 16 * (q & 1);
        f32x4 v = red[((wid * NW + 0) * 2 + (q >> 1)) * 64 + src];
#pragma unroll
        for (int i = 1; i < NW; ++i) v += red[((wid * NW + i) * 2 + (q >> 1)) * 64 + src];
        v *= ws2;
        if (mine) gemv3_epilogue<1, E3_RESID>(a, wid, tile, lane, &v, &pre);
      }
      VA_WAIT_VM(0);
      VA_STAMP(stamps, 6);                               // done (wave 0: epilogue stores acknowledged)
      VA_STAMP_FLUSH(stamps, 11);
    } else {
      // ============================================================== phase 3: the NEXT layer's qkv GEMV (gemv3_kernel<3, 8, 3, E3_STORE, true, 1, 0, WT, 2>)
      // Same pattern once more: the weights of the next layer's wqkv depend on nothing, so every wave requests its slice (3 tiles x 3
      // k-groups, straight into the registers phase 2 has released) as soon as its phase-2 products are issued, wave 0 once it has
      // published; the h planes and the partial sums of squares phase 2's epilogues write are handed over inside the launch (192
      // producers = these same workgroups); behind the hand-off: one round trip for the planes, 18 products per wave, reduction,
      // rinv, store of the two K-half partials the attention kernel adds on load.  One launch and one kernel boundary less per layer.
      Gemv3Args aq = e.p3;
      // fp8 tile pairs hold two k-groups per fragment, and a K half is 24 k-groups: FOUR waves take six each (three fragments per
      // tile) — the slices, products and 4-wave reduction of gemv3_kernel<6, 4, 3, E3_STORE, true, 1, 0, 1, 2> — waves 4 .. 7 only
      // pass the barriers (and finish tiles: any wave can run an epilogue)
      constexpr int G = FP8 ? 6 : 3, NWQ = FP8 ? 4 : NW, GF = 3, TQ = 3, KQ = 1536, KGQ = KQ / 32;
      // EXPERIMENT build -DMLPE_ATT_LOCAL (ATT instances): the qkv work items of head h on the XCD of h's attention workgroups (XCD =
      // workgroup id % 8 = h % 8): item s 64 + 4 h + sub goes to workgroup (h % 8) + 8 ((h / 8) 12 + 4 s + sub), so that the hand-off
      // of the q / k / v quads stays inside one L2 (consumer loads sc0 instead of sc1)
      // Round 6 (e.qlocal, second flag word bit 23 — A/B): the same relabelling for the ordinary instances, ACROSS the kernel boundary: the
      // attention launch that follows puts (head h, row r) on XCD h % 8 too, so if an XCD's L2 keeps what its own workgroups stored
      // (plain stores, written back by the end-of-kernel release) the new q / k / v quads are L2 hits there instead of a memory-side read.
#ifdef MLPE_ATT_LOCAL
      const int qid = ATT ? (((bid >> 3) % 12) >> 2) * 64 + 4 * ((bid & 7) + 8 * ((bid >> 3) / 12)) + (((bid >> 3) % 12) & 3) : bid;
#elif defined(VAURA_EXPERIMENT_ENGINES)      // measured: no effect (profiles/r06_ab_qlocal.txt) — the kernel-start acquire leaves nothing of the previous launch in L2
      const int qid = (!ATT && e.qlocal) ? (((bid >> 3) % 12) >> 2) * 64 + 4 * ((bid & 7) + 8 * ((bid >> 3) / 12)) + (((bid >> 3) % 12) & 3) : bid;
#else
      const int qid = bid;
#endif
      const int ks = qid & 1, tile0q = (qid >> 1) * TQ, kgo = ks * G * NWQ;
      const bool actq = wid < NWQ;                     // this wave multiplies in the qkv phase
      const int w3 = (wid + bid) % NWQ;
      const int mq = lane & 15;
      const int lane16 = lane * 16;
      const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(aq.W), 0, -16, 0x00020000);
      u32x4 wq[TQ][GF][WH];                            // GF fragments per tile: k-groups (x planes), or fp8 k-group pairs
      auto wq_off = [&](int t, int f, int hh) {
        const size_t kg = FP8 ? (size_t)(tile0q + t) * (KGQ / 2) + (size_t)((kgo + w3 * G + 2 * f) >> 1)
                              : ((size_t)(tile0q + t) * KGQ + (size_t)(kgo + w3 * G + f)) * WH + hh;
        return (int)(kg * 1024);
      };
      auto load_wq_g = [&](auto gc) {
        constexpr int g = decltype(gc)::value;
#pragma unroll
        for (int t = 0; t < TQ; ++t) {
#pragma unroll
          for (int hh = 0; hh < WH; ++hh)
            wq[t][g][hh] = __builtin_amdgcn_raw_buffer_load_b128(qrs, lane16, wq_off(t, g, hh), 2 /* nt */);
        }
      };
      auto load_wq = [&]() {
        if (!actq) return;
        load_wq_g(std::integral_constant<int, 0>{});
        load_wq_g(std::integral_constant<int, 1>{});
        load_wq_g(std::integral_constant<int, 2>{});
      };
      if (lane == 0) __hip_atomic_store(arrive2 + wid, ltag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifdef MLPE_DIAG2      // second diagnostic stamp set (-DVAURA_STAMPS -DMLPE_DIAG2): the phase 2 -> 3 chain.  slot 1 <- phase-2 products done, 2 <- wave 0
      VA_STAMP(stamps, 1);   // saw all tiles + epilogue stores issued, 3 <- phase 2 published (drained + flag), 5 <- hand-off 2 passed, 6 <- done
#endif
      f32x4 wsq = f32x4{1.f, 1.f, 1.f, 1.f};
      const bool epq = wid < TQ * RBK;                 // this wave finishes tile (wid % TQ) of row block (wid / TQ) of the qkv phase
      const int rq = wid / TQ, tq = wid - rq * TQ;
      if (!epw && actq) {
        if (!(e.abl & 8)) {   // PRE3 of the three k-groups at once, the rest once wave 0's phase-2 stores are in the memory pipeline
          // in ninths (k-group g, tile t; unit = 3 g + t): PRE3U of them at once, the rest behind the hold
          constexpr int PRE3U = MlpeThrottle<WT, RBK>::PRE3U;
          auto unit = [&](auto uc) {
            constexpr int u = decltype(uc)::value, g = u / 3, t = u % 3;
#pragma unroll
            for (int hh = 0; hh < WH; ++hh)
              wq[t][g][hh] = __builtin_amdgcn_raw_buffer_load_b128(qrs, lane16, wq_off(t, g, hh), 2 /* nt */);
          };
          va_static_for9([&](auto uc) { if constexpr (decltype(uc)::value < PRE3U) unit(uc); });
          if (PRE3U < 9)
            while (__hip_atomic_load(arrive + 17, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != ltag) __builtin_amdgcn_s_sleep(1);
          va_static_for9([&](auto uc) { if constexpr (decltype(uc)::value >= PRE3U) unit(uc); });
        } else {
          load_wq();
        }
      }
      if (epq) wsq = *reinterpret_cast<const f32x4*>(aq.wscale + (size_t)(tile0q + tq) * 16 + 4 * q);
      if (epw) {
        mlpe_wait_words(arrive2, ltag);
        const int m = lane & 15;
        const bool mine = (m >> 3) == h;
        const int src = (m & 7) + 16 * (q & 1);
        f32x4 v = red[((wid * NW + 0) * 2 + (q >> 1)) * 64 + src];
#pragma unroll
        for (int i = 1; i < NW; ++i) v += red[((wid * NW + i) * 2 + (q >> 1)) * 64 + src];
        v *= ws2;
        if (mine && wid * 16 < a.rows) gemv3_epilogue<1, E3_RESID>(a, wid, tile, lane, &v, &pre);
#ifdef MLPE_DIAG2
        VA_STAMP(stamps, 2);
#endif
        if (wid == 0 && !MLPE_REL && lane == 0) __hip_atomic_store(arrive + 17, ltag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        // publish phase 2: h, its partial sums of squares and its planes are out (write-through), drained, then the flag (two row
        // blocks: wave 1 drains its block's stores and tells wave 0, which publishes for both)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (RBK == 2) {
          if (wid == 1) {
            if (lane == 0) __hip_atomic_store(arrive + 19, ltag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
          } else {
            while (__hip_atomic_load(arrive + 19, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != ltag) __builtin_amdgcn_s_sleep(1);
          }
        }
        if (wid == 0 && lane == 0) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(e.flags + 256 + bid), "v"(epoch) : "memory");
        if (wid == 0 && MLPE_REL && lane == 0) __hip_atomic_store(arrive + 17, ltag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifdef MLPE_DIAG2
        VA_STAMP(stamps, 3);
#endif
        load_wq();
      }
      (void)mlpe_poll_flags(e.flags + 256, 48, epoch, e, wid, lane);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#ifdef MLPE_DIAG2
      VA_STAMP(stamps, 5);
#endif
      const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(aq.XP), 0, aq.R * VA_NPL * (KQ / 8) * 256, 0x00020000);
      const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(aq.ss_in), 0, RBK * 96 * 16 * 4, 0x00020000);
      u32x4 xq[RBK][G][XPL];
#pragma unroll
      for (int r = 0; r < RBK; ++r) {
        const int xl16 = (actq && r * 16 + mq < aq.rows) ? lane16 : 0x7ffffff0;
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int p = 0; p < XPL; ++p)
            xq[r][g][p] = MLPE_XLOAD(hrs, xl16, (int)(((r * VA_NPL + p) * (KQ / 8) * 16 + (kgo + w3 * G + g) * 64) * 16), 16 /* sc1 */);
      }
      constexpr int NSSQ = KQ / 64;
      float ssq[NSSQ];
      if (epq) {
#pragma unroll
        for (int j = 0; j < NSSQ; ++j)                   // written by phase 2's epilogues in THIS launch: sc1
          ssq[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srs, ((rq * 96 + q + 4 * j) * 16 + mq) * 4, 0, 16 /* sc1 */));
      }
      __builtin_amdgcn_sched_barrier(0);
      f32x4 accq[RBK][TQ][NACC];
#pragma unroll
      for (int r = 0; r < RBK; ++r)
#pragma unroll
        for (int t = 0; t < TQ; ++t)
#pragma unroll
          for (int p = 0; p < NACC; ++p) accq[r][t][p] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (actq) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int t = 0; t < TQ; ++t) {
          f16x8 wf[F32 ? 2 : 1];
          if constexpr (FP8) {
            const u32x4 pr = wq[t][g / 2][0];
            wf[0] = (g & 1) ? fp8x8_to_f16(pr.z, pr.w) : fp8x8_to_f16(pr.x, pr.y);
          } else {
            wf[0] = __builtin_bit_cast(f16x8, wq[t][g][0]);
            if constexpr (F32) wf[1] = __builtin_bit_cast(f16x8, wq[t][g][WH - 1]);
          }
#pragma unroll
          for (int r = 0; r < RBK; ++r) mfma_group<WT>(wf, xq[r][g], accq[r][t]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      }
      if (actq) {
#pragma unroll
        for (int r = 0; r < RBK; ++r)
#pragma unroll
          for (int t = 0; t < TQ; ++t) red[((r * NW + wid) * TQ + t) * 64 + lane] = acc_sum<WT>(accq[r][t]);
      }
      float rinvq = 1.f;
      if (epq) {
        float ssp = 0.f;
#pragma unroll
        for (int j = 0; j < NSSQ; ++j) ssp += ssq[j];
        ssp += va_xor16(ssp);
        ssp += va_xor32(ssp);
        rinvq = 1.0f / sqrtf(ssp * (1.0f / (float)aq.k_total) + aq.eps);
      }
      // ATT: the waves without a tile to finish have nothing left to request for this phase: their share of the K / V rows now
      if (ATT && !epq) att_request();
      __syncthreads();
      if (epq && rq * 16 < aq.rows) {
        f32x4 sacc = red[((rq * NW + 0) * TQ + tq) * 64 + lane];
#pragma unroll
        for (int i = 1; i < NWQ; ++i) sacc += red[((rq * NW + i) * TQ + tq) * 64 + lane];
        sacc *= wsq;
        const f32x4 v = sacc * rinvq;
        if (ks > 0) aq.out = aq.out2;
#ifdef MLPE_ATT_LOCAL
        if constexpr (ATT) {     // plain store: the line stays in this XCD's L2, where the head's attention workgroups read it (sc1 loads)
          reinterpret_cast<f32x4*>(aq.out)[((size_t)rq * (aq.N / 4) + (size_t)(tile0q + tq) * 4) * 16 + lane] = v;
        } else
#endif
#ifdef VAURA_EXPERIMENT_ENGINES
        if (!ATT && e.qlocal) {  // plain store (see qid above)
          reinterpret_cast<f32x4*>(aq.out)[((size_t)rq * (aq.N / 4) + (size_t)(tile0q + tq) * 4) * 16 + lane] = v;
        } else
#endif
        gemv3_epilogue<1, E3_STORE>(aq, rq, tile0q + tq, lane, &v, nullptr);
      }
      if constexpr (ATT) {
        // publish the qkv tiles (write-through stores): waves 0 .. 2 stored one each — drained, waves 1 and 2 tell wave 0 (LDS words),
        // wave 0 stores the flag; then these waves request their share of the K / V rows (behind their stores: a wave's vector-memory
        // counter retires in order, a drain in front of the flag would otherwise wait for the rows)
        if (epq) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (wid > 0) {
            if (lane == 0) __hip_atomic_store(arrive + 20 + wid, ltag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
          } else {
            while (__hip_atomic_load(arrive + 21, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != ltag ||
                   __hip_atomic_load(arrive + 22, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != ltag)
              __builtin_amdgcn_s_sleep(1);
#ifdef MLPE_ATT_LOCAL
            if (lane == 0) asm volatile("global_store_dword %0, %1, off" ::"v"(e.flags + 512 + qid), "v"(epoch) : "memory");
#else
            if (lane == 0) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(e.flags + 512 + qid), "v"(epoch) : "memory");
#endif
          }
          att_request();
        }
      } else {
        VA_WAIT_VM(0);
        VA_STAMP(stamps, 6);
        VA_STAMP_FLUSH(stamps, 11);
      }
    }
  }

  if constexpr (ATT) att_phase();
}

#ifdef VAURA_EXPERIMENT_ENGINES
// measured-negative engines (the layer tail as one launch; DESIGN_HISTORY.md round 4): experiment builds only
// (python -m vaura_amd.csrc.build --tag engines -DVAURA_EXPERIMENT_ENGINES=1), never part of libvaura_hip.so
#include "experiments/tail_engine.h"
#endif
