// Launcher of the fp16-pair GEMV (gemv3_kernel.h): the decode-step instances, weight ingress, op-level entry points.
#include "gemv3_kernel.h"
#include "mlp_engine.h"
#include <cstdlib>

// measurement aid (vaura_set_debug_flags): bit 0 = keep the one-workgroup-per-tile kernels for wo / w2 (A/B of the row split),
// bit 2 = w1||w3 and w2 as two launches even where the one-launch MLP (mlp_engine.h) is eligible, bits 28..31 = its timing ablations,
// bit 4 = per-position prefill attention,
// bit 5 = 64-row prefill GEMM workgroups only, bit 6 = row f2's linears on the 128 x 96 conv tile instead of linear_pair_kernel,
// bit 7 = row f2's space attention with one thread per query instead of the MFMA kernel, bit 8 = 256-row tiles in
// linear_pair_kernel, bit 9 = unsplit CLS attention, bit 10 = time attention with one thread per (head, frame),
// bit 11 = space attention on the exact-fp32 MFMA instead of fp16 pairs, bit 13 = the codec's last conv without the LDS window
unsigned va_debug_flags = 0;
VA_STAMP_SETTER(vaura_stamps_set_gemv3)
unsigned va_debug_flags_get() { return va_debug_flags; }

// second flag word (vaura_set_debug_flags2), bit 0: more than one row block -> still ONE row block per weight pass (round 4's walk: the
// A/B of the two-row-block instances and the control of their bit-identity test); bit 1: the one-launch MLP refuses 17..32 rows;
// bit 2: EXPERIMENT, the next layer's attention as a fourth phase of the one-launch MLP (api.hip: measured slower);
// bit 3: fp8 weights keep round 4's one-workgroup-per-tile kernels for wo / w2 (the A/B of the fp8 row-split instances);
// bit 4: fp8 weights never take the one-launch MLP; bits 5, 6, 12..15: row f2's linear layers (vit.hip); bit 7 and bits 16..22: experiment
// builds only (-DVAURA_EXPERIMENT_ENGINES: hand-off 1 of the one-launch MLP polled per wave; Infinity-Cache warm-up by its idle
// workgroups — both measured negative in round 6)
unsigned va_debug_flags2 = 0;
unsigned va_debug_flags2_get() { return va_debug_flags2; }
static bool rb2(const Gemv3Args& a) { return a.R >= 2 && !(va_debug_flags2 & 1u); }

// RB2 = true: the instance exists with two row blocks per weight pass as well (the decode step's shapes), taken when there are >= 2
// (XBR = plane batches of THAT instance: with two row blocks' planes and accumulators in registers the K = 1536 instances of two and
// three tiles take the planes in three batches, two in flight — all up front they spill)
template <int WT, int G, int NW, int T, int EPI, bool NORM, int XB = 1, int KS = 1, bool RB2 = false, int XBR = XB>
static int launch3(const Gemv3Args& a, int64_t n_tiles, hipStream_t s) {
  if (n_tiles % T) return VAURA_ERR_SHAPE;
  if constexpr (RB2) {
    if (rb2(a)) {
      VA_LAUNCH((gemv3_kernel<G, NW, T, EPI, NORM, XBR, 0, WT, KS, 2>), dim3((unsigned)(n_tiles / T * KS)), dim3(NW * 64), 0, s, a.W, a.XP, a);
      return 0;
    }
  }
  VA_LAUNCH((gemv3_kernel<G, NW, T, EPI, NORM, XB, 0, WT, KS>), dim3((unsigned)(n_tiles / T * KS)), dim3(NW * 64), 0, s, a.W, a.XP, a);
  return 0;
}


template <int WT, int G2, int EPI, int XB, int NBF = 2, int NW = 8, bool RB2 = false>
static int launch3h(const Gemv3Args& a, int64_t n_tiles, hipStream_t s) {
  const int halves = (a.R == 1 && a.rows <= 8) ? 1 : 2;    // at most 8 live rows: the second half would multiply zeros
  if (halves == 2 && (n_tiles % 8)) return VAURA_ERR_SHAPE;
  if constexpr (RB2) {
    if (rb2(a)) {
      VA_LAUNCH((gemv3h_kernel<G2, NW, EPI, XB, WT, NBF, 2>), dim3((unsigned)(n_tiles * halves)), dim3(NW * 64), 0, s, a.W, a.XP, a, halves);
      return 0;
    }
  }
  VA_LAUNCH((gemv3h_kernel<G2, NW, EPI, XB, WT, NBF>), dim3((unsigned)(n_tiles * halves)), dim3(NW * 64), 0, s, a.W, a.XP, a, halves);
  return 0;
}

template <int WT>
static int dispatch3(const Gemv3Args& a, int64_t tiles, int64_t K, int epilogue, bool norm, hipStream_t s) {
  // two weight planes are 8 bytes per lane and k-group: two-tile workgroups take weights and planes in three batches (two in
  // flight) like the K = 4096 instances their four, so that the slice a wave holds fits the register file without spills
  constexpr int XB2 = WT == 2 ? 3 : 1, XB4 = 4;
  if constexpr (WT == 1 || WT == 3) {   // fp8 tile pairs (round 5; 3 = against the hi activation plane only, round 6): with 17..32 rows the residual GEMVs on the row-split pair kernel too — one workgroup
    // per tile takes in BOTH row blocks' planes (524 KB at K = 4096 next to 65 KB of weights): 217.5 -> 214.7 ms on the 32-row loop;
    // at 16 rows the one-workgroup-per-tile kernels stay (162.3 against 164.8 ms).  Second flag word, bit 3: never (the A/B)
    if (a.R >= 2 && !norm && !(va_debug_flags & 1u) && !(va_debug_flags2 & 8u) && tiles % 8 == 0 && epilogue == E3_RESID) {
      if (K == 1536) return launch3h<WT, 3, E3_RESID, 1, 2, 8, true>(a, tiles, s);
      if (K == 4096) return launch3h<WT, 8, E3_RESID, 1, 2, 8, true>(a, tiles, s);
    }
  }
  if constexpr (WT != 1 && WT != 3) {   // narrow outputs without a fused norm (wo, w2): two workgroups per tile, 8 rows each
    if (!norm && !(va_debug_flags & 1u) && tiles % 8 == 0 && (epilogue == E3_RESID || epilogue == E3_STORE)) {
      if (K == 1536 && epilogue == E3_RESID) return launch3h<WT, 3, E3_RESID, 1, 2, 8, true>(a, tiles, s);
      if (K == 1536 && epilogue == E3_STORE) return launch3h<WT, 3, E3_STORE, 1>(a, tiles, s);
      // fp32 weights at K = 4096: four batches of two k-group pairs, two in flight.  (Three in flight measured the same 11.4 us:
      // this instance is bound by the bytes through each CU — 262 KB of weights + 196 KB of planes per workgroup — not by latency.)
      if (K == 4096 && epilogue == E3_RESID) return launch3h<WT, 8, E3_RESID, (WT == 2 ? 4 : 1), 2, 8, true>(a, tiles, s);
      if (K == 4096 && epilogue == E3_STORE) return launch3h<WT, 8, E3_STORE, (WT == 2 ? 4 : 1)>(a, tiles, s);
    }
  }
  if (K == 1536) {
    if (epilogue == E3_STORE && norm) return launch3<WT, 6, 8, 2, E3_STORE, true, XB2>(a, tiles, s);
    if (epilogue == E3_STORE && !norm) return launch3<WT, 6, 8, 1, E3_STORE, false>(a, tiles, s);
    if (epilogue == E3_RESID && !norm) return launch3<WT, 6, 8, 1, E3_RESID, false, 1, 1, true>(a, tiles, s);
    if (epilogue == E3_SWIGLU && norm) return launch3<WT, 6, 8, 2, E3_SWIGLU, true, XB2, 1, true, 3>(a, tiles, s);
    if (epilogue == E3_LOGITS && norm) return launch3<WT, 6, 8, 3, E3_LOGITS, true, XB2, 1, true, 3>(a, tiles, s);
  } else if (K == 4096) {
    if (epilogue == E3_RESID && !norm) return launch3<WT, 16, 8, 1, E3_RESID, false, XB4, 1, true>(a, tiles, s);
    if (epilogue == E3_STORE && !norm) return launch3<WT, 16, 8, 1, E3_STORE, false, XB4>(a, tiles, s);
  }
  return VAURA_ERR_SHAPE;
}

template <int EPI, bool NORM>
static int launch_gemm3(const Gemv3Args& a, int64_t tiles, int64_t K, hipStream_t s) {
  const int gx = (int)(tiles / (G3M_NW * G3M_T)), gy4 = (a.R + G3M_RB - 1) / G3M_RB, gy8 = (a.R + 7) / 8;
  // 128-row workgroups once there are enough of them to fill the chip twice over (debug flag bit 14: at >= 64 row blocks whatever the
  // count, round 2's rule — wo and w2 of a 166-position prompt then run 126 workgroups on 256 CUs)
  const bool big = !(va_debug_flags & 32u) && ((va_debug_flags & 0x4000u) ? a.R >= 64 : gx * gy8 >= 320);
  const int remap = (va_debug_flags & 0x8000u) ? 0 : 1;               // bit 15: blocks in launch order
  const bool pf2 = !(va_debug_flags & 0x10000u) && (K / 32) % 2 == 0;     // bit 16: one weight k-group in flight
  const int gy = big ? gy8 : gy4;
  const dim3 grid((unsigned)(gx * gy)), block(G3M_NW * 64);
  if (a.wq == 1 || a.wq == 3) {      // (a prompt pass of the hi-plane-only fp8 storage multiplies both planes: the exact fp8 arithmetic)
    if (big) VA_LAUNCH((gemm3_kernel<EPI, NORM, 1, 8>), grid, block, 0, s, a, (int)K, gx, gy, remap);
    else VA_LAUNCH((gemm3_kernel<EPI, NORM, 1>), grid, block, 0, s, a, (int)K, gx, gy, remap);
  } else if (a.wq == 2) {
    if (big && pf2) VA_LAUNCH((gemm3_kernel<EPI, NORM, 2, 8, 2>), grid, block, 0, s, a, (int)K, gx, gy, remap);
    else if (big) VA_LAUNCH((gemm3_kernel<EPI, NORM, 2, 8, 1>), grid, block, 0, s, a, (int)K, gx, gy, remap);
    else if (pf2) VA_LAUNCH((gemm3_kernel<EPI, NORM, 2, G3M_RB, 2>), grid, block, 0, s, a, (int)K, gx, gy, remap);
    else VA_LAUNCH((gemm3_kernel<EPI, NORM, 2>), grid, block, 0, s, a, (int)K, gx, gy, remap);
  } else {
    if (big && pf2) VA_LAUNCH((gemm3_kernel<EPI, NORM, 0, 8, 2>), grid, block, 0, s, a, (int)K, gx, gy, remap);
    else if (big) VA_LAUNCH((gemm3_kernel<EPI, NORM, 0, 8, 1>), grid, block, 0, s, a, (int)K, gx, gy, remap);
    else if (pf2) VA_LAUNCH((gemm3_kernel<EPI, NORM, 0, G3M_RB, 2>), grid, block, 0, s, a, (int)K, gx, gy, remap);
    else VA_LAUNCH((gemm3_kernel<EPI, NORM, 0>), grid, block, 0, s, a, (int)K, gx, gy, remap);
  }
  return 0;
}

// LDS-DMA pipelined GEMM (gemm4_kernel): fp16-plane weights.  One launch over row blocks [rb_off, rb_off + n_rb).
template <int EPI, bool NORM, int WT, int RBW, int CT = G4_CT>
static int launch_gemm4_i(const Gemv3Args& a, int64_t K, int gx, int rb_off, int n_rb, hipStream_t s) {
  using SH = G4Shape<WT, RBW, CT>;
  static unsigned long long big = 0;
  if (va_big_lds_once(reinterpret_cast<const void*>(gemm4_kernel<EPI, NORM, WT, RBW, CT>), SH::LDS, &big)) return VAURA_ERR_STATE;
  const int gy = (n_rb + RBW - 1) / RBW;
  VA_LAUNCH((gemm4_kernel<EPI, NORM, WT, RBW, CT>), dim3((unsigned)(gx * gy)), dim3(G4_NW * 64), SH::LDS, s, a, (int)K, gx, gy,
            ((va_debug_flags & 0x8000u) ? 0 : 1) | ((va_debug_flags >> 17) & 2),      // bit 18: ablation (no DMA in the loop; timing only)
            rb_off);
  return 0;
}
template <int EPI, bool NORM, int WT>
static int launch_gemm4_h(const Gemv3Args& a, int64_t K, int gx, int rbw, int rb_off, int n_rb, hipStream_t s) {
  if (rbw == 8) return launch_gemm4_i<EPI, NORM, WT, 8>(a, K, gx, rb_off, n_rb, s);
  if (rbw == 6) return launch_gemm4_i<EPI, NORM, WT, 6>(a, K, gx, rb_off, n_rb, s);
  return launch_gemm4_i<EPI, NORM, WT, 4>(a, K, gx, rb_off, n_rb, s);
}
template <int EPI, bool NORM>
static int launch_gemm4(const Gemv3Args& a, int64_t tiles, int64_t K, hipStream_t s) {
  const int gx = (int)(tiles / G4_CT);
  // Rows per workgroup (64, 96 or 128): the height with the least work on the busiest CU — ceil(workgroups / 256) rounds of that many
  // row blocks — weighted by what the wave tile costs (64 rows: 64 x 32 wave tiles, 0.63 LDS reads per MFMA against 0.38 — but with one
  // weight plane two such workgroups share a CU and fill each other's barrier gaps; 96 rows: one dead DMA slot in eight).  And one
  // cut: whole rounds of tall workgroups, then the remaining rows in ONE round of a lower height (a second launch), when that beats a
  // last round that is mostly empty.  A 166-position prompt, two planes: wo / w2 (6 column tiles) 252 x 64 rows; qkv (18) 504 x 96
  // rows = 1.97 rounds; w1||w3 (32) 512 x 128 rows + 224 x 96 rows instead of 672 x 128 = 2.6 rounds.  Debug flag bit 5: 64 rows only;
  // bit 22: no cut.
  const int cand[3] = {8, 6, 4};
  const float eff[3] = {1.0f, 1.03f, a.wq == 2 ? 1.12f : 0.95f};
  auto rounds = [&](int rb, int h) { return (float)((gx * ((rb + h - 1) / h) + 255) / 256); };
  int h1 = 4, h2 = 0, cut = 0;
  if (!(va_debug_flags & 32u)) {
    float best = 1e30f;
    for (int i = 0; i < 3; ++i) {
      const float c1 = rounds(a.R, cand[i]) * cand[i] * eff[i];
      if (c1 < best) { best = c1; h1 = cand[i]; h2 = 0; cut = 0; }
      // (one plane: the co-resident 64-row pairs already smooth the last round; a cut measured 3 % slower there)
      if (cand[i] == 4 || a.wq != 2 || (va_debug_flags & 0x400000u)) continue;
      const int full = gx * ((a.R + cand[i] - 1) / cand[i]) / 256;        // whole rounds of this height
      const int gy1 = full * 256 / gx, rb1 = gy1 * cand[i];
      if (full < 1 || gy1 < 1 || rb1 >= a.R) continue;
      for (int k = 0; k < 3; ++k) {
        const float c2 = (float)full * cand[i] * eff[i] + rounds(a.R - rb1, cand[k]) * cand[k] * eff[k] + 0.3f;   // + a kernel boundary
        if (c2 < best) { best = c2; h1 = cand[i]; h2 = cand[k]; cut = rb1; }
      }
    }
  }
  int rc;
  // two planes, 64-row workgroups chosen (few column tiles): the same number of 128 x 128 workgroups instead (debug flag bit 23: no)
  if (a.wq == 2 && h1 == 4 && !h2 && !(va_debug_flags & (0x800000u | 32u)))
    return launch_gemm4_i<EPI, NORM, 2, 8, 8>(a, K, 2 * gx, 0, a.R, s);
  if (a.wq == 2) {
    rc = launch_gemm4_h<EPI, NORM, 2>(a, K, gx, h1, 0, h2 ? cut : a.R, s);
    if (!rc && h2) rc = launch_gemm4_h<EPI, NORM, 2>(a, K, gx, h2, cut, a.R - cut, s);
  } else {
    rc = launch_gemm4_h<EPI, NORM, 0>(a, K, gx, h1, 0, h2 ? cut : a.R, s);
    if (!rc && h2) rc = launch_gemm4_h<EPI, NORM, 0>(a, K, gx, h2, cut, a.R - cut, s);
  }
  return rc;
}

// many row blocks (a prompt being teacher-forced): GEMM tiling instead of the register-resident GEMV loop
static int dispatch_gemm3(const Gemv3Args& a, int64_t tiles, int64_t K, int epilogue, bool norm, hipStream_t s) {
  if (a.wq != 1 && a.wq != 3 && !(va_debug_flags & 0x20000u)) {     // debug flag bit 17: the register-staged gemm3_kernel for every storage
    if (epilogue == E3_STORE && norm) return launch_gemm4<E3_STORE, true>(a, tiles, K, s);
    if (epilogue == E3_STORE && !norm) return launch_gemm4<E3_STORE, false>(a, tiles, K, s);
    if (epilogue == E3_RESID && !norm) return launch_gemm4<E3_RESID, false>(a, tiles, K, s);
    if (epilogue == E3_SWIGLU && norm) return launch_gemm4<E3_SWIGLU, true>(a, tiles, K, s);
    if (epilogue == E3_LOGITS && norm) return launch_gemm4<E3_LOGITS, true>(a, tiles, K, s);
    return VAURA_ERR_SHAPE;
  }
  if (epilogue == E3_STORE && norm) return launch_gemm3<E3_STORE, true>(a, tiles, K, s);
  if (epilogue == E3_STORE && !norm) return launch_gemm3<E3_STORE, false>(a, tiles, K, s);
  if (epilogue == E3_RESID && !norm) return launch_gemm3<E3_RESID, false>(a, tiles, K, s);
  if (epilogue == E3_SWIGLU && norm) return launch_gemm3<E3_SWIGLU, true>(a, tiles, K, s);
  if (epilogue == E3_LOGITS && norm) return launch_gemm3<E3_LOGITS, true>(a, tiles, K, s);
  return VAURA_ERR_SHAPE;
}

// bytes of one weight element by storage (a.wq): one fp16 plane 2, fp8 1, two fp16 planes 4; float scale[N] follows the tiles
static const float* weight_scales(const Gemv3Args& a, int64_t n_weight_rows, int64_t K) {
  const size_t per = (a.wq == 1 || a.wq == 3) ? 1 : (a.wq == 2 ? 4 : 2);
  return reinterpret_cast<const float*>(static_cast<const char*>(a.W) + (size_t)n_weight_rows * (size_t)K * per);
}

int va_launch_gemv3(const Gemv3Args& a0, int64_t n_weight_rows, int64_t K, int epilogue, bool norm, hipStream_t s) {
  Gemv3Args a = a0;
  if (!a.W || !a.XP || a.rows <= 0 || (n_weight_rows % 16)) return VAURA_ERR_ARG;
  if (norm && (!a.ss_in || a.n_ss_in <= 0 || a.n_ss_in > 128 || (int64_t)a.n_ss_in * 16 > K)) return VAURA_ERR_ARG;   // one partial per 16 columns of the normed vector
  const int64_t tiles = n_weight_rows / 16;
  a.wscale = weight_scales(a, n_weight_rows, K);
  if (a.out2) {   // the caller asked for two K-half partials (decode qkv): fused norm, K = 1536 only
    if (K != 1536 || epilogue != E3_STORE || !norm || a.R >= 16) return VAURA_ERR_SHAPE;
    if (a.wq == 1) return launch3<1, 6, 4, 3, E3_STORE, true, 1, 2, true>(a, tiles, s);   // fp8 tile pairs hold two k-groups per lane: 4 waves x 6 groups per K half
    if (a.wq == 3) return launch3<3, 6, 4, 3, E3_STORE, true, 1, 2, true>(a, tiles, s);
    if (a.wq == 2) return launch3<2, 3, 8, 3, E3_STORE, true, 1, 2, true, 3>(a, tiles, s);
    return launch3<0, 3, 8, 3, E3_STORE, true, 1, 2, true>(a, tiles, s);
  }
  // GEMM tiling only when there are enough row blocks to fill the chip with 64 x 256 tiles (a prompt pass); a decode
  // step of a large batch (R = 2..15 row blocks) keeps the weight-stationary GEMV loop and its N/(16 T) workgroups
  if (a.R >= 16 && (K == 1536 || K == 4096) && tiles % (G3M_NW * G3M_T) == 0) return dispatch_gemm3(a, tiles, K, epilogue, norm, s);
  if (a.wq == 1) return dispatch3<1>(a, tiles, K, epilogue, norm, s);
  if (a.wq == 3) return dispatch3<3>(a, tiles, K, epilogue, norm, s);
  if (a.wq == 2) return dispatch3<2>(a, tiles, K, epilogue, norm, s);
  return dispatch3<0>(a, tiles, K, epilogue, norm, s);
}

// ---------------------------------------------------------------------------- the MLP half of a layer as one launch (mlp_engine.h)
// Eligible: one row block (1..16 decoder rows), fp16-plane weights, the shipped geometry, hand-off flags provided, a device with at
// least 256 CUs (every workgroup must be resident: consumers wait for producers inside the launch).  With at most 8 live rows the
// second row half of phases 2 / 3 multiplies zeros (the separate launches use one workgroup per tile there) — still the faster form:
// configs[3] (4 rows, 10.24 s) 33.7 k -> 36.9 k tokens/s.
bool va_mlp_engine_eligible(const vaura_decoder* d) {
  if (!d->ws_sync || !d->state || d->rows < 1 || d->rows > 32) return false;
  if (d->rows > 16 && (va_debug_flags2 & 2u)) return false;          // second flag word, bit 1: 17..32 rows keep the separate launches
  // fp8 tile pairs (round 5): the two-row-block instances only (17..32 rows: configs[4]'s per-GPU shape); second flag word, bit 4: no
  if (va_is_fp8(d->wdtype)) { if (d->rows <= 16 || (va_debug_flags2 & 16u)) return false; }
  else if (d->wdtype != VAURA_W_H1 && d->wdtype != VAURA_W_H2) return false;
  if (d->dims.d_model != 1536 || d->dims.ffn_dim != 4096) return false;
  static int cus[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  if (!cus[dev]) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    cus[dev] = n > 0 ? n : -1;
  }
  return cus[dev] >= 256;
}

template <int WT, bool QKV, int RBK = 1, bool ATT = false>
static int launch_mlp_engine_t(const MlpEngineArgs& e, hipStream_t s) {
  using SH = MlpEngineShape<WT, RBK>;
  static unsigned long long big = 0;
  if (va_big_lds_once(reinterpret_cast<const void*>(mlp_engine_kernel<WT, QKV, RBK, ATT>), SH::LDS, &big)) return VAURA_ERR_STATE;
  VA_LAUNCH((mlp_engine_kernel<WT, QKV, RBK, ATT>), dim3(256), dim3(MLPE_NW * 64), SH::LDS, s, e.p1.W, e.p1.XP, e.p2.W, e);
  return 0;
}

// aq != nullptr: the next layer's qkv GEMV (K-split, two partial outputs) as a third phase of the same launch
// att != nullptr (with aq, one row block): the next layer's attention as a fourth phase (flags: 704 words)
int va_launch_mlp_engine(const Gemv3Args& a13, const Gemv3Args& a2, const Gemv3Args* aq, uint32_t* flags, int32_t* state, int layer,
                         hipStream_t s, const VaEngineAttention* att, const void* warm_ptr, size_t warm_bytes, const void* warm0_ptr,
                         size_t warm0_bytes) {
  MlpEngineArgs e;
  // second flag word, bits 16..20: sixteenths (1..16) of `warm_bytes` the idle workgroups touch (0 = off); bit 21: start early
  const unsigned six = (va_debug_flags2 >> 16) & 31u;
  e.pf_ptr = static_cast<const unsigned char*>(warm_ptr);
  e.pf_lines = (warm_ptr && six) ? (int)((warm_bytes / 128) * (six > 16 ? 16 : six) / 16) : 0;
  e.pf_early = (int)((va_debug_flags2 >> 21) & 1u);
  e.pf_ptr0 = static_cast<const unsigned char*>(warm0_ptr);                 // bit 22: the first region (the next layer's wo), whole
  e.pf_lines0 = (warm0_ptr && (va_debug_flags2 & 0x400000u)) ? (int)(warm0_bytes / 128) : 0;
  e.att_rope = nullptr; e.att_kc = e.att_vc = e.att_out = nullptr; e.att_outp = nullptr; e.att_max_len = 0;
  if (att) {
    if (!aq || a13.R != 1 || !att->rope || !att->kc || !att->vc || !att->out || att->n_head != 16 || att->max_len > 256 || att->max_len < 1)
      return VAURA_ERR_ARG;
    e.att_rope = att->rope; e.att_kc = att->kc; e.att_vc = att->vc; e.att_out = att->out; e.att_outp = att->outp; e.att_max_len = att->max_len;
  }
  e.p1 = a13;
  e.p2 = a2;
  e.p3 = aq ? *aq : a2;
  if (aq) {
    if (!aq->W || !aq->out || !aq->out2 || !aq->ss_in || aq->XP != a2.outp || aq->ss_in != a2.ss_out || aq->R != a2.R) return VAURA_ERR_ARG;
    if (aq->wq != a2.wq || aq->N != 3 * 1536 || aq->k_total != 1536 || aq->n_ss_in != 96) return VAURA_ERR_SHAPE;
    e.p3.wscale = weight_scales(*aq, 3 * 1536, 1536);
  }
  if (!a13.W || !a13.XP || !a2.W || !a2.XP || !flags || !state || a13.R < 1 || a13.R > 2 || a2.R != a13.R) return VAURA_ERR_ARG;
  if (a13.wq != a2.wq || a13.wq < 0 || a13.wq > 3 || ((a13.wq == 1 || a13.wq == 3) && (a13.R != 2 || att)) || a13.N != 4096 || a2.N != 1536 || a13.k_total != 1536 ||
      a13.n_ss_in != 96) return VAURA_ERR_SHAPE;
  e.p1.wscale = weight_scales(a13, 2 * 4096, 1536);
  e.p2.wscale = weight_scales(a2, 1536, 4096);
  e.flags = flags; e.state = state; e.state_rw = state; e.layer = layer;
  e.abl = (int)((va_debug_flags >> 28) & 15u);      // bits 28..31: timing ablations of the engine (tools only)
  e.qlocal = (int)((va_debug_flags2 >> 23) & 1u);
  e.pollwave = ((va_debug_flags2 >> 7) & 1u) && !(e.abl & 4);    // second flag word, bit 7 (experiment builds): hand-off 1 polled per wave
  if (a13.R == 2) {       // 17..32 decoder rows: both row blocks per weight fragment
    if (a13.wq == 1) return aq ? launch_mlp_engine_t<1, true, 2>(e, s) : launch_mlp_engine_t<1, false, 2>(e, s);
    if (a13.wq == 3) return aq ? launch_mlp_engine_t<3, true, 2>(e, s) : launch_mlp_engine_t<3, false, 2>(e, s);
    if (aq) return a13.wq == 2 ? launch_mlp_engine_t<2, true, 2>(e, s) : launch_mlp_engine_t<0, true, 2>(e, s);
    return a13.wq == 2 ? launch_mlp_engine_t<2, false, 2>(e, s) : launch_mlp_engine_t<0, false, 2>(e, s);
  }
#ifdef VAURA_EXPERIMENT_ENGINES
  if (att) return a13.wq == 2 ? launch_mlp_engine_t<2, true, 1, true>(e, s) : launch_mlp_engine_t<0, true, 1, true>(e, s);
#else
  if (att) return VAURA_ERR_STATE;       // the attention phase exists in experiment builds only (DESIGN_HISTORY.md round 5)
#endif
  if (aq) return a13.wq == 2 ? launch_mlp_engine_t<2, true>(e, s) : launch_mlp_engine_t<0, true>(e, s);
  return a13.wq == 2 ? launch_mlp_engine_t<2, false>(e, s) : launch_mlp_engine_t<0, false>(e, s);
}

#ifdef VAURA_EXPERIMENT_ENGINES
template <int WT>
static int launch_tail_engine_t(const TailEngineArgs& e, hipStream_t s) {
  using SH = MlpEngineShape<WT>;
  static unsigned long long big = 0;
  if (va_big_lds_once(reinterpret_cast<const void*>(tail_engine_kernel<WT>), SH::LDS, &big)) return VAURA_ERR_STATE;
  VA_LAUNCH((tail_engine_kernel<WT>), dim3(256), dim3(MLPE_NW * 64), SH::LDS, s, e.p0.W, e.p0.XP, e.p1.W, e.p2.W, e);
  return 0;
}

// wo -> w1||w3 -> w2 of one layer as ONE launch (tail_engine_kernel); flags: 512 words
int va_launch_tail_engine(const Gemv3Args& awo, const Gemv3Args& a13, const Gemv3Args& a2, uint32_t* flags, int32_t* state, int layer,
                          hipStream_t s) {
  TailEngineArgs e;
  e.p0 = awo; e.p1 = a13; e.p2 = a2;
  if (!awo.W || !awo.XP || !a13.W || !a13.XP || !a2.W || !a2.XP || !flags || !state || awo.R != 1 || a13.R != 1 || a2.R != 1) return VAURA_ERR_ARG;
  if (!awo.res || !awo.out || !awo.outp || !awo.gain_out || !awo.ss_out || !a2.out || !a2.outp || !a2.gain_out || !a2.ss_out || !a13.outp ||
      !a13.ss_in) return VAURA_ERR_ARG;
  if (awo.wq != a13.wq || a13.wq != a2.wq || (a13.wq != 0 && a13.wq != 2) || awo.N != 1536 || a13.N != 4096 || a2.N != 1536 ||
      a13.k_total != 1536 || a13.n_ss_in != 96 || awo.out != a2.res || awo.outp != a13.XP || a13.outp != a2.XP) return VAURA_ERR_SHAPE;
  e.p0.wscale = weight_scales(awo, 1536, 1536);
  e.p1.wscale = weight_scales(a13, 2 * 4096, 1536);
  e.p2.wscale = weight_scales(a2, 1536, 4096);
  e.flags = flags; e.state = state; e.state_rw = state; e.layer = layer;
  e.abl = (int)((va_debug_flags >> 28) & 15u);
  return a13.wq == 2 ? launch_tail_engine_t<2>(e, s) : launch_tail_engine_t<0>(e, s);
}
#endif

// ---------------------------------------------------------------------------- fp16-plane weight ingress
// power-of-two row scale 2^E with amax / 2^E in [2^13, 2^14): both planes of a weight of ordinary size are normal fp16 numbers
// (hi ~ 2^13, lo ~ 2^2), four binades of headroom below fp16's 65504
__global__ void h_row_scale_kernel(const float* __restrict__ src, float* __restrict__ scale, int K) {
  const float* row = src + (size_t)blockIdx.x * K;
  float mx = 0.f;
  for (int k = threadIdx.x; k < K; k += blockDim.x) mx = fmaxf(mx, fabsf(row[k]));
  mx = wave_max(mx);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    mx = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
    float sc = 1.f;
    if (mx > 0.f) {
      int e;
      (void)frexpf(mx, &e);                     // mx = m * 2^e, m in [0.5, 1)
      sc = ldexpf(1.f, e - 14);
    }
    scale[blockIdx.x] = sc;
  }
}

// src row-major (N x K) fp32 -> MFMA-tile order, PL fp16 planes: [N/16][K/32][PL][64 lanes][8 fp16]; lane = n%16 + 16*((k%32)/8)
template <int PL>
__global__ void h_pack_kernel(const float* __restrict__ src, const float* __restrict__ scale, u32x4* __restrict__ dst, int64_t N, int64_t K) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t KG = K / 32;
  if (gid >= (N / 16) * KG * 64) return;
  const int lane = (int)(gid & 63);
  const int64_t kg = (gid >> 6) % KG, tile = (gid >> 6) / KG;
  const int64_t n = tile * 16 + (lane & 15);
  const float inv = 1.0f / scale[n];            // exact: power of two
  const float* p = src + n * K + kg * 32 + 8 * (lane >> 4);
  _Float16 h[8], l[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float v = p[i] * inv;
    h[i] = (_Float16)v;
    l[i] = (_Float16)(v - (float)h[i]);
  }
  dst[((tile * KG + kg) * PL + 0) * 64 + lane] = u32x4{pack_h2(h[0], h[1]), pack_h2(h[2], h[3]), pack_h2(h[4], h[5]), pack_h2(h[6], h[7])};
  if constexpr (PL == 2)
    dst[((tile * KG + kg) * PL + 1) * 64 + lane] = u32x4{pack_h2(l[0], l[1]), pack_h2(l[2], l[3]), pack_h2(l[4], l[5]), pack_h2(l[6], l[7])};
}

int va_pack_weight_h(const float* src, void* dst, int64_t N, int64_t K, int planes, hipStream_t s) {
  if ((N % 16) || (K % 32)) return VAURA_ERR_SHAPE;
  float* scale = reinterpret_cast<float*>(static_cast<char*>(dst) + (size_t)N * (size_t)K * 2 * planes);
  VA_LAUNCH(h_row_scale_kernel, dim3((unsigned)N), dim3(256), 0, s, src, scale, (int)K);
  const int64_t total = (N / 16) * (K / 32) * 64;
  if (planes == 2) VA_LAUNCH(h_pack_kernel<2>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, (const float*)scale, reinterpret_cast<u32x4*>(dst), N, K);
  else VA_LAUNCH(h_pack_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, (const float*)scale, reinterpret_cast<u32x4*>(dst), N, K);
  return 0;
}

// ---------------------------------------------------------------------------- fp8 weight ingress
// power-of-two row scale: the smallest 2^E with amax <= 448 * 2^E (448 = largest e4m3 value)
__global__ void fp8_row_scale_kernel(const float* __restrict__ src, float* __restrict__ scale, int K) {
  const float* row = src + (size_t)blockIdx.x * K;
  float mx = 0.f;
  for (int k = threadIdx.x; k < K; k += blockDim.x) mx = fmaxf(mx, fabsf(row[k]));
  mx = wave_max(mx);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    mx = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
    float sc = 1.f;
    if (mx > 0.f) {
      int e;
      const float m = frexpf(mx, &e);           // mx = m * 2^e, m in [0.5, 1);  448 = 0.875 * 2^9
      sc = ldexpf(1.f, m <= 0.875f ? e - 9 : e - 8);
    }
    scale[blockIdx.x] = sc;
  }
}

__global__ void fp8_pack_kernel(const float* __restrict__ src, const float* __restrict__ scale, u32x4* __restrict__ dst,
                                int64_t N, int K) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one lane's 16 bytes
  if (gid >= N * (K / 64) * 4) return;
  const int lane = (int)(gid & 63);
  const int64_t pair = gid >> 6;
  const int kg2 = (int)(pair % (K / 64));
  const int64_t tile = pair / (K / 64);
  const int64_t n = tile * 16 + (lane & 15);
  const float inv = 1.0f / scale[n];            // exact: power of two
  const float* p = src + n * K + (int64_t)kg2 * 64 + 8 * (lane >> 4);
  uint32_t w[4];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float* e = p + 32 * h + 4 * j;
      int v = __builtin_amdgcn_cvt_pk_fp8_f32(e[0] * inv, e[1] * inv, 0, false);
      v = __builtin_amdgcn_cvt_pk_fp8_f32(e[2] * inv, e[3] * inv, v, true);
      w[2 * h + j] = (uint32_t)v;
    }
  dst[gid] = u32x4{w[0], w[1], w[2], w[3]};
}

int va_pack_weight_fp8(const float* src, void* dst, int64_t N, int64_t K, hipStream_t s) {
  if ((N % 16) || (K % 64)) return VAURA_ERR_SHAPE;
  float* scale = reinterpret_cast<float*>(static_cast<char*>(dst) + (size_t)N * (size_t)K);
  VA_LAUNCH(fp8_row_scale_kernel, dim3((unsigned)N), dim3(256), 0, s, src, scale, (int)K);
  const int64_t total = N * (K / 64) * 4;
  VA_LAUNCH(fp8_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, (const float*)scale,
            reinterpret_cast<u32x4*>(dst), N, (int)K);
  return 0;
}

// ---------------------------------------------------------------------------- op-level access
// packed rows (rows x C) fp32 [* gain] -> split rows; optional per-16-column partial sums of squares
__global__ void split_rows_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, const float* __restrict__ gain,
                                  float* __restrict__ ss, int rows_p, int C) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one quad
  if (gid >= (int64_t)rows_p * (C / 4)) return;
  const int row = (int)(gid / (C / 4)), cq = (int)(gid % (C / 4));
  f32x4 v = reinterpret_cast<const f32x4*>(src)[packed_quad(row, cq, C)];
  if (ss) {
    float s = ((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]) + v[3] * v[3];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if ((cq & 3) == 0) ss[((size_t)(row >> 4) * (C / 16) + (cq >> 2)) * 16 + (row & 15)] = s;
  }
  if (gain) v *= *reinterpret_cast<const f32x4*>(gain + cq * 4);
  store_split4(dst, row, cq * 4, C, v);
}

extern "C" {

void vaura_set_debug_flags(unsigned flags) { va_debug_flags = flags; }
void vaura_set_debug_flags2(unsigned flags) { va_debug_flags2 = flags; }

int vaura_split_rows(const float* src, uint16_t* dst, const float* gain, float* ss, int64_t rows, int64_t C, vaura_stream_t s) {
  if (!src || !dst || rows <= 0 || C <= 0 || (C % 16)) return VAURA_ERR_ARG;
  const int rows_p = (int)((rows + 15) / 16 * 16);
  const int64_t total = (int64_t)rows_p * (C / 4);
  VA_LAUNCH(split_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(s), src, dst, gain, ss, rows_p, (int)C);
  return 0;
}

int vaura_gemv_pair(const void* w, int wdtype, const uint16_t* x_split, const float* ss_in, int n_ss_in, const float* residual, float* out,
                    float* out_khalf2, uint16_t* out_split, const float* gain_out, float* ss_out, int64_t rows, int64_t N, int64_t K, int epilogue,
                    float eps, vaura_stream_t s) {
  if (!w || !x_split || rows <= 0) return VAURA_ERR_ARG;
  if (wdtype != VAURA_W_H1 && !va_is_fp8(wdtype) && wdtype != VAURA_W_H2) return VAURA_ERR_DTYPE;
  Gemv3Args a;
  a.wq = wdtype == VAURA_W_FP8H ? 3 : (wdtype == VAURA_W_FP8 ? 1 : (wdtype == VAURA_W_H2 ? 2 : 0)); a.wscale = nullptr; a.out2 = out_khalf2;
  a.W = w; a.XP = x_split; a.ss_in = ss_in; a.n_ss_in = n_ss_in; a.res = residual; a.out = out; a.outp = out_split;
  a.gain_out = gain_out; a.ss_out = ss_out; a.rows = (int)rows; a.R = (int)((rows + 15) / 16);
  a.N = (int)(epilogue == E3_SWIGLU ? N / 2 : N); a.eps = eps; a.k_total = (int)K; a.out_scale = 1.f;
  if (epilogue == E3_RESID && !residual) return VAURA_ERR_ARG;
  return va_launch_gemv3(a, N, K, epilogue, ss_in != nullptr, as_stream(s));
}

}  // extern "C"
