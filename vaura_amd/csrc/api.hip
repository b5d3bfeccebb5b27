// Step driver: chains the kernels of one decode position and the generate() hot loop.
//   VAURAModel.generate loop           models/vaura_model.py:502-547
//   VAURAModel._sample_next_token      models/vaura_model.py:775-827
//   Transformer.inference              models/modules/sampler/llama.py:445-504
//   TransformerBlock.forward           llama.py:272-283
#include "common.h"
#include <algorithm>
#include "gemv3_kernel.h"
#include <vector>

enum { EPI_STORE = 0, EPI_RESID = 1, EPI_SWIGLU = 2, EPI_GELU = 3, EPI_LOGITS = 4 };

// The compiled kernels are specialised for the one geometry every shipped V-AURA config uses
// (configs/modules/samplers/llama_9cbs.yaml: 1536 / 4096 / 16 heads of 96 / 1024-entry codebooks); anything else
// is refused with VAURA_ERR_SHAPE rather than run on the wrong tiling.
static int check_decoder(const vaura_decoder* d) {
  if (!d || !d->layers_host || !d->heads || !d->final_norm || !d->tok_emb || !d->tok_proj_w || !d->tok_proj_b ||
      !d->tok_table || !d->empty_video || !d->rope || !d->cond_proj || !d->kcache || !d->vcache || !d->seq || !d->state || !d->ws_h ||
      !d->ws_qkv || !d->ws_attn || !d->ws_ffn || !d->ws_logits)
    return VAURA_ERR_ARG;
  const vaura_dims& m = d->dims;
  if (m.d_model != 1536 || m.ffn_dim != 4096 || m.n_head != 16 || m.vocab != 1024) return VAURA_ERR_SHAPE;
  if (m.cond_dim + m.tok_dim != m.d_model) return VAURA_ERR_SHAPE;
  if (d->rows != d->batch && d->rows != 2 * d->batch) return VAURA_ERR_ARG;
  if (d->seq_len > d->max_len || d->batch <= 0) return VAURA_ERR_ARG;
  if (d->plane_shift < 0 || d->plane_shift > 24 || (d->plane_shift && !d->ws_h_split)) return VAURA_ERR_ARG;
  if (d->kv_dtype < 0 || d->kv_dtype > 2) return VAURA_ERR_ARG;
  if (d->kv_dtype != 0 && (d->max_len > 256 || !d->ws_h_split)) return VAURA_ERR_SHAPE;      // fp16 / fp8 K / V: the pair path's single-round-trip attention only
  return 0;
}

// Optional per-launch timing (vaura_profile_loop): every launch of the selected kinds carries its own
// start/stop events (see VA_LAUNCH), i.e. the interval rocprofv3's kernel trace reports.
// A step stage may be more than one launch (range-split attention = split + combine): every launch gets its own
// pair, a stage's time is the sum over its launches, and `stages` counts the brackets.
thread_local int va_prof_kind = -1;   // per calling thread: one caller thread per device
struct StepProfiler {
  unsigned mask = 0;
  std::vector<hipEvent_t> ev[VAURA_K_COUNT];
  int64_t stages[VAURA_K_COUNT] = {};
  void before(int kind) {
    if (!(mask & (1u << kind))) return;
    ++stages[kind];
    va_prof_kind = kind;
  }
  void after(int) { va_prof_kind = -1; }
};
static thread_local StepProfiler* g_prof = nullptr;
static thread_local int64_t g_prof_outliers[VAURA_K_COUNT] = {};
void va_prof_events(hipEvent_t* a, hipEvent_t* b) {
  (void)hipEventCreate(a); (void)hipEventCreate(b);
  g_prof->ev[va_prof_kind].push_back(*a); g_prof->ev[va_prof_kind].push_back(*b);
}
#define PROF_B(kind) do { if (g_prof) g_prof->before(kind); } while (0)
#define PROF_A(kind) do { if (g_prof) g_prof->after(kind); } while (0)

static Gemv3Args g3(const void* W, const uint16_t* xp, const float* ss_in, const float* res, float* out, uint16_t* outp,
                    const float* gain_out, float* ss_out, const vaura_decoder* d, int N, int n_pos = 1) {
  Gemv3Args a;
  // VAURA_W_FP8: the four per-layer matrices are fp8; the codebook heads (final logits) stay one fp16 plane.
  // VAURA_W_H2: (hi, lo) fp16 planes — the A operands as loaded
  // VAURA_W_FP8H: those fp8 matrices multiplied against the hi activation plane only (wq 3; the heads keep both planes)
  a.wq = d->wdtype == VAURA_W_H2 ? 2 : ((va_is_fp8(d->wdtype) && W != d->heads) ? (d->wdtype == VAURA_W_FP8H ? 3 : 1) : 0);
  a.wscale = nullptr; a.out2 = nullptr;
  a.W = W; a.XP = xp; a.ss_in = ss_in; a.n_ss_in = d->dims.d_model / 16; a.res = res; a.out = out; a.outp = outp;
  a.gain_out = gain_out; a.ss_out = ss_out;
  a.R = n_pos * ((d->rows + 15) / 16);           // prefill: one group of row blocks per position
  a.rows = n_pos == 1 ? d->rows : a.R * 16;
  a.N = N; a.eps = d->dims.eps;
  a.k_total = d->dims.d_model;
  // plane_shift S: the two plane sets without a norm behind them (SwiGLU output, attention output) are stored times 2^-S, their
  // consumers' matrices (w2, wo) were packed times 2^S by the caller — exact both ways, and 2^S more head-room in the fp16 planes
  a.out_scale = (outp && outp == d->ws_ffn_split) ? ldexpf(1.f, -d->plane_shift) : 1.f;
  return a;
}

// Teacher-forced positions [p0, p0 + n) of the pattern sequence in ONE pass per layer (the prompt of the
// sliding-window caller, scripts/generate.py:327-365): every GEMV keeps its weight slice in registers and
// loops over the positions' row blocks, K/V of the whole chunk are rotated/appended first, then each
// (row, head, position) attends causally over the cache.  No heads, no sampling.
static int enqueue_prefill_chunk_bf16(const vaura_decoder* d, int p0, int n, hipStream_t s) {
  const vaura_dims& m = d->dims;
  const int D = m.d_model, F = m.ffn_dim;
  int rc = va_launch_embed(d, p0, n, s);
  if (rc) return rc;
  for (int l = 0; l < m.n_layer; ++l) {
    const vaura_layer_weights& L = d->layers_host[l];
    const float* next_attn_gain = (l + 1 < m.n_layer) ? d->layers_host[l + 1].attn_norm : d->final_norm;
    rc = va_launch_gemv3(g3(L.wqkv, d->ws_h_split, d->ws_ss, nullptr, d->ws_qkv, nullptr, nullptr, nullptr, d, 3 * D, n), 3 * D, D,
                         E3_STORE, true, s);
    if (rc) return rc;
    rc = va_launch_rope_append(d, l, p0, n, s);
    if (rc) return rc;
    rc = va_launch_attention_prefill(d, l, p0, n, s);
    if (rc) return rc;
    rc = va_launch_gemv3(g3(L.wo, d->ws_attn_split, nullptr, d->ws_h, d->ws_h, d->ws_h_split, L.ffn_norm, d->ws_ss, d, D, n), D, D,
                         E3_RESID, false, s);
    if (rc) return rc;
    rc = va_launch_gemv3(g3(L.w13, d->ws_h_split, d->ws_ss, nullptr, nullptr, d->ws_ffn_split, nullptr, nullptr, d, F, n), 2 * F, D,
                         E3_SWIGLU, true, s);
    if (rc) return rc;
    rc = va_launch_gemv3(g3(L.w2, d->ws_ffn_split, nullptr, d->ws_h, d->ws_h, d->ws_h_split, next_attn_gain, d->ws_ss, d, D, n), D, F,
                         E3_RESID, false, s);
    if (rc) return rc;
  }
  return 0;
}

// pair path (H1 / H2 / FP8 storage): activations travel as (hi, lo) fp16 planes, products on the fp16 MFMA
static int enqueue_step_bf16(const vaura_decoder* d, const vaura_sampling* sp, int sample, hipStream_t s) {
  const vaura_dims& m = d->dims;
  const int D = m.d_model, F = m.ffn_dim, H = m.n_head, hd = D / H;
  const int rows = d->rows;
  if (!d->ws_h_split || !d->ws_attn_split || !d->ws_ffn_split || !d->ws_ss) return VAURA_ERR_ARG;
  PROF_B(VAURA_K_EMBED);
  int rc = va_launch_embed(d, -1, 1, s);   // h, split(h * attn_norm[0]), ss partials
  PROF_A(VAURA_K_EMBED);
  if (rc) return rc;
  const size_t kv_layer = (size_t)rows * H * (size_t)d->max_len * hd;
  // K-split qkv (consumer-reduced): fewer than 16 row blocks (otherwise the GEMM tiling takes over)
  float* qkv2 = (d->ws_qkv2 && (rows + 15) / 16 < 16) ? d->ws_qkv2 : nullptr;
  // the MLP half of every layer as ONE launch (mlp_engine.h) where the shape is eligible and the caller provided the hand-off
  // flags (dec->ws_sync); debug flag bit 2: two launches whatever the shape (the A/B, and the path every other shape takes)
  const bool mlp_engine = !(va_debug_flags_get() & 4u) && va_mlp_engine_eligible(d);
  // ... and the NEXT layer's qkv GEMV as a third phase of that launch
  const bool fuse_qkv = mlp_engine && qkv2 && !(va_debug_flags_get() & 0x2u);   // debug flag bit 1: qkv stays its own launch
  bool qkv_done = false;                     // layer l's qkv partials were written by layer l - 1's engine launch
  // EXPERIMENT builds only (-DVAURA_EXPERIMENT_ENGINES; DESIGN_HISTORY.md rounds 4-5, all bit-identical and measured slower or no
  // faster): debug flag bit 12 = attention + wo as one launch, bit 3 = the whole layer tail as one launch, second flag word bit 2 = the
  // next layer's attention as a fourth phase of the one-launch MLP.  The product library compiles none of them.
#ifdef VAURA_EXPERIMENT_ENGINES
  const bool attn_wo = mlp_engine && rows <= 16 && H == 16 && d->max_len <= 256 && (va_debug_flags_get() & 0x1000u) && !(va_debug_flags_get() & 8u) &&
                       d->plane_shift == 0;   // the experiments' own attention epilogues store unscaled planes
  const bool fuse_attn = fuse_qkv && rows <= 16 && H == 16 && hd == 96 && d->max_len <= 256 && d->ws_attn_split && !attn_wo && d->plane_shift == 0 &&
                         !(va_debug_flags_get() & 8u) && (va_debug_flags2_get() & 4u);
  const bool tail_engine = mlp_engine && rows <= 16 && (va_debug_flags_get() & 8u);
#else
  constexpr bool attn_wo = false, fuse_attn = false;
#endif
  bool attn_done = false;                    // layer l's attention was computed by layer l - 1's engine launch
  for (int l = 0; l < m.n_layer; ++l) {
    const vaura_layer_weights& L = d->layers_host[l];
    const float* next_attn_gain = (l + 1 < m.n_layer) ? d->layers_host[l + 1].attn_norm : d->final_norm;
    if (!qkv_done) {
      PROF_B(VAURA_K_QKV);   // qkv = rinv * Wqkv.(g*h)                                  llama.py:280, 228
      Gemv3Args aq = g3(L.wqkv, d->ws_h_split, d->ws_ss, nullptr, d->ws_qkv, nullptr, nullptr, nullptr, d, 3 * D);
      aq.out2 = qkv2;        // two K-half partials, added by the attention kernel on load
      rc = va_launch_gemv3(aq, 3 * D, D, E3_STORE, true, s);
      PROF_A(VAURA_K_QKV);
      if (rc) return rc;
    }
    qkv_done = false;
#ifdef VAURA_EXPERIMENT_ENGINES
    const Gemv3Args awo0 = g3(L.wo, d->ws_attn_split, nullptr, d->ws_h, d->ws_h, d->ws_h_split, L.ffn_norm, d->ws_ss, d, D);
#endif
    if (attn_done) {
      attn_done = false;                     // (computed by the previous layer's one-launch MLP)
#ifdef VAURA_EXPERIMENT_ENGINES
    } else if (attn_wo) {
      PROF_B(VAURA_K_ATTN);
      rc = va_launch_attn_wo(d->ws_qkv, qkv2, d->rope, d->kcache + l * kv_layer, d->vcache + l * kv_layer, d->ws_attn, d->ws_attn_split,
                             rows, H, d->max_len, d->state, awo0, d->ws_sync + 512, l, s);
      PROF_A(VAURA_K_ATTN);
      if (rc) return rc;
#endif
    } else {
    PROF_B(VAURA_K_ATTN);  // rope + cache append + softmax(qK^T)V                     llama.py:234-257
    rc = va_launch_attention(d->ws_qkv, qkv2, d->rope, va_kv_layer(d, d->kcache, l), va_kv_layer(d, d->vcache, l), d->ws_attn,
                             d->ws_attn_split, rows, H, hd, d->max_len, d->state, 0, d->ws_attn_part,
                             d->ws_attn_part ? va_attention_splits(rows, H, d->max_len) : 1, s,
                             d->ws_sync ? d->ws_sync + 512 : nullptr, ldexpf(1.f, -d->plane_shift), d->kv_dtype);      // words 512 .. 767: arrival counts of the range-split attention (the MLP / tail engines use 0 .. 511; the attention + wo experiment uses 512 .. only with caches <= 256, where nothing is split)
    PROF_A(VAURA_K_ATTN);
    if (rc) return rc;
    }
    const Gemv3Args awo = g3(L.wo, d->ws_attn_split, nullptr, d->ws_h, d->ws_h, d->ws_h_split, L.ffn_norm, d->ws_ss, d, D);
#ifdef VAURA_EXPERIMENT_ENGINES
    if (tail_engine) {
      PROF_B(VAURA_K_W13);
      rc = va_launch_tail_engine(awo, g3(L.w13, d->ws_h_split, d->ws_ss, nullptr, nullptr, d->ws_ffn_split, nullptr, nullptr, d, F),
                                 g3(L.w2, d->ws_ffn_split, nullptr, d->ws_h, d->ws_h, d->ws_h_split, next_attn_gain, d->ws_ss, d, D),
                                 d->ws_sync, d->state, l, s);
      PROF_A(VAURA_K_W13);
      if (rc) return rc;
      continue;
    }
#endif
    if (!attn_wo) {
      PROF_B(VAURA_K_WO);    // h += Wo.attn ; emit split(h * ffn_norm) + ss               llama.py:259, 279
      rc = va_launch_gemv3(awo, D, D, E3_RESID, false, s);
      PROF_A(VAURA_K_WO);
      if (rc) return rc;
    }
    const Gemv3Args a13 = g3(L.w13, d->ws_h_split, d->ws_ss, nullptr, nullptr, d->ws_ffn_split, nullptr, nullptr, d, F);
    const Gemv3Args a2 = g3(L.w2, d->ws_ffn_split, nullptr, d->ws_h, d->ws_h, d->ws_h_split, next_attn_gain, d->ws_ss, d, D);
    if (mlp_engine) {
      // ONE launch: w1||w3 + SwiGLU -> in-launch hand-off of the ffn planes -> w2 + residual, w2's weights requested ahead of the
      // hand-off (csrc/mlp_engine.h); booked under w13 by the per-launch profiler
      Gemv3Args aqn;
      const bool with_qkv = fuse_qkv && l + 1 < m.n_layer;
      if (with_qkv) {
        aqn = g3(d->layers_host[l + 1].wqkv, d->ws_h_split, d->ws_ss, nullptr, d->ws_qkv, nullptr, nullptr, nullptr, d, 3 * D);
        aqn.out2 = qkv2;
      }
      VaEngineAttention att;
      const bool with_attn = with_qkv && fuse_attn;
      if (with_attn)
        att = VaEngineAttention{d->rope, d->kcache + (size_t)(l + 1) * kv_layer, d->vcache + (size_t)(l + 1) * kv_layer, d->ws_attn,
                                d->ws_attn_split, H, d->max_len};
      PROF_B(VAURA_K_W13);
      // what the idle workgroups of this launch may warm in the Infinity Cache: the NEXT layer's w1||w3 (its first consumer)
      const void* warm = l + 1 < m.n_layer ? d->layers_host[l + 1].w13 : nullptr;
      const size_t warm_bytes = (size_t)2 * F * D * (d->wdtype == VAURA_W_H2 ? 4 : (va_is_fp8(d->wdtype) ? 1 : 2));
      rc = va_launch_mlp_engine(a13, a2, with_qkv ? &aqn : nullptr, d->ws_sync, d->state, l, s, with_attn ? &att : nullptr, warm, warm_bytes,
                                l + 1 < m.n_layer ? d->layers_host[l + 1].wo : nullptr, warm_bytes * D / (2 * F));
      PROF_A(VAURA_K_W13);
      if (rc) return rc;
      qkv_done = with_qkv;
      attn_done = with_attn;
      continue;
    }
    PROF_B(VAURA_K_W13);   // ffn = silu(W1 x) * (W3 x), x = rmsnorm(h)                  llama.py:282, 177
    rc = va_launch_gemv3(a13, 2 * F, D, E3_SWIGLU, true, s);
    PROF_A(VAURA_K_W13);
    if (rc) return rc;
    PROF_B(VAURA_K_W2);    // h += W2.ffn ; emit split(h * next attention_norm) + ss     llama.py:177, 282
    rc = va_launch_gemv3(a2, D, F, E3_RESID, false, s);
    PROF_A(VAURA_K_W2);
    if (rc) return rc;
  }
  if (!sample) return va_launch_advance(d->state, -1, s);
  PROF_B(VAURA_K_HEADS);   // logits = heads . rmsnorm(h)                               llama.py:503-504
  rc = va_launch_gemv3(g3(d->heads, d->ws_h_split, d->ws_ss, nullptr, d->ws_logits, nullptr, nullptr, nullptr, d,
                          m.n_codebooks * m.vocab), (int64_t)m.n_codebooks * m.vocab, D, E3_LOGITS, true, s);
  PROF_A(VAURA_K_HEADS);
  if (rc) return rc;
  PROF_B(VAURA_K_SAMPLE);
  rc = va_launch_sample(d->ws_logits, d->batch, m.n_codebooks, m.vocab, sp, d->noise, d->batch * m.n_codebooks, d->state, 0,
                        nullptr, d->seq, d->timesteps, d->seq_len, d->state, s);
  PROF_A(VAURA_K_SAMPLE);
  return rc;
}

static int enqueue_step(const vaura_decoder* d, const vaura_sampling* sp, int sample, hipStream_t s) {
  // H1 / H2 / FP8: the pair path; F32 / BF16 tiles: the exact-fp32-MFMA GEMVs (gemv_kernel.h)
  if (d->wdtype == VAURA_W_H1 || d->wdtype == VAURA_W_H2 || va_is_fp8(d->wdtype)) return enqueue_step_bf16(d, sp, sample, s);
  if (d->wdtype != VAURA_W_F32 && d->wdtype != VAURA_W_BF16) return VAURA_ERR_DTYPE;
  const vaura_dims& m = d->dims;
  const int D = m.d_model, F = m.ffn_dim, H = m.n_head, hd = D / H;
  const int rows = d->rows;
  PROF_B(VAURA_K_EMBED);
  int rc = va_launch_embed(d, -1, 1, s);
  PROF_A(VAURA_K_EMBED);
  if (rc) return rc;
  const size_t kv_layer = (size_t)rows * H * (size_t)d->max_len * hd;
  for (int l = 0; l < m.n_layer; ++l) {
    const vaura_layer_weights& L = d->layers_host[l];
    // h -> qkv  (attention_norm fused)                                   llama.py:280, 228
    PROF_B(VAURA_K_QKV);
    rc = va_launch_gemv(L.wqkv, d->wdtype, d->ws_h, L.attn_norm, nullptr, d->ws_qkv, rows, 3 * D, D, EPI_STORE, m.eps, s);
    PROF_A(VAURA_K_QKV);
    if (rc) return rc;
    // rope + cache append + softmax(qK^T)V                               llama.py:234-257
    PROF_B(VAURA_K_ATTN);
    rc = va_launch_attention(d->ws_qkv, nullptr, d->rope, d->kcache + l * kv_layer, d->vcache + l * kv_layer, d->ws_attn, nullptr, rows,
                             H, hd, d->max_len, d->state, 0, d->ws_attn_part,
                             d->ws_attn_part ? va_attention_splits(rows, H, d->max_len) : 1, s);
    PROF_A(VAURA_K_ATTN);
    if (rc) return rc;
    // h += wo . attn                                                      llama.py:259, 279
    PROF_B(VAURA_K_WO);
    rc = va_launch_gemv(L.wo, d->wdtype, d->ws_attn, nullptr, d->ws_h, d->ws_h, rows, D, D, EPI_RESID, 0.f, s);
    PROF_A(VAURA_K_WO);
    if (rc) return rc;
    // ffn = silu(w1 x) * (w3 x)  (ffn_norm fused)                        llama.py:282, 177
    PROF_B(VAURA_K_W13);
    rc = va_launch_gemv(L.w13, d->wdtype, d->ws_h, L.ffn_norm, nullptr, d->ws_ffn, rows, 2 * F, D, EPI_SWIGLU, m.eps, s);
    PROF_A(VAURA_K_W13);
    if (rc) return rc;
    // h += w2 . ffn                                                       llama.py:177, 282
    PROF_B(VAURA_K_W2);
    rc = va_launch_gemv(L.w2, d->wdtype, d->ws_ffn, nullptr, d->ws_h, d->ws_h, rows, D, F, EPI_RESID, 0.f, s);
    PROF_A(VAURA_K_W2);
    if (rc) return rc;
  }
  if (!sample) return va_launch_advance(d->state, -1, s);
  // logits = heads . norm(h)                                              llama.py:503-504
  PROF_B(VAURA_K_HEADS);
  rc = va_launch_gemv(d->heads, d->wdtype, d->ws_h, d->final_norm, nullptr, d->ws_logits, rows, (int64_t)m.n_codebooks * m.vocab, D,
                      EPI_LOGITS, m.eps, s);
  PROF_A(VAURA_K_HEADS);
  if (rc) return rc;
  PROF_B(VAURA_K_SAMPLE);
  rc = va_launch_sample(d->ws_logits, d->batch, m.n_codebooks, m.vocab, sp, d->noise, d->batch * m.n_codebooks, d->state, 0,
                          nullptr, d->seq, d->timesteps, d->seq_len, d->state, s);
  PROF_A(VAURA_K_SAMPLE);
  return rc;
}

// one captured decode step; owned by the caller (one per decoder descriptor / sampling setup)
// One captured decode step, and the same step captured `multi` times in a row (the position, the step counter and the noise index
// live on the device, so every copy is the same list of launches): the loop replays the long graph while at least `multi` steps
// remain — one graph launch (host call, packet-queue doorbell, start-of-graph work on the GPU) per `multi` steps instead of per step.
#ifndef VA_GRAPH_STEPS
#define VA_GRAPH_STEPS 4
#endif
struct StepGraph {
  hipGraph_t graph = nullptr, graphm = nullptr;
  hipGraphExec_t exec = nullptr, execm = nullptr;
  int multi = 1;
  unsigned flags = 0, flags2 = 0;      // debug flags at build time: the lazily captured multi-step graph is the same variant
};

extern unsigned va_debug_flags, va_debug_flags2;   // gemv3.hip
static int build_multi(StepGraph* g, const vaura_decoder* dec, const vaura_sampling* sp, hipStream_t st) {
  hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) return (int)e;
  int rc = 0;
  const unsigned now = va_debug_flags, now2 = va_debug_flags2;
  va_debug_flags = g->flags;                        // the kernel variants of the handle, whatever the caller has set since
  va_debug_flags2 = g->flags2;
  for (int i = 0; i < g->multi && !rc; ++i) rc = enqueue_step(dec, sp, 1, st);
  va_debug_flags = now;
  va_debug_flags2 = now2;
  e = hipStreamEndCapture(st, &g->graphm);
  if (rc) return rc;
  if (e != hipSuccess) return (int)e;
  e = hipGraphInstantiate(&g->execm, g->graphm, nullptr, nullptr, 0);
  return e == hipSuccess ? 0 : (int)e;
}

extern "C" {

const char* vaura_version(void) { return "vaura_hip 0.1 (gfx950)"; }

size_t vaura_struct_size(int which) {
  switch (which) {
    case 0: return sizeof(vaura_dims);
    case 1: return sizeof(vaura_layer_weights);
    case 2: return sizeof(vaura_sampling);
    case 3: return sizeof(vaura_decoder);
    case 4: return sizeof(vaura_conv);
    case 5: return sizeof(vaura_codec);
    case 6: return sizeof(vaura_codec_encoder);
    case 7: return sizeof(vaura_vit);
    case 8: return sizeof(vaura_vit_block);
    default: return 0;
  }
}

int vaura_decode_step(const vaura_decoder* dec, const vaura_sampling* sp, int sample, vaura_stream_t s) {
  int rc = check_decoder(dec);
  if (rc) return rc;
  if (sample && !sp) return VAURA_ERR_ARG;
  return enqueue_step(dec, sp, sample, as_stream(s));
}

void vaura_step_graph_free(vaura_step_graph_t g_) {
  StepGraph* g = static_cast<StepGraph*>(g_);
  if (!g) return;
  if (g->exec) (void)hipGraphExecDestroy(g->exec);
  if (g->graph) (void)hipGraphDestroy(g->graph);
  if (g->execm) (void)hipGraphExecDestroy(g->execm);
  if (g->graphm) (void)hipGraphDestroy(g->graphm);
  delete g;
}

int vaura_step_graph_build(const vaura_decoder* dec, const vaura_sampling* sp, vaura_stream_t s, vaura_step_graph_t* out) {
  int rc = check_decoder(dec);
  if (rc) return rc;
  if (!sp || !out) return VAURA_ERR_ARG;
  *out = nullptr;
  StepGraph* g = new StepGraph();
  hipStream_t st = as_stream(s);
  hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) { delete g; return (int)e; }
  rc = enqueue_step(dec, sp, 1, st);
  e = hipStreamEndCapture(st, &g->graph);
  if (rc) { vaura_step_graph_free(g); return rc; }
  if (e != hipSuccess) { vaura_step_graph_free(g); return (int)e; }
  e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
  if (e != hipSuccess) { vaura_step_graph_free(g); return (int)e; }
  // steps per graph launch: bits 24..27 of vaura_set_debug_flags (0 = the default VA_GRAPH_STEPS), clamped to 1..16.  The long
  // graph is captured lazily by the first vaura_generate_loop that has at least that many steps to run (build_multi).
  g->flags = va_debug_flags_get();
  g->flags2 = va_debug_flags2_get();
  const int req = (int)((g->flags >> 24) & 15u);
  g->multi = req ? std::min(16, std::max(1, req)) : VA_GRAPH_STEPS;
  *out = g;
  return 0;
}

int vaura_generate_loop(const vaura_decoder* dec, const vaura_sampling* sp, int n_prefill, int n_steps,
                        vaura_step_graph_t graph, vaura_stream_t s) {
  int rc = check_decoder(dec);
  if (rc) return rc;
  if (!sp || n_prefill < 0 || n_steps < 0) return VAURA_ERR_ARG;
  // the loop feeds positions [0, n_prefill + n_steps) and the sampler writes seq[..., position + 1]: the last written
  // slot must exist (the reference's loop is range(start, S), vaura_model.py:502), and so must its K/V rows
  if (n_prefill + n_steps > dec->seq_len - 1 || n_prefill + n_steps > dec->max_len) return VAURA_ERR_ARG;
  hipStream_t st = as_stream(s);
  if (n_prefill > 0 && dec->ws_h_split && dec->prefill_positions > 0) {
    // the caller guarantees state[0] == 0 at entry (vaura_pattern_build + zeroed state)
    for (int p0 = 0; p0 < n_prefill; p0 += dec->prefill_positions) {
      const int n = (n_prefill - p0 < dec->prefill_positions) ? n_prefill - p0 : dec->prefill_positions;
      rc = enqueue_prefill_chunk_bf16(dec, p0, n, st);
      if (rc) return rc;
    }
    rc = va_launch_advance(dec->state, n_prefill, st);
    if (rc) return rc;
  } else {
    for (int i = 0; i < n_prefill; ++i) {
      rc = enqueue_step(dec, sp, 0, st);
      if (rc) return rc;
    }
  }
  if (graph) {
    StepGraph* g = static_cast<StepGraph*>(graph);
    if (!g->exec) return VAURA_ERR_STATE;
    int i = 0;
    if (!g->execm && g->multi > 1 && n_steps >= g->multi) {   // first loop long enough to use it: capture the multi-step graph now
      rc = build_multi(g, dec, sp, st);
      if (rc) return rc;
    }
    if (g->execm)
      for (; i + g->multi <= n_steps; i += g->multi) {
        hipError_t e = hipGraphLaunch(g->execm, st);
        if (e != hipSuccess) return (int)e;
      }
    for (; i < n_steps; ++i) {
      hipError_t e = hipGraphLaunch(g->exec, st);
      if (e != hipSuccess) return (int)e;
    }
    return 0;
  }
  for (int i = 0; i < n_steps; ++i) {
    rc = enqueue_step(dec, sp, 1, st);
    if (rc) return rc;
  }
  return 0;
}

int vaura_profile_loop(const vaura_decoder* dec, const vaura_sampling* sp, int n_steps, unsigned kind_mask,
                       double* total_ms_host, int64_t* launches_host, vaura_stream_t s) {
  int rc = check_decoder(dec);
  if (rc) return rc;
  if (!sp || n_steps <= 0 || !total_ms_host || !launches_host) return VAURA_ERR_ARG;
  StepProfiler prof;
  prof.mask = kind_mask;
  g_prof = &prof;
  hipStream_t st = as_stream(s);
  for (int i = 0; i < n_steps && !rc; ++i) rc = enqueue_step(dec, sp, 1, st);
  g_prof = nullptr;
  hipError_t e = hipStreamSynchronize(st);
  for (int k = 0; k < VAURA_K_COUNT; ++k) {
    const size_t n = prof.ev[k].size() / 2;
    std::vector<float> t(n, 0.f);
    for (size_t i = 0; i < n; ++i)
      if (hipEventElapsedTime(&t[i], prof.ev[k][2 * i], prof.ev[k][2 * i + 1]) != hipSuccess) t[i] = 0.f;
    for (hipEvent_t ev : prof.ev[k]) (void)hipEventDestroy(ev);
    // A launch whose interval is more than 10x its kind's median is a stalled queue (one 190 ms interval among 5 472 launches
    // of 5.5 us was seen once), not the kernel: it is counted at the median and reported by vaura_profile_outliers.
    double tot = 0.0;
    g_prof_outliers[k] = 0;
    if (n) {
      std::vector<float> srt(t);
      std::nth_element(srt.begin(), srt.begin() + n / 2, srt.end());
      const float med = srt[n / 2];
      for (size_t i = 0; i < n; ++i) {
        if (t[i] > 10.f * med) { ++g_prof_outliers[k]; tot += med; }
        else tot += t[i];
      }
    }
    total_ms_host[k] = tot;
    launches_host[k] = prof.stages[k];
  }
  if (rc) return rc;
  return e == hipSuccess ? 0 : (int)e;
}

void vaura_profile_outliers(int64_t* per_kind) {
  if (per_kind) for (int k = 0; k < VAURA_K_COUNT; ++k) per_kind[k] = g_prof_outliers[k];
}

}  // extern "C"
