/*
 * vaura_hip.h — C ABI of libvaura_hip.so, the MI355X (gfx950) generation hot path of V-AURA.
 *
 * The reference (ilpoviertola/V-AURA) has no native code and no FFI: its hot path is the Python
 * call chain  VAURAModel.generate() -> _sample_next_token() -> llama.Transformer.forward()
 * -> sample_top_k/top_p -> DacModelWrapper.decode().  This header is the native boundary a
 * maintainer binds *under* those Python plugin classes (ctypes stub: INTEGRATION.md).  Each entry
 * point cites the reference interface it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - extern "C", plain pointers / sizes / a POD descriptor struct; no torch types.
 *   - every pointer is a DEVICE pointer unless its name ends in _host.
 *   - returns 0 on success, a negative vaura_status on an argument error, a positive value = hipError_t.
 *   - never allocates device memory, never synchronises the stream, never throws.
 *   - everything is enqueued on the hipStream_t that is passed in (pass torch's current stream).
 *   - one caller thread per device; re-entrant across devices, not within one descriptor.
 *
 * Activation layout ("packed rows"): a (rows x C) fp32 matrix is stored in blocks of 16 rows as
 *     [row_block][C/4][16 rows][4 cols]   ->  float index  ((rb*(C/4) + c/4)*16 + r%16)*4 + c%4
 *   so that one 64-lane wavefront reads/writes one 16x16 MFMA operand tile as a contiguous 1 KiB.
 *   rows beyond the live ones in the last block must be zero.
 *
 * Streamed-weight layout ("MFMA tiles"): an (N x K) matrix, N%16==0, K%32==0, is stored as
 *     [N/16][K/32][64 lanes][8]  with lane = (n%16) + 16*((k%32)/8), element j = k%8
 *   (VAURA_W_F32: [N/16][K/32][2 halves][64 lanes][4] fp32;  VAURA_W_BF16: 8 bf16 per lane;
 *    VAURA_W_H1:  8 fp16 per lane = W[n,k] / scale[n], followed by float scale[N] (power of two, max|W[n,:]| / scale in [2^13, 2^14));
 *    VAURA_W_H2:  [N/16][K/32][2 planes][64 lanes][8 fp16]: hi = fp16(W/scale), lo = fp16(W/scale - hi), followed by scale[N];
 *    VAURA_W_FP8: [N/16][K/64][64 lanes][16 bytes] = the lane's 8 values of the even then of the odd k-group, K%64==0,
 *          followed by float scale[N], scale[n] = smallest power of two with max|W[n,:]| <= 448*scale[n]);
 *   see vaura_pack_weight().
 */
#ifndef VAURA_HIP_H
#define VAURA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* vaura_stream_t; /* hipStream_t */

typedef enum vaura_status {
  VAURA_OK = 0,
  VAURA_ERR_ARG = -1,       /* null pointer / size out of range            */
  VAURA_ERR_SHAPE = -2,     /* dims not supported by the compiled kernels  */
  VAURA_ERR_DTYPE = -3,
  VAURA_ERR_STATE = -4      /* e.g. step graph not built                   */
} vaura_status;

/* storage of the streamed matrices.
 *   VAURA_W_H2  (hi, lo) fp16 planes + power-of-two row scales: 22 significand bits, 4 bytes per weight — fp32 checkpoints on
 *               the fp16-pair decode kernels (gemv3_kernel.h); the default for real (fp32) checkpoints
 *   VAURA_W_H1  one fp16 plane + row scales, 2 bytes per weight: lossless for checkpoints whose weights fit 11 significand
 *               bits (bf16-representable ones: 8)
 *   VAURA_W_FP8 (BASELINE configs[4]; no reference counterpart): OCP e4m3 with one power-of-two scale per output row, for the
 *               four per-layer matrices; heads stay VAURA_W_H1.  Multiplied against BOTH activation planes: exactly the H1
 *               arithmetic on the dequantised checkpoint (tested as such)
 *   VAURA_W_FP8H (round 6, configs[4]'s measured configuration): the SAME packed bytes as VAURA_W_FP8, multiplied against the hi
 *               activation plane only (11-bit activations under 4-bit-significand weights): half the plane bytes through every CU,
 *               half the matrix instructions; tolerance against VAURA_W_FP8 / H1 reported by the tests and bench.py, not bit parity
 *   VAURA_W_F32 / VAURA_W_BF16: fp32 / bf16 MFMA tiles of the exact-fp32-MFMA GEMVs (gemv_kernel.h): the conditioning MLP, and
 *               the decode step when the split workspaces are NULL (an exact, slower cross-check)                        */
typedef enum vaura_wdtype { VAURA_W_F32 = 0, VAURA_W_BF16 = 1, VAURA_W_FP8 = 2, VAURA_W_H1 = 3, VAURA_W_H2 = 4, VAURA_W_FP8H = 5 } vaura_wdtype;

/* ---- model geometry: configs/modules/samplers/llama_9cbs.yaml:3-17 + sampler/llama.py:308-361 */
typedef struct vaura_dims {
  int32_t n_layer;      /* 24   */
  int32_t d_model;      /* 1536 */
  int32_t n_head;       /* 16   (head_dim = d_model / n_head = 96) */
  int32_t ffn_dim;      /* 4096 */
  int32_t n_codebooks;  /* 9    */
  int32_t vocab;        /* 1024 (special token id == vocab) */
  int32_t cond_dim;     /* 512  */
  int32_t tok_dim;      /* 1024 */
  int32_t cond_in;      /* 768  */
  int32_t codebook_dim; /* 8    */
  int32_t tokens_per_frame; /* 7, scripts/generate.py:216 */
  float   eps;          /* 1e-5 */
} vaura_dims;

/* ---- per-layer streamed weights, MFMA-tile layout, dtype = vaura_decoder.wdtype */
typedef struct vaura_layer_weights {
  const void*  wqkv;       /* (3*d_model x d_model)                       llama.py:211 */
  const void*  wo;         /* (d_model x d_model)                         llama.py:212 */
  const void*  w13;        /* (2*ffn x d_model): 16-row tiles interleaved w1,w3   llama.py:171-172 */
  const void*  w2;         /* (d_model x ffn)                             llama.py:173 */
  const float* attn_norm;  /* (d_model) gain                              llama.py:268 */
  const float* ffn_norm;   /* (d_model) gain                              llama.py:269 */
} vaura_layer_weights;

/* ---- sampling parameters: VAURAModel._sample_next_token, models/vaura_model.py:775-827 */
typedef struct vaura_sampling {
  int32_t use_sampling;   /* 0 -> greedy argmax of the logits (vaura_model.py:825) */
  float   temp;           /* <= 0 -> greedy                                        */
  int32_t top_k;          /* used when top_p <= 0 and top_k > 0 (utils/utils.py:163-178) */
  float   top_p;          /* > 0 wins over top_k (vaura_model.py:818-819; utils/utils.py:181-196) */
  float   cfg_scale;      /* > 1 -> rows [B,2B) are the null-condition branch (vaura_model.py:786-813) */
  uint64_t seed;          /* Philox key when noise == NULL                          */
  uint64_t clip_base;     /* global index of clip 0 (keeps draws invariant to batch sharding) */
  int32_t input_is_probs; /* 1: the input rows already are probabilities (utils/utils.py sample_top_k / sample_top_p / multinomial
                             take probs): no temperature, no softmax, no CFG mix.  0 in the decode loop                       */
  float   tie_eps;        /* near-tie detector (round 6), 0 = off: RELATIVE bound on the error of a logit as the plane storages deliver it
                             (x the row's largest |logit|, x (2 cfg_scale - 1) through the CFG mix).  A used decision whose own margin is
                             inside twice that bound — greedy: top-1 - top-2 of the mixed logits; sampled: the runner-up of argmax(p / q),
                             and the top-k threshold where it could change the draw — is counted in state[6] (state[7] = first such
                             step + 1) and raises VAURA_STATUS_NEAR_TIE.  The token chosen is never changed by the detector           */
} vaura_sampling;

/* ---- everything one decode step touches.  All buffers are owned by the caller (torch tensors). */
#define VAURA_STATUS_NONFINITE_LOGITS 1
#define VAURA_STATUS_HANDOFF_TIMEOUT 2    /* a consumer of the one-launch MLP (csrc/mlp_engine.h) gave up waiting for its producers */
#define VAURA_STATUS_NEAR_TIE 4           /* informational: >= 1 used decision of the sampler was inside the arithmetic's noise (vaura_sampling.tie_eps) */

typedef struct vaura_decoder {
  vaura_dims dims;
  int32_t wdtype;          /* vaura_wdtype of the streamed matrices                */
  int32_t batch;           /* B  = clips                                           */
  int32_t rows;            /* Bs = B, or 2B when cfg_scale > 1                     */
  int32_t max_len;         /* KV capacity in positions (>= S)                      */
  int32_t timesteps;       /* T  = max_new_tokens                                  */
  int32_t seq_len;         /* S  = T + n_codebooks                                 */
  int32_t n_cond_tokens;   /* Tv                                                   */
  int32_t prefill_positions; /* > 0: every ws_* buffer holds this many positions' worth of row blocks, so a
                                prompt is teacher-forced in chunks of that many positions per pass (bf16 path) */
  int32_t plane_shift;     /* S in [0, 24], pair path only (0 otherwise): the two activation plane sets that have no RMSNorm in front of
                              them — the attention output (ws_attn_split) and silu(w1 x) * (w3 x) (ws_ffn_split) — are stored times
                              2^-S, and the caller packed wo and w2 times 2^S (vaura_pack_weight of the scaled matrix: the row scales
                              are powers of two, so the tiles are the same bits).  Exact in both directions unless a plane value drops
                              into fp16's subnormals; buys 2^S of head-room before |activation| > 65504 raises
                              VAURA_STATUS_NONFINITE_LOGITS.  0 = the layout every parity number was taken on */
  int32_t kv_dtype;        /* 0: the K / V cache is fp32 (every parity number).  1 (round 6; the low-precision serving configuration, BASELINE
                              configs[4]): fp16 — kcache / vcache then point at (n_layer, rows, n_head, max_len, head_dim) HALVES holding
                              fp16(rotated k) / fp16(v); caches of at most 256 positions only (VAURA_ERR_SHAPE otherwise); tolerance reported.
                              2: OCP e4m3 bytes of the same layout (unscaled, saturating at +-448): a quarter of the fp32 stream, ~1e-2 class */

  const vaura_layer_weights* layers_host; /* HOST array [n_layer] of device pointers */
  const void*  heads;        /* (n_codebooks*vocab x d_model) MFMA tiles (VAURA_W_H1 when wdtype is FP8) llama.py:356-361 */
  const float* final_norm;   /* (d_model)                                          llama.py:355 */
  const float* tok_emb;      /* (K, vocab+1, codebook_dim)                         llama.py:392-404 */
  const float* tok_proj_w;   /* (K, tok_dim, codebook_dim) weight-norm folded      llama.py:405-409 */
  const float* tok_proj_b;   /* (K, tok_dim)                                                */
  const float* tok_table;    /* (K, vocab+1, tok_dim) = vaura_build_token_table(tok_emb, tok_proj_w, tok_proj_b) */
  const float* empty_video;  /* (cond_dim)                                         llama.py:336-338 */
  const float* rope;         /* (max_len, head_dim/2, 2) cos,sin                   llama.py:593-603 */
  const float* cond_proj;    /* packed rows (rows*Tv x cond_dim): vaura_prefill_cond output */

  float*   kcache;           /* (n_layer, rows, n_head, max_len, head_dim) fp32 (fp16 when kv_dtype = 1), rotated keys */
  float*   vcache;           /* same shape                                          */
  int32_t* seq;              /* (B, K, S) pattern sequence, -1 = unknown            vaura_model.py:485-493 */
  int32_t* state;            /* 8 words: [0]=position of the token being fed, [1]=arrival counter, [2]=step index, [3] sequence id,
                                [4]=STATUS bits, sticky until the caller clears them (VAURA_STATUS_*): the sampler raises
                                VAURA_STATUS_NONFINITE_LOGITS when a logit it is about to sample from is inf / NaN — which is
                                where every overflow of the fp16-plane activation format ends up (|activation| > 65504 -> inf
                                in the hi plane -> NaN in the residual stream).  [5]=launch-epoch counter of the in-launch hand-offs,
                                OWNED BY THE LIBRARY: whatever ends a decode step (sampler, teacher-forced advance) bumps it and
                                nothing rewinds it.  A hand-off epoch is (state[3] sequence id, state[5], layer): rewinding [0]
                                inside a sequence is safe; a NEW sequence, a state buffer that starts from zero again, or another
                                decoder instance in the same process must come with a new sequence id in [3] (10 bits are
                                used) — the arrival words of the hand-offs live in LDS and outlive launches.  [6] = near-tie decisions counted
                                since the caller last zeroed it, [7] = step index + 1 of the first of them (vaura_sampling.tie_eps) */
  const float* noise;        /* optional (n_steps, B*K, vocab) Exp(1) draws; NULL -> Philox */

  float* ws_h;               /* packed rows (rows x d_model) residual stream        */
  float* ws_qkv;             /* packed rows (rows x 3*d_model)                      */
  float* ws_qkv2;            /* optional, same shape (one position's worth): on the pair path the decode step's qkv GEMV
                                then runs as two K-half workgroup sets (ws_qkv, ws_qkv2) that attention adds on load   */
  float* ws_attn;            /* packed rows (rows x d_model)                        */
  float* ws_ffn;             /* packed rows (rows x ffn_dim)                        */
  float* ws_logits;          /* row-major (rows, K*vocab)                           */
  /* pair path (wdtype H1 / H2 / FP8): activations as (hi, lo) fp16 planes ("split rows", 2 * rows_padded * C fp16, see
   * vaura_amd/csrc/gemv3_kernel.h) and per-tile partial sums of squares for the fused RMSNorm.  NULL with wdtype F32 / BF16
   * selects the exact-fp32-MFMA step                                                                              */
  uint16_t* ws_h_split;      /* split rows (rows x d_model) of h * next_norm_gain   */
  uint16_t* ws_attn_split;   /* split rows (rows x d_model)                         */
  uint16_t* ws_ffn_split;    /* split rows (rows x ffn_dim)                         */
  float*    ws_ss;           /* (row_blocks, d_model/16, 16)                        */
  const float* first_norm;   /* layers[0].attn_norm (device), gain applied by the embed kernel */
  float*    ws_attn_part;    /* optional (rows, n_head, 8, head_dim + 8): partials of the range-split attention used when
                                rows*n_head < 256 and max_len > 256 (vaura_attention_splits); NULL -> never split       */
  uint32_t* ws_sync;         /* optional 768 words (zeroed once by the caller): producer flags of the launches that hand activations over INSIDE the
                                launch (csrc/mlp_engine.h: w1||w3 -> w2 -> next layer's qkv; csrc/attention.hip: attention -> wo), each phase's weight
                                stream running ahead of its hand-off; words 512 .. 767: arrival counts of the range-split attention (the last split
                                of a (row, head) merges the partials inside the launch; left at zero).  NULL -> every GEMV, the attention and its
                                merge are separate launches */
} vaura_decoder;

/* -------------------------------------------------------------------------------------------
 * Weight ingress (once, at load).  Replaces nn.Module.load_state_dict for the streamed matrices.
 * src: row-major fp32 (N x K) as stored in the reference checkpoint (nn.Linear.weight).          */
int vaura_pack_weight(const float* src, void* dst, int64_t N, int64_t K, int wdtype, vaura_stream_t s);
size_t vaura_packed_weight_bytes(int64_t N, int64_t K, int wdtype);

/* a4 DacEmbeddingProjection (llama.py:60-73) evaluated once for every (codebook, token):
 * table[k][tok][c] = sum_i w[k][c][i] * emb[k][tok][i] + b[k][c]   (K, vocab+1, tok_dim) fp32   */
int vaura_build_token_table(const float* tok_emb, const float* proj_w, const float* proj_b, float* table, int K,
                            int vocab1, int cdim, int tok_dim, vaura_stream_t s);

/* row-major (rows x C) fp32 <-> packed rows.  `rows_padded` = ceil(rows/16)*16 rows are written.   */
int vaura_pack_rows(const float* src, float* dst, int64_t rows, int64_t C, vaura_stream_t s);
int vaura_unpack_rows(const float* src, float* dst, int64_t rows, int64_t C, vaura_stream_t s);

/* -------------------------------------------------------------------------------------------
 * a5  AVCLIPEmbedder.forward / MLP (llama.py:79-92,136-141): fc2(gelu_tanh(fc1(x))), hoisted to
 * once per clip.  feats: packed rows (n_rows x 768); tmp: packed rows (n_rows x cond_dim);
 * out: packed rows (n_rows x cond_dim).  fc1/fc2: MFMA tiles, wdtype as given.                   */
int vaura_prefill_cond(const vaura_dims* d, const float* feats, const void* fc1, const void* fc2, int wdtype,
                       float* tmp, float* out, int64_t n_rows, vaura_stream_t s);

/* -------------------------------------------------------------------------------------------
 * a14 Pattern.build_pattern_sequence (codebook_patterns.py:180-207) for DelayedPatternProvider
 * (:390-406).  codes (B,K,T) int32 with -1 = unknown -> seq (B,K,T+K); special = d_codebook.     */
int vaura_pattern_build(const int32_t* codes, int32_t* seq, int B, int K, int T, int special, vaura_stream_t s);
/* a14 Pattern.revert_pattern_sequence (codebook_patterns.py:260-285) + the [..., :T] slice
 * (vaura_model.py:568-569).  seq (B,K,S) -> codes (B,K,T); positions with no source get `fill`.  */
int vaura_pattern_revert(const int32_t* seq, int32_t* codes, int B, int K, int T, int S, int fill, vaura_stream_t s);

/* -------------------------------------------------------------------------------------------
 * a2/a13/a15  logits (rows, K*vocab) -> next tokens.  Standalone form of the sampler used inside
 * vaura_decode_step (same kernel).  noise: (B*K, vocab) Exp(1) draws or NULL (Philox, step index
 * `step`).  tokens_out (B,K) int32.                                                              */
int vaura_sample(const float* logits, int B, int K, int vocab, const vaura_sampling* sp, const float* noise,
                 int64_t step, int32_t* tokens_out, vaura_stream_t s);

/* -------------------------------------------------------------------------------------------
 * a3..a12 + a2/a13/a15 for ONE position: Transformer.inference (llama.py:445-504) restricted to the
 * position state[0] with a K/V cache, followed by _sample_next_token and the fix-up
 * (vaura_model.py:536-544) that writes seq[..., state[0]+1] and advances state.
 * sample == 0: teacher-forced step (prompt prefill): no heads, no sampling, state still advances. */
int vaura_decode_step(const vaura_decoder* dec, const vaura_sampling* sp, int sample, vaura_stream_t s);

/* a1  the hot loop of VAURAModel.generate (vaura_model.py:502-547): `n_prefill` teacher-forced
 * positions then `n_steps` sampled ones, all enqueued back-to-back.  graph != NULL replays a captured single-step
 * hipGraph per position (the position lives in dec->state); NULL launches every kernel eagerly.
 * vaura_step_graph_build captures one step of (dec, sp) on `s` (not the legacy null stream) into a handle the caller
 * owns: it is valid for exactly the buffers / shapes / sampling parameters it was built from.  The loop replays the step
 * `m` at a time where at least m steps remain (one graph launch per m steps; m = 4, or bits 24..27 of vaura_set_debug_flags
 * at build time, 1..16): that longer graph is captured — same (dec, sp), on `s` — by the first vaura_generate_loop call with
 * n_steps >= m, so that call must use the dec / sp the handle was built from (it always must).  No process environment is read. */
typedef void* vaura_step_graph_t;
int vaura_generate_loop(const vaura_decoder* dec, const vaura_sampling* sp, int n_prefill, int n_steps,
                        vaura_step_graph_t graph, vaura_stream_t s);
int vaura_step_graph_build(const vaura_decoder* dec, const vaura_sampling* sp, vaura_stream_t s, vaura_step_graph_t* out);
void vaura_step_graph_free(vaura_step_graph_t graph);

/* Measurement aid (bench.py): runs `n_steps` sampled steps eagerly on `s` with a hipEvent pair
 * around every launch of the kernel kinds selected by `kind_mask` (bit = vaura_kernel_kind), then
 * synchronises `s` and reports, per kind, the summed elapsed ms and the launch count (HOST arrays of
 * VAURA_K_COUNT entries).  A stage that is several launches (range-split attention) reports the sum of
 * its launches per stage.  The only entry point that creates events / synchronises.              */
typedef enum vaura_kernel_kind {
  VAURA_K_EMBED = 0, VAURA_K_QKV = 1, VAURA_K_ATTN = 2, VAURA_K_WO = 3, VAURA_K_W13 = 4, VAURA_K_W2 = 5,
  VAURA_K_HEADS = 6, VAURA_K_SAMPLE = 7, VAURA_K_COUNT = 8
} vaura_kernel_kind;
int vaura_profile_loop(const vaura_decoder* dec, const vaura_sampling* sp, int n_steps, unsigned kind_mask,
                       double* total_ms_host, int64_t* launches_host, vaura_stream_t s);
/* Launches of the calling thread's last vaura_profile_loop whose interval exceeded 10x their kind's median (a stalled queue, not
 * the kernel): they were counted at the median in total_ms_host; per kind, HOST array of VAURA_K_COUNT entries.               */
void vaura_profile_outliers(int64_t* per_kind);

/* -------------------------------------------------------------------------------------------
 * op-level entry points (parity tests call the same kernels the step uses)                      */
/* out = epilogue( W x (x*gain) * rinv ):  epi 0 store, 1 +residual, 2 SwiGLU pairs, 3 gelu_tanh,
 * 4 row-major logits.  K must be one of the compiled depths (512, 768, 1024, 1536, 4096).        */
int vaura_gemv(const void* w, int wdtype, const float* x, const float* gain, const float* residual, float* out,
               int64_t rows, int64_t N, int64_t K, int epilogue, float eps, vaura_stream_t s);
/* pair form of vaura_gemv: x as (hi, lo) fp16 planes ("split rows"), weights as fp16 plane(s) or fp8, products on
 * v_mfma_f32_16x16x32_f16.  ss_in (row_blocks, n_ss_in, 16): partial sums of squares of the raw input
 * (fused RMSNorm) or NULL.  Optional outputs: fp32 packed rows, split rows of out*gain_out, partial
 * sums of squares of out.  epilogue: 0 store, 1 +residual, 2 SwiGLU pairs, 4 row-major logits.       */
int vaura_gemv_pair(const void* w, int wdtype /* H1 | H2 | FP8 */, const uint16_t* x_split, const float* ss_in, int n_ss_in, const float* residual, float* out,
                    float* out_khalf2 /* NULL, or (K = 1536, fused norm, store): `out` gets the partial over the first half
                                         of K and this buffer the second half's; the consumer adds them */,
                    uint16_t* out_split, const float* gain_out, float* ss_out, int64_t rows, int64_t N, int64_t K, int epilogue,
                    float eps, vaura_stream_t s);
/* packed rows (rows x C) fp32 [* gain] -> split rows (2 * rows_padded * C fp16: hi plane, lo plane) [+ partial sums of squares] */
int vaura_split_rows(const float* src, uint16_t* dst, const float* gain, float* ss, int64_t rows, int64_t C, vaura_stream_t s);

/* a8/a9 for one layer at position `pos` (host value): rope(q,k), append, softmax(qK^T/sqrt(hd)) V. */
int vaura_attention_step(const float* qkv, const float* rope, float* kcache, float* vcache, float* out,
                         int rows, int n_head, int head_dim, int max_len, int pos, vaura_stream_t s);

/* same step with the cached range of every (row, head) split over n_split (2..8) workgroups + a combine
 * pass; part: (rows, n_head, n_split, head_dim + 8) floats of scratch.                              */
int vaura_attention_step_split(const float* qkv, const float* rope, float* kcache, float* vcache, float* out, float* part,
                               int rows, int n_head, int head_dim, int max_len, int pos, int n_split, vaura_stream_t s);
/* workgroups per (row, head) the decode step uses for this shape when ws_attn_part is given (1 = no split) */
int vaura_attention_splits(int rows, int n_head, int max_len);

/* -------------------------------------------------------------------------------------------
 * a16 DacModelWrapper.decode (models/modules/dac/model.py:41-48): quantizer.from_codes + DAC
 * decoder (descript-audio-codec 1.0.0, un-vendored).  Weight-norm is folded by the caller.      */
typedef struct vaura_conv {
  const float* w;     /* conv: [taps][Cout][Cin]; transposed (stride r, k = 2r, pad r/2): [r][2][Cout][Cin] */
  const float* bias;  /* (Cout) */
  const float* wscale;/* codec precision 3 only, else NULL: (Cout) power-of-two scale of the layer's e4m3 weights; `w` is then the
                         packed fp8 stream [phase][step][Cout][4][32] described at vaura_codec.precision */
  int32_t cin, cout, taps, dilation, stride, _pad;
} vaura_conv;

typedef struct vaura_codec {
  int32_t n_codebooks, codebook_size, codebook_dim, latent_dim;
  int32_t n_blocks, n_units;        /* 4 decoder blocks, 3 residual units each */
  int32_t rates[4];
  const float* codebooks;           /* (K, size, dim)                    quantizer.quantizers[k].codebook */
  const float* out_proj_w;          /* (K, latent, dim) weight-norm folded                     .out_proj   */
  const float* out_proj_b;          /* (K, latent) */
  vaura_conv conv_in;               /* latent -> C0, k7                                  decoder.model.0   */
  const float* alpha_up[4];         /* Snake in front of each transposed conv   decoder.model.{b+1}.block.0 */
  vaura_conv up[4];                 /*                                           decoder.model.{b+1}.block.1 */
  const float* alpha_res[4][3][2];  /* Snakes of each residual unit              ...block.{u+2}.block.{0,2} */
  vaura_conv res[4][3][2];          /* {k7 dilated, k1}                          ...block.{u+2}.block.{1,3} */
  const float* alpha_out;           /* final Snake                               decoder.model.{n+1}        */
  vaura_conv conv_out;              /* C_last -> 1, k7, w as [7][C_last]; tanh   decoder.model.{n+2}        */
  float* ws[4];                     /* activation buffers, channels-last (B, L, C), ws_elems x 4 bytes each */
  size_t ws_elems;
  int32_t precision;                /* 0: fp32 activations/weights, exact v_mfma_f32_16x16x4_f32;
                                       1: (hi, lo) fp16 pairs, 3 x v_mfma_f32_16x16x32_f16 per product: conv weights
                                          (not conv_out) must then be given in pair layout
                                          [.. Cout][Cin/8][hi|lo][8] halves instead of [.. Cout][Cin] floats;
                                       2: as 1 with single-plane weights (the caller guarantees every lo plane is zero, e.g.
                                          fp8-quantised weights: e4m3 x power-of-two scale is exact in fp16): the lo-plane
                                          product is skipped, 2 MFMAs per product (BASELINE configs[4] codec part);
                                       3: block-scaled fp8 on v_mfma_scale_f32_16x16x128_f8f6f4 (BASELINE configs[4]
                                          "fp8 MFMA ... codec conv"): activations are e4m3 with one power-of-two scale per
                                          32 channels of a row (quantised by the producing kernel), weights e4m3 with one
                                          power-of-two scale per output channel (`wscale`).  conv_in keeps the layout of 2
                                          (its input comes from the quantizer); conv_out stays fp32.  Weight stream of
                                          every other conv: per phase, input channels in super-chunks of 128 (the last may
                                          hold nch = 1..3 blocks of 32); inside a super-chunk the k-blocks kb = tap*nch + ch
                                          are taken four per step, so step s holds, for every output channel, 4 x 32 bytes
                                          (kb = 4s .. 4s+3; zeros past the last k-block): [phase][step][Cout][4][32].
                                       4: "f16": buffers and weights as in 1, but ONE matrix instruction per product — hi(w) x hi(x),
                                          plain fp16 operands with fp32 accumulate (the arithmetic class the reference runs DAC in,
                                          models/vaura_model.py:92); the lo planes are written but not read                      */
  int32_t _pad1;
} vaura_codec;

/* codes (B, K, T) int32 -> wav (B, 1, T*prod(rates)) fp32 */
int vaura_dac_decode(const vaura_codec* c, const int32_t* codes, int B, int T, float* wav, vaura_stream_t s);
/* floats each of the 4 workspaces must hold for (B, T) */
size_t vaura_dac_workspace_elems(const vaura_codec* c, int B, int T);
/* Op-level access for parity tests: ONE convolution of the decoder (WNConv1d / WNConvTranspose1d of descript-audio-codec's
 * DecoderBlock / ResidualUnit) in the arithmetic of `precision` (vaura_codec.precision; weights laid out for it).
 * in (B, Lin, Cin) fp32 channels-last, already activated -> out (B, Lout, Cout) fp32 = conv(in) + bias,
 * Lout = Lin * stride.  `scratch` (>= B*Lin*Cin floats) receives the input in the precision's activation format.   */
int vaura_dac_conv(const vaura_conv* cv, int precision, const float* in, float* out, float* scratch, int B, int Lin,
                   vaura_stream_t s);
/* Op-level access for parity tests: the codec's activation, Snake1d of descript-audio-codec 1.0.0 (dac/nn/layers.py: x + (alpha + 1e-9)^-1 *
 * sin(alpha x)^2), exactly as every conv epilogue of the decoder / encoder applies it.  x, y (rows, C) fp32 channels-last, alpha (C).
 * The sine is an own restatement (period-pi reduction of sin^2 + a degree-9 odd polynomial, csrc/dac.hip::snake_sin2): max abs error
 * 2.5e-7 for |alpha x| < 1e6 — the library sinf was the largest single cost of the decode.                                             */
int vaura_snake(const float* x, const float* alpha, float* y, int64_t rows, int C, vaura_stream_t s);

/* -------------------------------------------------------------------------------------------
 * f4 DacModelWrapper.encode (models/modules/dac/model.py:30-39): DAC encoder + residual VQ (descript-audio-codec
 * 1.0.0, un-vendored: dac/model/dac.py Encoder, dac/nn/quantize.py).  Convolutions run on (hi, lo) fp16 pairs like the
 * decoder's precision 1: conv weights in pair layout, weight-norm folded by the caller.                           */
typedef struct vaura_codec_encoder {
  int32_t n_codebooks, codebook_size, codebook_dim, latent_dim;
  int32_t n_blocks, n_units;        /* 4 encoder blocks, 3 residual units each */
  int32_t rates[4];                 /* (2, 4, 8, 8) */
  int32_t enc_dim;                  /* 64: channels after the first conv; doubles per block */
  int32_t _pad0;
  const float* conv_in_w;           /* (7, enc_dim) fp32 = encoder.block.0 weight as [tap][Cout]          */
  const float* conv_in_b;           /* (enc_dim) */
  const float* alpha_res[4][3][2];  /* Snakes of each residual unit      encoder.block.{b+1}.block.{u}.block.{0,2} */
  vaura_conv res[4][3][2];          /* {k7 dilated, k1}                  ...block.{u}.block.{1,3}                   */
  const float* alpha_down[4];       /* Snake in front of each strided conv       encoder.block.{b+1}.block.3        */
  vaura_conv down[4];               /* encoder.block.{b+1}.block.4 (C -> 2C, k = 2r, stride r, pad r/2) restated as a
                                       3-tap stride-1 conv over rows of r*C channels (the (L, C) buffer read as
                                       (L/r, r*C)): cin = r*C, cout = 2C, taps = 3, dilation = 1, stride = 1,
                                       w'[tau+1][co][q*C + ci] = w[co][ci][tau*r + q + r/2] (0 outside [0, 2r))     */
  const float* alpha_out;           /* final Snake                               encoder.block.{n+1}                 */
  vaura_conv conv_out;              /* C_last -> latent, k3                      encoder.block.{n+2}                 */
  const float* in_proj_w;           /* (K, dim, latent) weight-norm folded       quantizer.quantizers[k].in_proj     */
  const float* in_proj_b;           /* (K, dim) */
  const float* codebooks;           /* (K, size, dim)                                                                */
  const float* out_proj_w;          /* (K, latent, dim)                                                              */
  const float* out_proj_b;          /* (K, latent) */
  float* ws[4];                     /* activation buffers, ws_elems x 4 bytes each */
  size_t ws_elems;
} vaura_codec_encoder;
/* wav (B, n_samples) fp32, n_samples a multiple of prod(rates) (DAC.preprocess zero-pads on the right; the caller does)
 * -> codes (B, K, n_samples / prod(rates)) int32 */
int vaura_dac_encode(const vaura_codec_encoder* c, const float* wav, int B, int64_t n_samples, int32_t* codes, vaura_stream_t s);
size_t vaura_dac_encode_workspace_elems(const vaura_codec_encoder* c, int B, int64_t n_samples);

/* -------------------------------------------------------------------------------------------
 * f3 (the step after the path) post-codec scaling: normalize_audio (utils/data_utils.py:407-466) as called by
 * scale_audio / save_results (scripts/generate.py:404, 440-461), per clip.  wav/out: (n_clips, n_samples) fp32 (may
 * alias).  strategy: 0 'clip' (the generate_*.yaml default: clamp to +-10^(-db/20)), 1 'peak', 2 'rms' (then clamp
 * to +-1), 3 'none' (copy).  normalize: the reference's flag (rescale only when it would otherwise clip, if 0).
 * scratch: vaura_audio_scratch_elems(n_clips) floats (strategies 1, 2).  'loudness': vaura_audio_loudness below.   */
typedef enum vaura_audio_strategy { VAURA_AUDIO_CLIP = 0, VAURA_AUDIO_PEAK = 1, VAURA_AUDIO_RMS = 2, VAURA_AUDIO_NONE = 3 } vaura_audio_strategy;
int vaura_audio_normalize(const float* wav, float* out, int n_clips, int64_t n_samples, int strategy, int normalize,
                          float peak_clip_headroom_db, float rms_headroom_db, float* scratch, vaura_stream_t s);
size_t vaura_audio_scratch_elems(int n_clips);
/* f3, strategy 'loudness' (utils/data_utils.py:453-458 -> normalize_loudness :347-387 -> _clip_wav :389-404): per clip, gain to
 * -loudness_headroom_db LKFS (ITU-R BS.1770-4 integrated loudness as torchaudio 2.2.1's transforms.Loudness computes it — a
 * third-party dependency absent from the reference tree: restated from the published algorithm, PARITY UNPINNED), optional tanh
 * compressor, clamp to [-1, 1]; clips below energy_floor rms (reference: 2e-3) or shorter than one 400 ms gating block are only clamped.
 * scratch: vaura_audio_loudness_scratch_elems(n_clips) floats; its first n_clips floats hold the applied gains afterwards. */
int vaura_audio_loudness(const float* wav, float* out, int n_clips, int64_t n_samples, int sample_rate, float loudness_headroom_db,
                         int compressor, float energy_floor, float* scratch, vaura_stream_t s);
size_t vaura_audio_loudness_scratch_elems(int n_clips);

/* -------------------------------------------------------------------------------------------
 * f2 (the step before the path) Segment-AVCLIP visual features: MotionFormer.forward
 * (models/modules/feature_extractors/avclip/motionformer.py:252-364) for the generate_*.yaml configuration: divided space-time
 * ViT-B/16 ('divided_224_16x4': motionformer_src/video_model_builder.py:174-268, vit_helper.py:392-472, 80-172, 523-557) + one
 * spatial nn.TransformerEncoderLayer per frame (motionformer.py:366-512), eval mode, no content mask.
 * Linear weights ("*_w") are in the codec's (hi, lo) fp16 PAIR layout [Cout][Cin/8][hi|lo][8] halves (row-major (Cout, Cin) of
 * nn.Linear.weight; the Conv3d weight flattened to (768, 1536)); everything else fp32.                                   */
typedef struct vaura_vit_attn {      /* DividedAttention (vit_helper.py:80-96) */
  const void* qkv_w; const float* qkv_b;      /* (3D, D) pair layout, (3D) */
  const void* proj_w; const float* proj_b;    /* (D, D) pair layout, (D)   */
} vaura_vit_attn;
typedef struct vaura_vit_block {     /* DividedSpaceTimeBlock (vit_helper.py:392-441) */
  const float *ln1_w, *ln1_b, *ln2_w, *ln2_b, *ln3_w, *ln3_b;      /* norm1 (space), norm2 (mlp), norm3 (time) */
  vaura_vit_attn space, time;                                       /* .attn, .timeattn */
  const void* fc1_w; const float* fc1_b; const void* fc2_w; const float* fc2_b;   /* mlp.fc1 (hidden, D), mlp.fc2 (D, hidden) */
} vaura_vit_block;
typedef struct vaura_vit {
  int32_t depth, dim, heads, hidden;           /* 12, 768, 12, 3072 */
  int32_t n_patches, n_frames;                 /* 196 tokens per frame, 8 token frames */
  int32_t in_chans, frames, img, patch, patch_t, patch_k;   /* 3, 16, 224, 16, 2, 1536 = 3*2*16*16 */
  float eps; int32_t _pad;                     /* 1e-6 */
  const void* pe_w; const float* pe_b;         /* patch_embed_3d.proj (D, patch_k) pair layout, (D) */
  const float *cls_token, *pos_embed, *temp_embed;     /* (D), (1 + n_patches, D), (n_frames, D) */
  const vaura_vit_block* blocks_host;          /* HOST array [depth] of device pointers */
  const float *norm_w, *norm_b;                /* final LayerNorm */
  const float* agg_cls;                        /* spatial_attn_agg.cls_token (D) */
  const float *agg_ln1_w, *agg_ln1_b, *agg_ln2_w, *agg_ln2_b;
  const void* agg_in_w; const float* agg_in_b;         /* self_attn.in_proj (3D, D) */
  const void* agg_out_w; const float* agg_out_b;       /* self_attn.out_proj (D, D) */
  const void* agg_l1_w; const float* agg_l1_b;         /* linear1 (hidden, D) */
  const void* agg_l2_w; const float* agg_l2_b;         /* linear2 (D, hidden) */
  /* workspaces, sizes from vaura_avclip_workspace_bytes(v, n_seg, i), i = 0..6 in this order */
  float* ws_x; float* ws_qkv; uint16_t* ws_a; uint16_t* ws_h; uint16_t* ws_p; float* ws_z; float* ws_s;
} vaura_vit;
/* frames (n_seg, 3, 16, 224, 224) fp32 (the (B, S) segments flattened) -> feats (n_seg, 8, 768) fp32 */
int vaura_avclip_forward(const vaura_vit* v, const float* frames, int n_seg, float* feats, vaura_stream_t s);
size_t vaura_avclip_workspace_bytes(const vaura_vit* v, int n_seg, int which);

/* Measurement aid (tools/pmc_driver, A/B timing): selects kernel variants for launches enqueued (or graphs captured) afterwards.
 * bit 0: wo / w2 GEMVs as one workgroup per column tile instead of the row-split pair; bit 4: prefill attention as one workgroup
 * per position instead of the MFMA kernel (tools/README.md lists every bit).
 * 0 = the product configuration.                                                          */
void vaura_set_debug_flags(unsigned flags);
/* A second word of the same kind (the first is full).  bit 0: with 17..32 decoder rows the GEMVs walk the two row blocks one after the
 * other (round 4) instead of taking both per weight pass; bit 1: the one-launch MLP refuses more than 16 rows; bit 2 (experiment, measured slower): the next layer's
 * attention as a fourth phase of the one-launch MLP; bits 3, 4: fp8 weights keep round 4's kernels for wo / w2 / never take the
 * one-launch MLP; bits 5, 6, 12..15: row f2's linear layers on round 4's kernel / with late fragment reads / column-tile panel width
 * (tools/README.md).  0 = the product. */
void vaura_set_debug_flags2(unsigned flags);
/* Host-side launch counters for tests that must know WHICH kernel instance a call took (read-and-clear; single caller thread):
 * 0 = codec conv launches on the 256-row workgroup instances (conv_pair_kernel<..., 8>, csrc/dac.hip) since the last read.
 * Unknown `which` -> -1. */
long long vaura_debug_counter(int which);

const char* vaura_version(void);
/* sizeof() of the descriptor structs as compiled into the library (0 dims, 1 layer_weights, 2 sampling, 3 decoder,
 * 4 conv, 5 codec, 6 codec_encoder, 7 vit, 8 vit_block): a binding checks its mirrored struct layouts against these before the first call.            */
size_t vaura_struct_size(int which);

#ifdef __cplusplus
}
#endif
#endif /* VAURA_HIP_H */
