// Per-step bookkeeping kernels around the layer stack: input embedding, next-token selection,
// delay-pattern build / revert.
//
//   embed    models/modules/sampler/llama.py:455-472, 555-586  (token projection sum + video concat)
//   sample   models/vaura_model.py:807-825, 536-544; utils/utils.py:139-196
//   pattern  models/modules/misc/codebook_patterns.py:137-285, 390-406 (delayed pattern, closed form)
#include "common.h"
#include "gemv3_kernel.h"

// ------------------------------------------------------------------------------------ embed
// Token projection table, built once per weight set with the arithmetic of the reference's
// DacEmbeddingProjection (llama.py:70-73): table[k][tok][c] = sum_i W_k[c][i] * emb_k[tok][i] + b_k[c]
__global__ void token_table_kernel(const float* __restrict__ tok_emb, const float* __restrict__ proj_w,
                                   const float* __restrict__ proj_b, float* __restrict__ table, int K, int vocab1,
                                   int cdim, int tok_dim) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)K * vocab1 * tok_dim;
  if (i >= total) return;
  const int c = (int)(i % tok_dim);
  const int tok = (int)((i / tok_dim) % vocab1);
  const int k = (int)(i / ((int64_t)tok_dim * vocab1));
  const float* e = tok_emb + ((size_t)k * vocab1 + tok) * cdim;
  const float* wr = proj_w + ((size_t)k * tok_dim + c) * cdim;
  float z = 0.f;
  for (int j = 0; j < cdim; ++j) z = fmaf(wr[j], e[j], z);
  table[i] = z + proj_b[(size_t)k * tok_dim + c];
}

// h0[row] = [ cond(row, pos // tpf) | sum_k table_k[tok(row % B, k, pos)] ]  -> packed rows (+ split rows
// of h0 * gain and per-16-column partial sums of squares for the first fused RMSNorm)
__global__ __launch_bounds__(64) void embed_kernel(
    const int32_t* __restrict__ seq, const int32_t* __restrict__ state, const float* __restrict__ cond_proj,
    const float* __restrict__ empty_video, const float* __restrict__ table, float* __restrict__ h,
    uint16_t* __restrict__ hsplit, const float* __restrict__ gain, float* __restrict__ ss, int B, int K, int S, int Tv,
    int tpf, int vocab1, int cond_dim, int tok_dim, int pos_host, int rows16) {
  const int row = blockIdx.x;
  const int cq = blockIdx.y * 64 + threadIdx.x;   // 4-column quad
  const int b = row % B;
  // decode: the position lives on the device; prefill: positions pos_host + blockIdx.z, one row block
  // (rows16 = padded row count) per position
  const int pos = pos_host >= 0 ? pos_host + (int)blockIdx.z : state[0];
  const int vrow = (int)blockIdx.z * rows16 + row;
  const int D = cond_dim + tok_dim;
  const int frame = pos / tpf;
  f32x4 o;
  if (cq < cond_dim / 4) {
    if (frame < Tv)
      o = reinterpret_cast<const f32x4*>(cond_proj)[packed_quad(row * Tv + frame, cq, cond_dim)];
    else
      o = reinterpret_cast<const f32x4*>(empty_video)[cq];
  } else {
    const int c0 = (cq - cond_dim / 4) * 4;
    o = f32x4{0.f, 0.f, 0.f, 0.f};
    int tok[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) tok[k] = (k < K) ? seq[((size_t)b * K + k) * S + pos] : 0;
    f32x4 e[16];
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if (k < K) e[k] = *reinterpret_cast<const f32x4*>(table + ((size_t)k * vocab1 + tok[k]) * tok_dim + c0);
#pragma unroll
    for (int k = 0; k < 16; ++k)    // same left-to-right order as the reference's sum([...]) (llama.py:455-460)
      if (k < K) o += e[k];
  }
  va_st16(reinterpret_cast<f32x4*>(h) + packed_quad(vrow, cq, D), o);
  if (hsplit) {
    float s = ((o[0] * o[0] + o[1] * o[1]) + o[2] * o[2]) + o[3] * o[3];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if ((threadIdx.x & 3) == 0) va_st4(ss + ((size_t)(vrow >> 4) * (D / 16) + (cq >> 2)) * 16 + (vrow & 15), s);
    const f32x4 u = o * *reinterpret_cast<const f32x4*>(gain + cq * 4);
    store_split4(hsplit, vrow, cq * 4, D, u);
  }
}

int va_launch_embed(const vaura_decoder* d, int pos_host, int n_pos, hipStream_t s) {
  const vaura_dims& m = d->dims;
  const int D = m.cond_dim + m.tok_dim;
  if (!d->tok_table || (D % 256)) return VAURA_ERR_SHAPE;
  const bool split = d->wdtype == VAURA_W_H1 || d->wdtype == VAURA_W_H2 || va_is_fp8(d->wdtype);   // pair path (api.hip enqueue_step)
  if (split && (!d->ws_h_split || !d->ws_ss || !d->first_norm)) return VAURA_ERR_ARG;
  VA_LAUNCH(embed_kernel, dim3(d->rows, D / 256, n_pos), dim3(64), 0, s, d->seq, d->state, d->cond_proj, d->empty_video,
            d->tok_table, d->ws_h, split ? d->ws_h_split : nullptr, d->first_norm, d->ws_ss, d->batch, m.n_codebooks,
            d->seq_len, d->n_cond_tokens, m.tokens_per_frame, m.vocab + 1, m.cond_dim, m.tok_dim, pos_host,
            (d->rows + 15) / 16 * 16);
  return 0;
}

extern "C" int vaura_build_token_table(const float* tok_emb, const float* proj_w, const float* proj_b, float* table, int K,
                                       int vocab1, int cdim, int tok_dim, vaura_stream_t s) {
  if (!tok_emb || !proj_w || !proj_b || !table || K <= 0) return VAURA_ERR_ARG;
  const int64_t total = (int64_t)K * vocab1 * tok_dim;
  VA_LAUNCH(token_table_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(s), tok_emb, proj_w, proj_b,
            table, K, vocab1, cdim, tok_dim);
  return 0;
}

// ------------------------------------------------------------------------------------ sampling
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

struct SampleArgs {
  const float* logits;   // (rows, K*V) row-major
  const float* noise;    // (steps, B*K, V) or null
  const int32_t* state;  // device {pos, arrivals, step} or null
  int32_t* state_rw;     // same buffer, writable, or null (standalone)
  int32_t* tokens_out;   // (B, K) or null
  int32_t* seq;          // (B, K, S) or null
  int B, K, V, T, S;
  int use_sampling, top_k, probs_in;
  float temp, top_p, cfg_scale;
  float tie_eps;         // near-tie detector (vaura_sampling.tie_eps): relative bound on a logit's error, 0 = off
  uint64_t seed, clip_base;
  long long step_host;
};

#define SMP_THREADS 256

__device__ __forceinline__ void block_argmax(float v, int i, float* sv, int* si, float& bv, int& bi) {
  // first-index-wins argmax over the block (torch.argmax returns the first maximal index)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(v, o, 64);
    const int oi = __shfl_xor(i, o, 64);
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = v; si[threadIdx.x >> 6] = i; }
  __syncthreads();
  bv = sv[0]; bi = si[0];
#pragma unroll
  for (int w = 1; w < SMP_THREADS / 64; ++w)
    if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
}

__device__ __forceinline__ float block_sum(float v, float* sv) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sv[threadIdx.x >> 6] = v;
  __syncthreads();
  return ((sv[0] + sv[1]) + sv[2]) + sv[3];
}
__device__ __forceinline__ float block_max(float v, float* sv) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sv[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
}
// two maxima with ONE barrier pair (the near-tie detector's scale next to the softmax's maximum)
__device__ __forceinline__ void block_max2(float& a, float& b, float* sv) {
  a = wave_max(a);
  b = wave_max(b);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = a; sv[4 + (threadIdx.x >> 6)] = b; }
  __syncthreads();
  a = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
  b = fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7]));
}
// the near-tie screen's four quantities with ONE barrier pair: two maxima and two counts
__device__ __forceinline__ void block_tie4(float& m0, float& m1, int& c0, int& c1, float* sv, int* si) {
  m0 = wave_max(m0);
  m1 = wave_max(m1);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { c0 += __shfl_xor(c0, o, 64); c1 += __shfl_xor(c1, o, 64); }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    const int wv = threadIdx.x >> 6;
    sv[wv] = m0; sv[4 + wv] = m1; si[wv] = c0; si[4 + wv] = c1;
  }
  __syncthreads();
  m0 = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
  m1 = fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7]));
  c0 = si[0] + si[1] + si[2] + si[3];
  c1 = si[4] + si[5] + si[6] + si[7];
}
__device__ __forceinline__ int block_count(int v, int* si) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) si[threadIdx.x >> 6] = v;
  __syncthreads();
  return si[0] + si[1] + si[2] + si[3];
}

// V == 1024 == 4 * SMP_THREADS: thread t owns candidates 4t .. 4t+3
__global__ __launch_bounds__(SMP_THREADS) void sample_kernel(const float* __restrict__ logits_q, const int32_t* __restrict__ state_q,
                                                             SampleArgs a) {
  a.logits = logits_q;   // explicit scalar copies: preloaded into SGPRs at wave launch (the struct is not)
  a.state = state_q;
  __shared__ float sv[8];
  __shared__ int si[8];
  __shared__ float sp[1024];   // top-p: sorted probabilities
  __shared__ int sidx[1024];   // top-p: their token ids
  __shared__ float skeep[1024];
  __shared__ uint32_t hist[256];
  __shared__ uint32_t s_sel[2];
  const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int pos = a.state ? a.state[0] : 0;
  const long long step = a.state ? (long long)a.state[2] : a.step_host;
  const int V = a.V;

  const f32x4 lc = *reinterpret_cast<const f32x4*>(a.logits + ((size_t)b * a.K + k) * V + 4 * tid);
  float x[4] = {lc[0], lc[1], lc[2], lc[3]};
  // near-tie detector: magnitude of the rows this decision is made from (both branches, before the mix)
  float amax = fmaxf(fmaxf(fabsf(lc[0]), fabsf(lc[1])), fmaxf(fabsf(lc[2]), fabsf(lc[3])));
  if (a.cfg_scale > 1.0f) {  // models/vaura_model.py:810-813
    const f32x4 lu = *reinterpret_cast<const f32x4*>(a.logits + ((size_t)(a.B + b) * a.K + k) * V + 4 * tid);
    amax = fmaxf(amax, fmaxf(fmaxf(fabsf(lu[0]), fabsf(lu[1])), fmaxf(fabsf(lu[2]), fabsf(lu[3]))));
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = lu[j] + (x[j] - lu[j]) * a.cfg_scale;
  }
  // Near-tie detector (round 6).  The plane storages carry 22-bit operands where the reference computes fp32: a logit reaches the
  // sampler with an error of at most ~tie_eps x (the row's largest |logit|), and the CFG mix s lc - (s - 1) lu multiplies that by up
  // to 2 s - 1.  A decision whose own margin is inside twice that bound could have gone the other way in the reference's arithmetic
  // (or in ANY fp32 summation order: the reference's logits themselves move by ~3e-6 with the prefix length it re-feeds).  Such
  // decisions are COUNTED (state[6]; state[7] = first such step + 1) and raise the sticky VAURA_STATUS_NEAR_TIE bit; the token chosen
  // is never changed here.  What the host does with it is policy (engine.py near_tie: report | rerun on the exact-fp32 twin).
  bool near_tie = false;
  float tie_delta = 0.f;       // absolute bound on a mixed logit's error
  const bool tie_on = a.tie_eps > 0.f && !a.probs_in;
  const float tie_mix = a.tie_eps * (a.cfg_scale > 1.0f ? 2.f * a.cfg_scale - 1.f : 1.f);
  if (tie_on && !(a.use_sampling && a.temp > 0.0f)) tie_delta = tie_mix * block_max(amax, sv);      // greedy: its own reduction; sampled: with the softmax's maximum
  // Range guard of the fp16-plane activation format (gemv3_kernel.h split2): an activation beyond fp16's 65504 becomes inf in its
  // hi plane, inf - inf = NaN in the lo plane, and from there NaN in the residual stream of that row for the rest of the clip —
  // so EVERY overflow anywhere in the step (or in a teacher-forced prefix, through the K/V cache) arrives here as a non-finite
  // logit.  Raise the sticky status bit the host checks after generate() instead of sampling from garbage.
  if (!(fabsf(x[0]) < INFINITY && fabsf(x[1]) < INFINITY && fabsf(x[2]) < INFINITY && fabsf(x[3]) < INFINITY)) {
    if (a.state_rw) __hip_atomic_fetch_or(&a.state_rw[4], VAURA_STATUS_NONFINITE_LOGITS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ... and keep the rest of this launch on finite numbers: the selection networks below (radix select on the IEEE bits, the
    // bitonic sort, argmax of p / q) are written for ordered values; on NaN they can return an index outside the codebook, which
    // the NEXT step's embedding gather would follow out of its table (round 5: a memory fault, seen on the x3000 checkpoint under
    // top-k sampling).  The token drawn from the sanitised row is meaningless — the status bit says so — but it is a valid id.
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = fabsf(x[j]) < INFINITY ? x[j] : 0.f;
  }

  int token;
  if (!(a.use_sampling && a.temp > 0.0f)) {
    float bv = x[0]; int bi = 4 * tid;
#pragma unroll
    for (int j = 1; j < 4; ++j) if (x[j] > bv) { bv = x[j]; bi = 4 * tid + j; }
    float rv; int ri;
    block_argmax(bv, bi, sv, si, rv, ri);
    token = ri;
    if (tie_delta > 0.f) {     // runner-up of the mixed logits: the argmax could flip when top-1 - top-2 < 2 delta
      float second = -INFINITY;
#pragma unroll
      for (int j = 0; j < 4; ++j) if (4 * tid + j != ri) second = fmaxf(second, x[j]);
      second = block_max(second, sv);
      near_tie = (rv - second) < 2.f * tie_delta;
    }
  } else {
    // softmax(logits / temp) — or the input rows themselves when they already are probabilities (utils/utils.py:139-196)
    float p[4];
    if (a.probs_in) {
#pragma unroll
      for (int j = 0; j < 4; ++j) p[j] = x[j];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) x[j] = x[j] / a.temp;
      float mx = fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3]));
      if (tie_on) {
        block_max2(mx, amax, sv);
        tie_delta = tie_mix * amax;
      } else {
        mx = block_max(mx, sv);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) p[j] = expf(x[j] - mx);
      const float den = block_sum((p[0] + p[1]) + (p[2] + p[3]), sv);
#pragma unroll
      for (int j = 0; j < 4; ++j) p[j] = p[j] / den;
    }

    // Exp(1) draws for this (clip, codebook, step)
    float q[4];
    if (a.noise) {
      const f32x4 nz = *reinterpret_cast<const f32x4*>(a.noise + (((size_t)step * a.B + b) * a.K + k) * V + 4 * tid);
      q[0] = nz[0]; q[1] = nz[1]; q[2] = nz[2]; q[3] = nz[3];
    } else {
      uint32_t r[4];
      const uint64_t clip = a.clip_base + (uint64_t)b;
      philox4x32_10((uint32_t)tid, (uint32_t)step, (uint32_t)(clip * (uint64_t)a.K + k), (uint32_t)((clip * a.K + k) >> 32),
                    (uint32_t)a.seed, (uint32_t)(a.seed >> 32), r);
#pragma unroll
      for (int j = 0; j < 4; ++j) q[j] = -logf(((float)r[j] + 0.5f) * 2.3283064365386963e-10f);
    }

    if (a.top_p > 0.0f) {
      // utils/utils.py:181-196 — sort descending (ties: lower id first), sequential cumsum, cut, renormalise,
      // draw in sorted space with the noise indexed by RANK, map back.
      // Bitonic network over the 1024 (probability, id) pairs, element 4 tid + j in thread tid's registers.  A stage of stride st compares
      // element i with i ^ st: strides 1, 2 stay inside a thread, 4 .. 128 are lane exchanges inside a wave (xor of the lane by st / 4),
      // only 256 and 512 (three of the 55 stages) cross waves and go through LDS with barriers.  Same comparisons as the LDS network
      // of rounds 1-3 (ties: lower id first): the same order, bit for bit (round 4: the top-p launch 52 -> 36.5 us; the network is ~20 us of
      // it, the sequential sum below ~10: tools/time_sampler.py).
      float sp_r[4];
      int id_r[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { sp_r[j] = p[j]; id_r[j] = 4 * tid + j; }
      auto cmpx = [&](int j, float py, int idy, int st, int sz) {
        const int i = 4 * tid + j;
        const bool lower = (i & st) == 0;
        const bool desc = ((i & ~st) & sz) == 0;            // direction of the pair = that of its lower index
        const bool x_first = (sp_r[j] > py) || (sp_r[j] == py && id_r[j] < idy);
        const bool keep_x = (lower == desc) ? x_first : !x_first;
        if (!keep_x) { sp_r[j] = py; id_r[j] = idy; }
      };
      for (int sz = 2; sz <= 1024; sz <<= 1) {
        for (int st = sz >> 1; st > 0; st >>= 1) {
          if (st == 1) {           // (static register indices: a run-time `j ^ st` would send the arrays to scratch)
            const float py[4] = {sp_r[1], sp_r[0], sp_r[3], sp_r[2]};
            const int iy[4] = {id_r[1], id_r[0], id_r[3], id_r[2]};
#pragma unroll
            for (int j = 0; j < 4; ++j) cmpx(j, py[j], iy[j], 1, sz);
          } else if (st == 2) {
            const float py[4] = {sp_r[2], sp_r[3], sp_r[0], sp_r[1]};
            const int iy[4] = {id_r[2], id_r[3], id_r[0], id_r[1]};
#pragma unroll
            for (int j = 0; j < 4; ++j) cmpx(j, py[j], iy[j], 2, sz);
          } else if (st < 256) {
            float py[4]; int iy[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { py[j] = __shfl_xor(sp_r[j], st >> 2, 64); iy[j] = __shfl_xor(id_r[j], st >> 2, 64); }
            // stride >= 4: whether this thread's four elements are the lower ones of their pairs, and the pairs' direction, do not depend on j
            const bool want_first = (((4 * tid) & st) == 0) == ((((4 * tid) & ~st) & sz) == 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const bool x_first = (sp_r[j] > py[j]) || (sp_r[j] == py[j] && id_r[j] < iy[j]);
              if (x_first != want_first) { sp_r[j] = py[j]; id_r[j] = iy[j]; }
            }
          } else {
            __syncthreads();                                 // (the previous exchange's reads are done)
#pragma unroll
            for (int j = 0; j < 4; ++j) { sp[4 * tid + j] = sp_r[j]; sidx[4 * tid + j] = id_r[j]; }
            __syncthreads();
            float py[4]; int iy[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { py[j] = sp[(4 * tid + j) ^ st]; iy[j] = sidx[(4 * tid + j) ^ st]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) cmpx(j, py[j], iy[j], st, sz);
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 4; ++j) { sp[4 * tid + j] = sp_r[j]; sidx[4 * tid + j] = id_r[j]; }
      __syncthreads();
      // torch.cumsum's order: one sequential chain of 1024 fp32 additions — kept in REGISTERS of wave 0 (lane l holds elements 16 l ..
      // 16 l + 15, the carry moves from lane to lane through an SGPR), not 1024 dependent LDS round trips of one thread
      if (tid < 64) {
        float v[16], keep[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 x = *reinterpret_cast<const f32x4*>(sp + 16 * tid + 4 * j);
          v[4 * j] = x[0]; v[4 * j + 1] = x[1]; v[4 * j + 2] = x[2]; v[4 * j + 3] = x[3];
        }
        float carry = 0.f;
        for (int l = 0; l < 64; ++l) {
          float cs = carry;
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            cs += v[j];
            if (tid == l) keep[j] = (cs - v[j] > a.top_p) ? 0.f : 1.f;
          }
          carry = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cs), l));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<f32x4*>(skeep + 16 * tid + 4 * j) = f32x4{keep[4 * j], keep[4 * j + 1], keep[4 * j + 2], keep[4 * j + 3]};
      }
      __syncthreads();
      float ps[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) ps[j] = sp[4 * tid + j] * skeep[4 * tid + j];
      const float den2 = block_sum((ps[0] + ps[1]) + (ps[2] + ps[3]), sv);
      float bv = -1.f; int bi = 0x7fffffff;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float r = (ps[j] / den2) / q[j];
        if (r > bv) { bv = r; bi = 4 * tid + j; }
      }
      float rv; int ri;
      block_argmax(bv, bi, sv, si, rv, ri);
      token = sidx[ri];
      if (tie_delta > 0.f) {   // the draw: argmax of p / q — a probability moves by a factor exp(+-delta / temp), the runner-up wins inside twice that
        float second = -1.f;   // (the nucleus cut itself is not screened: no shipped config samples with top-p)
#pragma unroll
        for (int j = 0; j < 4; ++j) if (4 * tid + j != ri) second = fmaxf(second, (ps[j] / den2) / q[j]);
        second = block_max(second, sv);
        near_tie = (rv - second) < rv * (2.f * tie_delta / a.temp);
      }
    } else {
      const float p_raw[4] = {p[0], p[1], p[2], p[3]};   // probabilities before the top-k mask (near-tie detector)
      float thr_keep = 0.f, den_keep = 1.f;
      if (a.top_k > 0) {
        // utils/utils.py:172-176 — threshold = k-th largest probability (bitwise binary search on the
        // IEEE bits: probabilities are >= 0 so integer order == float order), keep p >= threshold.
        const int kk = a.top_k < V ? a.top_k : V;
        uint32_t key[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) key[j] = __builtin_bit_cast(uint32_t, p[j]);
        // k-th largest key by radix-256 select: 4 passes of {LDS histogram of the next byte among the keys
        // that still match the chosen prefix, suffix-sum over the 256 bins, pick the byte holding rank `need`}
        uint32_t thr = 0;
        int need = kk;
#pragma unroll 1
        for (int pass = 0; pass < 4; ++pass) {
          const int shift = 24 - 8 * pass;
          hist[tid] = 0;
          __syncthreads();
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (pass == 0 || (key[j] >> (shift + 8)) == (thr >> (shift + 8))) atomicAdd(&hist[(key[j] >> shift) & 255u], 1u);
          __syncthreads();
          const int mine = (int)hist[255 - tid];       // reversed: an inclusive prefix scan gives suffix sums
          int scan = mine;
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(scan, o, 64);
            if ((tid & 63) >= o) scan += up;
          }
          if ((tid & 63) == 63) si[tid >> 6] = scan;
          __syncthreads();
          for (int w2 = 0; w2 < (tid >> 6); ++w2) scan += si[w2];
          if (scan >= need && scan - mine < need) {    // exactly one bin holds the rank
            s_sel[0] = thr | ((uint32_t)(255 - tid) << shift);
            s_sel[1] = (uint32_t)(need - (scan - mine));
          }
          __syncthreads();
          thr = s_sel[0];
          need = (int)s_sel[1];
        }
        const float thrf = __builtin_bit_cast(float, thr);
        thr_keep = thrf;
#pragma unroll
        for (int j = 0; j < 4; ++j) p[j] = p[j] * (p[j] >= thrf ? 1.0f : 0.0f);
        const float den2 = block_sum((p[0] + p[1]) + (p[2] + p[3]), sv);
        den_keep = den2;
#pragma unroll
        for (int j = 0; j < 4; ++j) p[j] = p[j] / den2;
      }
      float bv = -1.f; int bi = 0x7fffffff;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float r = p[j] / q[j];
        if (r > bv) { bv = r; bi = 4 * tid + j; }
      }
      float rv; int ri;
      block_argmax(bv, bi, sv, si, rv, ri);
      token = ri;
      if (tie_delta > 0.f) {
        // (1) the draw argmax(p / q): a probability moves by a factor exp(+-delta / temp) (the common normalisation cancels), so the
        //     runner-up wins when its ratio is within twice that of the winner's.  (2) the top-k threshold (keep p >= k-th largest):
        //     membership matters only through the draw — a candidate whose probability is within the band BELOW the threshold and whose
        //     ratio would have reached the winner's had it been kept, or a winner within the band ABOVE the threshold while such a
        //     candidate exists (it could have been the one left out).
        const float band = 2.f * tie_delta / a.temp;
        float second = -1.f, cand = -1.f;
        int nbelow = 0, wedge = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int i = 4 * tid + j;
          if (i != ri) second = fmaxf(second, p[j] / q[j]);
          if (thr_keep > 0.f) {
            if (p_raw[j] < thr_keep && p_raw[j] >= thr_keep * (1.f - band)) { ++nbelow; cand = fmaxf(cand, (p_raw[j] / den_keep) / q[j]); }
            if (i == ri && p_raw[j] <= thr_keep * (1.f + band)) wedge = 1;
          }
        }
        block_tie4(second, cand, nbelow, wedge, sv, si);             // one barrier pair for all four
        near_tie = (rv - second) < rv * band;
        if (thr_keep > 0.f && (cand >= rv * (1.f - band) || (nbelow > 0 && wedge > 0))) near_tie = true;
      }
    }
  }

  if (tid == 0) {
    if (a.tokens_out) a.tokens_out[b * a.K + k] = token;
    if (a.seq) {
      // vaura_model.py:536-544 — invalid pattern slots become the special token; known tokens are kept
      const int offset = pos + 1;
      const int t = offset - 1 - k;
      const int tok = (t >= 0 && t < a.T) ? token : V;
      if (offset < a.S) {   // a step past the end of the sequence (refused by vaura_generate_loop) must not write
        int32_t* slot = a.seq + ((size_t)b * a.K + k) * a.S + offset;
        if (*slot == -1) {
          *slot = tok;
          // near-tie detector: only decisions that are USED count (a valid pattern slot that was still unknown)
          if (near_tie && t >= 0 && t < a.T && a.state_rw) {
            __hip_atomic_fetch_or(&a.state_rw[4], VAURA_STATUS_NEAR_TIE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&a.state_rw[6], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int expect = 0;
            (void)__hip_atomic_compare_exchange_strong(&a.state_rw[7], &expect, (int)step + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
    }
    if (a.state_rw) {
      // No fences: nothing in THIS launch reads the token slots or the state written here; the next kernel sees them
      // because a kernel boundary publishes all stores.  The arrival counter is a device-scope atomic (coherent by
      // itself).  (An agent-scope release here costs an L2 write-back per workgroup: DESIGN.md §6.)
      const int arrived = __hip_atomic_fetch_add(&a.state_rw[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (arrived == (int)(gridDim.x * gridDim.y) - 1) {
        __hip_atomic_store(&a.state_rw[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.state_rw[0] = pos + 1;
        a.state_rw[2] = (int)step + 1;
        a.state_rw[5] = a.state_rw[5] + 1;     // launch-epoch counter of the in-launch hand-offs (common.h va_handoff_epoch): never rewound
      }
    }
  }
}

int va_launch_sample(const float* logits, int B, int K, int vocab, const vaura_sampling* sp, const float* noise,
                     int /*noise_rows_per_step*/, const int32_t* state, int64_t step_host, int32_t* tokens_out,
                     int32_t* seq, int T, int S, int32_t* state_rw, hipStream_t s) {
  if (!logits || !sp || B <= 0 || K <= 0) return VAURA_ERR_ARG;
  if (vocab != 1024) return VAURA_ERR_SHAPE;
  SampleArgs a;
  a.logits = logits; a.noise = noise; a.state = state; a.state_rw = state_rw; a.tokens_out = tokens_out; a.seq = seq;
  a.B = B; a.K = K; a.V = vocab; a.T = T; a.S = S;
  a.use_sampling = sp->use_sampling; a.top_k = sp->top_k; a.temp = sp->temp; a.top_p = sp->top_p;
  a.cfg_scale = sp->input_is_probs ? 1.0f : sp->cfg_scale; a.seed = sp->seed; a.clip_base = sp->clip_base; a.step_host = step_host;
  a.probs_in = sp->input_is_probs;
  a.tie_eps = sp->tie_eps > 0.f ? sp->tie_eps : 0.f;
  VA_LAUNCH(sample_kernel, dim3(K, B), dim3(SMP_THREADS), 0, s, a.logits, a.state, a);
  return 0;
}

__global__ void advance_kernel(int32_t* state, int set_to) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    state[0] = set_to >= 0 ? set_to : state[0] + 1;
    state[5] = state[5] + 1;                   // launch-epoch counter (common.h va_handoff_epoch)
  }
}
int va_launch_advance(int32_t* state, int set_to, hipStream_t s) {
  VA_LAUNCH(advance_kernel, dim3(1), dim3(64), 0, s, state, set_to);
  return 0;
}

// ------------------------------------------------------------------------------------ pattern
// delayed pattern, delays = 0..K-1: sequence step s of codebook q holds timestep t = s - 1 - q.
__global__ void pattern_build_kernel(const int32_t* __restrict__ codes, int32_t* __restrict__ seq, int B, int K, int T,
                                     int special) {
  const int S = T + K;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * K * S) return;
  const int s = (int)(i % S), q = (int)((i / S) % K), b = (int)(i / ((int64_t)S * K));
  const int t = s - 1 - q;
  seq[i] = (t >= 0 && t < T) ? codes[((size_t)b * K + q) * T + t] : special;
}
__global__ void pattern_revert_kernel(const int32_t* __restrict__ seq, int32_t* __restrict__ codes, int B, int K, int T,
                                      int S, int fill) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * K * T) return;
  const int t = (int)(i % T), q = (int)((i / T) % K), b = (int)(i / ((int64_t)T * K));
  const int s = t + 1 + q;
  codes[i] = (s < S) ? seq[((size_t)b * K + q) * S + s] : fill;
}

extern "C" {

int vaura_pattern_build(const int32_t* codes, int32_t* seq, int B, int K, int T, int special, vaura_stream_t s) {
  if (!codes || !seq || B <= 0 || K <= 0 || T <= 0) return VAURA_ERR_ARG;
  const int64_t n = (int64_t)B * K * (T + K);
  VA_LAUNCH(pattern_build_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(s), codes, seq, B, K, T, special);
  return 0;
}

int vaura_pattern_revert(const int32_t* seq, int32_t* codes, int B, int K, int T, int S, int fill, vaura_stream_t s) {
  if (!codes || !seq || B <= 0 || K <= 0 || T <= 0 || S <= 0) return VAURA_ERR_ARG;
  const int64_t n = (int64_t)B * K * T;
  VA_LAUNCH(pattern_revert_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(s), seq, codes, B, K, T, S, fill);
  return 0;
}

int vaura_sample(const float* logits, int B, int K, int vocab, const vaura_sampling* sp, const float* noise,
                 int64_t step, int32_t* tokens_out, vaura_stream_t s) {
  if (!tokens_out) return VAURA_ERR_ARG;
  return va_launch_sample(logits, B, K, vocab, sp, noise, B * K, nullptr, step, tokens_out, nullptr, 0, 0, nullptr, as_stream(s));
}

}  // extern "C"
