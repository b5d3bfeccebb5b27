// Segment-AVCLIP visual feature extractor: the step BEFORE the decode path.
//   MotionFormer.forward / forward_segments            models/modules/feature_extractors/avclip/motionformer.py:252-364
//   VisionTransformer.forward_features                 .../motionformer_src/video_model_builder.py:174-268
//   PatchEmbed3D, DividedSpaceTimeBlock, DividedAttention, Mlp      .../motionformer_src/vit_helper.py:523-557, 392-472, 80-172, 475-498
//   SpatialTransformerEncoderLayer (nn.TransformerEncoderLayer, norm_first, GELU)   motionformer.py:366-512
// for the configuration the generate_*.yaml files use (divided space-time ViT-B/16, 16 x 224 x 224 segments -> 1 + 8 x 196
// tokens, spatial aggregation by one encoder layer, no temporal / global aggregation).
//
// Bound: MFMA.  Every nn.Linear and the 3-D patch embedding (a GEMM over 3 x 2 x 16 x 16 = 1536-element patches) run on the
// codec's pair GEMM (dac.hip conv_pair_kernel: activations and weights as (hi, lo) fp16 pairs, 22 significand bits, fp32
// accumulate) with bias / residual / exact-GELU epilogues; LayerNorm, the three attention patterns of the divided block and
// the aggregation layer's one-query attention are fp32 kernels (KBs to a few MB of data each):
//   cls      one query (the CLS token of a sequence) over all keys of the sequence
//   time     a patch token over CLS + the 8 tokens at its spatial location           (9 keys)
//   space    a patch token over CLS + the 196 tokens of its frame                    (197 keys, K / V of the frame in LDS)
// Token rows: X[(seg * 1569 + 0)] = CLS, X[seg * 1569 + 1 + f * 196 + n] = patch (frame f, location n) — the reference's
// flatten(2) order (f, h, w).
#include "common.h"

typedef _Float16 vf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void put_pair8(uint16_t* base, size_t row, int oct, int C, const float* v) {
  vf16x8 hi, lo;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    hi[i] = (_Float16)v[i];
    lo[i] = (_Float16)(v[i] - (float)hi[i]);
  }
  vf16x8* dst = reinterpret_cast<vf16x8*>(base + ((row * (size_t)(C >> 3) + (size_t)oct) * 2) * 8);
  dst[0] = hi;
  dst[1] = lo;
}

// ---------------------------------------------------------------------------------------------- tokenisation
// frames (n_seg, 3, 16, 224, 224) fp32 -> patch matrix in pair layout (n_seg * 1568, 1536): k = ((c * 2 + dt) * 16 + dy) * 16 + dx,
// the flattening order of the Conv3d weight (768, 3, 2, 16, 16) (vit_helper.py:543-548)
__global__ __launch_bounds__(256) void vit_patchify_kernel(const float* __restrict__ frames, uint16_t* __restrict__ P, int n_seg, int C,
                                                           int T, int HW, int pt, int ps) {
  const int gh = HW / ps;                       // 14
  const int K = C * pt * ps * ps;               // 1536
  const int64_t total = (int64_t)n_seg * (T / pt) * gh * gh * (K / 8);
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  const int oct = (int)(gid % (K / 8));
  const int64_t tok = gid / (K / 8);
  const int pw = (int)(tok % gh), ph = (int)((tok / gh) % gh), tf = (int)((tok / (gh * gh)) % (T / pt));
  const int seg = (int)(tok / ((int64_t)gh * gh * (T / pt)));
  const int k0 = oct * 8;
  const int dx = k0 % ps, dy = (k0 / ps) % ps, dt = (k0 / (ps * ps)) % pt, c = k0 / (ps * ps * pt);
  const float* src = frames + ((((size_t)seg * C + c) * T + (size_t)(tf * pt + dt)) * HW + (size_t)(ph * ps + dy)) * HW + (size_t)(pw * ps + dx);
  const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
  const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  put_pair8(P, (size_t)tok, oct, K, v);
}

// X[seg, 0] = cls + pos[0];  X[seg, 1 + f * n + i] += pos[1 + i] + temp[f]      (video_model_builder.py:240-249, 'separate')
__global__ __launch_bounds__(256) void vit_embed_kernel(float* __restrict__ X, const float* __restrict__ cls, const float* __restrict__ pos,
                                                        const float* __restrict__ temp, int n_seg, int nf, int np, int D) {
  const int L = 1 + nf * np;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // one float4
  if (gid >= (int64_t)n_seg * L * (D / 4)) return;
  const int cq = (int)(gid % (D / 4));
  const int64_t row = gid / (D / 4);
  const int r = (int)(row % L);
  f32x4* x = reinterpret_cast<f32x4*>(X) + gid;
  if (r == 0) {
    *x = reinterpret_cast<const f32x4*>(cls)[cq] + reinterpret_cast<const f32x4*>(pos)[cq];
  } else {
    const int f = (r - 1) / np, i = (r - 1) % np;
    *x = *x + (reinterpret_cast<const f32x4*>(pos)[(size_t)(1 + i) * (D / 4) + cq] + reinterpret_cast<const f32x4*>(temp)[(size_t)f * (D / 4) + cq]);
  }
}

// ---------------------------------------------------------------------------------------------- LayerNorm
// One wave per row, D = 768 (12 values per lane).  out = (x - mean) * rsqrt(var + eps) * w + b (biased variance, like torch).
// map 0: dst row = src row.  map 1 (final norm -> aggregation layer input): src rows are the patch tokens of X (CLS skipped),
// dst row = (seg * nf + f) * (np + 1) + 1 + i: slot 0 of every (segment, frame) sequence is left for the aggregation CLS token.
__global__ __launch_bounds__(256) void vit_ln_kernel(const float* __restrict__ X, const float* __restrict__ w, const float* __restrict__ b,
                                                     float* __restrict__ out_f32, uint16_t* __restrict__ out_pair, int64_t rows, int D,
                                                     float eps, int map, int nf, int np) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  int64_t src = r, dst = r;
  if (map == 1) {
    const int per = nf * np;
    const int64_t seg = r / per;
    const int t = (int)(r % per), f = t / np, i = t % np;
    src = seg * (per + 1) + 1 + t;
    dst = (seg * nf + f) * (np + 1) + 1 + i;
  }
  // lane holds octet `lane` and, where it exists, octet 64 + lane  (768 / 8 = 96 octets: lanes 0..31 hold two)
  const int nq = D / 8;
  const bool two = lane + 64 < nq;
  const float* xr = X + src * D;
  float v[16];
  {
    const f32x4 a0 = reinterpret_cast<const f32x4*>(xr)[2 * lane], a1 = reinterpret_cast<const f32x4*>(xr)[2 * lane + 1];
    v[0] = a0[0]; v[1] = a0[1]; v[2] = a0[2]; v[3] = a0[3]; v[4] = a1[0]; v[5] = a1[1]; v[6] = a1[2]; v[7] = a1[3];
    f32x4 b0 = f32x4{0.f, 0.f, 0.f, 0.f}, b1 = b0;
    if (two) { b0 = reinterpret_cast<const f32x4*>(xr)[2 * (lane + 64)]; b1 = reinterpret_cast<const f32x4*>(xr)[2 * (lane + 64) + 1]; }
    v[8] = b0[0]; v[9] = b0[1]; v[10] = b0[2]; v[11] = b0[3]; v[12] = b1[0]; v[13] = b1[1]; v[14] = b1[2]; v[15] = b1[3];
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float d = v[i] - mean;
    q += (i < 8 || two) ? d * d : 0.f;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
  for (int part = 0; part < 2; ++part) {
    if (part == 1 && !two) break;
    const int o = lane + 64 * part;
    float y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) y[i] = (v[8 * part + i] - mean) * rstd * w[8 * o + i] + b[8 * o + i];
    if (out_f32) {
      reinterpret_cast<f32x4*>(out_f32 + dst * D)[2 * o] = f32x4{y[0], y[1], y[2], y[3]};
      reinterpret_cast<f32x4*>(out_f32 + dst * D)[2 * o + 1] = f32x4{y[4], y[5], y[6], y[7]};
    }
    if (out_pair) put_pair8(out_pair, (size_t)dst, o, D, y);
  }
}

// rows dst[i * stride] <- vec (D floats): CLS slots of the aggregation sequences, residual rows of the aggregation layer
__global__ __launch_bounds__(256) void vit_fill_rows_kernel(float* __restrict__ dst, const float* __restrict__ vec, int64_t n, int64_t stride, int D) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n * (D / 4)) return;
  const int cq = (int)(gid % (D / 4));
  reinterpret_cast<f32x4*>(dst + (gid / (D / 4)) * stride * D)[cq] = reinterpret_cast<const f32x4*>(vec)[cq];
}

// ---------------------------------------------------------------------------------------------- attention
// QKV rows are (3 * D) floats: [q | k | v], each (heads x 64); q is scaled by 64^-0.5 = 0.125 here (vit_helper.py:119).
#define VHD 64

// One query = row seq * Lseq (the CLS token) of sequence `seq`, keys = the Lseq rows of that sequence.  grid (heads, n_seq).
// out_pair row = seq * out_stride.
__global__ __launch_bounds__(256) void vit_cls_attn_kernel(const float* __restrict__ qkv, uint16_t* __restrict__ out_pair, int Lseq, int D,
                                                           int64_t out_stride) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sc = sm;                 // Lseq scores -> probabilities
  float* red = sm + ((Lseq + 3) & ~3);   // 4 x 64 partial outputs, then 8 reduction slots
  const int h = blockIdx.x, seq = blockIdx.y, tid = threadIdx.x;
  const size_t row0 = (size_t)seq * Lseq;
  const float* base = qkv + row0 * 3 * D;
  const float* qp = base + h * VHD;
  float q[VHD];
#pragma unroll
  for (int i = 0; i < VHD / 4; ++i) {
    const f32x4 t = reinterpret_cast<const f32x4*>(qp)[i];
    q[4 * i] = t[0] * 0.125f; q[4 * i + 1] = t[1] * 0.125f; q[4 * i + 2] = t[2] * 0.125f; q[4 * i + 3] = t[3] * 0.125f;
  }
  float mx = -INFINITY;
  for (int j = tid; j < Lseq; j += 256) {
    const float* kp = base + (size_t)j * 3 * D + D + h * VHD;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VHD / 4; ++i) {
      const f32x4 t = reinterpret_cast<const f32x4*>(kp)[i];
      s = fmaf(q[4 * i], t[0], s); s = fmaf(q[4 * i + 1], t[1], s); s = fmaf(q[4 * i + 2], t[2], s); s = fmaf(q[4 * i + 3], t[3], s);
    }
    sc[j] = s;
    mx = fmaxf(mx, s);
  }
  float* rs = red + 4 * VHD;
  mx = wave_max(mx);
  if ((tid & 63) == 0) rs[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(rs[0], rs[1]), fmaxf(rs[2], rs[3]));
  float sum = 0.f;
  for (int j = tid; j < Lseq; j += 256) {
    const float e = expf(sc[j] - mx);
    sc[j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  if ((tid & 63) == 0) rs[4 + (tid >> 6)] = sum;
  __syncthreads();
  const float inv = 1.0f / (((rs[4] + rs[5]) + rs[6]) + rs[7]);
  // P.V: thread (d = tid & 63, part = tid >> 6) sums keys j = part, part + 4, ... (64 consecutive floats per key: coalesced)
  const int d = tid & 63, part = tid >> 6;
  float acc = 0.f;
  for (int j = part; j < Lseq; j += 4) acc = fmaf(sc[j], base[(size_t)j * 3 * D + 2 * D + h * VHD + d], acc);
  red[part * VHD + d] = acc;
  __syncthreads();
  if (tid < 8) {
    float y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int dd = tid * 8 + i;
      y[i] = (((red[dd] + red[VHD + dd]) + red[2 * VHD + dd]) + red[3 * VHD + dd]) * inv;
    }
    put_pair8(out_pair, (size_t)seq * out_stride, h * (VHD / 8) + tid, D, y);
  }
}

// time attention: query (seg, f, n) over keys {CLS, (seg, f', n) for f' in 0..nf-1}.  grid (np, n_seg), thread = (head, f).
template <int NFT>
__global__ __launch_bounds__(128) void vit_time_attn_kernel(const float* __restrict__ qkv, uint16_t* __restrict__ out_pair, int np,
                                                            int heads, int D) {
  constexpr int nf = NFT;
  const int n = blockIdx.x, seg = blockIdx.y, tid = threadIdx.x;
  const int h = tid / nf, f = tid % nf;
  if (h >= heads) return;
  const int L = 1 + nf * np;
  const size_t row0 = (size_t)seg * L;
  const size_t qrow = row0 + 1 + (size_t)f * np + n;
  const float* qp = qkv + qrow * 3 * D + h * VHD;
  float q[VHD];
#pragma unroll
  for (int i = 0; i < VHD / 4; ++i) {
    const f32x4 t = reinterpret_cast<const f32x4*>(qp)[i];
    q[4 * i] = t[0] * 0.125f; q[4 * i + 1] = t[1] * 0.125f; q[4 * i + 2] = t[2] * 0.125f; q[4 * i + 3] = t[3] * 0.125f;
  }
  float sc[NFT + 1];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j <= nf; ++j) {
    const size_t kr = j == 0 ? row0 : row0 + 1 + (size_t)(j - 1) * np + n;
    const float* kp = qkv + kr * 3 * D + D + h * VHD;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VHD / 4; ++i) {
      const f32x4 t = reinterpret_cast<const f32x4*>(kp)[i];
      s = fmaf(q[4 * i], t[0], s); s = fmaf(q[4 * i + 1], t[1], s); s = fmaf(q[4 * i + 2], t[2], s); s = fmaf(q[4 * i + 3], t[3], s);
    }
    sc[j] = s;
    mx = fmaxf(mx, s);
  }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j <= nf; ++j) { sc[j] = expf(sc[j] - mx); sum += sc[j]; }
  const float inv = 1.0f / sum;
  float o[VHD];
#pragma unroll
  for (int i = 0; i < VHD; ++i) o[i] = 0.f;
#pragma unroll
  for (int j = 0; j <= nf; ++j) {
    const size_t kr = j == 0 ? row0 : row0 + 1 + (size_t)(j - 1) * np + n;
    const float* vp = qkv + kr * 3 * D + 2 * D + h * VHD;
    const float pj = sc[j] * inv;
#pragma unroll
    for (int i = 0; i < VHD / 4; ++i) {
      const f32x4 t = reinterpret_cast<const f32x4*>(vp)[i];
      o[4 * i] = fmaf(pj, t[0], o[4 * i]); o[4 * i + 1] = fmaf(pj, t[1], o[4 * i + 1]);
      o[4 * i + 2] = fmaf(pj, t[2], o[4 * i + 2]); o[4 * i + 3] = fmaf(pj, t[3], o[4 * i + 3]);
    }
  }
#pragma unroll
  for (int oc = 0; oc < VHD / 8; ++oc) put_pair8(out_pair, qrow, h * (VHD / 8) + oc, D, o + 8 * oc);
}

// ---- round-2 forms of the CLS and time patterns: 16 lanes per (query, key row), each lane 4 of the head's 64 channels, so every
// load is a 256-byte run of one row (the kernels above read 16-byte pieces of 64 different rows per instruction) and a dot
// product is 4 fmaf + a 4-step xor reduction inside the 16-lane row.
__device__ __forceinline__ float row16_sum(float v) {
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
  return v;
}
__device__ __forceinline__ void put_pair4(uint16_t* base, size_t row, int c0, int C, const f32x4 v) {   // c0 % 4 == 0
  typedef _Float16 vf16x4 __attribute__((ext_vector_type(4)));
  vf16x4 hi, lo;
#pragma unroll
  for (int r = 0; r < 4; ++r) { hi[r] = (_Float16)v[r]; lo[r] = (_Float16)(v[r] - (float)hi[r]); }
  vf16x4* dst = reinterpret_cast<vf16x4*>(base + ((row * (size_t)(C >> 3) + (size_t)(c0 >> 3)) * 2) * 8 + (c0 & 7));
  dst[0] = hi;
  dst[2] = lo;
}

// CLS query of sequence `seq` over keys [split * chunk, ...) of its Lseq rows.  grid (heads, n_seq, nsplit).  nsplit == 1: the
// normalised result goes to out_pair row seq * out_stride; else the partial (max, sum, 64 unnormalised outputs) goes to
// part[((seq * heads + h) * nsplit + split) * 66] and vit_cls_combine_kernel finishes.
__global__ __launch_bounds__(256) void vit_cls_attn_split_kernel(const float* __restrict__ qkv, uint16_t* __restrict__ out_pair,
                                                                 float* __restrict__ part, int Lseq, int D, int64_t out_stride, int chunk) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sc = sm;                          // chunk scores -> probabilities
  float* red = sm + chunk;                 // 16 x 64 partial outputs | 16 partial sums | 4 wave maxima
  const int h = blockIdx.x, seq = blockIdx.y, split = blockIdx.z, nsplit = gridDim.z, tid = threadIdx.x;
  const int sub = tid & 15, ks = tid >> 4;
  const size_t row0 = (size_t)seq * Lseq;
  const float* base = qkv + row0 * 3 * D + h * VHD + 4 * sub;
  const int j0 = split * chunk, j1 = min(Lseq, j0 + chunk);
  f32x4 q = *reinterpret_cast<const f32x4*>(base);
  q = q * 0.125f;
  float mx = -INFINITY;
  for (int j = j0 + ks; j < j1; j += 16) {
    const f32x4 k = *reinterpret_cast<const f32x4*>(base + (size_t)j * 3 * D + D);
    const float s = row16_sum(fmaf(q[3], k[3], fmaf(q[2], k[2], fmaf(q[1], k[1], q[0] * k[0]))));
    if (sub == 0) sc[j - j0] = s;
    mx = fmaxf(mx, s);
  }
  float* rl = red + 16 * VHD;
  float* rm = rl + 16;
  mx = wave_max(mx);
  if ((tid & 63) == 0) rm[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(rm[0], rm[1]), fmaxf(rm[2], rm[3]));
  f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
  float l = 0.f;
  for (int j = j0 + ks; j < j1; j += 16) {
    const float e = expf(sc[j - j0] - mx);
    l += e;
    const f32x4 v = *reinterpret_cast<const f32x4*>(base + (size_t)j * 3 * D + 2 * D);
    o[0] = fmaf(e, v[0], o[0]); o[1] = fmaf(e, v[1], o[1]); o[2] = fmaf(e, v[2], o[2]); o[3] = fmaf(e, v[3], o[3]);
  }
  *reinterpret_cast<f32x4*>(red + ks * VHD + 4 * sub) = o;
  if (sub == 0) rl[ks] = l;
  __syncthreads();
  if (tid < VHD) {
    float y = 0.f, lt = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { y += red[i * VHD + tid]; lt += rl[i]; }
    if (nsplit == 1) {
      y *= 1.0f / lt;
      const _Float16 hi = (_Float16)y, lo = (_Float16)(y - (float)hi);
      _Float16* dst = reinterpret_cast<_Float16*>(out_pair) + (((size_t)seq * out_stride * (size_t)(D >> 3) + (size_t)(h * (VHD / 8) + (tid >> 3))) * 2) * 8 + (tid & 7);
      dst[0] = hi;
      dst[8] = lo;
    } else {
      float* p = part + ((size_t)(seq * gridDim.x + h) * nsplit + split) * 66;
      p[2 + tid] = y;
      if (tid == 0) { p[0] = mx; p[1] = lt; }
    }
  }
}

// grid (heads, n_seq), 64 threads: merges the nsplit partials of one (sequence, head)
__global__ __launch_bounds__(64) void vit_cls_combine_kernel(const float* __restrict__ part, uint16_t* __restrict__ out_pair, int nsplit, int D,
                                                             int64_t out_stride) {
  const int h = blockIdx.x, seq = blockIdx.y, tid = threadIdx.x;
  const float* p = part + (size_t)(seq * gridDim.x + h) * nsplit * 66;
  float M = -INFINITY;
  for (int i = 0; i < nsplit; ++i) M = fmaxf(M, p[i * 66]);
  float L = 0.f, y = 0.f;
  for (int i = 0; i < nsplit; ++i) {
    const float w = expf(p[i * 66] - M);
    L = fmaf(w, p[i * 66 + 1], L);
    y = fmaf(w, p[i * 66 + 2 + tid], y);
  }
  y *= 1.0f / L;
  const _Float16 hi = (_Float16)y, lo = (_Float16)(y - (float)hi);
  _Float16* dst = reinterpret_cast<_Float16*>(out_pair) + (((size_t)seq * out_stride * (size_t)(D >> 3) + (size_t)(h * (VHD / 8) + (tid >> 3))) * 2) * 8 + (tid & 7);
  dst[0] = hi;
  dst[8] = lo;
}

// time pattern, 16 lanes per (frame f, head h) query of location n: grid (heads / 2, np, n_seg), 256 threads = 2 heads x 8 frames x 16
template <int NFT>
__global__ __launch_bounds__(256) void vit_time_attn16_kernel(const float* __restrict__ qkv, uint16_t* __restrict__ out_pair, int np, int D) {
  constexpr int nf = NFT;
  const int tid = threadIdx.x, sub = tid & 15, pr = tid >> 4;
  const int f = pr % nf, h = blockIdx.x * 2 + pr / nf;
  const int n = blockIdx.y, seg = blockIdx.z;
  const int L = 1 + nf * np;
  const size_t row0 = (size_t)seg * L;
  const size_t qrow = row0 + 1 + (size_t)f * np + n;
  const float* col = qkv + h * VHD + 4 * sub;
  f32x4 q = *reinterpret_cast<const f32x4*>(col + qrow * 3 * D);
  q = q * 0.125f;
  float sc[NFT + 1];
  f32x4 vv[NFT + 1];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j <= nf; ++j) {
    const size_t kr = j == 0 ? row0 : row0 + 1 + (size_t)(j - 1) * np + n;
    const f32x4 k = *reinterpret_cast<const f32x4*>(col + kr * 3 * D + D);
    vv[j] = *reinterpret_cast<const f32x4*>(col + kr * 3 * D + 2 * D);
    sc[j] = row16_sum(fmaf(q[3], k[3], fmaf(q[2], k[2], fmaf(q[1], k[1], q[0] * k[0]))));
    mx = fmaxf(mx, sc[j]);
  }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j <= nf; ++j) { sc[j] = expf(sc[j] - mx); sum += sc[j]; }
  const float inv = 1.0f / sum;
  f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j <= nf; ++j) {
    const float pj = sc[j] * inv;
    o[0] = fmaf(pj, vv[j][0], o[0]); o[1] = fmaf(pj, vv[j][1], o[1]); o[2] = fmaf(pj, vv[j][2], o[2]); o[3] = fmaf(pj, vv[j][3], o[3]);
  }
  put_pair4(out_pair, qrow, h * VHD + 4 * sub, D, o);
}

// space attention: query (seg, f, n) over keys {CLS, (seg, f, n') for n' in 0..np-1}.  grid (heads, nf, n_seg); the frame's K and
// V rows of this head (np + 1 <= 197 rows x 64) are staged in LDS once, every thread is one query and reads them as broadcasts;
// online softmax over chunks of 16 keys.
#define VS_CH 16
__global__ __launch_bounds__(256) void vit_space_attn_kernel(const float* __restrict__ qkv, uint16_t* __restrict__ out_pair, int nf, int np, int D) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int nk = np + 1;
  float* Ks = sm;                           // nk x 64
  float* Vs = sm + (size_t)nk * VHD;
  const int h = blockIdx.x, f = blockIdx.y, seg = blockIdx.z, tid = threadIdx.x;
  const int L = 1 + nf * np;
  const size_t row0 = (size_t)seg * L;
  for (int u = tid; u < nk * (VHD / 4); u += 256) {
    const int j = u / (VHD / 4), c = u % (VHD / 4);
    const size_t kr = j == 0 ? row0 : row0 + 1 + (size_t)f * np + (j - 1);
    reinterpret_cast<f32x4*>(Ks)[u] = reinterpret_cast<const f32x4*>(qkv + kr * 3 * D + D + h * VHD)[c];
    reinterpret_cast<f32x4*>(Vs)[u] = reinterpret_cast<const f32x4*>(qkv + kr * 3 * D + 2 * D + h * VHD)[c];
  }
  __syncthreads();
  if (tid >= np) return;
  const size_t qrow = row0 + 1 + (size_t)f * np + tid;
  const float* qp = qkv + qrow * 3 * D + h * VHD;
  float q[VHD], o[VHD];
#pragma unroll
  for (int i = 0; i < VHD / 4; ++i) {
    const f32x4 t = reinterpret_cast<const f32x4*>(qp)[i];
    q[4 * i] = t[0] * 0.125f; q[4 * i + 1] = t[1] * 0.125f; q[4 * i + 2] = t[2] * 0.125f; q[4 * i + 3] = t[3] * 0.125f;
  }
#pragma unroll
  for (int i = 0; i < VHD; ++i) o[i] = 0.f;
  float m = -INFINITY, l = 0.f;
  for (int j0 = 0; j0 < nk; j0 += VS_CH) {
    float sc[VS_CH];
    float cm = -INFINITY;
#pragma unroll
    for (int u = 0; u < VS_CH; ++u) {
      const int j = j0 + u;
      float s = -INFINITY;
      if (j < nk) {
        s = 0.f;
#pragma unroll
        for (int i = 0; i < VHD / 4; ++i) {
          const f32x4 t = reinterpret_cast<const f32x4*>(Ks + (size_t)j * VHD)[i];
          s = fmaf(q[4 * i], t[0], s); s = fmaf(q[4 * i + 1], t[1], s); s = fmaf(q[4 * i + 2], t[2], s); s = fmaf(q[4 * i + 3], t[3], s);
        }
      }
      sc[u] = s;
      cm = fmaxf(cm, s);
    }
    const float mn = fmaxf(m, cm);
    const float fs = expf(m - mn);          // exp(-inf) = 0 on the first chunk
    l *= fs;
#pragma unroll
    for (int i = 0; i < VHD; ++i) o[i] *= fs;
#pragma unroll
    for (int u = 0; u < VS_CH; ++u) {
      const int j = j0 + u;
      if (j < nk) {
        const float e = expf(sc[u] - mn);
        l += e;
#pragma unroll
        for (int i = 0; i < VHD / 4; ++i) {
          const f32x4 t = reinterpret_cast<const f32x4*>(Vs + (size_t)j * VHD)[i];
          o[4 * i] = fmaf(e, t[0], o[4 * i]); o[4 * i + 1] = fmaf(e, t[1], o[4 * i + 1]);
          o[4 * i + 2] = fmaf(e, t[2], o[4 * i + 2]); o[4 * i + 3] = fmaf(e, t[3], o[4 * i + 3]);
        }
      }
    }
    m = mn;
  }
  const float inv = 1.0f / l;
#pragma unroll
  for (int i = 0; i < VHD; ++i) o[i] *= inv;
#pragma unroll
  for (int oc = 0; oc < VHD / 8; ++oc) put_pair8(out_pair, qrow, h * (VHD / 8) + oc, D, o + 8 * oc);
}

// The same space attention on the matrix cores, exact fp32 (v_mfma_f32_16x16x4_f32 = an fmaf chain).  One workgroup per (head,
// frame, segment) as above; K (stride 65 floats) and V (stride 68) of the frame in LDS, both bank-conflict-free for the fragment
// reads below; a wave takes 16 queries at a time:
//   S^T tile (16 keys x 16 queries) = K_tile . Q^T over d: A = K[key = lane&15][d], B = Q[q = lane&15][d] with the k slots of lane
//     group g standing for d = 16g .. 16g+15 (any bijection will do, A and B use the same) -> a lane ends up with the scores of
//     ITS query (lane&15) against keys 4g + reg of every tile: all 13 tiles stay in registers (52), so softmax is one pass —
//     max and sum over a query's keys are two lane shuffles (xor 16, xor 32) plus register work;
//   O^T tile (16 d x 16 queries) = V^T . P^T over keys: A = V[key = 4g + s][d = lane&15], B = P[q = lane&15][key = 4g + s] — exactly
//     the register the lane already holds (the accumulator-as-operand idiom): no transposition, no LDS round trip for P.
#define VSK 65
#define VSV 68
#define VS_NKT 13                                // key tiles: np + 1 <= 208
__global__ __launch_bounds__(512) void vit_space_attn_mfma_kernel(const float* __restrict__ qkv, uint16_t* __restrict__ out_pair, int nf, int np,
                                                                  int D) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int nk = np + 1;
  float* Ks = sm;                                // 208 x VSK
  float* Vs = sm + VS_NKT * 16 * VSK;            // 208 x VSV
  const int h = blockIdx.x, f = blockIdx.y, seg = blockIdx.z, tid = threadIdx.x;
  const int L = 1 + nf * np;
  const size_t row0 = (size_t)seg * L;
  for (int u = tid; u < VS_NKT * 16 * (VHD / 4); u += 512) {
    const int j = u / (VHD / 4), c = u % (VHD / 4);
    f32x4 kq = f32x4{0.f, 0.f, 0.f, 0.f}, vq = kq;
    if (j < nk) {
      const size_t kr = j == 0 ? row0 : row0 + 1 + (size_t)f * np + (j - 1);
      kq = reinterpret_cast<const f32x4*>(qkv + kr * 3 * D + D + h * VHD)[c];
      vq = reinterpret_cast<const f32x4*>(qkv + kr * 3 * D + 2 * D + h * VHD)[c];
    }
    float* kd = Ks + j * VSK + 4 * c;
    kd[0] = kq[0]; kd[1] = kq[1]; kd[2] = kq[2]; kd[3] = kq[3];
    *reinterpret_cast<f32x4*>(Vs + j * VSV + 4 * c) = vq;
  }
  __syncthreads();
  const int lane = tid & 63, wv = tid >> 6, r16 = lane & 15, g = lane >> 4;
  const int nqt = (np + 15) / 16;
  for (int qt = wv; qt < nqt; qt += 8) {      // 13 query tiles over 8 waves (two waves per SIMD)
    const int qi = qt * 16 + r16;
    const size_t qrow = row0 + 1 + (size_t)f * np + (qi < np ? qi : np - 1);
    float q[16];
    {
      const f32x4* qp = reinterpret_cast<const f32x4*>(qkv + qrow * 3 * D + h * VHD + 16 * g);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 t = qp[i];
        q[4 * i] = t[0] * 0.125f; q[4 * i + 1] = t[1] * 0.125f; q[4 * i + 2] = t[2] * 0.125f; q[4 * i + 3] = t[3] * 0.125f;
      }
    }
    f32x4 st[VS_NKT];
    float m = -INFINITY;
    // two key tiles at a time: two independent accumulation chains keep the matrix pipe issuing back to back
#pragma unroll
    for (int kt = 0; kt < VS_NKT; kt += 2) {
      f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
      const float* kp0 = Ks + (kt * 16 + r16) * VSK + 16 * g;
      const float* kp1 = kp0 + 16 * VSK;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kp0[s], q[s], acc0, 0, 0, 0);
        if (kt + 1 < VS_NKT) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kp1[s], q[s], acc1, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (kt * 16 + 4 * g + r >= nk) acc0[r] = -INFINITY;
        m = fmaxf(m, acc0[r]);
      }
      st[kt] = acc0;
      if (kt + 1 < VS_NKT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if ((kt + 1) * 16 + 4 * g + r >= nk) acc1[r] = -INFINITY;
          m = fmaxf(m, acc1[r]);
        }
        st[kt + 1] = acc1;
      }
      __builtin_amdgcn_sched_barrier(0);     // keep the fully unrolled body from hoisting every tile's LDS reads (spills)
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < VS_NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = expf(st[kt][r] - m);
        st[kt][r] = e;
        l += e;
      }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    f32x4 oacc[VHD / 16];
#pragma unroll
    for (int dt = 0; dt < VHD / 16; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < VS_NKT; ++kt) {
      const float* vp = Vs + (kt * 16 + 4 * g) * VSV + r16;
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int dt = 0; dt < VHD / 16; ++dt)      // four independent chains
          oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[s * VSV + dt * 16], st[kt][s], oacc[dt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int dt = 0; dt < VHD / 16; ++dt) {
      const f32x4 acc = oacc[dt];
      if (qi < np) {      // the lane holds channels c0 .. c0+3 of its query: half an octet of the pair layout
        const int c0 = h * VHD + dt * 16 + 4 * g;
        typedef _Float16 vf16x4 __attribute__((ext_vector_type(4)));
        vf16x4 hi, lo;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float o = acc[r] * inv;
          hi[r] = (_Float16)o;
          lo[r] = (_Float16)(o - (float)hi[r]);
        }
        vf16x4* dst = reinterpret_cast<vf16x4*>(out_pair + ((qrow * (size_t)(D >> 3) + (size_t)(c0 >> 3)) * 2) * 8 + (c0 & 7));
        dst[0] = hi;
        dst[2] = lo;      // + 8 halves: the lo plane of the same octet
      }
    }
  }
}

// The same dataflow on fp16 pairs (x = hi + lo, 22 significand bits; three v_mfma_f32_16x16x32_f16 per product like every linear
// layer of this file): 5x fewer matrix cycles than the exact-fp32 instruction.
// LDS images (round 4): K AND V as hi / lo planes in ONE row-major form, [key][64 channels] halves with 160-byte rows.
//  * S = K . Q^T reads a K fragment (key r16 of a tile, k-octet 4 st + g) with one ds_read_b128: rows 160 B = 40 banks apart, so the 16
//    rows of a service group (8 of octet g, the other 8 of octet g + 1: MI355X_MICROARCH.md, LDS) start on the 16 different multiples
//    of 4 banks — conflict-free.
//  * O^T = V^T . P^T sums over keys: a lane's 8 k slots are keys 4 g .. 4 g + 3 of two adjacent key tiles — the registers it already
//    holds — and the matching V^T fragment comes from the row-major image with ds_read_b64_tr_b16 (the transposing read of gfx950:
//    per 16-lane group a block of 4 keys x 16 channels, delivered channel-major); the two blocks of a 32-lane half are 8 consecutive
//    keys of the same channels = 8 rows x 32 B on bank bases 0, 40, 16, 56, 32, 8, 48, 24: conflict-free.
//  * staging is the same for both: one octet of one key per item, (hi, lo) as two ds_write_b128 to consecutive 16-byte slots.
// Round 3 kept K k-major ([k-octet][key]: the 8 lanes that stage one key hit one bank, 8-way) and V transposed ([channel][key]: 56
// two-byte stores per thread, most of them conflicting): SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.60 (profiles/r03_avclip_mfma.json).
#define VP_KEYS (VS_NKT * 16)                      // 208
#define VP_ROWH 80                                 // halves per key row (160 B)
#define VP_VROWS (((VS_NKT + 1) / 2) * 32)         // 224: the P.V instruction walks 32 keys at a time; rows past the last tile are zeros
__global__ __launch_bounds__(512) void vit_space_attn_pair_kernel(const float* __restrict__ qkv, uint16_t* __restrict__ out_pair, int nf, int np,
                                                                  int D) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  typedef __fp16 fp4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
  _Float16* Kh = reinterpret_cast<_Float16*>(smraw);                       // [208][80]
  _Float16* Kl = Kh + VP_KEYS * VP_ROWH;
  _Float16* Vh = Kl + VP_KEYS * VP_ROWH;                                   // [224][80]
  _Float16* Vl = Vh + VP_VROWS * VP_ROWH;
  const int nk = np + 1;
  const int h = blockIdx.x, f = blockIdx.y, seg = blockIdx.z, tid = threadIdx.x;
  const int L = 1 + nf * np;
  const size_t row0 = (size_t)seg * L;
  // Staging: EVERY global load of this thread is requested before the first is converted (4 x 2 quads of K and of V: 64 registers).
  constexpr int KIT = (VP_KEYS * 8 + 511) / 512;
  f32x4 kreg[KIT][2], vreg[KIT][2];
#pragma unroll
  for (int it = 0; it < KIT; ++it) {       // one octet of one key per item
    const int u = tid + it * 512, j = min(u >> 3, nk - 1), kq = u & 7;
    const size_t kr = j == 0 ? row0 : row0 + 1 + (size_t)f * np + (j - 1);
    const f32x4* kp = reinterpret_cast<const f32x4*>(qkv + kr * 3 * D + D + h * VHD + 8 * kq);
    kreg[it][0] = kp[0];
    kreg[it][1] = kp[1];
  }
#pragma unroll
  for (int it = 0; it < KIT; ++it) {
    const int u = tid + it * 512, j = min(u >> 3, nk - 1), kq = u & 7;
    const size_t kr = j == 0 ? row0 : row0 + 1 + (size_t)f * np + (j - 1);
    const f32x4* vp = reinterpret_cast<const f32x4*>(qkv + kr * 3 * D + 2 * D + h * VHD + 8 * kq);
    vreg[it][0] = vp[0];
    vreg[it][1] = vp[1];
  }
  auto park = [&](_Float16* hp, _Float16* lp, const f32x4 a, const f32x4 b, int j, int kq) {
    h8 hi, lo;
    const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float x = j < nk ? v[i] : 0.f;          // keys >= nk: zeros
      hi[i] = (_Float16)x;
      lo[i] = (_Float16)(x - (float)hi[i]);
    }
    *reinterpret_cast<h8*>(hp + j * VP_ROWH + 8 * kq) = hi;
    *reinterpret_cast<h8*>(lp + j * VP_ROWH + 8 * kq) = lo;
  };
#pragma unroll
  for (int it = 0; it < KIT; ++it) {
    const int u = tid + it * 512;
    if (u >= VP_KEYS * 8) break;
    park(Kh, Kl, kreg[it][0], kreg[it][1], u >> 3, u & 7);
  }
#pragma unroll
  for (int it = 0; it < KIT; ++it) {
    const int u = tid + it * 512;
    if (u >= VP_KEYS * 8) break;
    park(Vh, Vl, vreg[it][0], vreg[it][1], u >> 3, u & 7);
  }
  if (tid < (VP_VROWS - VP_KEYS) * 8) {             // V rows 208 .. 223: multiplied by P = 0, so they must be finite
    const h8 z = h8{(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    *reinterpret_cast<h8*>(Vh + (VP_KEYS + (tid >> 3)) * VP_ROWH + 8 * (tid & 7)) = z;
    *reinterpret_cast<h8*>(Vl + (VP_KEYS + (tid >> 3)) * VP_ROWH + 8 * (tid & 7)) = z;
  }
  __syncthreads();
  const int lane = tid & 63, wv = tid >> 6, r16 = lane & 15, g = lane >> 4;
  const int nqt = (np + 15) / 16;
  for (int qt = wv; qt < nqt; qt += 8) {
    const int qi = qt * 16 + r16;
    const size_t qrow = row0 + 1 + (size_t)f * np + (qi < np ? qi : np - 1);
    h8 qh[2], ql[2];
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const f32x4* qp = reinterpret_cast<const f32x4*>(qkv + qrow * 3 * D + h * VHD + 32 * st + 8 * g);
      const f32x4 a = qp[0], b = qp[1];
      const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float x = v[i] * 0.125f;
        qh[st][i] = (_Float16)x;
        ql[st][i] = (_Float16)(x - (float)qh[st][i]);
      }
    }
    f32x4 sc[VS_NKT];
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < VS_NKT; ++kt) {
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const h8 kh = *reinterpret_cast<const h8*>(Kh + (kt * 16 + r16) * VP_ROWH + 8 * (4 * st + g));
        const h8 kl = *reinterpret_cast<const h8*>(Kl + (kt * 16 + r16) * VP_ROWH + 8 * (4 * st + g));
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl, qh[st], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, ql[st], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, qh[st], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (kt * 16 + 4 * g + r >= nk) acc[r] = -INFINITY;
        m = fmaxf(m, acc[r]);
      }
      sc[kt] = acc;
      if (kt & 1) __builtin_amdgcn_sched_barrier(0);
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < VS_NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = expf(sc[kt][r] - m);
        sc[kt][r] = e;
        l += e;
      }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    f32x4 oacc[VHD / 16];
#pragma unroll
    for (int dt = 0; dt < VHD / 16; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mp = 0; mp < (VS_NKT + 1) / 2; ++mp) {       // 32 keys per instruction: key tiles 2 mp and 2 mp + 1
      h8 ph, pl;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p0 = sc[2 * mp][r], p1 = 2 * mp + 1 < VS_NKT ? sc[2 * mp + 1][r] : 0.f;
        ph[r] = (_Float16)p0; pl[r] = (_Float16)(p0 - (float)ph[r]);
        ph[r + 4] = (_Float16)p1; pl[r + 4] = (_Float16)(p1 - (float)ph[r + 4]);
      }
#pragma unroll
      for (int dt = 0; dt < VHD / 16; ++dt) {
        // transposing reads: lane 4 q + p of a 16-lane group addresses key 32 mp + 4 g + q, channels 16 dt + 4 p .. + 3, and receives
        // channel 16 dt + r16 of the group's four keys (EXEC is all ones here: the loop over q tiles is wave-uniform)
        const int off = (32 * mp + 4 * g + (r16 >> 2)) * VP_ROWH + dt * 16 + 4 * (r16 & 3);
        typedef __attribute__((address_space(3))) fp4* lds_fp4;
        const h4 a0 = __builtin_bit_cast(h4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp4)(Vh + off)));
        const h4 a1 = __builtin_bit_cast(h4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp4)(Vh + off + 16 * VP_ROWH)));
        const h4 b0 = __builtin_bit_cast(h4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp4)(Vl + off)));
        const h4 b1 = __builtin_bit_cast(h4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp4)(Vl + off + 16 * VP_ROWH)));
        const h8 vh = h8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        const h8 vl = h8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, ph, oacc[dt], 0, 0, 0);
        oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, pl, oacc[dt], 0, 0, 0);
        oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, ph, oacc[dt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (qi < np) {
#pragma unroll
      for (int dt = 0; dt < VHD / 16; ++dt) put_pair4(out_pair, qrow, h * VHD + dt * 16 + 4 * g, D, oacc[dt] * inv);
    }
  }
}

// ---------------------------------------------------------------------------------------------- driver
static int ln(const vaura_vit* v, const float* X, const float* w, const float* b, float* of, uint16_t* op, int64_t rows, int map, hipStream_t s) {
  VA_LAUNCH(vit_ln_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, X, w, b, of, op, rows, v->dim, v->eps, map, v->n_frames, v->n_patches);
  return 0;
}

// CLS pattern launcher: long sequences are split over 8 workgroups per (sequence, head) — 384 workgroups of 1 569 keys each left
// a third of the chip idle and took 205 us for 308 MB — with the partials in `part` (>= n_seq * heads * 8 * 66 floats).
static int cls_attention(const vaura_vit* v, const float* qkv, uint16_t* out, float* part, int n_seq, int Lseq, int64_t out_stride, hipStream_t s) {
  const int D = v->dim;
  if (va_debug_flags_get() & 512) {     // debug flag bit 9: the one-workgroup-per-(sequence, head) kernel
    const size_t sm_cls = sizeof(float) * (size_t)(((Lseq + 3) & ~3) + 4 * VHD + 8);
    VA_LAUNCH(vit_cls_attn_kernel, dim3(v->heads, (unsigned)n_seq), dim3(256), sm_cls, s, qkv, out, Lseq, D, out_stride);
    return 0;
  }
  const int nsplit = (Lseq > 512 && part) ? 8 : 1;
  const int chunk = (((Lseq + nsplit - 1) / nsplit) + 15) & ~15;
  const size_t smb = sizeof(float) * (size_t)(chunk + 16 * VHD + 16 + 4);
  VA_LAUNCH(vit_cls_attn_split_kernel, dim3(v->heads, (unsigned)n_seq, nsplit), dim3(256), smb, s, qkv, out, part, Lseq, D, out_stride, chunk);
  if (nsplit > 1) VA_LAUNCH(vit_cls_combine_kernel, dim3(v->heads, (unsigned)n_seq), dim3(64), 0, s, (const float*)part, out, nsplit, D, out_stride);
  return 0;
}

static int divided_attention(const vaura_vit* v, const vaura_vit_attn& at, const float* ln_w, const float* ln_b, bool time, int n_seg,
                             hipStream_t s) {
  const int D = v->dim, L = 1 + v->n_frames * v->n_patches;
  const int64_t N = (int64_t)n_seg * L;
  int rc = ln(v, v->ws_x, ln_w, ln_b, nullptr, v->ws_a, N, 0, s);
  if (rc) return rc;
  rc = va_launch_linear_pair(v->ws_a, (const uint16_t*)at.qkv_w, at.qkv_b, nullptr, v->ws_qkv, nullptr, 2, 1, (int)N, (int)N, 0, D, 3 * D, s);
  if (rc) return rc;
  rc = cls_attention(v, (const float*)v->ws_qkv, v->ws_a, v->ws_s /* free until the aggregation layer */, n_seg, L, (int64_t)L, s);
  if (rc) return rc;
  if (time) {
    if ((va_debug_flags_get() & 1024) || (v->heads & 1))      // debug flag bit 10: one thread per (head, frame)
      VA_LAUNCH(vit_time_attn_kernel<8>, dim3(v->n_patches, n_seg), dim3(128), 0, s, (const float*)v->ws_qkv, v->ws_a, v->n_patches,
                v->heads, D);
    else
      VA_LAUNCH(vit_time_attn16_kernel<8>, dim3(v->heads / 2, v->n_patches, n_seg), dim3(256), 0, s, (const float*)v->ws_qkv, v->ws_a,
                v->n_patches, D);
  } else {
    if (v->n_patches + 1 <= VS_NKT * 16 && !(va_debug_flags_get() & (128 | 2048))) {   // the fp16-pair MFMA kernel
      const size_t sm = (size_t)2 * (VP_KEYS + VP_VROWS) * VP_ROWH * 2;               // 135 KB of the CU's 160 KB
      static unsigned long long big_lds_p = 0;
      if (va_big_lds_once(reinterpret_cast<const void*>(vit_space_attn_pair_kernel), sm, &big_lds_p)) return VAURA_ERR_STATE;
      VA_LAUNCH(vit_space_attn_pair_kernel, dim3(v->heads, v->n_frames, n_seg), dim3(512), sm, s, (const float*)v->ws_qkv, v->ws_a, v->n_frames,
                v->n_patches, D);
    } else if (v->n_patches + 1 <= VS_NKT * 16 && !(va_debug_flags_get() & 128)) {   // debug flag bit 11: exact-fp32 MFMA; bit 7: one thread per query
      const size_t sm = sizeof(float) * (size_t)(VS_NKT * 16) * (VSK + VSV);       // 108 KB of the CU's 160 KB
      static unsigned long long big_lds_m = 0;
      if (va_big_lds_once(reinterpret_cast<const void*>(vit_space_attn_mfma_kernel), sm, &big_lds_m)) return VAURA_ERR_STATE;
      VA_LAUNCH(vit_space_attn_mfma_kernel, dim3(v->heads, v->n_frames, n_seg), dim3(512), sm, s, (const float*)v->ws_qkv, v->ws_a, v->n_frames,
                v->n_patches, D);
    } else {
      const size_t sm = sizeof(float) * 2 * (size_t)(v->n_patches + 1) * VHD;      // 100.9 KB of the CU's 160 KB
      static unsigned long long big_lds = 0;
      if (va_big_lds_once(reinterpret_cast<const void*>(vit_space_attn_kernel), sm, &big_lds)) return VAURA_ERR_STATE;
      VA_LAUNCH(vit_space_attn_kernel, dim3(v->heads, v->n_frames, n_seg), dim3(256), sm, s, (const float*)v->ws_qkv, v->ws_a, v->n_frames,
                v->n_patches, D);
    }
  }
  // x = x + proj(attention)      vit_helper.py:452-468
  return va_launch_linear_pair(v->ws_a, (const uint16_t*)at.proj_w, at.proj_b, v->ws_x, v->ws_x, nullptr, 2, 1, (int)N, (int)N, 0, D, D, s);
}

extern "C" {

size_t vaura_avclip_workspace_bytes(const vaura_vit* v, int n_seg, int which) {
  if (!v || n_seg <= 0) return 0;
  const size_t L = 1 + (size_t)v->n_frames * v->n_patches, N = (size_t)n_seg * L, D = v->dim;
  const size_t nq = (size_t)n_seg * v->n_frames, NA = nq * (v->n_patches + 1);
  switch (which) {
    case 0: return N * D * 4;                                         // ws_x    fp32 token rows
    case 1: return N * 3 * D * 4 > NA * 3 * D * 4 ? N * 3 * D * 4 : NA * 3 * D * 4;   // ws_qkv
    case 2: return (N > NA ? N : NA) * D * 4;                         // ws_a    pair layout (4 bytes per element)
    case 3: return N * v->hidden * 4;                                 // ws_h    pair layout
    case 4: return (size_t)n_seg * v->n_frames * v->n_patches * v->patch_k * 4;      // ws_p    pair layout patches
    case 5: return NA * D * 4;                                        // ws_z    fp32 aggregation sequences
    case 6: return nq * D * 4 * 4 + nq * v->hidden * 4;               // ws_s    small rows: r0 | x0 | a0 (pair) | spare | h0 (pair)
    default: return 0;
  }
}

int vaura_avclip_forward(const vaura_vit* v, const float* frames, int n_seg, float* feats, vaura_stream_t s_) {
  if (!v || !frames || !feats || n_seg <= 0 || !v->blocks_host) return VAURA_ERR_ARG;
  if (v->dim != 768 || v->heads * VHD != v->dim || (v->hidden % 96) || (v->patch_k % 32) || v->n_frames != 8 || v->heads * v->n_frames > 128 ||
      v->n_patches > 255)
    return VAURA_ERR_SHAPE;
  if (!v->ws_x || !v->ws_qkv || !v->ws_a || !v->ws_h || !v->ws_p || !v->ws_z || !v->ws_s) return VAURA_ERR_ARG;
  hipStream_t s = as_stream(s_);
  const int D = v->dim, nf = v->n_frames, np = v->n_patches, L = 1 + nf * np;
  const int64_t N = (int64_t)n_seg * L;
  int rc;
  // ---- tokens: 3-D patch embedding as a GEMM into rows 1.. of every sequence, then CLS + positional embeddings
  {
    const int64_t total = (int64_t)n_seg * nf * np * (v->patch_k / 8);
    VA_LAUNCH(vit_patchify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, frames, v->ws_p, n_seg, v->in_chans, v->frames,
              v->img, v->patch_t, v->patch);
    rc = va_launch_linear_pair(v->ws_p, (const uint16_t*)v->pe_w, v->pe_b, nullptr, v->ws_x, nullptr, 2, n_seg, nf * np, L, 1, v->patch_k, D, s);
    if (rc) return rc;
    const int64_t q4 = N * (D / 4);
    VA_LAUNCH(vit_embed_kernel, dim3((unsigned)((q4 + 255) / 256)), dim3(256), 0, s, v->ws_x, v->cls_token, v->pos_embed, v->temp_embed, n_seg,
              nf, np, D);
  }
  // ---- 12 divided space-time blocks                                  vit_helper.py:443-472
  for (int i = 0; i < v->depth; ++i) {
    const vaura_vit_block& b = v->blocks_host[i];
    rc = divided_attention(v, b.time, b.ln3_w, b.ln3_b, true, n_seg, s);
    if (rc) return rc;
    rc = divided_attention(v, b.space, b.ln1_w, b.ln1_b, false, n_seg, s);
    if (rc) return rc;
    rc = ln(v, v->ws_x, b.ln2_w, b.ln2_b, nullptr, v->ws_a, N, 0, s);
    if (rc) return rc;
    rc = va_launch_linear_pair(v->ws_a, (const uint16_t*)b.fc1_w, b.fc1_b, nullptr, nullptr, v->ws_h, 1, 1, (int)N, (int)N, 0, D, v->hidden, s);
    if (rc) return rc;
    rc = va_launch_linear_pair(v->ws_h, (const uint16_t*)b.fc2_w, b.fc2_b, v->ws_x, v->ws_x, nullptr, 2, 1, (int)N, (int)N, 0, v->hidden, D, s);
    if (rc) return rc;
  }
  // ---- final norm without the CLS token, regrouped as (segment, frame) sequences of 1 + np rows    motionformer.py:311-330
  const int64_t nq = (int64_t)n_seg * nf, NA = nq * (np + 1);
  rc = ln(v, v->ws_x, v->norm_w, v->norm_b, v->ws_z, nullptr, (int64_t)n_seg * nf * np, 1, s);
  if (rc) return rc;
  VA_LAUNCH(vit_fill_rows_kernel, dim3((unsigned)((nq * (D / 4) + 255) / 256)), dim3(256), 0, s, v->ws_z, v->agg_cls, nq, (int64_t)(np + 1), D);
  // ---- spatial aggregation: one pre-norm encoder layer; only its CLS row is used                   motionformer.py:399-448
  float* r0 = v->ws_s;                                  // residual rows (the CLS token) -> x0 after the attention
  float* x0 = r0 + nq * D;
  uint16_t* a0 = reinterpret_cast<uint16_t*>(x0 + nq * D);
  uint16_t* h0 = reinterpret_cast<uint16_t*>(x0 + 3 * nq * D);
  rc = ln(v, v->ws_z, v->agg_ln1_w, v->agg_ln1_b, nullptr, v->ws_a, NA, 0, s);
  if (rc) return rc;
  rc = va_launch_linear_pair(v->ws_a, (const uint16_t*)v->agg_in_w, v->agg_in_b, nullptr, v->ws_qkv, nullptr, 2, 1, (int)NA, (int)NA, 0, D, 3 * D, s);
  if (rc) return rc;
  rc = cls_attention(v, (const float*)v->ws_qkv, a0, nullptr, (int)nq, np + 1, (int64_t)1, s);     // 197 keys: one workgroup each
  if (rc) return rc;
  VA_LAUNCH(vit_fill_rows_kernel, dim3((unsigned)((nq * (D / 4) + 255) / 256)), dim3(256), 0, s, r0, v->agg_cls, nq, (int64_t)1, D);
  rc = va_launch_linear_pair(a0, (const uint16_t*)v->agg_out_w, v->agg_out_b, r0, x0, nullptr, 2, 1, (int)nq, (int)nq, 0, D, D, s);
  if (rc) return rc;
  rc = ln(v, x0, v->agg_ln2_w, v->agg_ln2_b, nullptr, a0, nq, 0, s);
  if (rc) return rc;
  rc = va_launch_linear_pair(a0, (const uint16_t*)v->agg_l1_w, v->agg_l1_b, nullptr, nullptr, h0, 1, 1, (int)nq, (int)nq, 0, D, v->hidden, s);
  if (rc) return rc;
  return va_launch_linear_pair(h0, (const uint16_t*)v->agg_l2_w, v->agg_l2_b, x0, feats, nullptr, 2, 1, (int)nq, (int)nq, 0, v->hidden, D, s);
}

}  // extern "C"
