// Weight-streaming skinny GEMM for the decode step:  out(rows x N) = epilogue( X(rows x K) . W(N x K)^T )
//
// Replaces, for ONE position per sequence, the nn.Linear calls of the reference decoder
//   wqkv / wo            models/modules/sampler/llama.py:228, 259
//   w1, w3, w2 (SwiGLU)  llama.py:176-177
//   lm_heads             llama.py:503-504
//   cls_embeddings MLP   llama.py:88-92
// with RMSNorm (llama.py:153-158) fused in: the gain multiplies x on load, and because
// W.(g*x*rinv) == rinv * W.(g*x) the per-row rsqrt(mean(x^2)+eps) is applied in the epilogue from
// a sum of squares the kernel accumulates over the x it streams anyway.
//
// Bound: HBM (each weight byte is read once per step and used for <=16*R rows).
// Design (MI355X):
//   * weights are pre-packed so that one wavefront load instruction is a contiguous 1 KiB that IS
//     the A operand of v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain) for 8 consecutive k-steps;
//     bf16 storage is widened to fp32 in-register (exact), so both storage types compute the same
//     real numbers when the weights are bf16-representable.
//   * a workgroup owns T 16-row output tiles over the whole K; its NW waves split K, issue ALL of
//     their weight loads up front (register-resident stream, no LDS round trip), then the x loads;
//     partial 16x16 tiles are summed across waves through LDS in a fixed order (deterministic).
//   * activations use the packed-rows layout so x fragments and output tiles are contiguous 1 KiB.
#include "common.h"

enum { EPI_STORE = 0, EPI_RESID = 1, EPI_SWIGLU = 2, EPI_GELU = 3, EPI_LOGITS = 4 };

struct GemvArgs {
  const void* W;
  const float* X;
  const float* gain;
  const float* res;
  float* out;
  int rows;      // live rows
  int R;         // row blocks = ceil(rows/16)
  int N;         // output width seen by the epilogue's consumer (SWIGLU: N/2)
  float eps;
};

__device__ __forceinline__ float silu_f(float a) { return a / (1.0f + expf(-a)); }
__device__ __forceinline__ float gelu_tanh_f(float x) {
  const float kBeta = 0.7978845608028654f;  // sqrt(2/pi)
  const float kKappa = 0.044715f;
  float inner = kBeta * (x + kKappa * (x * x * x));
  return 0.5f * x * (1.0f + tanhf(inner));
}

template <bool BF16, int G, int NW, int T, int EPI, bool NORM>
__global__ __launch_bounds__(NW * 64) void gemv_kernel(GemvArgs a) {
  constexpr int K = 32 * G * NW;
  constexpr int KG = K / 32;
  __shared__ f32x4 red[NW][T][64];
  __shared__ float ssr[NW][16];
  __shared__ float gs[NORM ? K : 4];

  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const int m = lane & 15;   // activation row inside the block / weight row inside the tile
  const int q = lane >> 4;   // which 8-wide k sub-group of the 32-wide k-group
  const int tile0 = blockIdx.x * T;

  // ---- 1. issue every weight load of this wave (HBM stream)
  u32x4 wb[T][G][BF16 ? 1 : 2];
#pragma unroll
  for (int t = 0; t < T; ++t) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const size_t kg = (size_t)(tile0 + t) * KG + (size_t)(w * G + g);
      if constexpr (BF16) {
        wb[t][g][0] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a.W) + kg * 64 + lane);
      } else {
        wb[t][g][0] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a.W) + (kg * 2 + 0) * 64 + lane);
        wb[t][g][1] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a.W) + (kg * 2 + 1) * 64 + lane);
      }
    }
  }
  if constexpr (NORM) {
    for (int i = threadIdx.x; i < K / 4; i += NW * 64)
      reinterpret_cast<f32x4*>(gs)[i] = reinterpret_cast<const f32x4*>(a.gain)[i];
    __syncthreads();
  }

  for (int rb = 0; rb < a.R; ++rb) {
    // ---- 2. x fragments of this wave's K slice (L2-resident, packed rows)
    const f32x4* Xp = reinterpret_cast<const f32x4*>(a.X) + (size_t)rb * (K / 4) * 16;
    f32x4 xv[G][2];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int kq = (w * G + g) * 8 + 2 * q;
      xv[g][0] = Xp[(size_t)kq * 16 + m];
      xv[g][1] = Xp[(size_t)(kq + 1) * 16 + m];
    }
    f32x4 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float ss = 0.f;

    // ---- 3. 8 MFMA k-steps per 32-wide k-group
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float xs[8] = {xv[g][0][0], xv[g][0][1], xv[g][0][2], xv[g][0][3],
                     xv[g][1][0], xv[g][1][1], xv[g][1][2], xv[g][1][3]};
      if constexpr (NORM) {
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(&gs[(w * G + g) * 32 + 8 * q]);
        const f32x4 g1 = *reinterpret_cast<const f32x4*>(&gs[(w * G + g) * 32 + 8 * q + 4]);
#pragma unroll
        for (int j = 0; j < 8; ++j) ss = fmaf(xs[j], xs[j], ss);
#pragma unroll
        for (int j = 0; j < 4; ++j) { xs[j] *= g0[j]; xs[4 + j] *= g1[j]; }
      }
#pragma unroll
      for (int t = 0; t < T; ++t) {
        float wv[8];
        if constexpr (BF16) {
          const u32x4 u = wb[t][g][0];
          wv[0] = bf16_lo(u.x); wv[1] = bf16_hi(u.x); wv[2] = bf16_lo(u.y); wv[3] = bf16_hi(u.y);
          wv[4] = bf16_lo(u.z); wv[5] = bf16_hi(u.z); wv[6] = bf16_lo(u.w); wv[7] = bf16_hi(u.w);
        } else {
          // whole-vector bit_cast: element-wise __builtin_bit_cast(float, u.y) of an ext-vector member
          // was observed to read lane element 0 for every member (hipcc 7.2)
          const f32x4 f0 = __builtin_bit_cast(f32x4, wb[t][g][0]);
          const f32x4 f1 = __builtin_bit_cast(f32x4, wb[t][g][1]);
          wv[0] = f0[0]; wv[1] = f0[1]; wv[2] = f0[2]; wv[3] = f0[3];
          wv[4] = f1[0]; wv[5] = f1[1]; wv[6] = f1[2]; wv[7] = f1[3];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j], xs[j], acc[t], 0, 0, 0);
      }
    }

    // ---- 4. cross-wave reduction (fixed order) + epilogue by wave 0
#pragma unroll
    for (int t = 0; t < T; ++t) red[w][t][lane] = acc[t];
    if constexpr (NORM) {
      ss += __shfl_xor(ss, 16, 64);
      ss += __shfl_xor(ss, 32, 64);
      if (q == 0) ssr[w][m] = ss;
    }
    __syncthreads();
    if (w == 0) {
      float rinv = 1.f;
      if constexpr (NORM) {
        float tot = 0.f;
#pragma unroll
        for (int i = 0; i < NW; ++i) tot += ssr[i][m];
        rinv = 1.0f / sqrtf(tot * (1.0f / (float)K) + a.eps);
      }
      f32x4 v[T];
#pragma unroll
      for (int t = 0; t < T; ++t) {
        f32x4 sacc = red[0][t][lane];
#pragma unroll
        for (int i = 1; i < NW; ++i) sacc += red[i][t][lane];
        v[t] = sacc * rinv;
      }
      // lane holds out[row = rb*16 + m][n = 16*tile + 4*q + r], r = 0..3
      if constexpr (EPI == EPI_SWIGLU) {
        static_assert(T == 2 || EPI != EPI_SWIGLU, "SwiGLU needs a (w1, w3) tile pair");
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = silu_f(v[0][r]) * v[T - 1][r];
        const int tile = blockIdx.x;  // tile of the ffn dimension
        reinterpret_cast<f32x4*>(a.out)[((size_t)rb * (a.N / 4) + (size_t)tile * 4) * 16 + lane] = o;
      } else {
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const int tile = tile0 + t;
          if constexpr (EPI == EPI_LOGITS) {
            const int row = rb * 16 + m;
            if (row < a.rows)
              *reinterpret_cast<f32x4*>(a.out + (size_t)row * a.N + (size_t)tile * 16 + 4 * q) = v[t];
          } else {
            const size_t idx = ((size_t)rb * (a.N / 4) + (size_t)tile * 4) * 16 + lane;
            f32x4 o = v[t];
            if constexpr (EPI == EPI_RESID) o += reinterpret_cast<const f32x4*>(a.res)[idx];
            if constexpr (EPI == EPI_GELU) {
#pragma unroll
              for (int r = 0; r < 4; ++r) o[r] = gelu_tanh_f(o[r]);
            }
            reinterpret_cast<f32x4*>(a.out)[idx] = o;
          }
        }
      }
    }
    if (rb + 1 < a.R) __syncthreads();
  }
}

// --------------------------------------------------------------------------------- dispatch
template <bool BF16, int G, int NW, int T, int EPI, bool NORM>
static int launch_one(const GemvArgs& a, int64_t n_tiles, hipStream_t s) {
  if (n_tiles % T) return VAURA_ERR_SHAPE;
  hipLaunchKernelGGL((gemv_kernel<BF16, G, NW, T, EPI, NORM>), dim3((unsigned)(n_tiles / T)), dim3(NW * 64), 0, s, a);
  VA_CHECK_LAUNCH();
  return 0;
}

template <bool BF16>
static int dispatch(const GemvArgs& a, int64_t Nrows_w, int64_t K, int epi, bool norm, hipStream_t s) {
  const int64_t tiles = Nrows_w / 16;
  if (K == 1536) {
    if (epi == EPI_STORE && norm) return launch_one<BF16, 12, 4, 1, EPI_STORE, true>(a, tiles, s);
    if (epi == EPI_STORE && !norm) return launch_one<BF16, 12, 4, 1, EPI_STORE, false>(a, tiles, s);
    if (epi == EPI_RESID && !norm) return launch_one<BF16, 6, 8, 1, EPI_RESID, false>(a, tiles, s);
    if (epi == EPI_SWIGLU && norm) return launch_one<BF16, 12, 4, 2, EPI_SWIGLU, true>(a, tiles, s);
    if (epi == EPI_LOGITS && norm) return launch_one<BF16, 12, 4, 2, EPI_LOGITS, true>(a, tiles, s);
    if (epi == EPI_LOGITS && !norm) return launch_one<BF16, 12, 4, 2, EPI_LOGITS, false>(a, tiles, s);
  } else if (K == 4096) {
    if (epi == EPI_RESID && !norm) return launch_one<BF16, 8, 16, 1, EPI_RESID, false>(a, tiles, s);
    if (epi == EPI_STORE && !norm) return launch_one<BF16, 8, 16, 1, EPI_STORE, false>(a, tiles, s);
  } else if (K == 768) {
    if (epi == EPI_GELU && !norm) return launch_one<BF16, 6, 4, 1, EPI_GELU, false>(a, tiles, s);
    if (epi == EPI_STORE && !norm) return launch_one<BF16, 6, 4, 1, EPI_STORE, false>(a, tiles, s);
  } else if (K == 512) {
    if (epi == EPI_STORE && !norm) return launch_one<BF16, 4, 4, 1, EPI_STORE, false>(a, tiles, s);
    if (epi == EPI_GELU && !norm) return launch_one<BF16, 4, 4, 1, EPI_GELU, false>(a, tiles, s);
  }
  return VAURA_ERR_SHAPE;
}

int va_launch_gemv(const void* w, int wdtype, const float* x, const float* gain, const float* residual, float* out,
                   int64_t rows, int64_t N, int64_t K, int epilogue, float eps, hipStream_t s) {
  if (!w || !x || !out || rows <= 0 || N <= 0 || (N % 16) || (K % 32)) return VAURA_ERR_ARG;
  if (epilogue == EPI_RESID && !residual) return VAURA_ERR_ARG;
  if (epilogue == EPI_SWIGLU && (N % 32)) return VAURA_ERR_SHAPE;
  GemvArgs a;
  a.W = w; a.X = x; a.gain = gain; a.res = residual; a.out = out;
  a.rows = (int)rows; a.R = (int)((rows + 15) / 16);
  a.N = (int)(epilogue == EPI_SWIGLU ? N / 2 : N);
  a.eps = eps;
  if (wdtype == VAURA_W_BF16) return dispatch<true>(a, N, K, epilogue, gain != nullptr, s);
  if (wdtype == VAURA_W_F32) return dispatch<false>(a, N, K, epilogue, gain != nullptr, s);
  return VAURA_ERR_DTYPE;
}

// --------------------------------------------------------------------------------- packing
// src row-major (N x K) fp32 -> MFMA-tile order.  One thread per (tile, kgroup, lane): 8 elements.
template <bool BF16>
__global__ void pack_weight_kernel(const float* __restrict__ src, void* __restrict__ dst, int64_t N, int64_t K) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t KG = K / 32;
  const int64_t total = (N / 16) * KG * 64;
  if (gid >= total) return;
  const int lane = (int)(gid & 63);
  const int64_t kg = (gid >> 6) % KG;
  const int64_t tile = (gid >> 6) / KG;
  const int64_t n = tile * 16 + (lane & 15);
  const int64_t k = kg * 32 + 8 * (lane >> 4);
  const float* p = src + n * K + k;
  if constexpr (BF16) {
    uint4 u;
    u.x = (uint32_t)f32_to_bf16_rne(p[0]) | ((uint32_t)f32_to_bf16_rne(p[1]) << 16);
    u.y = (uint32_t)f32_to_bf16_rne(p[2]) | ((uint32_t)f32_to_bf16_rne(p[3]) << 16);
    u.z = (uint32_t)f32_to_bf16_rne(p[4]) | ((uint32_t)f32_to_bf16_rne(p[5]) << 16);
    u.w = (uint32_t)f32_to_bf16_rne(p[6]) | ((uint32_t)f32_to_bf16_rne(p[7]) << 16);
    reinterpret_cast<uint4*>(dst)[(tile * KG + kg) * 64 + lane] = u;
  } else {
    f32x4* d = reinterpret_cast<f32x4*>(dst);
    d[((tile * KG + kg) * 2 + 0) * 64 + lane] = f32x4{p[0], p[1], p[2], p[3]};
    d[((tile * KG + kg) * 2 + 1) * 64 + lane] = f32x4{p[4], p[5], p[6], p[7]};
  }
}

__global__ void pack_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t rows, int64_t C, int unpack) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one float4 quad of the padded matrix
  const int64_t rows_p = (rows + 15) / 16 * 16;
  const int64_t CQ = C / 4;
  if (gid >= rows_p * CQ) return;
  const int row = (int)(gid / CQ);
  const int cq = (int)(gid % CQ);
  const size_t pq = packed_quad(row, cq, (int)C);
  if (unpack) {
    if (row < rows) reinterpret_cast<f32x4*>(dst)[(size_t)row * CQ + cq] = reinterpret_cast<const f32x4*>(src)[pq];
  } else {
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (row < rows) v = reinterpret_cast<const f32x4*>(src)[(size_t)row * CQ + cq];
    reinterpret_cast<f32x4*>(dst)[pq] = v;
  }
}

extern "C" {

size_t vaura_packed_weight_bytes(int64_t N, int64_t K, int wdtype) {
  return (size_t)N * (size_t)K * (wdtype == VAURA_W_BF16 ? 2 : 4);
}

int vaura_pack_weight(const float* src, void* dst, int64_t N, int64_t K, int wdtype, vaura_stream_t s) {
  if (!src || !dst || N <= 0 || K <= 0 || (N % 16) || (K % 32)) return VAURA_ERR_ARG;
  const int64_t total = (N / 16) * (K / 32) * 64;
  const unsigned blocks = (unsigned)((total + 255) / 256);
  if (wdtype == VAURA_W_BF16)
    hipLaunchKernelGGL(pack_weight_kernel<true>, dim3(blocks), dim3(256), 0, as_stream(s), src, dst, N, K);
  else if (wdtype == VAURA_W_F32)
    hipLaunchKernelGGL(pack_weight_kernel<false>, dim3(blocks), dim3(256), 0, as_stream(s), src, dst, N, K);
  else
    return VAURA_ERR_DTYPE;
  VA_CHECK_LAUNCH();
  return 0;
}

static int pack_rows_impl(const float* src, float* dst, int64_t rows, int64_t C, int unpack, vaura_stream_t s) {
  if (!src || !dst || rows <= 0 || C <= 0 || (C % 4)) return VAURA_ERR_ARG;
  const int64_t total = ((rows + 15) / 16 * 16) * (C / 4);
  hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(s), src, dst, rows, C, unpack);
  VA_CHECK_LAUNCH();
  return 0;
}
int vaura_pack_rows(const float* src, float* dst, int64_t rows, int64_t C, vaura_stream_t s) {
  return pack_rows_impl(src, dst, rows, C, 0, s);
}
int vaura_unpack_rows(const float* src, float* dst, int64_t rows, int64_t C, vaura_stream_t s) {
  return pack_rows_impl(src, dst, rows, C, 1, s);
}

int vaura_gemv(const void* w, int wdtype, const float* x, const float* gain, const float* residual, float* out,
               int64_t rows, int64_t N, int64_t K, int epilogue, float eps, vaura_stream_t s) {
  return va_launch_gemv(w, wdtype, x, gain, residual, out, rows, N, K, epilogue, eps, as_stream(s));
}

int vaura_prefill_cond(const vaura_dims* d, const float* feats, const void* fc1, const void* fc2, int wdtype,
                       float* tmp, float* out, int64_t n_rows, vaura_stream_t s) {
  if (!d || !feats || !fc1 || !fc2 || !tmp || !out || n_rows <= 0) return VAURA_ERR_ARG;
  int rc = va_launch_gemv(fc1, wdtype, feats, nullptr, nullptr, tmp, n_rows, d->cond_dim, d->cond_in, EPI_GELU, 0.f, as_stream(s));
  if (rc) return rc;
  return va_launch_gemv(fc2, wdtype, tmp, nullptr, nullptr, out, n_rows, d->cond_dim, d->cond_dim, EPI_STORE, 0.f, as_stream(s));
}

}  // extern "C"
