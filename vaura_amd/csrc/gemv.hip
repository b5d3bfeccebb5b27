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

#include "gemv_kernel.h"

// --------------------------------------------------------------------------------- dispatch
template <bool BF16, int G, int NW, int T, int EPI, bool NORM>
static int launch_one(const GemvArgs& a, int64_t n_tiles, hipStream_t s) {
  if (n_tiles % T) return VAURA_ERR_SHAPE;
  VA_LAUNCH((gemv_kernel<BF16, G, NW, T, EPI, NORM>), dim3((unsigned)(n_tiles / T)), dim3(NW * 64), 0, s, a);
  return 0;
}

template <bool BF16>
static int dispatch(const GemvArgs& a, int64_t Nrows_w, int64_t K, int epi, bool norm, hipStream_t s) {
  const int64_t tiles = Nrows_w / 16;
  if (K == 1536) {
    if (epi == EPI_STORE && norm) return launch_one<BF16, 6, 8, 1, EPI_STORE, true>(a, tiles, s);
    if (epi == EPI_STORE && !norm) return launch_one<BF16, 6, 8, 1, EPI_STORE, false>(a, tiles, s);
    if (epi == EPI_RESID && !norm) return launch_one<BF16, 6, 8, 1, EPI_RESID, false>(a, tiles, s);
    if (epi == EPI_SWIGLU && norm) return launch_one<BF16, 6, 8, 2, EPI_SWIGLU, true>(a, tiles, s);
    if (epi == EPI_LOGITS && norm) return launch_one<BF16, 6, 8, 2, EPI_LOGITS, true>(a, tiles, s);
    if (epi == EPI_LOGITS && !norm) return launch_one<BF16, 6, 8, 2, EPI_LOGITS, false>(a, tiles, s);
  } else if (K == 4096) {
    if (epi == EPI_RESID && !norm) return launch_one<BF16, 16, 8, 1, EPI_RESID, false>(a, tiles, s);
    if (epi == EPI_STORE && !norm) return launch_one<BF16, 16, 8, 1, EPI_STORE, false>(a, tiles, s);
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
  if (va_is_fp8(wdtype)) return (size_t)N * (size_t)K + (size_t)N * sizeof(float);   // tile pairs + row scales
  if (wdtype == VAURA_W_H1 || wdtype == VAURA_W_H2) return (size_t)N * (size_t)K * (wdtype == VAURA_W_H1 ? 2 : 4) + (size_t)N * sizeof(float);
  return (size_t)N * (size_t)K * (wdtype == VAURA_W_BF16 ? 2 : 4);
}

int vaura_pack_weight(const float* src, void* dst, int64_t N, int64_t K, int wdtype, vaura_stream_t s) {
  if (!src || !dst || N <= 0 || K <= 0 || (N % 16) || (K % 32)) return VAURA_ERR_ARG;
  if (va_is_fp8(wdtype)) return va_pack_weight_fp8(src, dst, N, K, as_stream(s));
  if (wdtype == VAURA_W_H1 || wdtype == VAURA_W_H2) return va_pack_weight_h(src, dst, N, K, wdtype == VAURA_W_H1 ? 1 : 2, as_stream(s));
  const int64_t total = (N / 16) * (K / 32) * 64;
  const unsigned blocks = (unsigned)((total + 255) / 256);
  if (wdtype == VAURA_W_BF16) {
    VA_LAUNCH(pack_weight_kernel<true>, dim3(blocks), dim3(256), 0, as_stream(s), src, dst, N, K);
  } else if (wdtype == VAURA_W_F32) {
    VA_LAUNCH(pack_weight_kernel<false>, dim3(blocks), dim3(256), 0, as_stream(s), src, dst, N, K);
  } else {
    return VAURA_ERR_DTYPE;
  }
  return 0;
}

static int pack_rows_impl(const float* src, float* dst, int64_t rows, int64_t C, int unpack, vaura_stream_t s) {
  if (!src || !dst || rows <= 0 || C <= 0 || (C % 4)) return VAURA_ERR_ARG;
  const int64_t total = ((rows + 15) / 16 * 16) * (C / 4);
  VA_LAUNCH(pack_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(s), src, dst, rows, C, unpack);
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
