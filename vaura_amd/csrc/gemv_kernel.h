// GEMV kernel template (see gemv.hip for the design notes); shared with tools/microbench.
#pragma once
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

// ABL: ablation bits for tools/microbench only (0 in the product): 1 = no MFMA, 2 = no x loads, 4 = no weight loads
template <bool BF16, int G, int NW, int T, int EPI, bool NORM, int ABL = 0>
__global__ __launch_bounds__(NW * 64) void gemv_kernel(GemvArgs a) {
  constexpr int K = 32 * G * NW;
  constexpr int KG = K / 32;
  __shared__ f32x4 red[NW][T][64];
  __shared__ float ssr[NW][16];

  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const int m = lane & 15;   // activation row inside the block / weight row inside the tile
  const int q = lane >> 4;   // which 8-wide k sub-group of the 32-wide k-group
  const int tile0 = blockIdx.x * T;

  // ---- 1. small, cache-resident operands first (vmcnt retires in order: anything issued behind the
  //         weight stream would wait for all of it).  The norm gain goes through LDS: every lane of a
  //         16-row group needs the same 8 gains, and per-lane global loads of them cost the CU's
  //         vector-memory path (64 B/clk) 4x the bytes of the weights themselves.
  __shared__ float gs[NORM ? K : 4];
  f32x4 gstage[NORM ? (K / 4 + NW * 64 - 1) / (NW * 64) : 1];
  if constexpr (NORM) {
#pragma unroll
    for (int i = 0; i < (K / 4 + NW * 64 - 1) / (NW * 64); ++i) {
      const int idx = threadIdx.x + i * NW * 64;
      if (idx < K / 4) gstage[i] = reinterpret_cast<const f32x4*>(a.gain)[idx];
    }
  }
  f32x4 xv[G][2];
  auto load_x = [&](int rb) {
    const f32x4* Xp = reinterpret_cast<const f32x4*>(a.X) + (size_t)rb * (K / 4) * 16;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int kq = (w * G + g) * 8 + 2 * q;
      if constexpr (ABL & 2) {
        xv[g][0] = xv[g][1] = f32x4{(float)lane, 1.f, 2.f, (float)g};
      } else {
        xv[g][0] = Xp[(size_t)kq * 16 + m];
        xv[g][1] = Xp[(size_t)(kq + 1) * 16 + m];
      }
    }
  };
  load_x(0);

  // ---- 2. the whole weight slice of this wave, streamed once from HBM (non-temporal)
  u32x4 wb[T][G][BF16 ? 1 : 2];
#pragma unroll
  for (int g = 0; g < G; ++g) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const size_t kg = (size_t)(tile0 + t) * KG + (size_t)(w * G + g);
      if constexpr (ABL & 4) {
        wb[t][g][0] = u32x4{(uint32_t)lane, 1u, (uint32_t)g, 3u};
        if constexpr (!BF16) wb[t][g][1] = wb[t][g][0];
      } else if constexpr (BF16) {
        wb[t][g][0] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a.W) + kg * 64 + lane);
      } else {
        wb[t][g][0] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a.W) + (kg * 2 + 0) * 64 + lane);
        wb[t][g][1] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a.W) + (kg * 2 + 1) * 64 + lane);
      }
    }
  }

  if constexpr (NORM) {
#pragma unroll
    for (int i = 0; i < (K / 4 + NW * 64 - 1) / (NW * 64); ++i) {
      const int idx = threadIdx.x + i * NW * 64;
      if (idx < K / 4) reinterpret_cast<f32x4*>(gs)[idx] = gstage[i];
    }
    __syncthreads();
  }

  auto row_block = [&](const int rb) {
    // two accumulators per tile: v_mfma_f32_16x16x4_f32 has a 40-cycle dependent latency at a 32-cycle issue
    f32x4 acc[T][2];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t][0] = acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    float ss = 0.f;

    // ---- 3. 8 MFMA k-steps per 32-wide k-group, consumed in arrival order of the weight stream
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float xs[8] = {xv[g][0][0], xv[g][0][1], xv[g][0][2], xv[g][0][3],
                     xv[g][1][0], xv[g][1][1], xv[g][1][2], xv[g][1][3]};
      if constexpr (NORM) {
#pragma unroll
        for (int j = 0; j < 8; ++j) ss = fmaf(xs[j], xs[j], ss);
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(&gs[(w * G + g) * 32 + 8 * q]);
        const f32x4 g1 = *reinterpret_cast<const f32x4*>(&gs[(w * G + g) * 32 + 8 * q + 4]);
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
        for (int j = 0; j < 8; ++j) {
          if constexpr (ABL & 1) {
            asm volatile("" ::"v"(wv[j]), "v"(xs[j]));
          } else {
            acc[t][j & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j], xs[j], acc[t][j & 1], 0, 0, 0);
          }
        }
      }
      // keep this group's MFMAs here: without the fence hipcc hoists every load-dependent VALU op above
      // the first MFMA and waits vmcnt(0) for the whole weight slice before any matrix work starts
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- 4. cross-wave reduction (fixed order) + epilogue by wave 0
#pragma unroll
    for (int t = 0; t < T; ++t) red[w][t][lane] = acc[t][0] + acc[t][1];
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
  };
  // The decode step has ONE row block: keep it straight-line (inside a loop hipcc hoists the weight
  // widening out of the loop and parks a vmcnt(0) in the loop header, serialising stream and MFMA).
  row_block(0);
  for (int rb = 1; rb < a.R; ++rb) {
    __syncthreads();
    load_x(rb);
    row_block(rb);
  }
}

