// EXPERIMENT (measured negative, round 4; DESIGN_HISTORY.md): attention + wo as one launch.  Compiled only into experiment builds
// (-DVAURA_EXPERIMENT_ENGINES), included from attention.hip behind its kernels (it reuses attention256_body); reachable there through
// debug flag bit 12.  Bit-identical to the product's two launches, slower.
#pragma once

// ---------------------------------------------------------------------------------------------------
// Attention + wo as ONE launch (the decode step's eligible shape: 16 heads x 16 row slots = 256 workgroups, cache <= 256 positions,
// fp16-plane weights; va_mlp_engine_eligible): phase 1 = attention256_body of this (head, row), unchanged; its output planes are
// handed over inside the launch (write-through stores, drained, one epoch-valued sc1 flag per workgroup; wave 0 of a consumer polls
// the 256 flags, a raw barrier releases the others; every load of the planes sc1 — csrc/mlp_engine.h has the protocol); phase 2 =
// wo + residual + ffn_norm gain / partial sums of squares / planes (gemv3h_kernel<3, 8, E3_RESID>: same products, same order) on
// workgroups 0..191 = (tile, row half).  wo's weights depend on nothing: every wave requests its slice right behind its K/V rows,
// so that stream runs under the attention (which leaves the HBM pipe almost idle) — one launch and one kernel boundary less per layer.
struct AttnWoArgs {
  Gemv3Args wo;            // W, XP (= the attention's planes), res / out (h), outp (h planes), gain_out (ffn_norm), ss_out, wscale
  uint32_t* flags;         // [256]
  const int32_t* state;
  int32_t* state_rw;
  int layer, abl;
  int rows;                // live decoder rows (the grid always has 16 row slots: phase 2 needs its 192 workgroups)
};

template <int WT>
__global__ __launch_bounds__(ATT1_THREADS) void attn_wo_kernel(
    const int32_t* __restrict__ pos_dev, float* __restrict__ kcache, float* __restrict__ vcache, const float* __restrict__ qkv,
    const float* __restrict__ qkv2, const float* __restrict__ rope, int n_head, int max_len, float* __restrict__ out,
    uint16_t* __restrict__ outp, AttnWoArgs e) {
  constexpr int HD = 96, QUADS = HD / 4, NW = ATT1_THREADS / 64, NACC = 2;
  constexpr bool F32 = WT == 2;
  constexpr int WH = F32 ? 2 : 1, BS = 1024 * WH, G2 = 3, K = 64 * G2 * NW, KG = K / 32;
  __shared__ f32x4 sqkv[3 * QUADS + 64];
  __shared__ f32x4 wacc[NW][QUADS];
  __shared__ float wm[NW], wl[NW];
  __shared__ f32x4 red[NW * 2 * 64];
  const int pos = pos_dev[0];
  float* kc = kcache + ((size_t)blockIdx.y * n_head + blockIdx.x) * (size_t)max_len * HD;
  float* vc = vcache + ((size_t)blockIdx.y * n_head + blockIdx.x) * (size_t)max_len * HD;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bid = blockIdx.y * gridDim.x + blockIdx.x;
  const uint32_t epoch = va_handoff_epoch(e.state, e.layer);
  const bool narrow = bid < 192;
  const int h = (bid >> 3) & 1, tile = (bid & 7) + 8 * (bid >> 4);
  const int la = lane & 7, sb = (lane >> 3) & 1, q = lane >> 4, m = lane & 15;
  const int w = (wid + tile) % NW;
  Gemv3Args a = e.wo;
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.W), 0, -16, 0x00020000);
  u32x4 wo_w[G2][2][WH];
  EpiPre pre;
  pre.have = false;
  f32x4 ws = f32x4{1.f, 1.f, 1.f, 1.f};
  auto hook = [&]() {
    if (!narrow) return;
    const int voffw0 = (la + 16 * q) * 16 + sb * BS;
#pragma unroll
    for (int g = 0; g < G2; ++g) {
      const int soff = (tile * KG + 2 * (w * G2 + g)) * BS;
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int hh = 0; hh < WH; ++hh)
          wo_w[g][nh][hh] = __builtin_amdgcn_raw_buffer_load_b128(wrs, voffw0 + nh * 128, soff + hh * 1024, 2 /* nt */);
    }
    if (wid == 0) {       // the epilogue's residual rows, gain and row scales (earlier KERNELS wrote them)
      if (((lane >> 3) & 1) == h) pre = gemv3_epilogue_prefetch<E3_RESID>(a, 0, tile, lane);
      ws = *reinterpret_cast<const f32x4*>(a.wscale + (size_t)tile * 16 + 4 * q);
    }
  };
  if ((int)blockIdx.y >= e.rows) {
    hook();                // a row slot without a sequence: no attention, but this workgroup still owns a wo tile and a flag
  } else {
    switch ((pos + 63) >> 6) {
      case 0: attention256_body<HD, 0>(qkv, qkv2, rope, kc, vc, out, outp, n_head, pos, sqkv, wacc, wm, wl, 1.f, hook); break;
      case 1: attention256_body<HD, 1>(qkv, qkv2, rope, kc, vc, out, outp, n_head, pos, sqkv, wacc, wm, wl, 1.f, hook); break;
      case 2: attention256_body<HD, 2>(qkv, qkv2, rope, kc, vc, out, outp, n_head, pos, sqkv, wacc, wm, wl, 1.f, hook); break;
      case 3: attention256_body<HD, 3>(qkv, qkv2, rope, kc, vc, out, outp, n_head, pos, sqkv, wacc, wm, wl, 1.f, hook); break;
      default: attention256_body<HD, 4>(qkv, qkv2, rope, kc, vc, out, outp, n_head, pos, sqkv, wacc, wm, wl, 1.f, hook); break;
    }
  }
  // publish: threads 0..23 (wave 0) stored this (row, head)'s output and planes (write-through); drained, then the flag
  if (wid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(e.flags + bid), "v"(epoch) : "memory");
  }
  if (!narrow) return;
  (void)mlpe_poll_flags(e.flags, 64, epoch, e, wid, lane);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // ---- phase 2: wo + residual (gemv3h_kernel<3, 8, E3_RESID>): the weights are in registers since the attention's start
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.XP), 0, a.R * VA_NPL * (K / 8) * 256, 0x00020000);
  const int voffx = (sb * 64 + q * 16 + la + 8 * h) * 16;
  u32x4 wo_x[G2][VA_NPL];
#pragma unroll
  for (int g = 0; g < G2; ++g)
#pragma unroll
    for (int p = 0; p < VA_NPL; ++p)
      wo_x[g][p] = __builtin_amdgcn_raw_buffer_load_b128(xrs, la + 8 * h < a.rows ? voffx : 0x7ffffff0,
                                                         (p * (K / 8) * 16 + (w * G2 + g) * 128) * 16, 16 /* sc1 */);
  __builtin_amdgcn_sched_barrier(0);
  f32x4 acc[2][NACC];
#pragma unroll
  for (int nh = 0; nh < 2; ++nh)
#pragma unroll
    for (int p = 0; p < NACC; ++p) acc[nh][p] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < G2; ++g) {
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
      f16x8 wf[F32 ? 2 : 1];
      wf[0] = __builtin_bit_cast(f16x8, wo_w[g][nh][0]);
      if constexpr (F32) wf[1] = __builtin_bit_cast(f16x8, wo_w[g][nh][WH - 1]);
      mfma_group<WT>(wf, wo_x[g], acc[nh]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int nh = 0; nh < 2; ++nh) {
    const f32x4 v = acc_sum<WT>(acc[nh]);
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float x = v[r];
      o[r] = x + va_dpp<VA_DPP_ROR8>(va_xor32(x));
    }
    red[(wid * 2 + nh) * 64 + lane] = o;
  }
  __syncthreads();
  if (wid == 0) {
    const bool mine = (m >> 3) == h;
    const int src = (m & 7) + 16 * (q & 1);
    f32x4 v = red[(0 * 2 + (q >> 1)) * 64 + src];
#pragma unroll
    for (int i = 1; i < NW; ++i) v += red[(i * 2 + (q >> 1)) * 64 + src];
    v *= ws;
    if (mine) gemv3_epilogue<1, E3_RESID>(a, 0, tile, lane, &v, &pre);
  }
}

// attention (single-round-trip kernel) + wo of one layer as one launch; the caller checked va_mlp_engine_eligible, rows in 1..16,
// n_head == 16, max_len <= 256.  flags: 256 words.  awo.wscale is filled in here (the scales follow the packed tiles).
int va_launch_attn_wo(const float* qkv, const float* qkv2, const float* rope, float* kc, float* vc, float* out, uint16_t* outp,
                      int rows, int n_head, int max_len, const int32_t* state, const Gemv3Args& awo, uint32_t* flags, int layer, hipStream_t s) {
  if (!qkv || !rope || !kc || !vc || !out || !outp || !state || !flags || !awo.W || awo.XP != outp || !awo.res || !awo.out || !awo.outp ||
      !awo.gain_out || !awo.ss_out || awo.R != 1) return VAURA_ERR_ARG;
  if (n_head != 16 || max_len > 256 || rows < 1 || rows > 16 || awo.N != 1536 || (awo.wq != 0 && awo.wq != 2)) return VAURA_ERR_SHAPE;
  AttnWoArgs e;
  e.rows = rows;
  e.wo = awo;
  e.wo.wscale = reinterpret_cast<const float*>(static_cast<const char*>(awo.W) + (size_t)1536 * 1536 * (awo.wq == 2 ? 4 : 2));
  e.flags = flags; e.state = state; e.state_rw = const_cast<int32_t*>(state); e.layer = layer;
  e.abl = (int)((va_debug_flags_get() >> 28) & 15u);
  if (awo.wq == 2)
    VA_LAUNCH(attn_wo_kernel<2>, dim3(n_head, 16), dim3(ATT1_THREADS), 0, s, state, kc, vc, qkv, qkv2, rope, n_head, max_len, out, outp, e);
  else
    VA_LAUNCH(attn_wo_kernel<0>, dim3(n_head, 16), dim3(ATT1_THREADS), 0, s, state, kc, vc, qkv, qkv2, rope, n_head, max_len, out, outp, e);
  return 0;
}

