// EXPERIMENT (measured negative, round 4; DESIGN_HISTORY.md): the whole layer tail as one launch.  Compiled only into experiment
// builds (-DVAURA_EXPERIMENT_ENGINES: `python -m vaura_amd.csrc.build --tag engines -DVAURA_EXPERIMENT_ENGINES=1`), included from
// mlp_engine.h; reachable there through debug flag bit 3.  Bit-identical to the product's launches, no faster.
#pragma once

struct TailEngineArgs {
  Gemv3Args p0;            // wo: W, XP (attention planes), res / out (h), outp (h planes), gain_out (ffn_norm), ss_out, N = d_model, wscale
  Gemv3Args p1;            // w1||w3: XP (h planes), ss_in, outp (ffn planes), N = ffn_dim, eps, k_total, wscale
  Gemv3Args p2;            // w2: XP (ffn planes), out (h), outp (h planes), gain_out (next attention_norm), ss_out, N = d_model, wscale
  uint32_t* flags;         // [512]: [0, 192) phase-0 producers, [256, 512) phase-1 producers
  const int32_t* state;
  int32_t* state_rw;
  int layer;
  int abl;                 // timing ablations (tools only): 1 = no flag waits (wrong results), 2 = phase 1's weights requested at kernel start
};

// =====================================================================================================================================
// The TAIL of a decoder layer — h += wo.attn ; h += w2( silu(w1 x) * (w3 x) ), x = rmsnorm(h) (llama.py:259, 279, 282) — as ONE launch:
//   phase 0   wo GEMV + residual + ffn_norm gain / partial sums of squares / planes   (gemv3h_kernel<3, 8, E3_RESID>, workgroups 0..191)
//   hand-off  h planes (16 rows x 1536 x (hi, lo) = 98 KB) + the partial sums: 192 producers -> all 256 workgroups
//   phase 1   w1||w3 + SwiGLU           (gemv3_kernel<6, 8, 2, E3_SWIGLU, true>, all 256 workgroups)
//   hand-off  ffn planes: 256 producers -> workgroups 0..191
//   phase 2   w2 + residual             (gemv3h_kernel<8, 8, E3_RESID>)
// What it buys over mlp_engine_kernel: w1||w3's weights — 2/5 of a layer's bytes, 7.8 us of HBM stream — depend on nothing, so every
// wave requests its whole slice (96 registers) at kernel start and the stream runs under phase 0 and the first hand-off; behind that
// hand-off phase 1 is one round trip for the planes plus its products.  The residual rows phase 2 adds are the ones the SAME wave
// produced in phase 0 ((tile, row half) is the workgroup's in both): they stay in registers.  Same products, same order, same
// epilogue arithmetic as the separate launches: bit-identical.
// In-order vector-memory counters shape the roles (MI355X_MICROARCH.md): wave 0 — the wave that publishes and polls — requests its
// own slice of the next phase's weights only AFTER it has published (a drain in front of a flag would wait for them), the other seven
// waves as early as they can.
template <int WT>
__global__ __launch_bounds__(MLPE_NW * 64) void tail_engine_kernel(const void* __restrict__ WOq, const uint16_t* __restrict__ XAq,
                                                                   const void* __restrict__ W13q, const void* __restrict__ W2q,
                                                                   TailEngineArgs e) {
  using SH = MlpEngineShape<WT>;
  constexpr bool F32 = WT == 2;
  constexpr int WH = SH::WH, NW = MLPE_NW, NACC = 2, BS = 1024 * WH;
  extern __shared__ __attribute__((aligned(16))) unsigned char mlpe_lds[];
  unsigned char* ring = mlpe_lds;
  f32x4* red = reinterpret_cast<f32x4*>(mlpe_lds + NW * SH::WAVE_RING);            // [NW][2][64], all three phases
  unsigned* arrive0 = reinterpret_cast<unsigned*>(mlpe_lds + NW * SH::WAVE_RING + SH::RED);
  unsigned* arrive1 = arrive0 + NW;

  VA_STAMP_DECL(stamps);
  VA_STAMP(stamps, 0);
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bid = blockIdx.x;
  const uint32_t epoch = va_handoff_epoch(e.state, e.layer);
  // LDS arrival words carry the epoch mixed with THIS workgroup's id: on a busy chip the 256 workgroups of a launch do not all start at
  // once, and a late one can land on a CU another workgroup of the SAME launch has just left — whose arrival words hold this very
  // epoch (round 4: found by the concurrent-load test; on an idle chip every workgroup has a CU of its own and it never shows)
  const uint32_t ltag = epoch ^ ((uint32_t)(blockIdx.x + 1) * 0x9E3779B1u);
  uint32_t* flags0 = e.flags;            // [192] phase-0 producers
  uint32_t* flags1 = e.flags + 256;      // [256] phase-1 producers
  const bool narrow = bid < 192;         // this workgroup owns a (tile, row half) of the 1536-wide outputs (phases 0 and 2)
  const int h = (bid >> 3) & 1;
  const int tile = (bid & 7) + 8 * (bid >> 4);
  const int la = lane & 7, sb = (lane >> 3) & 1, q = lane >> 4, m = lane & 15;
  const int lane16 = lane * 16;
  const int voffw0 = (la + 16 * q) * 16 + sb * BS;
  const int wn = (wid + tile) % NW;                        // K slice of this wave in the narrow phases (de-phased per tile)
  const int w1 = (wid + bid) % NW;                         // ... and in phase 1
  constexpr int T = 2, G = 6, K1 = 1536, KG1 = K1 / 32;
  const int tile0 = bid * T;

  // ---- wave 0: what the three epilogues need and earlier KERNELS wrote (residual rows, gains, row scales): first thing
  EpiPre pre0, pre2;
  pre0.have = pre2.have = false;
  f32x4 ws0 = f32x4{1.f, 1.f, 1.f, 1.f}, ws2 = ws0, ws1[T] = {ws0, ws0};
  if (wid == 0) {
    if (narrow && ((lane >> 3) & 1) == h) {
      pre0 = gemv3_epilogue_prefetch<E3_RESID>(e.p0, 0, tile, lane);                  // h rows + ffn_norm gain
      pre2.gain = *reinterpret_cast<const f32x4*>(e.p2.gain_out + tile * 16 + 4 * q);  // next layer's attention_norm gain (phase 2)
      pre2.have = true;
    }
    if (narrow) {
      ws0 = *reinterpret_cast<const f32x4*>(e.p0.wscale + (size_t)tile * 16 + 4 * q);
      ws2 = *reinterpret_cast<const f32x4*>(e.p2.wscale + (size_t)tile * 16 + 4 * q);
    }
#pragma unroll
    for (int t = 0; t < T; ++t) ws1[t] = *reinterpret_cast<const f32x4*>(e.p1.wscale + (size_t)(tile0 + t) * 16 + 4 * q);
  }

  // ---- phase-1 weights of this wave: [2 tiles][6 k-groups][planes] straight into registers, requested at kernel start by waves
  //      1..7 (behind their phase-0 operands), by wave 0 once it has published phase 0
  const __amdgpu_buffer_rsrc_t w13rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(W13q), 0, -16, 0x00020000);
  u32x4 wb[T][G][WH];
  auto load_w13 = [&]() {
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const size_t kg = (size_t)(tile0 + t) * KG1 + (size_t)(w1 * G + g);
#pragma unroll
        for (int hh = 0; hh < WH; ++hh)
          wb[t][g][hh] = __builtin_amdgcn_raw_buffer_load_b128(w13rs, lane16, (int)((kg * WH + hh) * 1024), 2 /* nt */);
      }
  };

  // ================================================================ phase 0: wo + residual (gemv3h_kernel<3, 8, E3_RESID>)
  f32x4 hkeep = f32x4{0.f, 0.f, 0.f, 0.f};                 // wave 0: the new residual rows of this (tile, row half)
  if (narrow) {
    Gemv3Args a = e.p0;
    a.W = WOq;
    a.XP = XAq;
    constexpr int G2 = 3, K = 64 * G2 * NW, KG = K / 32;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.W), 0, -16, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.XP), 0, a.R * VA_NPL * (K / 8) * 256, 0x00020000);
    const int voffx = (sb * 64 + q * 16 + la + 8 * h) * 16;
    u32x4 wo_w[G2][2][WH], wo_x[G2][VA_NPL];
#pragma unroll
    for (int g = 0; g < G2; ++g) {
      const int soff = (tile * KG + 2 * (wn * G2 + g)) * BS;
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int hh = 0; hh < WH; ++hh)
          wo_w[g][nh][hh] = __builtin_amdgcn_raw_buffer_load_b128(wrs, voffw0 + nh * 128, soff + hh * 1024, 2 /* nt */);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < G2; ++g)
#pragma unroll
      for (int p = 0; p < VA_NPL; ++p)
        wo_x[g][p] = __builtin_amdgcn_raw_buffer_load_b128(xrs, la + 8 * h < a.rows ? voffx : 0x7ffffff0,
                                                           (p * (K / 8) * 16 + (wn * G2 + g) * 128) * 16, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (wid != 0 && (e.abl & 2)) load_w13();               // (ablation bit 1: right behind this wave's phase-0 requests — slower: see below)
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
    if (lane == 0) __hip_atomic_store(arrive0 + wid, ltag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    VA_STAMP(stamps, 1);                                   // phase 0: products done
    // phase 1's weight stream starts once this wave's phase-0 operands have LANDED (its products are issued): requested at kernel
    // start the 50 MB compete with phase 0's 9 MB for the same HBM pipe and phase 0 — the head of the whole chain — slows down
    if (wid != 0 && !(e.abl & 2)) load_w13();
    if (wid == 0) {
      mlpe_wait_words(arrive0, ltag);
      const bool mine = (m >> 3) == h;
      const int src = (m & 7) + 16 * (q & 1);
      f32x4 v = red[(0 * 2 + (q >> 1)) * 64 + src];
#pragma unroll
      for (int i = 1; i < NW; ++i) v += red[(i * 2 + (q >> 1)) * 64 + src];
      v *= ws0;
      if (mine) gemv3_epilogue<1, E3_RESID>(a, 0, tile, lane, &v, &pre0, &hkeep);
      // publish phase 0: h (fp32), its partial sums of squares and its planes are out (write-through), drained, then the flag
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(flags0 + bid), "v"(epoch) : "memory");
      VA_STAMP(stamps, 2);                                 // wave 0: phase 0 published
    }
  } else {
    load_w13();            // no phase-0 tile: at once (measured: held back until the hand-off these 64 workgroups become phase 1's laggards)
  }
  if (wid == 0 && narrow) load_w13();                      // wave 0: once it has published, BEFORE it polls (behind the poll: +3 % on the loop)

  // ---- hand-off 0: wave 0 polls the 192 phase-0 flags (lanes 0..47: four each), bounded; the raw barrier releases the others.
  //      `red` is free again behind it too (wave 0 summed the phase-0 tiles before it published).
  const bool broken0 = mlpe_poll_flags(flags0, 48, epoch, e, wid, lane);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  VA_STAMP(stamps, 3);                                     // hand-off 0 passed

  // ================================================================ phase 1: w1||w3 + SwiGLU (gemv3_kernel<6, 8, 2, E3_SWIGLU, true>)
  {
    Gemv3Args a = e.p1;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.XP), 0, a.R * VA_NPL * (K1 / 8) * 256, 0x00020000);
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.ss_in), 0, 96 * 16 * 4, 0x00020000);
    u32x4 xb[G][VA_NPL];
    const int xl16 = m < a.rows ? lane16 : 0x7ffffff0;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int p = 0; p < VA_NPL; ++p)
        xb[g][p] = __builtin_amdgcn_raw_buffer_load_b128(xrs, xl16, (int)((p * (K1 / 8) * 16 + (w1 * G + g) * 64) * 16), 16 /* sc1 */);
    constexpr int NSS = K1 / 64;
    float ssv[NSS];
    if (wid == 0) {
#pragma unroll
      for (int j = 0; j < NSS; ++j)                        // the producers' partial sums: written in THIS launch -> sc1 as well
        ssv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srs, ((q + 4 * j) * 16 + m) * 4, 0, 16 /* sc1 */));
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[T][NACC];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int p = 0; p < NACC; ++p) acc[t][p] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
      for (int t = 0; t < T; ++t) {
        f16x8 wf[F32 ? 2 : 1];
        wf[0] = __builtin_bit_cast(f16x8, wb[t][g][0]);
        if constexpr (F32) wf[1] = __builtin_bit_cast(f16x8, wb[t][g][WH - 1]);
        mfma_group<WT>(wf, xb[g], acc[t]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < T; ++t) red[(wid * T + t) * 64 + lane] = acc_sum<WT>(acc[t]);
    if (lane == 0) __hip_atomic_store(arrive1 + wid, ltag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    VA_STAMP(stamps, 4);                                   // phase 1: products done
    if (wid == 0) {
      float ssp = 0.f;
#pragma unroll
      for (int j = 0; j < NSS; ++j) ssp += ssv[j];
      ssp += va_xor16(ssp);
      ssp += va_xor32(ssp);
      const float rinv = 1.0f / sqrtf(ssp * (1.0f / (float)a.k_total) + a.eps);
      mlpe_wait_words(arrive1, ltag);
      f32x4 v[T];
#pragma unroll
      for (int t = 0; t < T; ++t) {
        f32x4 sacc = red[(0 * T + t) * 64 + lane];
#pragma unroll
        for (int i = 1; i < NW; ++i) sacc += red[(i * T + t) * 64 + lane];
        sacc *= ws1[t];
        v[t] = sacc * rinv;
      }
      gemv3_epilogue<T, E3_SWIGLU>(a, 0, tile0, lane, v, nullptr);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(flags1 + bid), "v"(epoch) : "memory");
      VA_STAMP(stamps, 5);                                 // wave 0: phase 1 published
    }
  }
  if (!narrow) {
    VA_STAMP_FLUSH(stamps, 12);
    return;
  }

  // ================================================================ phase 2: w2 + residual (gemv3h_kernel<8, 8, E3_RESID>)
  {
    Gemv3Args a = e.p2;
    a.W = W2q;
    constexpr int G2 = SH::G2, PL = SH::PL;
    constexpr int K = 64 * G2 * NW, KG = K / 32;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.W), 0, -16, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.XP), 0, a.R * VA_NPL * (K / 8) * 256, 0x00020000);
    const int voffx = (sb * 64 + q * 16 + la + 8 * h) * 16;
    unsigned char* myring = ring + wid * SH::WAVE_RING;
    // run-ahead: this wave's w2 slice (pairs [0, PL) by LDS-DMA, the rest into the registers phase 1 released); wave 0 comes here
    // once it has published phase 1
    u32x4 wreg[G2 - PL > 0 ? G2 - PL : 1][2][WH];
    {
      const unsigned char* src = static_cast<const unsigned char*>(a.W) + ((size_t)tile * KG + 2 * (size_t)(wn * G2)) * BS + lane * 16;
#pragma unroll
      for (int c = 0; c < 2 * PL; ++c)
#pragma unroll
        for (int hh = 0; hh < WH; ++hh)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)(c * WH + hh) * 1024),
                                           (__attribute__((address_space(3))) void*)(myring + (c * WH + hh) * 1024), 16, 0, 2 /* nt */);
#pragma unroll
      for (int j = PL; j < G2; ++j) {
        const int soff = (tile * KG + 2 * (wn * G2 + j)) * BS;
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
          for (int hh = 0; hh < WH; ++hh)
            wreg[j - PL][nh][hh] = __builtin_amdgcn_raw_buffer_load_b128(wrs, voffw0 + nh * 128, soff + hh * 1024, 2 /* nt */);
      }
    }
    (void)mlpe_poll_flags(flags1, 64, epoch, e, wid, lane, broken0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    VA_STAMP(stamps, 6);                                   // hand-off 1 passed
    u32x4 xb[G2][VA_NPL];
    {
      const int vx = la + 8 * h < a.rows ? voffx : 0x7ffffff0;
#pragma unroll
      for (int j = 0; j < G2; ++j)
#pragma unroll
        for (int p = 0; p < VA_NPL; ++p)
          xb[j][p] = __builtin_amdgcn_raw_buffer_load_b128(xrs, vx, (p * (K / 8) * 16 + (wn * G2 + j) * 128) * 16, 16 /* sc1 */);
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the LDS-DMA'd ring fragments of this wave have landed (see mlp_engine_kernel)
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[2][NACC];
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
      for (int p = 0; p < NACC; ++p) acc[nh][p] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < G2; ++j) {
#pragma unroll
      for (int nh = 0; nh < 2; ++nh) {
        f16x8 wf[F32 ? 2 : 1];
        if (j < PL) {
          const u32x4* fr = reinterpret_cast<const u32x4*>(myring + ((2 * j + sb) * WH) * 1024) + (la + 8 * nh + 16 * q);
          wf[0] = __builtin_bit_cast(f16x8, fr[0]);
          if constexpr (F32) wf[1] = __builtin_bit_cast(f16x8, fr[64]);
        } else {
          wf[0] = __builtin_bit_cast(f16x8, wreg[j - PL][nh][0]);
          if constexpr (F32) wf[1] = __builtin_bit_cast(f16x8, wreg[j - PL][nh][WH - 1]);
        }
        mfma_group<WT>(wf, xb[j], acc[nh]);
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
      v *= ws2;
      pre2.res = hkeep;                                    // the rows this wave produced in phase 0
      if (mine) gemv3_epilogue<1, E3_RESID>(a, 0, tile, lane, &v, &pre2);
    }
    VA_WAIT_VM(0);
    VA_STAMP_FLUSH(stamps, 12);                            // t7 of the record = done
  }
}
