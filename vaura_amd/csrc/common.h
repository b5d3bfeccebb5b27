// Shared device helpers for libvaura_hip (gfx950 / CDNA4 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include "../../include/vaura_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define VA_CHECK_LAUNCH()                      \
  do {                                         \
    hipError_t e__ = hipGetLastError();        \
    if (e__ != hipSuccess) return (int)e__;    \
  } while (0)
// launch + check; the sticky per-thread error is cleared first so that a stale error left by some
// earlier, unrelated runtime call (observed: hipErrorNoDevice from a device probe) is not reported here.
// When vaura_profile_loop arms va_prof_start/stop the launch carries its own start/stop events
// (hipExtLaunchKernelGGL): they time exactly the kernel, like rocprofv3's kernel trace does.
extern thread_local int va_prof_kind;
void va_prof_events(hipEvent_t* start, hipEvent_t* stop);
#define VA_LAUNCH(kern, grid, block, smem, stream, ...)                                              \
  do {                                                                                               \
    (void)hipGetLastError();                                                                         \
    if (va_prof_kind >= 0) {                                                                         \
      hipEvent_t va_e0, va_e1;                                                                       \
      va_prof_events(&va_e0, &va_e1);                                                                \
      hipExtLaunchKernelGGL(kern, grid, block, smem, stream, va_e0, va_e1, 0, __VA_ARGS__);         \
    } else {                                                                                         \
      hipLaunchKernelGGL(kern, grid, block, smem, stream, __VA_ARGS__);                              \
    }                                                                                                \
    VA_CHECK_LAUNCH();                                                                               \
  } while (0)

// packed-rows index (see vaura_hip.h): float index of (row, col) in a (rows x C) matrix
__host__ __device__ __forceinline__ size_t packed_index(int row, int col, int C) {
  return ((((size_t)(row >> 4) * (size_t)(C >> 2) + (size_t)(col >> 2)) << 4) + (size_t)(row & 15)) * 4 + (size_t)(col & 3);
}
// float4 index of (row, 4-col quad cq) in packed rows
__host__ __device__ __forceinline__ size_t packed_quad(int row, int cq, int C) {
  return (((size_t)(row >> 4) * (size_t)(C >> 2) + (size_t)cq) << 4) + (size_t)(row & 15);
}

__device__ __forceinline__ float bf16_lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

// ---- cross-lane exchanges on the VALU (DPP modifiers + gfx950's v_permlane{16,32}_swap) instead of ds_bpermute: __shfl_xor
// compiles to an LDS-crossbar round trip (~60+ cycles of latency each, and the LDS pipe is shared by the 8 waves of a workgroup);
// the decode attention kernel issued ~50 of them per wave, the GEMV epilogues 4-8 on their post-barrier critical path.
template <int CTRL>
__device__ __forceinline__ float va_dpp(float x) {   // x as seen through DPP control CTRL (all rows / banks enabled)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
#define VA_DPP_XOR1 0xB1    /* quad_perm [1,0,3,2] */
#define VA_DPP_XOR2 0x4E    /* quad_perm [2,3,0,1] */
#define VA_DPP_HALF_MIRROR 0x141   /* lane i <-> 7 - i of its 8: pairs the two quads (== xor 4 on quad-uniform values) */
#define VA_DPP_ROR8 0x128   /* rotate a 16-lane row by 8 == xor 8 */
// the value of lane ^ 16 / lane ^ 32.  v_permlane{16,32}_swap exchanges the odd rows of its first register with the even rows of
// its second.  PRECONDITION of va_xor16 / va_xor32 / wave_sum / wave_max: 1-D blocks whose size is a multiple of 64 (lane = threadIdx.x
// & 63: every kernel of this library is launched that way — the lane bit costs nothing, an mbcnt pair per call would) and the
// partner lane active (full waves at the call site).  tools/microbench/lane_exchange_probe.hip checks all six exchanges on device.
// The butterfly order of wave_sum / wave_max is 1, 2, 4, 8, 16, 32 since round 3 (round 2: 32 .. 1): fp32 sums round differently from
// round-2 records (LayerNorm in vit, the RVQ argmin distances, the sampler's softmax denominator, rinv) — bit-exactness claims against
// round-2 outputs do not carry over; the goldens (reference-generated) are what pins them.  Written as inline asm: with the builtin, hipcc 7.2 folds `select(lane bit, r[0], r[1])` of swap(x, x) to r[0] (and
// may give both operands one physical register) — tools/microbench/lane_exchange_probe.hip is the passing check of THIS form.  The
// s_nop covers the VALU-write -> permlane-read wait states the compiler would otherwise insert.
__device__ __forceinline__ float va_xor16(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return ((threadIdx.x >> 4) & 1) ? a : b;
}
__device__ __forceinline__ float va_xor32(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return ((threadIdx.x >> 5) & 1) ? a : b;
}
// all-reduce over the 64 lanes, butterfly order 1, 2, 4, 8, 16, 32 (after the xor-1 and xor-2 steps the values are uniform per quad,
// so the half-mirror pairs exactly what xor 4 would)
__device__ __forceinline__ float wave_sum(float v) {
  v += va_dpp<VA_DPP_XOR1>(v);
  v += va_dpp<VA_DPP_XOR2>(v);
  v += va_dpp<VA_DPP_HALF_MIRROR>(v);
  v += va_dpp<VA_DPP_ROR8>(v);
  v += va_xor16(v);
  v += va_xor32(v);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, va_dpp<VA_DPP_XOR1>(v));
  v = fmaxf(v, va_dpp<VA_DPP_XOR2>(v));
  v = fmaxf(v, va_dpp<VA_DPP_HALF_MIRROR>(v));
  v = fmaxf(v, va_dpp<VA_DPP_ROR8>(v));
  v = fmaxf(v, va_xor16(v));
  v = fmaxf(v, va_xor32(v));
  return v;
}

// round-to-nearest-even fp32 -> bf16 bits (inputs are finite weights)
__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

inline hipStream_t as_stream(vaura_stream_t s) { return (hipStream_t)s; }

// Opt a kernel into more than 64 KB of dynamic LDS — once per DEVICE (function attributes are per device: an engine on a second
// GPU of the same process needs its own call).  `done` is a per-call-site bit mask of devices already set up.
inline int va_big_lds_once(const void* fn, size_t bytes, unsigned long long* done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return VAURA_ERR_STATE;
  if (!((*done >> dev) & 1ull)) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return VAURA_ERR_STATE;
    *done |= 1ull << dev;
  }
  return 0;
}


// ---- output stores of the decode-step kernels: write-through (sc0 sc1), so that a kernel leaves no dirty lines in its XCD's L2
// for the end-of-kernel release to write back (MI355X_MICROARCH.md: a release costs ~1.7 us clean, ~6.5 us behind freshly dirtied
// lines): -1 % on the decode loop.  Inline asm because no builtin takes a per-lane address with cache-policy bits; hipcc does not
// know the statement is a store of 4 registers, so the string itself ends with the wait state that keeps the next instruction from
// overwriting the data registers before the store has read them (cdna_hip_programming.md, inline asm: stores).
// -DVAURA_PLAIN_STORES (experiment build, python -m vaura_amd.csrc.build --plain-stores): ordinary stores, for the A/B.
#ifndef VAURA_PLAIN_STORES
__device__ __forceinline__ void va_st16(void* p, const f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void va_st8(void* p, const uint2 v) { asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void va_st4(void* p, const float v) { asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
#else
__device__ __forceinline__ void va_st16(void* p, const f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void va_st8(void* p, const uint2 v) { *reinterpret_cast<uint2*>(p) = v; }
__device__ __forceinline__ void va_st4(void* p, const float v) { *reinterpret_cast<float*>(p) = v; }
#endif

// ---- in-kernel time stamps: DIAGNOSTIC BUILD ONLY (python -m vaura_amd.csrc.build --stamps -> libvaura_hip_stamps.so; the product
// library is compiled without VAURA_STAMPS and contains none of this).  Every wave of the decode-step kernels reads
// s_memrealtime (100 MHz, one counter for the whole chip: comparable across CUs and across launches) at fixed points and writes
// ONE 128-byte record per wave to a side buffer that no kernel reads: {kind | block << 8 | xcc << 40 | wave << 48, t0 .. t6, t7}.  tools/pmc_driver --stamps
// collects them, tools/stamp_report.py turns them into the per-phase shares of profiles/r03_stage_stamps.json.  The stamps
// drain the memory pipeline where they wait, so this build's run TIME means nothing; its phase SHARES do.
#ifdef VAURA_STAMPS
static __device__ unsigned long long* va_stamp_ptr = nullptr;   // [0] = record counter, [1] = capacity, [2] = 1: every wave writes a record (0: wave 0 only), records from [8]
struct VaStamps { unsigned long long t[7]; };
__device__ __forceinline__ unsigned long long va_now() {
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define VA_STAMP(st, i) do { __builtin_amdgcn_sched_barrier(0); (st).t[i] = va_now(); __builtin_amdgcn_sched_barrier(0); } while (0)
// wait until at most n vector-memory operations of this wave are outstanding (loads retire in order)
#define VA_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
__device__ __forceinline__ void va_stamp_flush(const VaStamps& st, int kind) {
  unsigned long long* p = va_stamp_ptr;
  if (!p || (threadIdx.x & 63) != 0) return;                 // lane 0: one 128-byte record per wave
  if (p[2] == 0 && threadIdx.x != 0) return;                 // [2] = 0: wave 0 only (least perturbation: read spans and gaps from this mode)
  const unsigned long long slot = atomicAdd(p, 1ull);
  const unsigned long long t7 = va_now();                    // the slot counter's round trip is over: what follows is fire-and-forget
  if (slot >= p[1]) return;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  unsigned long long* r = p + 8 + slot * 16;
  r[0] = (unsigned long long)kind | ((unsigned long long)(blockIdx.x + gridDim.x * blockIdx.y) << 8) | ((unsigned long long)(xcc & 15u) << 40) |
         ((unsigned long long)(threadIdx.x >> 6) << 48);
#pragma unroll
  for (int i = 0; i < 7; ++i) r[1 + i] = st.t[i];
  r[8] = t7;
}
#define VA_STAMP_DECL(st) VaStamps st = {}
#define VA_STAMP_FLUSH(st, kind) va_stamp_flush(st, kind)
#define VA_STAMP_SETTER(name) extern "C" int name(unsigned long long* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(va_stamp_ptr), &p, sizeof p); }
#else
#define VA_STAMP(st, i) do { } while (0)
#define VA_WAIT_VM(n) do { } while (0)
#define VA_STAMP_DECL(st) do { } while (0)
#define VA_STAMP_FLUSH(st, kind) do { } while (0)
#define VA_STAMP_SETTER(name)
#endif

// ---- launchers implemented across the .hip files (host side, internal)
int va_launch_gemv(const void* w, int wdtype, const float* x, const float* gain, const float* residual, float* out,
                   int64_t rows, int64_t N, int64_t K, int epilogue, float eps, hipStream_t s);
int va_attention_splits(int rows, int n_head, int max_len);   // workgroups per (row, head) for this shape
int va_launch_attention(const float* qkv, const float* qkv2, const float* rope, float* kc, float* vc, float* out,
                        uint16_t* outp, int rows, int n_head, int head_dim, int max_len, const int32_t* pos_dev, int pos_host,
                        float* part, int n_split, hipStream_t s, uint32_t* arrivals = nullptr, float pscale = 1.f, int kv_half = 0);
// K / V cache of one layer: the layer stride in ELEMENTS, the elements fp32 or (vaura_decoder.kv_dtype = 1 / 2) fp16 / fp8 e4m3
static inline float* va_kv_layer(const vaura_decoder* d, float* base, int layer) {
  const size_t kv_layer = (size_t)d->rows * d->dims.n_head * (size_t)d->max_len * (size_t)(d->dims.d_model / d->dims.n_head);
  if (d->kv_dtype == 1) return reinterpret_cast<float*>(reinterpret_cast<uint16_t*>(base) + layer * kv_layer);
  if (d->kv_dtype == 2) return reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(base) + layer * kv_layer);
  return base + layer * kv_layer;
}
struct Gemv3Args;
static inline bool va_is_fp8(int wdtype) { return wdtype == VAURA_W_FP8 || wdtype == VAURA_W_FP8H; }   // e4m3 tile pairs (both activation arithmetics)
unsigned va_debug_flags2_get();  // vaura_set_debug_flags2: the second word
unsigned va_debug_flags_get();   // vaura_set_debug_flags (gemv3.hip): kernel-variant switches for A/B measurements
int va_pack_weight_fp8(const float* src, void* dst, int64_t N, int64_t K, hipStream_t s);
int va_pack_weight_h(const float* src, void* dst, int64_t N, int64_t K, int planes, hipStream_t s);   // fp16 plane(s) + row scales
int va_launch_gemv3(const Gemv3Args& a, int64_t n_weight_rows, int64_t K, int epilogue, bool norm, hipStream_t s);
bool va_mlp_engine_eligible(const vaura_decoder* d);
struct VaEngineAttention {     // the next layer's attention as a fourth phase of the one-launch MLP (mlp_engine.h, ATT instances)
  const float* rope;
  float* kc;                   // K / V cache of THAT layer: (rows, n_head, max_len, 96)
  float* vc;
  float* out;                  // fp32 packed rows (rows x d_model)
  uint16_t* outp;              // planes for wo
  int n_head, max_len;
};
int va_launch_mlp_engine(const Gemv3Args& a13, const Gemv3Args& a2, const Gemv3Args* aq_next, uint32_t* flags, int32_t* state, int layer,
                         hipStream_t s, const VaEngineAttention* att = nullptr, const void* warm_ptr = nullptr, size_t warm_bytes = 0,
                         const void* warm0_ptr = nullptr, size_t warm0_bytes = 0);
// experiment builds only (-DVAURA_EXPERIMENT_ENGINES: csrc/experiments/)
int va_launch_attn_wo(const float* qkv, const float* qkv2, const float* rope, float* kc, float* vc, float* out, uint16_t* outp,
                      int rows, int n_head, int max_len, const int32_t* state, const Gemv3Args& awo, uint32_t* flags, int layer, hipStream_t s);
int va_launch_tail_engine(const Gemv3Args& awo, const Gemv3Args& a13, const Gemv3Args& a2, uint32_t* flags, int32_t* state, int layer,
                          hipStream_t s);
int va_launch_embed(const vaura_decoder* d, int pos_host, int n_pos, hipStream_t s);
int va_launch_sample(const float* logits, int B, int K, int vocab, const vaura_sampling* sp, const float* noise,
                     int noise_rows_per_step, const int32_t* state, int64_t step_host, int32_t* tokens_out,
                     int32_t* seq, int T, int S, int32_t* state_rw, hipStream_t s);
int va_launch_advance(int32_t* state, int set_to, hipStream_t s);
int va_launch_linear_pair(const uint16_t* in, const uint16_t* w, const float* bias, const float* res, float* out_raw,
                          uint16_t* out_act, int act, int B, int Lin, int Lout, int oshift, int Cin, int Cout, hipStream_t s);
int va_launch_rope_append(const vaura_decoder* d, int layer, int p0, int n_pos, hipStream_t s);
int va_launch_attention_prefill(const vaura_decoder* d, int layer, int p0, int n_pos, hipStream_t s);

// Epoch of an in-launch hand-off (mlp_engine.h, the attention + wo experiment): neither the flag words in global memory nor the arrival
// words in LDS are ever reset — a hand-off passes when every producer's word holds THIS launch's epoch — so no two launches that can
// see each other's words may share an epoch.  epoch = (sequence id state[3], 10 bits | launch-epoch counter state[5], 17 bits | layer)
// + 1.  state[5] is OWNED BY THE LIBRARY: bumped by whatever ends a decode step (the sampler's last workgroup, advance_kernel), never
// rewound — a caller that rewinds the position (state[0]) within a sequence still gets fresh epochs (round 4's epoch used state[0] and
// matched stale words when a (sequence, position) step was re-run).  state[3] is the CALLER's: a new value for every sequence start
// and for every decoder instance of the process — LDS keeps its words across launches and instances, so a state buffer that starts
// from zero again must come with a new sequence id (DecoderEngine._reset_state: a process-wide counter).
__device__ __forceinline__ uint32_t va_handoff_epoch(const int32_t* state, int layer) {
  return ((((uint32_t)state[3] & 0x3ffu) << 22) | (((uint32_t)state[5] & 0x1ffffu) << 5) | ((uint32_t)layer & 31u)) + 1u;
}
