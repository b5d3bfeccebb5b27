// Does an `sc1` load return FRESH data when the reading XCD's L2 (and the CU's L1) hold an older copy of the line?
//
// The one-launch MLP (csrc/mlp_engine.h) hands activations over inside a launch with write-through (sc0 sc1) stores, a drained flag and
// sc1 consumer loads (MI355X_MICROARCH.md "hand-offs measured with sc1 loads in place of the acquire").  Its third phase re-reads, with
// sc1 loads, ADDRESSES that the first phase of the same launch read with PLAIN loads 15 us earlier (the h planes: rewritten in between by
// the second phase's epilogues on other XCDs).  Per-XCD L2s are not coherent with each other: if an sc1 load could be served from the
// reader's own, older L2 copy, that phase would be safe only by eviction.  This probe asks the hardware:
//   every round:  ALL 256 workgroups plain-load the whole buffer (value = round - 1: now warm in every L1 and every XCD's L2, nothing
//                 else streams in between) -> grid barrier -> workgroup 0 rewrites the buffer with sc0 sc1 stores (value = round), drains,
//                 publishes a flag -> every workgroup polls the flag with sc1 loads, then sc1-loads the whole buffer and counts words
//                 that are not `round` (stale) -> grid barrier.
// Reported per XCD of the reader.   hipcc -O3 --offload-arch=gfx950 l2_warm_handoff_probe.hip -o l2_warm_handoff_probe && ./l2_warm_handoff_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e__)); return 2; } } while (0)

constexpr int MAXWORDS = 64 * 1024;   // up to 256 KB: the size of the ffn planes (larger than a CU's 32 KB L1: a streaming read of it self-evicts);
                                      // the 4 KB case stays L1-resident and is the control that plain loads DO go stale
constexpr int NWG = 256, NT = 256;

__device__ __forceinline__ uint32_t ld_sc1(const uint32_t* p) {
  uint32_t v;
  asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void st_wt(uint32_t* p, uint32_t v) { asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }

// grid barrier on a monotonic counter (agent-scope atomics; every workgroup resident: 256 workgroups of 100 KB LDS on 256 CUs)
// every spin is bounded (a hung GPU box is worse than a failed probe): ~2 s, then word [32] of `ctr` is raised and all waits fall through
__device__ void grid_barrier(uint32_t* ctr, uint32_t target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int spin = 0; ld_sc1(ctr) < target && !ld_sc1(ctr + 32); ++spin) {
      if (spin > 2000000) { st_wt(ctr + 32, 1u); break; }
      __builtin_amdgcn_s_sleep(4);
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(NT) void probe(uint32_t* buf, uint32_t* flag, uint32_t* ctr, uint32_t* stale_per_xcd, uint32_t* plain_stale_per_xcd,
                                            int rounds, int mode, int WORDS) {
  extern __shared__ unsigned char pad[];    // 100 KB: one workgroup per CU
  if (threadIdx.x == 0) pad[0] = 0;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 15u;
  uint32_t bars = 0;
  for (int r = 1; r <= rounds; ++r) {
    // (1) warm: plain loads of the whole buffer by every workgroup (expects r - 1; a mismatch here would be a bug of the probe)
    uint32_t acc = 0;
    for (int i = threadIdx.x; i < WORDS; i += NT) acc += buf[i] != (uint32_t)(r - 1);
    if (acc) atomicAdd(plain_stale_per_xcd + 8 + xcc, acc);
    grid_barrier(ctr, ++bars * NWG);
    // (2) workgroup 0 rewrites it write-through, drains, publishes
    if (blockIdx.x == 0) {
      for (int i = threadIdx.x; i < WORDS; i += NT) st_wt(buf + i, (uint32_t)r);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) st_wt(flag, (uint32_t)r);
    }
    // (3) consumers: poll the flag (sc1), then read the buffer
    if (threadIdx.x == 0)
      for (int spin = 0; ld_sc1(flag) != (uint32_t)r && !ld_sc1(ctr + 32); ++spin) {
        if (spin > 2000000) { st_wt(ctr + 32, 1u); break; }
        __builtin_amdgcn_s_sleep(2);
      }
    __syncthreads();
    uint32_t stale = 0, pstale = 0;
    for (int i = threadIdx.x; i < WORDS; i += NT) {
      if (mode == 0) stale += ld_sc1(buf + i) != (uint32_t)r;                  // what the engine does
      else pstale += buf[i] != (uint32_t)r;                                    // control: plain loads (expected stale: L1 / L2 copies)
    }
    if (stale) atomicAdd(stale_per_xcd + xcc, stale);
    if (pstale) atomicAdd(plain_stale_per_xcd + xcc, pstale);
    grid_barrier(ctr, ++bars * NWG);
  }
}

int main() {
  uint32_t *buf, *flag, *ctr, *stale, *pstale;
  CK(hipMalloc(&buf, MAXWORDS * 4)); CK(hipMalloc(&flag, 256)); CK(hipMalloc(&ctr, 256)); CK(hipMalloc(&stale, 64)); CK(hipMalloc(&pstale, 64));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  const int rounds = 200;
  for (int WORDS : {MAXWORDS, 1024})
  for (int mode = 0; mode < 2; ++mode) {
    CK(hipMemset(buf, 0, MAXWORDS * 4)); CK(hipMemset(flag, 0, 256)); CK(hipMemset(ctr, 0, 256)); CK(hipMemset(stale, 0, 64)); CK(hipMemset(pstale, 0, 64));
    probe<<<NWG, NT, 100 * 1024>>>(buf, flag, ctr, stale, pstale, rounds, mode, WORDS);
    CK(hipDeviceSynchronize());
    uint32_t hs[16], hp[16];
    CK(hipMemcpy(hs, stale, 64, hipMemcpyDeviceToHost)); CK(hipMemcpy(hp, pstale, 64, hipMemcpyDeviceToHost));
    uint32_t aborted = 0;
    CK(hipMemcpy(&aborted, ctr + 32, 4, hipMemcpyDeviceToHost));
    if (aborted) printf("PROBE ABORTED: a wait gave up (not every workgroup resident?)\n");
    unsigned long long tot = 0, ptot = 0, warm_bad = 0;
    for (int x = 0; x < 8; ++x) { tot += hs[x]; ptot += hp[x]; warm_bad += hp[8 + x]; }
    printf("%s consumer loads, %d rounds x 256 workgroups x %d words: stale words %llu (per reader XCD:", mode == 0 ? "sc1  " : "plain", rounds, WORDS,
           mode == 0 ? tot : ptot);
    for (int x = 0; x < 8; ++x) printf(" %u", mode == 0 ? hs[x] : hp[x]);
    printf(")   warm-phase mismatches %llu\n", warm_bad);
    if (mode == 0 && tot) { printf("RESULT: sc1 loads CAN return stale data from a warm L2\n"); }
  }
  return 0;
}
