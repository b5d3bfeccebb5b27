// Stand-alone microbenchmark of the weight-streaming GEMV variants (development tool, not product).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I../../vaura_amd/csrc gemv_bench.hip -o gemv_bench
// Each timed launch streams a DIFFERENT weight set (cycling through > 256 MiB) so the Infinity Cache
// cannot serve it, like the real decode step where every matrix is read once per step.
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include "gemv_kernel.h"
#include "gemv3_kernel.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e, __FILE__, __LINE__); exit(1); } } while (0)

template <int G, int NW, int T, bool NT>
__global__ __launch_bounds__(NW * 64) void stream_only(const u32x4* __restrict__ W, float* out) {
  constexpr int KG = G * NW;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int tile0 = blockIdx.x * T;
  u32x4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const size_t kg = (size_t)(tile0 + t) * KG + (size_t)(w * G + g);
      u32x4 v = NT ? __builtin_nontemporal_load(W + kg * 64 + lane) : W[kg * 64 + lane];
      acc ^= v;
    }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[threadIdx.x] = 1.f;
}

struct Case { const char* name; double bytes; };

// touch one dword per `stride` bytes of [p, p+bytes): pulls the lines into the memory-side Infinity Cache
__global__ void prefetch_kernel(const char* p, size_t bytes, int stride, float* sink) {
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * (size_t)stride;
  const size_t step = (size_t)gridDim.x * blockDim.x * (size_t)stride;
  uint32_t acc = 0;
  for (; i < bytes; i += step) acc ^= *reinterpret_cast<const uint32_t*>(p + i);
  if (acc == 0x12345678u) sink[threadIdx.x] = 1.f;
}

// time per launch inside a replayed hipGraph of `nsets` back-to-back launches (how the product runs them)
static hipStream_t g_stream;
template <typename F>
static double time_launches_impl(F&& launch, int nsets, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipGraph_t graph; hipGraphExec_t exec;
  CK(hipStreamBeginCapture(g_stream, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < nsets; ++i) launch(i);
  CK(hipStreamEndCapture(g_stream, &graph));
  CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  CK(hipGraphLaunch(exec, g_stream));
  CK(hipStreamSynchronize(g_stream));
  const int reps = (iters + nsets - 1) / nsets;
  CK(hipEventRecord(e0, g_stream));
  for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(exec, g_stream));
  CK(hipEventRecord(e1, g_stream));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
  return 1e3 * ms / (reps * nsets);  // us per launch
}

static const char* g_filter = nullptr;
static const char* g_current = "";
#define REPORT(name, us, bytes) do { g_current = name; report(name, us, bytes); } while (0)
#define time_launches(...) (g_filter && !strstr(g_current, g_filter) ? -1.0 : time_launches_impl(__VA_ARGS__))
int main(int argc, char** argv) {
  if (argc > 1) g_filter = argv[1];
  const int rows = 16;
  CK(hipStreamCreate(&g_stream));
  const int NSETS = 24;
  // the five decode GEMVs: (name, N, K)
  struct Shape { const char* name; int N, K; } shapes[] = {
      {"qkv", 4608, 1536}, {"wo", 1536, 1536}, {"w13", 8192, 1536}, {"w2", 1536, 4096}, {"heads", 9216, 1536}};
  size_t maxW = (size_t)9216 * 1536 * 2;
  char* Wbuf;
  CK(hipMalloc(&Wbuf, maxW * NSETS));
  CK(hipMemset(Wbuf, 0x3c, maxW * NSETS));
  float *x, *gain, *res, *out;
  CK(hipMalloc(&x, 16 * 4096 * 4)); CK(hipMemset(x, 0, 16 * 4096 * 4));
  CK(hipMalloc(&gain, 4096 * 4)); CK(hipMemset(gain, 0, 4096 * 4));
  CK(hipMalloc(&res, 16 * 9216 * 4)); CK(hipMemset(res, 0, 16 * 9216 * 4));
  CK(hipMalloc(&out, 16 * 9216 * 4));
  const int iters = 240;

  auto args = [&](int set, int N, int K, bool norm, bool resid) {
    GemvArgs a;
    a.W = Wbuf + (size_t)set * maxW; a.X = x; a.gain = norm ? gain : nullptr; a.res = resid ? res : nullptr; a.out = out;
    a.rows = rows; a.R = 1; a.N = N; a.eps = 1e-5f;
    return a;
  };
  auto report = [&](const char* what, double us, double bytes) {
    if (us < 0) return;
    printf("%-44s %8.2f us  %7.2f TB/s\n", what, us, bytes / us * 1e-6);
  };

  // ---- floor: pure streaming with the GEMV's own access pattern
  {
    double b = 8192.0 * 1536 * 2;
    REPORT("stream w13 G6 NW8 T2 nt (256 WG x 512)", time_launches([&](int s) {
      hipLaunchKernelGGL((stream_only<6, 8, 2, true>), dim3(256), dim3(512), 0, g_stream, (const u32x4*)(Wbuf + (size_t)s * maxW), out); }, NSETS, iters), b);
    REPORT("stream w13 G6 NW8 T2 plain", time_launches([&](int s) {
      hipLaunchKernelGGL((stream_only<6, 8, 2, false>), dim3(256), dim3(512), 0, g_stream, (const u32x4*)(Wbuf + (size_t)s * maxW), out); }, NSETS, iters), b);
    REPORT("stream w13 G6 NW8 T1 nt (512 WG x 512)", time_launches([&](int s) {
      hipLaunchKernelGGL((stream_only<6, 8, 1, true>), dim3(512), dim3(512), 0, g_stream, (const u32x4*)(Wbuf + (size_t)s * maxW), out); }, NSETS, iters), b);
    REPORT("stream w13 G12 NW4 T1 nt (512 WG x 256)", time_launches([&](int s) {
      hipLaunchKernelGGL((stream_only<12, 4, 1, true>), dim3(512), dim3(256), 0, g_stream, (const u32x4*)(Wbuf + (size_t)s * maxW), out); }, NSETS, iters), b);
    REPORT("stream w13 G3 NW16 T1 nt (512 WG x 1024)", time_launches([&](int s) {
      hipLaunchKernelGGL((stream_only<3, 16, 1, true>), dim3(512), dim3(1024), 0, g_stream, (const u32x4*)(Wbuf + (size_t)s * maxW), out); }, NSETS, iters), b);
    REPORT("stream w13 G3 NW16 T4 nt (128 WG x 1024)", time_launches([&](int s) {
      hipLaunchKernelGGL((stream_only<3, 16, 4, true>), dim3(128), dim3(1024), 0, g_stream, (const u32x4*)(Wbuf + (size_t)s * maxW), out); }, NSETS, iters), b);
    double bq = 4608.0 * 1536 * 2;
    REPORT("stream qkv G6 NW8 T1 nt (288 WG x 512)", time_launches([&](int s) {
      hipLaunchKernelGGL((stream_only<6, 8, 1, true>), dim3(288), dim3(512), 0, g_stream, (const u32x4*)(Wbuf + (size_t)s * maxW), out); }, NSETS, iters), bq);
    double bo = 1536.0 * 1536 * 2;
    REPORT("stream wo G6 NW8 T1 nt (96 WG x 512)", time_launches([&](int s) {
      hipLaunchKernelGGL((stream_only<6, 8, 1, true>), dim3(96), dim3(512), 0, g_stream, (const u32x4*)(Wbuf + (size_t)s * maxW), out); }, NSETS, iters), bo);
    REPORT("stream wo G3 NW16 T1 nt (96 WG x 1024)", time_launches([&](int s) {
      hipLaunchKernelGGL((stream_only<3, 16, 1, true>), dim3(96), dim3(1024), 0, g_stream, (const u32x4*)(Wbuf + (size_t)s * maxW), out); }, NSETS, iters), bo);
    double b2 = 1536.0 * 4096 * 2;
    REPORT("stream w2 G16 NW8 T1 nt (96 WG x 512)", time_launches([&](int s) {
      hipLaunchKernelGGL((stream_only<16, 8, 1, true>), dim3(96), dim3(512), 0, g_stream, (const u32x4*)(Wbuf + (size_t)s * maxW), out); }, NSETS, iters), b2);
    REPORT("stream w2 G8 NW16 T1 nt (96 WG x 1024)", time_launches([&](int s) {
      hipLaunchKernelGGL((stream_only<8, 16, 1, true>), dim3(96), dim3(1024), 0, g_stream, (const u32x4*)(Wbuf + (size_t)s * maxW), out); }, NSETS, iters), b2);
    REPORT("empty-ish launch (1 WG)", time_launches([&](int s) {
      hipLaunchKernelGGL((stream_only<1, 1, 1, true>), dim3(1), dim3(64), 0, g_stream, (const u32x4*)(Wbuf + (size_t)s * maxW), out); }, NSETS, iters), 1.0);
  }
  // ---- the real kernels
  REPORT("gemv qkv  <6,8,1,STORE,norm>", time_launches([&](int s) {
    hipLaunchKernelGGL((gemv_kernel<true, 6, 8, 1, EPI_STORE, true>), dim3(288), dim3(512), 0, g_stream, args(s, 4608, 1536, true, false)); }, NSETS, iters), 4608.0 * 1536 * 2);
  REPORT("  qkv ablate: no MFMA", time_launches([&](int s) {
    hipLaunchKernelGGL((gemv_kernel<true, 6, 8, 1, EPI_STORE, true, 1>), dim3(288), dim3(512), 0, g_stream, args(s, 4608, 1536, true, false)); }, NSETS, iters), 4608.0 * 1536 * 2);
  REPORT("  qkv ablate: no x loads", time_launches([&](int s) {
    hipLaunchKernelGGL((gemv_kernel<true, 6, 8, 1, EPI_STORE, true, 2>), dim3(288), dim3(512), 0, g_stream, args(s, 4608, 1536, true, false)); }, NSETS, iters), 4608.0 * 1536 * 2);
  REPORT("  qkv ablate: no MFMA, no x", time_launches([&](int s) {
    hipLaunchKernelGGL((gemv_kernel<true, 6, 8, 1, EPI_STORE, true, 3>), dim3(288), dim3(512), 0, g_stream, args(s, 4608, 1536, true, false)); }, NSETS, iters), 4608.0 * 1536 * 2);
  REPORT("  qkv ablate: no W loads", time_launches([&](int s) {
    hipLaunchKernelGGL((gemv_kernel<true, 6, 8, 1, EPI_STORE, true, 4>), dim3(288), dim3(512), 0, g_stream, args(s, 4608, 1536, true, false)); }, NSETS, iters), 4608.0 * 1536 * 2);
  REPORT("  qkv ablate: no W, no x (MFMA only)", time_launches([&](int s) {
    hipLaunchKernelGGL((gemv_kernel<true, 6, 8, 1, EPI_STORE, true, 6>), dim3(288), dim3(512), 0, g_stream, args(s, 4608, 1536, true, false)); }, NSETS, iters), 4608.0 * 1536 * 2);
  REPORT("  qkv no-norm", time_launches([&](int s) {
    hipLaunchKernelGGL((gemv_kernel<true, 6, 8, 1, EPI_STORE, false>), dim3(288), dim3(512), 0, g_stream, args(s, 4608, 1536, false, false)); }, NSETS, iters), 4608.0 * 1536 * 2);
  REPORT("  qkv <12,4,1> (256 thr)", time_launches([&](int s) {
    hipLaunchKernelGGL((gemv_kernel<true, 12, 4, 1, EPI_STORE, true>), dim3(288), dim3(256), 0, g_stream, args(s, 4608, 1536, true, false)); }, NSETS, iters), 4608.0 * 1536 * 2);
  REPORT("  qkv <3,16,1> (1024 thr)", time_launches([&](int s) {
    hipLaunchKernelGGL((gemv_kernel<true, 3, 16, 1, EPI_STORE, true>), dim3(288), dim3(1024), 0, g_stream, args(s, 4608, 1536, true, false)); }, NSETS, iters), 4608.0 * 1536 * 2);
  REPORT("  qkv <6,8,2> (144 WG)", time_launches([&](int s) {
    hipLaunchKernelGGL((gemv_kernel<true, 6, 8, 2, EPI_STORE, true>), dim3(144), dim3(512), 0, g_stream, args(s, 4608, 1536, true, false)); }, NSETS, iters), 4608.0 * 1536 * 2);
  REPORT("gemv wo   <6,8,1,RESID>", time_launches([&](int s) {
    hipLaunchKernelGGL((gemv_kernel<true, 6, 8, 1, EPI_RESID, false>), dim3(96), dim3(512), 0, g_stream, args(s, 1536, 1536, false, true)); }, NSETS, iters), 1536.0 * 1536 * 2);
  REPORT("gemv w13  <6,8,2,SWIGLU,norm>", time_launches([&](int s) {
    hipLaunchKernelGGL((gemv_kernel<true, 6, 8, 2, EPI_SWIGLU, true>), dim3(256), dim3(512), 0, g_stream, args(s, 4096, 1536, true, false)); }, NSETS, iters), 8192.0 * 1536 * 2);
  REPORT("gemv w2   <16,8,1,RESID>", time_launches([&](int s) {
    hipLaunchKernelGGL((gemv_kernel<true, 16, 8, 1, EPI_RESID, false>), dim3(96), dim3(512), 0, g_stream, args(s, 1536, 4096, false, true)); }, NSETS, iters), 1536.0 * 4096 * 2);
  REPORT("gemv heads<6,8,2,LOGITS,norm>", time_launches([&](int s) {
    hipLaunchKernelGGL((gemv_kernel<true, 6, 8, 2, EPI_LOGITS, true>), dim3(288), dim3(512), 0, g_stream, args(s, 9216, 1536, true, false)); }, NSETS, iters), 9216.0 * 1536 * 2);
  // ---- bf16-MFMA kernels on split rows
  uint16_t* xs; float* ssb; uint16_t* osp;
  CK(hipMalloc(&xs, 16 * 4096 * 6)); CK(hipMemset(xs, 0, 16 * 4096 * 6));
  CK(hipMalloc(&ssb, 96 * 16 * 4 * 4)); CK(hipMemset(ssb, 0, 96 * 16 * 4 * 4));
  CK(hipMalloc(&osp, 16 * 9216 * 6));
  auto a3 = [&](int set, int N, bool norm, bool resid, bool split) {
    Gemv3Args a;
    a.W = Wbuf + (size_t)set * maxW; a.XP = xs; a.ss_in = norm ? ssb : nullptr; a.n_ss_in = 96; a.res = resid ? res : nullptr;
    a.out = out; a.outp = split ? osp : nullptr; a.gain_out = split ? gain : nullptr; a.ss_out = split ? ssb + 96 * 16 : nullptr;
    a.rows = rows; a.R = 1; a.N = N; a.eps = 1e-5f; a.k_total = 1536;
    return a;
  };
#define G3(name, G, NW, T, EPI, NORM, XB, ABL, grid, N, resid, split, bytes) \
  REPORT(name, time_launches([&](int s) { hipLaunchKernelGGL((gemv3_kernel<G, NW, T, EPI, NORM, XB, ABL>), dim3(grid), dim3(NW * 64), 0, g_stream, a3(s, N, NORM, resid, split).W, a3(s, N, NORM, resid, split).XP, a3(s, N, NORM, resid, split)); }, NSETS, iters), bytes)
  const double bq = 4608.0 * 1536 * 2, bo = 1536.0 * 1536 * 2, b13 = 8192.0 * 1536 * 2, b2 = 1536.0 * 4096 * 2, bh = 9216.0 * 1536 * 2;
  G3("g3 qkv <6,8,1> norm", 6, 8, 1, E3_STORE, true, 1, 0, 288, 4608, false, false, bq);
  G3("   qkv same-phase slices (old)", 6, 8, 1, E3_STORE, true, 1, 8, 288, 4608, false, false, bq);
  G3("   qkv ablate no MFMA", 6, 8, 1, E3_STORE, true, 1, 1, 288, 4608, false, false, bq);
  G3("   qkv ablate no x", 6, 8, 1, E3_STORE, true, 1, 2, 288, 4608, false, false, bq);
  G3("   qkv ablate no W", 6, 8, 1, E3_STORE, true, 1, 4, 288, 4608, false, false, bq);
  G3("   qkv ablate no W no x", 6, 8, 1, E3_STORE, true, 1, 6, 288, 4608, false, false, bq);
  G3("   qkv <6,8,2> 144 WG", 6, 8, 2, E3_STORE, true, 1, 0, 144, 4608, false, false, bq);
  G3("   qkv <12,4,1> 288 WG x256", 12, 4, 1, E3_STORE, true, 1, 0, 288, 4608, false, false, bq);
  G3("   qkv <12,4,2> 144 WG x256", 12, 4, 2, E3_STORE, true, 1, 0, 144, 4608, false, false, bq);
  G3("   qkv <3,16,1> 288 WG x1024", 3, 16, 1, E3_STORE, true, 1, 0, 288, 4608, false, false, bq);
  G3("   qkv <3,16,2> 144 WG x1024", 3, 16, 2, E3_STORE, true, 1, 0, 144, 4608, false, false, bq);
  G3("   qkv <3,16,4> 72 WG x1024", 3, 16, 4, E3_STORE, true, 1, 0, 72, 4608, false, false, bq);
  // x as fp32 split in registers (timing only): product instances vs the same with ABL 16, and the 2-plane load-only bound
  G3("fx qkv <6,8,2> product shape", 6, 8, 2, E3_STORE, true, 1, 0, 144, 4608, false, false, bq);
  G3("fx qkv <6,8,2> fp32 x split in registers", 6, 8, 2, E3_STORE, true, 1, 16, 144, 4608, false, false, bq);
  G3("fx w13 <6,8,2> product shape", 6, 8, 2, E3_SWIGLU, true, 1, 0, 256, 4096, false, true, b13);
  G3("fx w13 <6,8,2> fp32 x split in registers", 6, 8, 2, E3_SWIGLU, true, 1, 16, 256, 4096, false, true, b13);
  G3("fx heads <6,8,3> product shape", 6, 8, 3, E3_LOGITS, true, 1, 0, 192, 9216, false, false, bh);
  G3("fx heads <6,8,3> fp32 x split in registers", 6, 8, 3, E3_LOGITS, true, 1, 16, 192, 9216, false, false, bh);
  G3("g3 wo  <6,8,1> resid+split", 6, 8, 1, E3_RESID, false, 1, 0, 96, 1536, true, true, bo);
  G3("   wo  <3,16,1>", 3, 16, 1, E3_RESID, false, 1, 0, 96, 1536, true, true, bo);
  G3("g3 w13 <6,8,2> swiglu", 6, 8, 2, E3_SWIGLU, true, 1, 0, 256, 4096, false, true, b13);
  G3("   w13 <3,16,2>", 3, 16, 2, E3_SWIGLU, true, 1, 0, 256, 4096, false, true, b13);
  G3("   w13 <12,4,2>", 12, 4, 2, E3_SWIGLU, true, 1, 0, 256, 4096, false, true, b13);
  G3("g3 w2  <8,16,1> xb2", 8, 16, 1, E3_RESID, false, 2, 0, 96, 1536, true, true, b2);
  G3("   w2  ablate no x", 8, 16, 1, E3_RESID, false, 2, 2, 96, 1536, true, true, b2);
  G3("g3 heads <6,8,2>", 6, 8, 2, E3_LOGITS, true, 1, 0, 288, 9216, false, false, bh);
  G3("   heads <3,16,2>", 3, 16, 2, E3_LOGITS, true, 1, 0, 288, 9216, false, false, bh);
  // ---- next-layer prefetch into the Infinity Cache on a second stream (fork/join inside the graph)
  {
    hipStream_t s2; CK(hipStreamCreate(&s2));
    auto timed_pair = [&](const char* name, int stride, int pf_blocks, bool do_prefetch) {
      if (g_filter && !strstr(name, g_filter)) return;
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      std::vector<hipEvent_t> fork(NSETS), join(NSETS);
      for (int i = 0; i < NSETS; ++i) { CK(hipEventCreate(&fork[i])); CK(hipEventCreate(&join[i])); }
      hipGraph_t graph; hipGraphExec_t exec;
      CK(hipStreamBeginCapture(g_stream, hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < NSETS; ++i) {
        if (do_prefetch) {
          CK(hipEventRecord(fork[i], g_stream));
          CK(hipStreamWaitEvent(s2, fork[i], 0));
          hipLaunchKernelGGL(prefetch_kernel, dim3(pf_blocks), dim3(256), 0, s2, Wbuf + (size_t)((i + 1) % NSETS) * maxW,
                             (size_t)(8192.0 * 1536 * 2), stride, out);
          CK(hipEventRecord(join[i], s2));
        }
        hipLaunchKernelGGL((gemv3_kernel<6, 8, 2, E3_SWIGLU, true, 1, 0>), dim3(256), dim3(512), 0, g_stream, a3(i, 4096, true, false, true).W, a3(i, 4096, true, false, true).XP, a3(i, 4096, true, false, true));
        hipLaunchKernelGGL((gemv3_kernel<6, 8, 1, E3_RESID, false, 1, 0>), dim3(96), dim3(512), 0, g_stream, a3(i, 1536, false, true, true).W, a3(i, 1536, false, true, true).XP, a3(i, 1536, false, true, true));
        if (do_prefetch) CK(hipStreamWaitEvent(g_stream, join[i], 0));
      }
      CK(hipStreamEndCapture(g_stream, &graph));
      CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
      CK(hipGraphLaunch(exec, g_stream)); CK(hipStreamSynchronize(g_stream));
      CK(hipEventRecord(e0, g_stream));
      for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(exec, g_stream));
      CK(hipEventRecord(e1, g_stream)); CK(hipEventSynchronize(e1));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%-44s %8.2f us per (w13 + wo) pair\n", name, 1e3 * ms / (10 * NSETS));
    };
    timed_pair("pf none", 0, 0, false);
    timed_pair("pf stride128 256 blocks", 128, 256, true);
    timed_pair("pf stride64 256 blocks", 64, 256, true);
    timed_pair("pf stride128 64 blocks", 128, 64, true);
    timed_pair("pf stride16 256 blocks (all bytes)", 16, 256, true);
  }
  G3("nt w13 <6,8,2> x plain", 6, 8, 2, E3_SWIGLU, true, 1, 0, 256, 4096, false, true, b13);
  G3("nt qkv <6,8,2> x plain", 6, 8, 2, E3_STORE, true, 1, 0, 144, 4608, false, false, bq);
  // ---- tile-count sweep after the rinv fix
  G3("sw qkv <6,8,2> 144", 6, 8, 2, E3_STORE, true, 1, 0, 144, 4608, false, false, bq);
  G3("sw qkv <6,8,3> 96", 6, 8, 3, E3_STORE, true, 1, 0, 96, 4608, false, false, bq);
  G3("sw qkv <12,4,2> 144", 12, 4, 2, E3_STORE, true, 1, 0, 144, 4608, false, false, bq);
  G3("sw qkv <3,16,2> 144", 3, 16, 2, E3_STORE, true, 1, 0, 144, 4608, false, false, bq);
  G3("sw w13 <6,8,4> 128", 6, 8, 4, E3_SWIGLU, true, 1, 0, 128, 4096, false, true, b13);
  G3("sw w13 <3,16,2> 256", 3, 16, 2, E3_SWIGLU, true, 1, 0, 256, 4096, false, true, b13);
  G3("sw w13 <3,16,4> 128", 3, 16, 4, E3_SWIGLU, true, 1, 0, 128, 4096, false, true, b13);
  G3("sw heads <6,8,4> 144", 6, 8, 4, E3_LOGITS, true, 1, 0, 144, 9216, false, false, bh);
  G3("sw heads <6,8,3> 192", 6, 8, 3, E3_LOGITS, true, 1, 0, 192, 9216, false, false, bh);
  G3("sw heads <3,16,4> 144", 3, 16, 4, E3_LOGITS, true, 1, 0, 144, 9216, false, false, bh);
  G3("sw wo <12,4,1> 96x256", 12, 4, 1, E3_RESID, false, 1, 0, 96, 1536, true, true, bo);
  G3("sw wo <3,16,1> 96x1024", 3, 16, 1, E3_RESID, false, 1, 0, 96, 1536, true, true, bo);
  G3("sw w2 <8,16,1> xb4 96", 8, 16, 1, E3_RESID, false, 4, 0, 96, 1536, true, true, b2);
  G3("sw w2 <8,16,1> xb1 96", 8, 16, 1, E3_RESID, false, 1, 0, 96, 1536, true, true, b2);
  G3("sw w2 <16,8,1> xb2 96x512", 16, 8, 1, E3_RESID, false, 2, 0, 96, 1536, true, true, b2);
  G3("sw w2 <16,8,1> xb4 96x512", 16, 8, 1, E3_RESID, false, 4, 0, 96, 1536, true, true, b2);
  return 0;
}
