// Stand-alone driver of the PRODUCT library for counter collection:  rocprofv3 --pmc ... -- ./pmc_driver <libvaura_hip.so> [...]
// rocprofv3 --pmc under python crashes on this image, and a profile of a separately compiled microbenchmark says nothing about
// the shipped code object — so this program dlopen()s the very libvaura_hip.so the plugins load and calls vaura_decode_step
// (include/vaura_hip.h) on a full-size decoder descriptor (24 layers, 1536/4096, 16 rows = 8 clips with CFG) filled with seeded
// values.  No python, no env/bash hop: the program itself goes after `--`.
//
//   pmc_driver <lib> [--weights h1|h2] [--steps 24] [--pos0 100] [--rows 16]
//   pmc_driver <lib> --time 7 [--flags 0,1,...] [--weights ...]     A/B timing, no profiler: for every debug-flag set
//       (vaura_set_debug_flags) one captured step graph; `--time` rounds of the full 228-step loop per variant, INTERLEAVED in
//       one process (median / min reported), then the per-stage averages of one eager pass (vaura_profile_loop)
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <chrono>
#include <string>
#include <thread>
#include <vector>
#include "../include/vaura_hip.h"

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e__)); exit(2); } } while (0)

__global__ void fill_f32(float* p, size_t n, float scale, float offset, uint32_t seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = offset + scale * ((float)(h & 0xffff) / 32768.0f - 1.0f);
  }
}
// fp16 plane images: a tile image is a permutation of the matrix, so any finite values are a valid weight set.  `planes` = 2: even
// 1 KiB blocks are hi planes (|v| < 4096), odd ones lo planes (|v| < 2)
__global__ void fill_f16(_Float16* p, size_t n, int planes, uint32_t seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const bool lo = planes == 2 && ((i >> 9) & 1);
    p[i] = (_Float16)((lo ? 2.0f : 4096.0f) * ((float)(h & 0xffff) / 32768.0f - 1.0f));
  }
}
// fp8 e4m3 tile pairs: any byte but the two NaN codes (0x7f, 0xff); exponents capped so that 1536-deep sums stay far from fp16's range
__global__ void fill_fp8(uint8_t* p, size_t n, uint32_t seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (uint8_t)((h & 0x80u) | ((h >> 8) & 0x3fu));      // |v| < 2
  }
}
__global__ void fill_const(float* p, size_t n, float v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void fill_i32(int32_t* p, size_t n, int32_t v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

static uint32_t g_seed = 1;
static const char* wname(int wd) { return wd == VAURA_W_H2 ? "h2" : (wd == VAURA_W_FP8 ? "fp8" : (wd == VAURA_W_FP8H ? "fp8h" : "h1")); }
static float* dev_f32(size_t n, float scale, float offset = 0.f) {
  float* p; CK(hipMalloc(&p, n * 4));
  fill_f32<<<1024, 256>>>(p, n, scale, offset, g_seed++);
  return p;
}
static void* dev_weight(size_t N, size_t K, int wd) {   // N x K fp16 plane(s) in MFMA-tile order + float scale[N] (2^-17: weights of +-0.03)
  if (wd == VAURA_W_FP8 || wd == VAURA_W_FP8H) {        // fp8 tile pairs + float scale[N] (2^-6: weights of +-0.03)
    char* p; CK(hipMalloc(&p, N * K + N * 4));
    fill_fp8<<<1024, 256>>>((uint8_t*)p, N * K, g_seed++);
    fill_const<<<64, 256>>>((float*)(p + N * K), N, 0.015625f);
    return p;
  }
  const int planes = wd == VAURA_W_H2 ? 2 : 1;
  char* p; CK(hipMalloc(&p, N * K * 2 * planes + N * 4));
  fill_f16<<<1024, 256>>>((_Float16*)p, N * K * planes, planes, g_seed++);
  fill_const<<<64, 256>>>((float*)(p + N * K * 2 * planes), N, 7.62939453125e-06f);
  return p;
}
template <typename T> static T* dev_zero(size_t n) { T* p; CK(hipMalloc(&p, n * sizeof(T))); CK(hipMemset(p, 0, n * sizeof(T))); return p; }

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s <libvaura_hip.so> [--weights h1|h2] [--steps N] [--pos0 P] [--rows R]\n", argv[0]); return 1; }
  int wd = VAURA_W_H1, steps = 24, pos0 = 100, rows = 16, rounds = 0, chains = 1, kv16 = 0, stride = 1;
  const char* stamps_out = nullptr;
  std::vector<unsigned> variants{0u};
  std::vector<unsigned> variants2{0u};        // the second flag word of each variant: --flags F or F:F2 (vaura_set_debug_flags2)
  for (int i = 2; i + 1 < argc; i += 2) {
    if (!strcmp(argv[i], "--time")) rounds = atoi(argv[i + 1]);
    if (!strcmp(argv[i], "--flags")) {
      variants.clear();
      variants2.clear();
      for (char* t = strtok(argv[i + 1], ","); t; t = strtok(nullptr, ",")) {
        char* colon = nullptr;
        variants.push_back((unsigned)strtoul(t, &colon, 0));
        variants2.push_back(colon && *colon == ':' ? (unsigned)strtoul(colon + 1, nullptr, 0) : 0u);
      }
    }
    if (!strcmp(argv[i], "--weights")) wd = !strcmp(argv[i + 1], "fp8h") ? VAURA_W_FP8H : (!strcmp(argv[i + 1], "fp8") ? VAURA_W_FP8 :
                                             ((!strcmp(argv[i + 1], "h2") || !strcmp(argv[i + 1], "f32")) ? VAURA_W_H2 : VAURA_W_H1));   // h1 | h2 | fp8 | fp8h
    else if (!strcmp(argv[i], "--steps")) steps = atoi(argv[i + 1]);
    else if (!strcmp(argv[i], "--pos0")) pos0 = atoi(argv[i + 1]);
    else if (!strcmp(argv[i], "--rows")) rows = atoi(argv[i + 1]);
    else if (!strcmp(argv[i], "--chains")) chains = atoi(argv[i + 1]);
    else if (!strcmp(argv[i], "--stride")) stride = atoi(argv[i + 1]);              // counter passes: positions pos0, pos0 + stride, ... (the caches are pre-filled)
    else if (!strcmp(argv[i], "--kv")) kv16 = !strcmp(argv[i + 1], "f16") ? 1 : (!strcmp(argv[i + 1], "f8") ? 2 : 0);   // fp16 / e4m3 K / V cache (vaura_decoder.kv_dtype)
    else if (!strcmp(argv[i], "--stamps")) stamps_out = argv[i + 1];
  }
  void* lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
  auto step = (int (*)(const vaura_decoder*, const vaura_sampling*, int, vaura_stream_t))dlsym(lib, "vaura_decode_step");
  auto ssize = (size_t (*)(int))dlsym(lib, "vaura_struct_size");
  if (!step || !ssize || ssize(3) != sizeof(vaura_decoder)) { fprintf(stderr, "library / header mismatch\n"); return 1; }

  const int NL = 24, D = 1536, F = 4096, H = 16, K = 9, V = 1024, Tv = 32, T = 220, S = T + K, ML = 256;
  const int batch = rows / 2, rp = (rows + 15) / 16 * 16;
  vaura_decoder d;
  memset(&d, 0, sizeof d);
  d.dims = vaura_dims{NL, D, H, F, K, V, 512, 1024, 768, 8, 7, 1e-5f};
  d.wdtype = wd; d.batch = batch; d.rows = rows; d.max_len = ML; d.timesteps = T; d.seq_len = S; d.n_cond_tokens = Tv;
  d.prefill_positions = 0;
  d.kv_dtype = kv16;           // (the caches below are allocated at the fp32 size either way)
  std::vector<vaura_layer_weights> lw(NL);
  for (int l = 0; l < NL; ++l) {
    lw[l].wqkv = dev_weight(3 * D, D, wd); lw[l].wo = dev_weight(D, D, wd);
    lw[l].w13 = dev_weight(2 * F, D, wd); lw[l].w2 = dev_weight(D, F, wd);
    lw[l].attn_norm = dev_f32(D, 0.2f, 1.0f); lw[l].ffn_norm = dev_f32(D, 0.2f, 1.0f);
  }
  d.layers_host = lw.data();
  d.heads = dev_weight((size_t)K * V, D, (wd == VAURA_W_FP8 || wd == VAURA_W_FP8H) ? VAURA_W_H1 : wd);     // fp8 storages keep one-plane heads
  d.final_norm = dev_f32(D, 0.2f, 1.0f);
  d.tok_emb = dev_f32((size_t)K * (V + 1) * 8, 1.f); d.tok_proj_w = dev_f32((size_t)K * 1024 * 8, 0.3f); d.tok_proj_b = dev_f32((size_t)K * 1024, 0.02f);
  d.tok_table = dev_f32((size_t)K * (V + 1) * 1024, 0.5f);
  d.empty_video = dev_f32(512, 0.02f);
  d.rope = dev_f32((size_t)ML * 48 * 2, 0.7f);
  d.cond_proj = dev_f32((size_t)((rows * Tv + 15) / 16 * 16) * 512, 0.3f);
  d.kcache = dev_f32((size_t)NL * rows * H * ML * 96, 0.5f); d.vcache = dev_f32((size_t)NL * rows * H * ML * 96, 0.5f);
  int32_t* seq; CK(hipMalloc(&seq, (size_t)batch * K * S * 4)); fill_i32<<<64, 256>>>(seq, (size_t)batch * K * S, 7);
  d.seq = seq;
  d.state = dev_zero<int32_t>(8);
  d.noise = nullptr;
  d.ws_h = dev_zero<float>((size_t)rp * D); d.ws_qkv = dev_zero<float>((size_t)rp * 3 * D); d.ws_qkv2 = dev_zero<float>((size_t)rp * 3 * D);
  d.ws_attn = dev_zero<float>((size_t)rp * D); d.ws_ffn = dev_zero<float>((size_t)rp * F); d.ws_logits = dev_zero<float>((size_t)rows * K * V);
  d.ws_h_split = dev_zero<uint16_t>((size_t)rp * 3 * D); d.ws_attn_split = dev_zero<uint16_t>((size_t)rp * 3 * D);
  d.ws_ffn_split = dev_zero<uint16_t>((size_t)rp * 3 * F); d.ws_ss = dev_zero<float>((size_t)(rp / 16) * (D / 16) * 16);
  d.first_norm = lw[0].attn_norm;
  d.ws_attn_part = nullptr;
  d.ws_sync = dev_zero<uint32_t>(768);
  vaura_sampling sp{1, 1.0f, 250, 0.0f, 6.0f, 1234ull, 0ull};
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  CK(hipDeviceSynchronize());
  if (stamps_out) {
    // <lib> must be the DIAGNOSTIC build (python -m vaura_amd.csrc.build --stamps): wave 0 of every workgroup of the decode-step
    // kernels writes s_memrealtime stamps to this side buffer (csrc/common.h); `steps` graph-replayed decode steps at pos0.
    auto set_g = (int (*)(unsigned long long*))dlsym(lib, "vaura_stamps_set_gemv3");
    auto set_a = (int (*)(unsigned long long*))dlsym(lib, "vaura_stamps_set_attention");
    auto gbuild = (int (*)(const vaura_decoder*, const vaura_sampling*, vaura_stream_t, vaura_step_graph_t*))dlsym(lib, "vaura_step_graph_build");
    auto gloop = (int (*)(const vaura_decoder*, const vaura_sampling*, int, int, vaura_step_graph_t, vaura_stream_t))dlsym(lib, "vaura_generate_loop");
    if (!set_g || !set_a || !gbuild || !gloop) { fprintf(stderr, "%s is not a stamps build\n", argv[1]); return 1; }
    const size_t cap = 1500000;
    unsigned long long* buf; CK(hipMalloc(&buf, (8 + cap * 16) * 8)); CK(hipMemset(buf, 0, (8 + cap * 16) * 8));
    unsigned long long hdr[3] = {0, cap, getenv("PMC_STAMP_ALL_WAVES") ? 1ull : 0ull};
    vaura_step_graph_t g;
    if (auto setf = (void (*)(unsigned))dlsym(lib, "vaura_set_debug_flags")) setf(variants[0]);    // --flags F: the variant to stamp
    if (auto setf2 = (void (*)(unsigned))dlsym(lib, "vaura_set_debug_flags2")) setf2(variants2[0]);
    if (gbuild(&d, &sp, st, &g)) { fprintf(stderr, "graph build\n"); return 3; }
    const int32_t st0[4] = {pos0, 0, 0, 1};
    CK(hipMemcpy(d.state, st0, sizeof st0, hipMemcpyHostToDevice));
    if (gloop(&d, &sp, 0, 4, g, st)) return 3;               // warm-up, unstamped
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(buf, hdr, sizeof hdr, hipMemcpyHostToDevice));
    if (set_g(buf) || set_a(buf)) { fprintf(stderr, "stamp setters failed\n"); return 3; }
    CK(hipMemcpy(d.state, st0, sizeof st0, hipMemcpyHostToDevice));
    if (gloop(&d, &sp, 0, steps, g, st)) return 3;
    CK(hipStreamSynchronize(st));
    std::vector<unsigned long long> host(8 + cap * 16);
    CK(hipMemcpy(host.data(), buf, host.size() * 8, hipMemcpyDeviceToHost));
    const size_t n = std::min<size_t>(host[0], cap);
    FILE* f = fopen(stamps_out, "wb");
    if (!f) { perror(stamps_out); return 1; }
    fwrite(host.data() + 8, 128, n, f);
    fclose(f);
    printf("stamps: %zu records of %d steps at position %d (rows %d, weights %s) -> %s\n", n, steps, pos0, rows, wname(wd), stamps_out);
    return 0;
  }
  if (chains > 1) {
    // Concurrency experiment: the batch as `chains` independent decode chains (rows / chains rows each: clips are independent,
    // SURVEY.md 8e), each with its own stream, step graph, K/V cache and workspaces, sharing the weights; one host thread per
    // chain replays its graph.  Wall time until ALL chains have finished their 228 steps, against one chain with all rows.
    auto gbuild = (int (*)(const vaura_decoder*, const vaura_sampling*, vaura_stream_t, vaura_step_graph_t*))dlsym(lib, "vaura_step_graph_build");
    auto gloop = (int (*)(const vaura_decoder*, const vaura_sampling*, int, int, vaura_step_graph_t, vaura_stream_t))dlsym(lib, "vaura_generate_loop");
    const int crow = rows / chains, cb = crow / 2, crp = (crow + 15) / 16 * 16;
    std::vector<vaura_decoder> ds(chains, d);
    std::vector<hipStream_t> sts(chains);
    std::vector<vaura_step_graph_t> gs(chains);
    for (int c = 0; c < chains; ++c) {
      vaura_decoder& e = ds[c];
      e.batch = cb; e.rows = crow;
      e.cond_proj = dev_f32((size_t)((crow * Tv + 15) / 16 * 16) * 512, 0.3f);
      e.kcache = dev_f32((size_t)NL * crow * H * ML * 96, 0.5f); e.vcache = dev_f32((size_t)NL * crow * H * ML * 96, 0.5f);
      int32_t* sq; CK(hipMalloc(&sq, (size_t)cb * K * S * 4)); fill_i32<<<64, 256>>>(sq, (size_t)cb * K * S, 7);
      e.seq = sq; e.state = dev_zero<int32_t>(8);
      e.ws_h = dev_zero<float>((size_t)crp * D); e.ws_qkv = dev_zero<float>((size_t)crp * 3 * D); e.ws_qkv2 = dev_zero<float>((size_t)crp * 3 * D);
      e.ws_attn = dev_zero<float>((size_t)crp * D); e.ws_ffn = dev_zero<float>((size_t)crp * F); e.ws_logits = dev_zero<float>((size_t)crow * K * V);
      e.ws_h_split = dev_zero<uint16_t>((size_t)crp * 3 * D); e.ws_attn_split = dev_zero<uint16_t>((size_t)crp * 3 * D);
      e.ws_ffn_split = dev_zero<uint16_t>((size_t)crp * 3 * F); e.ws_ss = dev_zero<float>((size_t)(crp / 16) * (D / 16) * 16);
      CK(hipStreamCreateWithFlags(&sts[c], hipStreamNonBlocking));
      CK(hipDeviceSynchronize());
      const int rc = gbuild(&e, &sp, sts[c], &gs[c]);
      if (rc) { fprintf(stderr, "graph build chain %d: %d\n", c, rc); return 3; }
    }
    const int n = S - 1;
    const bool one_thread = getenv("PMC_ONE_THREAD") != nullptr;
    std::vector<float> wall;
    for (int r = -1; r < (rounds > 0 ? rounds : 5); ++r) {
      int32_t zero[4] = {0, 0, 0, 1};
      for (int c = 0; c < chains; ++c) CK(hipMemcpy(ds[c].state, zero, sizeof zero, hipMemcpyHostToDevice));
      CK(hipDeviceSynchronize());
      const auto h0 = std::chrono::steady_clock::now();
      if (one_thread) {
        // one host thread, steps interleaved over the chains
        for (int i = 0; i < n; ++i)
          for (int c = 0; c < chains; ++c)
            if (gloop(&ds[c], &sp, 0, 1, gs[c], sts[c])) { fprintf(stderr, "loop\n"); return 3; }
      } else {
        std::vector<std::thread> th;
        for (int c = 0; c < chains; ++c)
          th.emplace_back([&, c] { (void)hipSetDevice(0); if (gloop(&ds[c], &sp, 0, n, gs[c], sts[c])) fprintf(stderr, "loop failed\n"); });
        for (auto& t : th) t.join();
      }
      const auto h1 = std::chrono::steady_clock::now();
      CK(hipDeviceSynchronize());
      const auto h2 = std::chrono::steady_clock::now();
      if (r >= 0) wall.push_back(std::chrono::duration<float, std::milli>(h2 - h0).count());
      if (r == 0) printf("host enqueue %.3f ms, wall %.3f ms\n", std::chrono::duration<float, std::milli>(h1 - h0).count(), wall.back());
    }
    std::sort(wall.begin(), wall.end());
    printf("chains %d x rows %d (%s): %d-step loops of ALL chains: median wall %.3f ms, min %.3f ms\n", chains, crow,
           one_thread ? "one host thread" : "one host thread per chain", n, wall[wall.size() / 2], wall[0]);
    return 0;
  }
  if (rounds > 0) {
    auto gbuild = (int (*)(const vaura_decoder*, const vaura_sampling*, vaura_stream_t, vaura_step_graph_t*))dlsym(lib, "vaura_step_graph_build");
    auto gloop = (int (*)(const vaura_decoder*, const vaura_sampling*, int, int, vaura_step_graph_t, vaura_stream_t))dlsym(lib, "vaura_generate_loop");
    auto setf1 = (void (*)(unsigned))dlsym(lib, "vaura_set_debug_flags");
    auto setf2 = (void (*)(unsigned))dlsym(lib, "vaura_set_debug_flags2");      // absent in experiment builds of older trees
    size_t cur_v = 0;
    auto setf = [&](unsigned f) { setf1(f); if (setf2) setf2(f == 0 && cur_v >= variants2.size() ? 0u : variants2[cur_v]); };
    auto prof = (int (*)(const vaura_decoder*, const vaura_sampling*, int, unsigned, double*, int64_t*, vaura_stream_t))dlsym(lib, "vaura_profile_loop");
    if (!gbuild || !gloop || !setf1 || !prof) { fprintf(stderr, "missing symbols\n"); return 1; }
    const int n = S - 1;
    std::vector<vaura_step_graph_t> graphs(variants.size());
    for (size_t v = 0; v < variants.size(); ++v) {
      cur_v = v;
      setf(variants[v]);
      const int rc = gbuild(&d, &sp, st, &graphs[v]);
      if (rc) { fprintf(stderr, "graph build (flags %u): %d\n", variants[v], rc); return 3; }
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<std::vector<float>> ms(variants.size()), host_ms(variants.size());
    const bool eager = getenv("PMC_EAGER") != nullptr;     // time eager launches instead of graph replays
    int32_t zero[4] = {0, 0, 0, 0};
    for (int r = -1; r < rounds; ++r)          // round -1 = warm-up
      for (size_t v = 0; v < variants.size(); ++v) {
        zero[3] = (zero[3] + 1) & 0x7FF;       // new sequence id per loop (epochs of the in-launch hand-offs)
        CK(hipMemcpyAsync(d.state, zero, sizeof zero, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        const auto h0 = std::chrono::steady_clock::now();
        cur_v = v;
        if (eager) setf(variants[v]);
        const int rc = gloop(&d, &sp, 0, n, eager ? nullptr : graphs[v], st);
        const auto h1 = std::chrono::steady_clock::now();
        if (rc) { fprintf(stderr, "generate_loop: %d\n", rc); return 3; }
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        int32_t sw = 0;                        // sticky status word (include/vaura_hip.h): a hand-off that gave up stops every later one from waiting
        CK(hipMemcpy(&sw, d.state + 4, sizeof sw, hipMemcpyDeviceToHost));
        if (sw) { fprintf(stderr, "STATUS WORD 0x%x after a loop of flags %u:%u: the timing of this run is INVALID\n", sw, variants[v], variants2[v]); return 5; }
        if (r >= 0) { ms[v].push_back(t); host_ms[v].push_back(std::chrono::duration<float, std::milli>(h1 - h0).count()); }
      }
    static const char* kinds[8] = {"embed", "qkv", "attn", "wo", "w13", "w2", "heads", "sample"};
    for (size_t v = 0; v < variants.size(); ++v) {
      std::sort(ms[v].begin(), ms[v].end());
      std::sort(host_ms[v].begin(), host_ms[v].end());
      printf("host enqueue of the loop (%s): median %.3f ms\n", eager ? "eager launches" : "graph replays", host_ms[v][host_ms[v].size() / 2]);
      cur_v = v;
      setf(variants[v]);
      zero[3] = (zero[3] + 1) & 0x7FF;
      CK(hipMemcpy(d.state, zero, sizeof zero, hipMemcpyHostToDevice));
      double tot[8]; int64_t cnt[8];
      const int rc = prof(&d, &sp, n, 0xFF, tot, cnt, st);
      if (rc) { fprintf(stderr, "profile_loop: %d\n", rc); return 3; }
      printf("flags %u:%u weights %s rows %d: loop of %d steps median %.3f ms min %.3f ms (%.1f us/step) |", variants[v], variants2[v],
             wname(wd), rows, n, ms[v][ms[v].size() / 2], ms[v][0], 1e3 * ms[v][ms[v].size() / 2] / n);
      for (int k = 0; k < 8; ++k) printf(" %s %.2f", kinds[k], 1e3 * tot[k] / (cnt[k] ? cnt[k] : 1));
      printf("\n");
    }
    setf1(0);
    if (setf2) setf2(0);
    return 0;
  }
  const int32_t st0[4] = {pos0, 0, 0, 1};
  CK(hipMemcpy(d.state, st0, sizeof st0, hipMemcpyHostToDevice));
  for (int i = 0; i < steps; ++i) {
    if (stride > 1) {      // a SAMPLE of cache lengths over the whole loop (every K / V row exists: the caches were filled at start) — a counter
      CK(hipStreamSynchronize(st));                                  // pass over all 228 x 76 dispatches takes ~15 min per counter under rocprofv3
      const int32_t p = pos0 + i * stride;
      CK(hipMemcpy(d.state, &p, sizeof p, hipMemcpyHostToDevice));
    }
    const int rc = step(&d, &sp, 1, st);
    if (rc) { fprintf(stderr, "vaura_decode_step: %d\n", rc); return 3; }
  }
  CK(hipStreamSynchronize(st));
  int32_t st1[4];
  CK(hipMemcpy(st1, d.state, sizeof st1, hipMemcpyDeviceToHost));
  const int last = pos0 + (steps - 1) * stride;
  printf("pmc_driver: %d steps, weights %s, rows %d, positions %d..%d stride %d\n", steps, wname(wd), rows, pos0, last, stride);
  return st1[0] == last + 1 ? 0 : 4;
}
