// Stand-alone driver of the PRODUCT library's MFMA-bound stages for counter collection (rocprofv3 --pmc crashes under python on
// this image):   rocprofv3 --kernel-trace --pmc <counters> -- ./mfma_driver <libvaura_hip.so> codec|avclip [clips] [precision]
//
//   codec    vaura_dac_decode on the DAC-44k geometry (1536 -> 96 channels, rates 8,8,4,2, 9 codebooks), `clips` x 220 frames,
//            codec precision 0 fp32-MFMA | 1 fp16 pairs (default) | 2 one-plane weights | 4 single fp16 plane ("f16")
//   avclip   vaura_avclip_forward on ViT-B/16 divided space-time, `clips` x 4 segments of 16 x 224 x 224 frames
//   prefill  the four GEMMs of one teacher-forced prompt pass of the sliding-window caller (vaura_gemv_pair at 166 positions x
//            2 `clips` rows: qkv, wo, w1||w3, w2 of one layer), `precision` = weight storage 3 one fp16 plane | 4 two planes
//
// Counters do not depend on the VALUES, so every weight / bias / alpha pointer of the descriptors points into one buffer of small
// finite noise (fp16 values |v| < 0.05; read as fp32 they are finite too): no checkpoint, no packing code to keep in sync.  Shapes,
// strides, layouts and launch geometry are exactly the product's: the descriptors are filled like vaura_amd/engine.py fills them.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../include/vaura_hip.h"

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e__)); exit(2); } } while (0)

__global__ void fill_noise(_Float16* p, size_t n, uint32_t seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (_Float16)(0.05f * ((float)(h & 0xffff) / 32768.0f - 1.0f));
  }
}
__global__ void fill_f32(float* p, size_t n, float scale, uint32_t seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = scale * ((float)(h & 0xffff) / 32768.0f - 1.0f);
  }
}
__global__ void fill_codes(int32_t* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (int32_t)((i * 2654435761u >> 7) & 1023u);
}

static char* g_noise = nullptr;
static size_t g_noise_bytes = 0, g_cursor = 0;
static const void* noise(size_t bytes) {      // a (reused when exhausted) slice of the noise buffer, 256-byte aligned
  bytes = (bytes + 255) & ~(size_t)255;
  if (g_cursor + bytes > g_noise_bytes) g_cursor = 0;
  if (bytes > g_noise_bytes) { fprintf(stderr, "noise buffer too small for %zu bytes\n", bytes); exit(2); }
  const void* p = g_noise + g_cursor;
  g_cursor += bytes;
  return p;
}
static const float* ones(size_t n) {          // Snake alphas / LayerNorm gains: O(1) values
  float* p; CK(hipMalloc(&p, n * 4));
  fill_f32<<<64, 256>>>(p, n, 0.25f, 77u);
  std::vector<float> h(n, 1.0f);
  CK(hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice));
  return p;
}

static void conv(vaura_conv& cv, int cin, int cout, int taps, int dil, int stride) {
  cv.cin = cin; cv.cout = cout; cv.taps = taps; cv.dilation = dil; cv.stride = stride; cv._pad = 0;
  const size_t elems = (size_t)(stride > 1 ? stride * 2 : taps) * cout * cin;
  cv.w = (const float*)noise(elems * 4);       // fp32 [taps][Cout][Cin] or the pair layout: 4 bytes per weight either way
  cv.bias = (const float*)noise((size_t)cout * 4);
  cv.wscale = nullptr;
}

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s <libvaura_hip.so> codec|avclip|prefill [clips] [precision] [repeats]\n", argv[0]); return 1; }
  const bool codec = !strcmp(argv[2], "codec"), prefill = !strcmp(argv[2], "prefill");
  const int clips = argc > 3 ? atoi(argv[3]) : 8;
  const int precision = argc > 4 ? atoi(argv[4]) : 1;
  const int repeats = argc > 5 ? atoi(argv[5]) : 3;
  void* lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
  auto ssize = (size_t (*)(int))dlsym(lib, "vaura_struct_size");
  if (!ssize || ssize(5) != sizeof(vaura_codec) || ssize(7) != sizeof(vaura_vit) || ssize(8) != sizeof(vaura_vit_block)) { fprintf(stderr, "library / header mismatch\n"); return 1; }
  g_noise_bytes = (size_t)512 << 20;
  CK(hipMalloc(&g_noise, g_noise_bytes));
  fill_noise<<<2048, 256>>>((_Float16*)g_noise, g_noise_bytes / 2, 12345u);
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms = 0;
  if (prefill) {
    typedef int (*gemv_pair_t)(const void*, int, const uint16_t*, const float*, int, const float*, float*, float*, uint16_t*, const float*,
                               float*, int64_t, int64_t, int64_t, int, float, vaura_stream_t);
    auto gp = (gemv_pair_t)dlsym(lib, "vaura_gemv_pair");
    auto wbytes = (size_t (*)(int64_t, int64_t, int))dlsym(lib, "vaura_packed_weight_bytes");
    if (!gp || !wbytes) return 1;
    const int wd = (argc > 4 && atoi(argv[4]) == 3) ? VAURA_W_H1 : VAURA_W_H2;
    const int64_t rows = (int64_t)166 * 2 * clips, rp = (rows + 15) / 16 * 16, D = 1536, F = 4096;
    struct G { const char* name; int64_t N, K; int epi; bool norm; } gs[4] = {
        {"qkv", 3 * D, D, 0 /* store */, true}, {"wo", D, D, 1 /* + residual */, false}, {"w13", 2 * F, D, 2 /* SwiGLU */, true}, {"w2", D, F, 1 /* + residual */, false}};
    // packed weights = noise halves followed by N power-of-two row scales; planes = noise halves; sums of squares positive
    float* ss; CK(hipMalloc(&ss, (size_t)(rp / 16) * (F / 16) * 16 * 4)); fill_f32<<<256, 256>>>(ss, (size_t)(rp / 16) * (F / 16) * 16, 0.0f, 1u);
    {
      std::vector<float> one((size_t)(rp / 16) * (F / 16) * 16, 16.0f);
      CK(hipMemcpy(ss, one.data(), one.size() * 4, hipMemcpyHostToDevice));
    }
    float *res, *out, *oss; uint16_t* osp;
    CK(hipMalloc(&res, (size_t)rp * F * 4)); fill_f32<<<2048, 256>>>(res, (size_t)rp * F, 1.0f, 21u);
    CK(hipMalloc(&out, (size_t)rp * 3 * D * 4)); CK(hipMalloc(&osp, (size_t)rp * 2 * 3 * D * 2)); CK(hipMalloc(&oss, (size_t)(rp / 16) * (3 * D / 16) * 16 * 4));
    const float* gout = ones(3 * D);
    void* wp[4];
    for (int i = 0; i < 4; ++i) {
      const size_t wb = wbytes(gs[i].N, gs[i].K, wd);
      CK(hipMalloc(&wp[i], wb));
      fill_noise<<<2048, 256>>>((_Float16*)wp[i], (wb - (size_t)gs[i].N * 4) / 2, 100u + i);
      std::vector<float> sc((size_t)gs[i].N, 1.0f / 8192.0f);
      CK(hipMemcpy((char*)wp[i] + wb - (size_t)gs[i].N * 4, sc.data(), sc.size() * 4, hipMemcpyHostToDevice));
    }
    const uint16_t* xs = (const uint16_t*)noise((size_t)rp * 2 * F * 2);
    CK(hipDeviceSynchronize());
    for (int r = 0; r < repeats; ++r) {
      if (r == repeats - 1) CK(hipEventRecord(e0, st));
      for (int i = 0; i < 4; ++i) {
        const bool sw = gs[i].epi == 2 /* SwiGLU */, rs = gs[i].epi == 1 /* + residual */;
        const int rc = gp(wp[i], wd, xs, gs[i].norm ? ss : nullptr, gs[i].norm ? (int)(gs[i].K / 16) : 0, rs ? res : nullptr, sw ? nullptr : out, nullptr, osp,
                          gout, rs ? oss : nullptr, rows, gs[i].N, gs[i].K, gs[i].epi, 1e-5f, st);
        if (rc) { fprintf(stderr, "vaura_gemv_pair(%s): %d\n", gs[i].name, rc); return 3; }
      }
    }
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<float> h(1024);
    CK(hipMemcpy(h.data(), out, 4096, hipMemcpyDeviceToHost));
    bool finite = true;
    for (float v : h) finite = finite && (v == v) && v > -1e30f && v < 1e30f;
    double fl = 0;
    for (int i = 0; i < 4; ++i) fl += 2.0 * rows * gs[i].N * gs[i].K;
    printf("mfma_driver prefill: %lld rows, weights %s, one layer's four GEMMs %.1f us (%.1f TF-equiv), output finite: %d\n", (long long)rows,
           wd == VAURA_W_H1 ? "h1" : "h2", ms * 1e3, fl / (ms * 1e-3) / 1e12, (int)finite);
    return finite ? 0 : 4;
  }
  if (codec) {
    auto decode = (int (*)(const vaura_codec*, const int32_t*, int, int, float*, vaura_stream_t))dlsym(lib, "vaura_dac_decode");
    auto wse = (size_t (*)(const vaura_codec*, int, int))dlsym(lib, "vaura_dac_workspace_elems");
    if (!decode || !wse) return 1;
    const int T = 220, rates[4] = {8, 8, 4, 2}, dils[3] = {1, 3, 9};
    vaura_codec c;
    memset(&c, 0, sizeof c);
    c.n_codebooks = 9; c.codebook_size = 1024; c.codebook_dim = 8; c.latent_dim = 1024; c.n_blocks = 4; c.n_units = 3;
    c.precision = precision;
    float* cb; CK(hipMalloc(&cb, (size_t)9 * 1024 * 8 * 4)); fill_f32<<<64, 256>>>(cb, (size_t)9 * 1024 * 8, 0.5f, 3u);
    c.codebooks = cb;
    float* opw; CK(hipMalloc(&opw, (size_t)9 * 1024 * 8 * 4)); fill_f32<<<64, 256>>>(opw, (size_t)9 * 1024 * 8, 0.2f, 4u);
    c.out_proj_w = opw; c.out_proj_b = (const float*)noise(9 * 1024 * 4);
    int C = 1536;
    conv(c.conv_in, 1024, C, 7, 1, 1);
    for (int b = 0; b < 4; ++b) {
      c.rates[b] = rates[b];
      c.alpha_up[b] = ones(C);
      conv(c.up[b], C, C / 2, 2, 1, rates[b]);
      C /= 2;
      for (int u = 0; u < 3; ++u) {
        c.alpha_res[b][u][0] = ones(C); conv(c.res[b][u][0], C, C, 7, dils[u], 1);
        c.alpha_res[b][u][1] = ones(C); conv(c.res[b][u][1], C, C, 1, 1, 1);
      }
    }
    c.alpha_out = ones(C);
    conv(c.conv_out, C, 1, 7, 1, 1);
    float* wout; CK(hipMalloc(&wout, (size_t)7 * C * 4)); fill_f32<<<8, 256>>>(wout, (size_t)7 * C, 0.1f, 5u);
    c.conv_out.w = wout;
    c.ws_elems = wse(&c, clips, T);
    for (int i = 0; i < 4; ++i) CK(hipMalloc(&c.ws[i], c.ws_elems * 4));
    int32_t* codes; CK(hipMalloc(&codes, (size_t)clips * 9 * T * 4)); fill_codes<<<64, 256>>>(codes, (size_t)clips * 9 * T);
    float* wav; CK(hipMalloc(&wav, (size_t)clips * T * 512 * 4));
    CK(hipDeviceSynchronize());
    for (int r = 0; r < repeats; ++r) {
      if (r == repeats - 1) CK(hipEventRecord(e0, st));
      const int rc = decode(&c, codes, clips, T, wav, st);
      if (rc) { fprintf(stderr, "vaura_dac_decode: %d\n", rc); return 3; }
    }
    if (const char* so = getenv("MFMA_STAMPS")) {
      // <lib> = the diagnostic build (python -m vaura_amd.csrc.build --stamps): ONE more decode with the conv kernels' in-kernel
      // s_memrealtime stamps collected (csrc/common.h; wave 0 of every workgroup, every wave with PMC_STAMP_ALL_WAVES) -> file
      auto set_d = (int (*)(unsigned long long*))dlsym(lib, "vaura_stamps_set_dac");
      if (!set_d) { fprintf(stderr, "%s is not a stamps build\n", argv[1]); return 1; }
      CK(hipStreamSynchronize(st));
      const size_t cap = 1500000;
      unsigned long long* buf; CK(hipMalloc(&buf, (8 + cap * 16) * 8)); CK(hipMemset(buf, 0, (8 + cap * 16) * 8));
      unsigned long long hdr[3] = {0, cap, getenv("PMC_STAMP_ALL_WAVES") ? 1ull : 0ull};
      CK(hipMemcpy(buf, hdr, sizeof hdr, hipMemcpyHostToDevice));
      if (set_d(buf)) return 3;
      if (decode(&c, codes, clips, T, wav, st)) return 3;
      CK(hipStreamSynchronize(st));
      std::vector<unsigned long long> host(8 + cap * 16);
      CK(hipMemcpy(host.data(), buf, host.size() * 8, hipMemcpyDeviceToHost));
      const size_t n = std::min<size_t>(host[0], cap);
      FILE* f = fopen(so, "wb");
      if (!f) { perror(so); return 1; }
      fwrite(host.data() + 8, 128, n, f);
      fclose(f);
      printf("stamps: %zu records -> %s\n", n, so);
      if (set_d(nullptr)) return 3;
    }
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<float> h(1024);
    CK(hipMemcpy(h.data(), wav, 4096, hipMemcpyDeviceToHost));
    bool finite = true;
    for (float v : h) finite = finite && (v == v) && v > -1e30f && v < 1e30f;
    printf("mfma_driver codec: %d clips x %d frames, precision %d, last decode %.3f ms (%.1f TFLOP/s of the 353.8 GFLOP per clip), output finite: %d\n",
           clips, T, precision, ms, clips * 353.8e9 / (ms * 1e-3) / 1e12, (int)finite);
    return finite ? 0 : 4;
  }
  auto fwd = (int (*)(const vaura_vit*, const float*, int, float*, vaura_stream_t))dlsym(lib, "vaura_avclip_forward");
  auto wsb = (size_t (*)(const vaura_vit*, int, int))dlsym(lib, "vaura_avclip_workspace_bytes");
  if (!fwd || !wsb) return 1;
  const int D = 768, Hd = 3072, depth = 12, n_seg = clips * 4;
  vaura_vit v;
  memset(&v, 0, sizeof v);
  v.depth = depth; v.dim = D; v.heads = 12; v.hidden = Hd; v.n_patches = 196; v.n_frames = 8;
  v.in_chans = 3; v.frames = 16; v.img = 224; v.patch = 16; v.patch_t = 2; v.patch_k = 1536; v.eps = 1e-6f;
  auto lin = [&](int out, int in) { return noise((size_t)out * in * 4); };
  auto vec = [&](int n) { return (const float*)noise((size_t)n * 4); };
  v.pe_w = lin(D, 1536); v.pe_b = vec(D);
  v.cls_token = vec(D); v.pos_embed = vec(197 * D); v.temp_embed = vec(8 * D);
  std::vector<vaura_vit_block> blocks(depth);
  for (auto& b : blocks) {
    b.ln1_w = ones(D); b.ln1_b = vec(D); b.ln2_w = ones(D); b.ln2_b = vec(D); b.ln3_w = ones(D); b.ln3_b = vec(D);
    for (vaura_vit_attn* a : {&b.space, &b.time}) { a->qkv_w = lin(3 * D, D); a->qkv_b = vec(3 * D); a->proj_w = lin(D, D); a->proj_b = vec(D); }
    b.fc1_w = lin(Hd, D); b.fc1_b = vec(Hd); b.fc2_w = lin(D, Hd); b.fc2_b = vec(D);
  }
  v.blocks_host = blocks.data();
  v.norm_w = ones(D); v.norm_b = vec(D); v.agg_cls = vec(D);
  v.agg_ln1_w = ones(D); v.agg_ln1_b = vec(D); v.agg_ln2_w = ones(D); v.agg_ln2_b = vec(D);
  v.agg_in_w = lin(3 * D, D); v.agg_in_b = vec(3 * D); v.agg_out_w = lin(D, D); v.agg_out_b = vec(D);
  v.agg_l1_w = lin(Hd, D); v.agg_l1_b = vec(Hd); v.agg_l2_w = lin(D, Hd); v.agg_l2_b = vec(D);
  void** wsp[7] = {(void**)&v.ws_x, (void**)&v.ws_qkv, (void**)&v.ws_a, (void**)&v.ws_h, (void**)&v.ws_p, (void**)&v.ws_z, (void**)&v.ws_s};
  for (int i = 0; i < 7; ++i) CK(hipMalloc(wsp[i], wsb(&v, n_seg, i)));
  const size_t fr = (size_t)n_seg * 3 * 16 * 224 * 224;
  float* frames; CK(hipMalloc(&frames, fr * 4)); fill_f32<<<2048, 256>>>(frames, fr, 1.0f, 9u);
  float* feats; CK(hipMalloc(&feats, (size_t)n_seg * 8 * D * 4));
  CK(hipDeviceSynchronize());
  for (int r = 0; r < repeats; ++r) {
    if (r == repeats - 1) CK(hipEventRecord(e0, st));
    const int rc = fwd(&v, frames, n_seg, feats, st);
    if (rc) { fprintf(stderr, "vaura_avclip_forward: %d\n", rc); return 3; }
  }
  CK(hipEventRecord(e1, st));
  CK(hipStreamSynchronize(st));
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<float> h(1024);
  CK(hipMemcpy(h.data(), feats, 4096, hipMemcpyDeviceToHost));
  bool finite = true;
  for (float x : h) finite = finite && (x == x) && x > -1e30f && x < 1e30f;
  printf("mfma_driver avclip: %d clips x 4 segments, last forward %.3f ms (%.1f TFLOP/s of the 355 GFLOP of linears per segment), output finite: %d\n",
         clips, ms, n_seg * 355e9 / (ms * 1e-3) / 1e12, (int)finite);
  return finite ? 0 : 4;
}
