// Probes the operand layout of v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 operands) with one-hot data:
//   1. which (lane group, byte) of A meets which (lane group, byte) of B  (the k index of every operand byte),
//   2. which output rows / columns and which k a lane's scale operand multiplies.
// hipcc --offload-arch=gfx950 -O2 tools/microbench/mfma_mx8_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// one wave per block; per-lane raw registers from memory: a[blk][lane][8], b likewise, sa/sb [blk][lane]; out[blk][lane][4]
__global__ void k(const int* a, const int* b, const int* sa, const int* sb, float* out) {
  const int l = threadIdx.x; const size_t o = (size_t)blockIdx.x * 64 + l;
  i32x8 av, bv;
  for (int v = 0; v < 8; ++v) { av[v] = a[o * 8 + v]; bv[v] = b[o * 8 + v]; }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc, 0, 0, 0, sa[o], 0, sb[o]);
  for (int i = 0; i < 4; ++i) out[o * 4 + i] = acc[i];
}

int main() {
  // ---- experiment 1: k index of every byte.  Block (gA*32+jA): A one-hot (value 1.0) at byte jA of every lane of group gA;
  // B: byte j of lane group g holds the value 1 + (something identifying (g, j))?  fp8 cannot hold 128 distinct exact
  // values in one product, so two passes: B = 2^(j % 8) on group g only (pass per g and per j / 8).
  const int NB = 128 * 16;   // (gA, jA) x (gB, jhi)
  std::vector<int> a((size_t)NB * 64 * 8, 0), b((size_t)NB * 64 * 8, 0), sa((size_t)NB * 64, 127), sb((size_t)NB * 64, 127);
  auto setbyte = [](std::vector<int>& v, size_t blk, int lane, int byte, uint8_t val) {
    int& w = v[(blk * 64 + lane) * 8 + byte / 4];
    w = (w & ~(0xff << (8 * (byte % 4)))) | ((int)val << (8 * (byte % 4)));
  };
  const uint8_t pow2[8] = {0x38, 0x40, 0x48, 0x50, 0x58, 0x60, 0x68, 0x70};   // 1, 2, 4, ..., 128
  for (int gA = 0; gA < 4; ++gA) for (int jA = 0; jA < 32; ++jA) for (int gB = 0; gB < 4; ++gB) for (int jh = 0; jh < 4; ++jh) {
    const size_t blk = ((size_t)(gA * 32 + jA) * 4 + gB) * 4 + jh;
    for (int r = 0; r < 16; ++r) {
      setbyte(a, blk, gA * 16 + r, jA, 0x38);
      for (int e = 0; e < 8; ++e) setbyte(b, blk, gB * 16 + r, jh * 8 + e, pow2[e]);
    }
  }
  int *da, *db, *dsa, *dsb; float* dout;
  hipMalloc(&da, a.size() * 4); hipMalloc(&db, b.size() * 4); hipMalloc(&dsa, sa.size() * 4); hipMalloc(&dsb, sb.size() * 4);
  hipMalloc(&dout, (size_t)NB * 64 * 4 * 4);
  hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dsa, sa.data(), sa.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), sb.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(NB), dim3(64), 0, 0, da, db, dsa, dsb, dout);
  std::vector<float> out((size_t)NB * 64 * 4);
  hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
  printf("experiment 1: A byte (group, byte) meets B byte (group, byte):\n");
  int same = 0, total = 0;
  for (int gA = 0; gA < 4; ++gA) for (int jA = 0; jA < 32; ++jA) {
    int fg = -1, fj = -1, hits = 0;
    for (int gB = 0; gB < 4; ++gB) for (int jh = 0; jh < 4; ++jh) {
      const size_t blk = ((size_t)(gA * 32 + jA) * 4 + gB) * 4 + jh;
      const float c = out[(blk * 64 + 0) * 4 + 0];   // C[0][0]
      if (c != 0.f) { int e = 0; while ((1 << e) < (int)c) ++e; fg = gB; fj = jh * 8 + e; ++hits; }
    }
    ++total; same += (hits == 1 && fg == gA && fj == jA);
    if (!(hits == 1 && fg == gA && fj == jA)) printf("  A(%d,%2d) -> B(%d,%2d) hits %d\n", gA, jA, fg, fj, hits);
  }
  printf("  %d of %d operand bytes meet the same (group, byte) of the other operand\n", same, total);

  // ---- experiment 2: scales.  All data 1.0 (C = 128 everywhere at scale 1).  Block l (0..63): lane l's A scale = 128 (x2);
  // block 64 + l: lane l's B scale = 128.  Print which C entries changed and to what.
  const int NB2 = 128;
  std::vector<int> a2((size_t)NB2 * 64 * 8, 0x38383838), b2((size_t)NB2 * 64 * 8, 0x38383838), sa2((size_t)NB2 * 64, 127), sb2((size_t)NB2 * 64, 127);
  for (int l = 0; l < 64; ++l) { sa2[(size_t)l * 64 + l] = 128; sb2[(size_t)(64 + l) * 64 + l] = 128; }
  hipMemcpy(da, a2.data(), a2.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db, b2.data(), b2.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dsa, sa2.data(), sa2.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dsb, sb2.data(), sb2.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(NB2), dim3(64), 0, 0, da, db, dsa, dsb, dout);
  hipMemcpy(out.data(), dout, (size_t)NB2 * 64 * 4 * 4, hipMemcpyDeviceToHost);
  for (int which = 0; which < 2; ++which) {
    printf("experiment 2%c: lane l's %c scale doubled -> changed C entries (row = 4*(lane>>4)+reg, col = lane&15):\n", 'a' + which, which ? 'B' : 'A');
    for (int l = 0; l < 64; ++l) {
      const size_t blk = which * 64 + l;
      int nchg = 0, r0 = -1, c0 = -1, r1 = -1, c1 = -1; float val = 0;
      for (int ln = 0; ln < 64; ++ln) for (int i = 0; i < 4; ++i) {
        const float c = out[(blk * 64 + ln) * 4 + i];
        if (c != 128.f) { const int row = 4 * (ln >> 4) + i, col = ln & 15; if (!nchg) { r0 = row; c0 = col; } r1 = row; c1 = col; val = c; ++nchg; }
      }
      if (l < 20 || l % 16 == 0 || l == 63) printf("  lane %2d: %3d entries changed, rows %d..%d cols %d..%d, value %g\n", l, nchg, r0, r1, c0, c1, val);
    }
  }
  // ---- experiment 3: which k a lane's A scale covers: A = 1 everywhere, B one-hot on (group gB, byte 0 and byte 16), lane 0's A scale doubled
  printf("experiment 3: lane (row 0, group g)'s A scale doubled; B one-hot at (group gB, byte jB) -> C[0][0] (1 = not covered, 2 = covered)\n");
  const int NB3 = 4 * 4 * 32;
  std::vector<int> a3((size_t)NB3 * 64 * 8, 0x38383838), b3((size_t)NB3 * 64 * 8, 0), sa3((size_t)NB3 * 64, 127), sb3((size_t)NB3 * 64, 127);
  for (int g = 0; g < 4; ++g) for (int gB = 0; gB < 4; ++gB) for (int jB = 0; jB < 32; ++jB) {
    const size_t blk = ((size_t)g * 4 + gB) * 32 + jB;
    sa3[blk * 64 + g * 16] = 128;
    for (int r = 0; r < 16; ++r) setbyte(b3, blk, gB * 16 + r, jB, 0x38);
  }
  hipFree(da); hipFree(db); hipFree(dsa); hipFree(dsb); hipFree(dout);
  hipMalloc(&da, a3.size() * 4); hipMalloc(&db, b3.size() * 4); hipMalloc(&dsa, sa3.size() * 4); hipMalloc(&dsb, sb3.size() * 4);
  hipMalloc(&dout, (size_t)NB3 * 64 * 4 * 4);
  hipMemcpy(da, a3.data(), a3.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db, b3.data(), b3.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dsa, sa3.data(), sa3.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dsb, sb3.data(), sb3.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(NB3), dim3(64), 0, 0, da, db, dsa, dsb, dout);
  out.resize((size_t)NB3 * 64 * 4);
  hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
  for (int g = 0; g < 4; ++g) {
    printf("  scale lane group %d covers:", g);
    for (int gB = 0; gB < 4; ++gB) {
      int lo = -1, hi = -1, n = 0;
      for (int jB = 0; jB < 32; ++jB) if (out[((((size_t)g * 4 + gB) * 32 + jB) * 64) * 4] == 2.f) { if (lo < 0) lo = jB; hi = jB; ++n; }
      if (n) printf(" B group %d bytes %d..%d (%d)", gB, lo, hi, n);
    }
    printf("\n");
  }
  return 0;
}
