// Checks, with exact integer data, the operand layout conv_mx8_kernel (csrc/dac.hip) assumes for
// v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands:
//   lane l = (r = l&15, G = l>>4) holds A[row r][k] and B[k][col r] with k = 16G + j for bytes j = 0..15 of its 8 VGPRs and
//   k = 64 + 16G + (j - 16) for bytes 16..31 (found with tools/microbench/mfma_mx8_probe.hip),
//   its scale operand (E8M0 in byte 0, op_sel 0) multiplies k = 32G .. 32G+31 of its row / column,
//   C/D: col = l&15, row = 4*(l>>4) + reg.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/microbench/mfma_mx8_layout.hip -o /tmp/mx8 && /tmp/mx8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdint>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k(const uint8_t* A, const uint8_t* B, const uint8_t* sa, const uint8_t* sb, float* C) {
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  i32x8 a, b;
  for (int v = 0; v < 8; ++v) {
    int wa = 0, wb = 0;
    for (int e = 0; e < 4; ++e) {
      const int j = 4 * v + e, kk = j < 16 ? 16 * g + j : 64 + 16 * g + (j - 16);
      wa |= (int)A[r * 128 + kk] << (8 * e);
      wb |= (int)B[kk * 16 + r] << (8 * e);
    }
    a[v] = wa; b[v] = wb;
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 0, (int)sa[r * 4 + g], 0, (int)sb[r * 4 + g]);
  for (int i = 0; i < 4; ++i) C[(4 * g + i) * 16 + r] = acc[i];
}

static uint8_t enc(int v) {   // e4m3 of a small integer
  static const uint8_t tab[9] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4a, 0x4c, 0x4e, 0x50};
  return v < 0 ? (uint8_t)(0x80 | tab[-v]) : tab[v];
}

int main() {
  uint8_t hA[16 * 128], hB[128 * 16], hsa[64], hsb[64];
  int iA[16 * 128], iB[128 * 16];
  srand(7);
  for (int i = 0; i < 16 * 128; ++i) { iA[i] = rand() % 17 - 8; hA[i] = enc(iA[i]); iB[i] = rand() % 17 - 8; hB[i] = enc(iB[i]); }
  for (int i = 0; i < 64; ++i) { hsa[i] = 124 + rand() % 7; hsb[i] = 124 + rand() % 7; }
  uint8_t *dA, *dB, *dsa, *dsb; float* dC;
  (void)hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dsa, 64); hipMalloc(&dsb, 64); hipMalloc(&dC, 256 * 4);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipMemcpy(dsa, hsa, 64, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 64, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dC);
  float hC[256];
  hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double ref = 0;
      for (int g = 0; g < 4; ++g) {
        long s = 0;
        for (int kk = 32 * g; kk < 32 * g + 32; ++kk) s += (long)iA[i * 128 + kk] * iB[kk * 16 + j];
        ref += std::ldexp((double)s, (int)hsa[i * 4 + g] - 127 + (int)hsb[j * 4 + g] - 127);
      }
      if ((double)hC[i * 16 + j] != ref) { if (bad < 5) printf("C[%d][%d] = %g, expected %g\n", i, j, hC[i * 16 + j], ref); ++bad; }
    }
  printf("mfma_scale 16x16x128 e4m3 layout + per-lane block scales: %s (%d mismatches)\n", bad ? "MISMATCH" : "as assumed", bad);
  return bad != 0;
}
