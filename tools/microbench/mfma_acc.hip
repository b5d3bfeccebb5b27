// How exact is v_mfma_f32_16x16x32_bf16's accumulation?  (development probe)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <cstdint>
#include <cstring>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d line %d\n", (int)e, __LINE__); exit(1); } } while (0)

// A: (16 x K) bf16 row-major, B: (16 x K) bf16 row-major (i.e. B^T), out (mode, 16, 16)
__global__ void probe(const uint16_t* A, const uint16_t* B, float* out, int K) {
  const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
  f32x4 chain = {0, 0, 0, 0}, fresh = {0, 0, 0, 0}, pair = {0, 0, 0, 0};
  for (int k0 = 0; k0 < K; k0 += 32) {
    u32x4 a = *reinterpret_cast<const u32x4*>(A + (size_t)r * K + k0 + 8 * q);
    u32x4 b = *reinterpret_cast<const u32x4*>(B + (size_t)r * K + k0 + 8 * q);
    bf16x8 af = __builtin_bit_cast(bf16x8, a), bf = __builtin_bit_cast(bf16x8, b);
    chain = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, chain, 0, 0, 0);
    f32x4 z = {0, 0, 0, 0};
    z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, z, 0, 0, 0);
    fresh += z;
  }
  for (int i = 0; i < 4; ++i) {
    const int row = 4 * q + i, col = r;
    out[0 * 256 + row * 16 + col] = chain[i];
    out[1 * 256 + row * 16 + col] = fresh[i];
  }
}

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
  const int K = 1536;
  std::vector<uint16_t> A(16 * K), B(16 * K);
  srand(1);
  for (auto& v : A) v = f2bf(((rand() / (float)RAND_MAX) - 0.5f) * 0.08f);
  for (auto& v : B) v = f2bf(((rand() / (float)RAND_MAX) - 0.5f) * 4.0f);
  uint16_t *dA, *dB; float* dO;
  CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, B.size() * 2)); CK(hipMalloc(&dO, 3 * 256 * 4));
  CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dO, K);
  std::vector<float> O(3 * 256);
  CK(hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost));
  double e_chain = 0, e_fresh = 0, e_f32 = 0, mag = 0, bias_chain = 0, bias_fresh = 0;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double ref = 0; float f32 = 0.f;
      for (int k = 0; k < K; ++k) {
        ref += (double)bf2f(A[i * K + k]) * (double)bf2f(B[j * K + k]);
        f32 = fmaf(bf2f(A[i * K + k]), bf2f(B[j * K + k]), f32);
      }
      mag = fmax(mag, fabs(ref));
      e_chain = fmax(e_chain, fabs(O[i * 16 + j] - ref));
      e_fresh = fmax(e_fresh, fabs(O[256 + i * 16 + j] - ref));
      e_f32 = fmax(e_f32, fabs(f32 - ref));
      bias_chain += O[i * 16 + j] - ref; bias_fresh += O[256 + i * 16 + j] - ref;
    }
  printf("max |ref| %.4f\n", mag);
  printf("chained MFMA accumulate : max err %.3e  mean signed %.3e\n", e_chain, bias_chain / 256);
  printf("fresh C=0 + fp32 add    : max err %.3e  mean signed %.3e\n", e_fresh, bias_fresh / 256);
  printf("fp32 fmaf chain (CPU)   : max err %.3e\n", e_f32);
  return 0;
}
