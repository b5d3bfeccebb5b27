// What do the VALU lane exchanges of csrc/common.h return on this chip?  Every lane holds its own id; the probe prints, per helper,
// whether lane i received lane (i ^ k).   hipcc -O3 --offload-arch=gfx950 -I../../vaura_amd/csrc -I../../include lane_exchange_probe.hip
#include "common.h"
#include <cstdio>
__global__ void probe(float* out) {
  const float x = (float)threadIdx.x;
  out[threadIdx.x] = va_xor16(x);
  out[64 + threadIdx.x] = va_xor32(x);
  out[128 + threadIdx.x] = va_dpp<VA_DPP_XOR1>(x);
  out[192 + threadIdx.x] = va_dpp<VA_DPP_XOR2>(x);
  out[256 + threadIdx.x] = va_dpp<VA_DPP_HALF_MIRROR>(x);
  out[320 + threadIdx.x] = va_dpp<VA_DPP_ROR8>(x);
  out[384 + threadIdx.x] = wave_sum(x);
  out[448 + threadIdx.x] = wave_max(x);
  // raw semantics with DISTINCT operands: first = lane, second = lane + 100
  const auto r = __builtin_amdgcn_permlane16_swap(threadIdx.x, threadIdx.x + 100u, false, false);
  out[512 + threadIdx.x] = (float)r[0];
  out[576 + threadIdx.x] = (float)r[1];
}
int main() {
  float* d; hipMalloc(&d, 640 * 4);
  probe<<<1, 64>>>(d);
  float h[640]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char* names[] = {"xor16", "xor32", "dpp xor1", "dpp xor2", "half_mirror (i^7)", "ror8 (i^8)"};
  const int k[] = {16, 32, 1, 2, 7, 8};
  int bad = 0;
  for (int t = 0; t < 6; ++t) {
    int ok = 1;
    for (int i = 0; i < 64; ++i) ok &= (int)h[t * 64 + i] == (i ^ k[t]);
    printf("%-20s %s   lanes 0,1,8,16,17,32,48: %g %g %g %g %g %g %g\n", names[t], ok ? "ok" : "WRONG", h[t*64], h[t*64+1], h[t*64+8], h[t*64+16], h[t*64+17], h[t*64+32], h[t*64+48]);
    bad += !ok;
  }
  printf("wave_sum %g (expect 2016)  wave_max %g (expect 63)\n", h[384], h[448]);
  printf("permlane16_swap(lane, lane + 100): r[0] lanes 0,16,32,48: %g %g %g %g   r[1]: %g %g %g %g\n", h[512], h[528], h[544], h[560], h[576], h[592], h[608], h[624]);
  return bad;
}
