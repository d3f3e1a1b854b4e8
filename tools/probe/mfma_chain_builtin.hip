// Does hipcc insert the wait states a mixed K=32 / K=16 dependent MFMA chain needs on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void chain1(float* out) {  // one chain
  h8 a8, b8; h4 a4, b4;
  for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)1.0f; b8[i] = (_Float16)1.0f; }
  for (int i = 0; i < 4; ++i) { a4[i] = (_Float16)1.0f; b4[i] = (_Float16)1.0f; }
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc, 0, 0, 0);
  out[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];  // 4 * 96
}
__global__ void chain2(float* out) {  // two interleaved chains
  h8 a8, b8; h4 a4, b4;
  for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)1.0f; b8[i] = (_Float16)1.0f; }
  for (int i = 0; i < 4; ++i) { a4[i] = (_Float16)1.0f; b4[i] = (_Float16)1.0f; }
  f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc0, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc1, 0, 0, 0);
  acc0 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc0, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc1, 0, 0, 0);
  acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc0, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc1, 0, 0, 0);
  out[threadIdx.x] = acc0[0] + acc1[3];  // 160
}
int main() {
  float* dev; hipMalloc(&dev, 256);
  std::vector<float> h(64);
  hipLaunchKernelGGL(chain1, dim3(1), dim3(64), 0, 0, dev);
  hipMemcpy(h.data(), dev, 256, hipMemcpyDeviceToHost);
  printf("chain1: got %.0f expect 384\n", h[5]);
  hipLaunchKernelGGL(chain2, dim3(1), dim3(64), 0, 0, dev);
  hipMemcpy(h.data(), dev, 256, hipMemcpyDeviceToHost);
  printf("chain2: got %.0f expect 160\n", h[5]);
  return 0;
}
