// v_permlane16_swap / v_permlane32_swap (gfx950) as the split-fp16 dense kernel uses them: with both operands the same
// register, max / sum of the two results is the reduction over lanes l ^ 16 (resp. l ^ 32).  Prints PASS / FAIL.
//   hipcc --offload-arch=gfx950 -O2 -o permlane_swap_probe permlane_swap_probe.hip && ./permlane_swap_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

__device__ __forceinline__ void swap16(float& a, float& b) { asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void swap32(float& a, float& b) { asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }

// the same through inline asm (what the kernel uses)
__global__ void probe_asm(const float* in, float* out16, float* out32, float* quad_max, float* quad_sum) {
  const int l = threadIdx.x;
  float a = in[l], b = in[l];
  swap16(a, b);
  out16[2 * l] = a; out16[2 * l + 1] = b;
  float c = in[l], d = in[l];
  swap32(c, d);
  out32[2 * l] = c; out32[2 * l + 1] = d;
  float m0 = fmaxf(a, b), m1 = m0;
  swap32(m0, m1);
  quad_max[l] = fmaxf(m0, m1);
  float s0 = a + b, s1 = s0;
  swap32(s0, s1);
  quad_sum[l] = s0 + s1;
}

__global__ void probe(const float* in, float* out16, float* out32, float* quad_max, float* quad_sum) {
  const int l = threadIdx.x;
  float v = in[l];
  unsigned u = __builtin_bit_cast(unsigned, v);
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  out16[2 * l] = __builtin_bit_cast(float, r[0]);
  out16[2 * l + 1] = __builtin_bit_cast(float, r[1]);
  auto q = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  out32[2 * l] = __builtin_bit_cast(float, q[0]);
  out32[2 * l + 1] = __builtin_bit_cast(float, q[1]);
  float m = fmaxf(__builtin_bit_cast(float, r[0]), __builtin_bit_cast(float, r[1]));
  unsigned um = __builtin_bit_cast(unsigned, m);
  auto q2 = __builtin_amdgcn_permlane32_swap(um, um, false, false);
  quad_max[l] = fmaxf(__builtin_bit_cast(float, q2[0]), __builtin_bit_cast(float, q2[1]));
  float s = __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
  unsigned us = __builtin_bit_cast(unsigned, s);
  auto q3 = __builtin_amdgcn_permlane32_swap(us, us, false, false);
  quad_sum[l] = __builtin_bit_cast(float, q3[0]) + __builtin_bit_cast(float, q3[1]);
}

int main() {
  float h[64], *d, *o16, *o32, *qm, *qs;
  for (int i = 0; i < 64; ++i) h[i] = (float)((i * 37) % 64) + 0.25f * (i % 3);
  hipMalloc(&d, 256); hipMalloc(&o16, 512); hipMalloc(&o32, 512); hipMalloc(&qm, 256); hipMalloc(&qs, 256);
  hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  for (int variant = 0; variant < 2; ++variant) {
  if (variant == 0) hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o16, o32, qm, qs);
  else hipLaunchKernelGGL(probe_asm, dim3(1), dim3(64), 0, 0, d, o16, o32, qm, qs);
  float r16[128], r32[128], m[64], s[64];
  hipMemcpy(r16, o16, 512, hipMemcpyDeviceToHost); hipMemcpy(r32, o32, 512, hipMemcpyDeviceToHost);
  hipMemcpy(m, qm, 256, hipMemcpyDeviceToHost); hipMemcpy(s, qs, 256, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    // {r0, r1} must be {in[l], in[l ^ 16]} in some order, likewise for 32
    const bool ok16 = (r16[2 * l] == h[l] && r16[2 * l + 1] == h[l ^ 16]) || (r16[2 * l] == h[l ^ 16] && r16[2 * l + 1] == h[l]);
    const bool ok32 = (r32[2 * l] == h[l] && r32[2 * l + 1] == h[l ^ 32]) || (r32[2 * l] == h[l ^ 32] && r32[2 * l + 1] == h[l]);
    float em = fmaxf(fmaxf(h[l & 15], h[(l & 15) + 16]), fmaxf(h[(l & 15) + 32], h[(l & 15) + 48]));
    float es = (h[l & 15] + h[(l & 15) + 16]) + (h[(l & 15) + 32] + h[(l & 15) + 48]);
    const bool okq = m[l] == em && fabsf(s[l] - es) < 1e-4f;
    if (!(ok16 && ok32 && okq)) { ++bad; if (bad < 6) printf("lane %d: 16 {%g %g} 32 {%g %g} max %g (%g) sum %g (%g)\n", l, r16[2*l], r16[2*l+1], r32[2*l], r32[2*l+1], m[l], em, s[l], es); }
  }
  printf("%s: %s\n", variant ? "inline asm" : "builtin (same value for both operands)", bad ? "FAIL" : "PASS");
  }
  return 0;
}
