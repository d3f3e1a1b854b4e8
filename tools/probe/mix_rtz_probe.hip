// Does v_fma_mixlo_f16 round its f16 result with MODE.FP_ROUND[3:2] (the f16/f64 field), and does `clamp` give [0,1]?
// The FFN epilogue wants: hi = RTZ_f16(v*sc); hi = max(hi, 0) (packed); lo = clamp(f16(v*sc - hi)) -- 2.5 VALU ops per
// value, relu included, valid only if hi is rounded toward zero (then lo >= 0 for v >= 0 and clamp zeroes it for v < 0).
//   hipcc --offload-arch=gfx950 -O2 -o mix_rtz_probe mix_rtz_probe.hip && ./mix_rtz_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void probe(const float* in, float sc, unsigned* out_hi, unsigned* out_lo, int rtz) {
  const int i = threadIdx.x;
  float v0 = in[2 * i], v1 = in[2 * i + 1];
  unsigned hi = 0, lo = 0;
  if (rtz) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3");
  asm volatile(
      "v_fma_mixlo_f16 %0, %2, %4, 0\n\t"
      "v_fma_mixhi_f16 %0, %3, %4, 0\n\t"
      "v_pk_max_f16 %0, %0, 0\n\t"
      "s_nop 1\n\t"
      "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1] clamp\n\t"
      "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1] clamp\n\t"
      : "+v"(hi), "+v"(lo)
      : "v"(v0), "v"(v1), "s"(sc));
  if (rtz) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 0");
  out_hi[i] = hi;
  out_lo[i] = lo;
}

static float h2f(unsigned short h) {
  _Float16 x;
  __builtin_memcpy(&x, &h, 2);
  return (float)x;
}

int main() {
  const int n = 128;
  std::vector<float> in(n);
  unsigned s = 12345;
  for (int i = 0; i < n; ++i) {
    s = s * 1664525u + 1013904223u;
    in[i] = ((float)(s >> 8) / 16777216.f - 0.4f) * 37.f * (i % 7 == 0 ? 1e-3f : 1.f);
  }
  float *d_in;
  unsigned *d_hi, *d_lo;
  hipMalloc(&d_in, n * 4); hipMalloc(&d_hi, n * 2); hipMalloc(&d_lo, n * 2);
  hipMemcpy(d_in, in.data(), n * 4, hipMemcpyHostToDevice);
  const float sc = 0.125f;
  for (int rtz = 0; rtz < 2; ++rtz) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(n / 2), 0, 0, d_in, sc, d_hi, d_lo, rtz);
    std::vector<unsigned> hi(n / 2), lo(n / 2);
    hipMemcpy(hi.data(), d_hi, n * 2, hipMemcpyDeviceToHost);
    hipMemcpy(lo.data(), d_lo, n * 2, hipMemcpyDeviceToHost);
    double worst = 0, worst_small = 0; int neg_lo = 0, hi_above = 0;
    for (int i = 0; i < n; ++i) {
      const float h = h2f((unsigned short)(hi[i / 2] >> (16 * (i % 2)))), l = h2f((unsigned short)(lo[i / 2] >> (16 * (i % 2))));
      const double want = std::fmax((double)in[i] * sc, 0.0);
      if (l < 0) ++neg_lo;
      if (h > want) ++hi_above;
      // values whose lo half is a normal fp16 (hi >= 2^-3): relative error; smaller ones: absolute error in fp16 denormal steps
      if (want >= 0.125) worst = std::fmax(worst, std::fabs((double)h + l - want) / want);
      else if (want > 0) worst_small = std::fmax(worst_small, std::fabs((double)h + l - want) / std::ldexp(1.0, -24));
      else if (h != 0 || l != 0) worst = 1.0;
    }
    printf("%s: worst relative error of hi+lo vs relu(v*sc) = %.3g (2^-21 = %.3g), small values: %.2f fp16 denormal steps; "
           "hi > exact in %d of %d; negative lo: %d\n",
           rtz ? "MODE[3:2]=RTZ" : "default (RNE)", worst, std::ldexp(1.0, -21), worst_small, hi_above, n, neg_lo);
  }
  return 0;
}
