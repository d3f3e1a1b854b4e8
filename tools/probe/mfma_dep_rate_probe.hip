// Hardware probe (gfx950): issue rate of v_mfma_f32_16x16x32_f16 as a function of how many INDEPENDENT accumulators the instruction
// stream rotates through - 1 (every MFMA depends on the one before), 2, 4, 8, 16 - with 1 and 2 waves per SIMD.  Cycles per MFMA per
// SIMD from s_memtime (100 MHz constant clock -> ns; printed as ns per MFMA and, with the measured shader clock, cycles).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_dep_rate tools/probe/mfma_dep_rate_probe.hip && /tmp/mfma_dep_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
#include <vector>

#define M(c) "v_mfma_f32_16x16x32_f16 a[" #c ":" #c "+3], v[0:3], v[4:7], a[" #c ":" #c "+3]\n\t"
// 48 MFMAs rotating through C accumulators
#define ROT1 M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0)
#define ROT2 M(0) M(4) M(0) M(4) M(0) M(4) M(0) M(4) M(0) M(4) M(0) M(4) M(0) M(4) M(0) M(4)
#define ROT4 M(0) M(4) M(8) M(12) M(0) M(4) M(8) M(12) M(0) M(4) M(8) M(12) M(0) M(4) M(8) M(12)
#define ROT8 M(0) M(4) M(8) M(12) M(16) M(20) M(24) M(28) M(0) M(4) M(8) M(12) M(16) M(20) M(24) M(28)
#define ROT16 M(0) M(4) M(8) M(12) M(16) M(20) M(24) M(28) M(32) M(36) M(40) M(44) M(48) M(52) M(56) M(60)
// the pattern of attend_fold_h3_kernel's inner loop: two accumulators, three dependent products each, interleaved, eight times
#define CLOB "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", \
  "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30",   \
  "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49",   \
  "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "memory"

template <int C>
__global__ void __launch_bounds__(256) probe(unsigned long long* out, int iters) {
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (C == 1) asm volatile(ROT1 ROT1 ROT1 ::: CLOB);
    if (C == 2) asm volatile(ROT2 ROT2 ROT2 ::: CLOB);
    if (C == 4) asm volatile(ROT4 ROT4 ROT4 ::: CLOB);
    if (C == 8) asm volatile(ROT8 ROT8 ROT8 ::: CLOB);
    if (C == 16) asm volatile(ROT16 ROT16 ROT16 ::: CLOB);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

int main() {
  unsigned long long* d;
  hipMalloc(&d, 8);
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto run = [&](auto kern, int c, int wgs_per_cu) {
    // 256 threads = one wave per SIMD; 2 workgroups per CU = two waves per SIMD
    hipLaunchKernelGGL(kern, dim3(256 * wgs_per_cu), dim3(256), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256 * wgs_per_cu), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas_per_simd = 48.0 * iters * wgs_per_cu;
    printf("accumulators %2d, waves/SIMD %d: %.2f ns per MFMA per SIMD (%.1f TFLOP/s chip)\n", c, wgs_per_cu, ms * 1e6 / mfmas_per_simd,
           mfmas_per_simd * 1024 * 16384 / (ms * 1e-3) / 1e12);
  };
  for (int w = 1; w <= 2; ++w) {
    run(probe<1>, 1, w);
    run(probe<2>, 2, w);
    run(probe<4>, 4, w);
    run(probe<8>, 8, w);
    run(probe<16>, 16, w);
  }
  return 0;
}
