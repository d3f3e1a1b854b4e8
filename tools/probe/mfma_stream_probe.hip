// Hardware probe (gfx950): what does a long stream of v_mfma_f32_16x16x32_f16 cost per instruction when EVERY CU runs
// it (4 waves per CU, one per SIMD), with random-looking operands vs zero operands, short vs long streams?  Separates
// "the schedule loses issue slots" from "the chip does not sustain 16 cycles per MFMA under this load".
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int KIND>
__global__ void __launch_bounds__(256) k(long long* out, const unsigned* init, int iters) {
  unsigned long long t0 = 0, t1 = 0;
  // operands: v[0:7] = A (hi, lo), v[8:31] = B for three token tiles (hi, lo): from memory so they look like data
  unsigned r[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) r[i] = init[(threadIdx.x * 32 + i) % 4096];
  asm volatile(
      "v_mov_b32 v0, %0\n\tv_mov_b32 v1, %1\n\tv_mov_b32 v2, %2\n\tv_mov_b32 v3, %3\n\tv_mov_b32 v4, %4\n\tv_mov_b32 v5, %5\n\tv_mov_b32 v6, %6\n\tv_mov_b32 v7, %7\n\t"
      "v_mov_b32 v8, %8\n\tv_mov_b32 v9, %9\n\tv_mov_b32 v10, %10\n\tv_mov_b32 v11, %11\n\tv_mov_b32 v12, %12\n\tv_mov_b32 v13, %13\n\tv_mov_b32 v14, %14\n\tv_mov_b32 v15, %15\n\t"
      "v_mov_b32 v16, %16\n\tv_mov_b32 v17, %17\n\tv_mov_b32 v18, %18\n\tv_mov_b32 v19, %19\n\tv_mov_b32 v20, %20\n\tv_mov_b32 v21, %21\n\tv_mov_b32 v22, %22\n\tv_mov_b32 v23, %23\n\t"
      "v_mov_b32 v24, %24\n\tv_mov_b32 v25, %25\n\tv_mov_b32 v26, %26\n\tv_mov_b32 v27, %27\n\tv_mov_b32 v28, %28\n\tv_mov_b32 v29, %29\n\tv_mov_b32 v30, %30\n\tv_mov_b32 v31, %31\n\t"
      :: "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]), "v"(r[8]), "v"(r[9]), "v"(r[10]), "v"(r[11]),
         "v"(r[12]), "v"(r[13]), "v"(r[14]), "v"(r[15]), "v"(r[16]), "v"(r[17]), "v"(r[18]), "v"(r[19]), "v"(r[20]), "v"(r[21]), "v"(r[22]), "v"(r[23]),
         "v"(r[24]), "v"(r[25]), "v"(r[26]), "v"(r[27]), "v"(r[28]), "v"(r[29]), "v"(r[30]), "v"(r[31])
      : "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23",
        "v24","v25","v26","v27","v28","v29","v30","v31");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\t" : "=s"(t0));
  for (int it = 0; it < iters; ++it) {
    // 36 MFMAs: 4 weight pairs x (hh, hl, lh) x 3 token tiles, accumulators a[0:47]
    asm volatile(
        ".rept 4\n\t"
        "v_mfma_f32_16x16x32_f16 a[0:3], v[0:3], v[8:11], a[0:3]\n\t"
        "v_mfma_f32_16x16x32_f16 a[4:7], v[0:3], v[16:19], a[4:7]\n\t"
        "v_mfma_f32_16x16x32_f16 a[8:11], v[0:3], v[24:27], a[8:11]\n\t"
        "v_mfma_f32_16x16x32_f16 a[0:3], v[0:3], v[12:15], a[0:3]\n\t"
        "v_mfma_f32_16x16x32_f16 a[4:7], v[0:3], v[20:23], a[4:7]\n\t"
        "v_mfma_f32_16x16x32_f16 a[8:11], v[0:3], v[28:31], a[8:11]\n\t"
        "v_mfma_f32_16x16x32_f16 a[0:3], v[4:7], v[8:11], a[0:3]\n\t"
        "v_mfma_f32_16x16x32_f16 a[4:7], v[4:7], v[16:19], a[4:7]\n\t"
        "v_mfma_f32_16x16x32_f16 a[8:11], v[4:7], v[24:27], a[8:11]\n\t"
        ".endr\n\t" ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11");
    if constexpr (KIND == 1) {  // accumulators reset now and then so values stay finite
      if ((it & 63) == 63) asm volatile(".rept 1\n\tv_accvgpr_write_b32 a0, 0\n\t.endr" ::: "a0");
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\t" : "=s"(t1));
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (long long)(t1 - t0);
}

int main() {
  long long* dev; hipMalloc(&dev, 64 * 8);
  unsigned* init; hipMalloc(&init, 4096 * 4);
  unsigned h[4096];
  for (int mode = 0; mode < 2; ++mode) {
    // mode 0: zero operands; mode 1: random fp16 pairs of magnitude ~1 (0x3c00 +- mantissa noise)
    srand(1);
    for (int i = 0; i < 4096; ++i) { unsigned a = 0x3800 + (rand() & 0x7ff), b = 0xb800 + (rand() & 0x7ff); h[i] = mode ? (a | (b << 16)) : 0u; }
    hipMemcpy(init, h, sizeof(h), hipMemcpyHostToDevice);
    for (int iters : {8, 64, 512, 4096}) {
      for (int blocks : {1, 256}) {
        hipMemset(dev, 0, 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k<0>), dim3(blocks), dim3(256), 0, 0, dev, init, iters);   // warm
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<0>), dim3(blocks), dim3(256), 0, 0, dev, init, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, dev, 8, hipMemcpyDeviceToHost);
        printf("%s operands, %4d CUs busy, %5d x 36 MFMAs: %9lld cycles = %.2f per MFMA; launch %.3f ms -> %.2f GHz\n", mode ? "random" : "zero  ",
               blocks, iters, c, (double)c / (36.0 * iters), ms, c / (ms * 1e6));
      }
    }
  }
  return 0;
}
