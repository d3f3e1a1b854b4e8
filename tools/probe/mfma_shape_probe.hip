// Hardware probe (gfx950): the SAME FLOPs as a stream of v_mfma_f32_16x16x32_f16 (the shape every kernel here uses) and as a
// stream of v_mfma_f32_32x32x16_f16, on one CU and on all 256, zero and random operands: cycles per instruction and the clock
// the chip sustains.  Both shapes are 1024 FLOP per cycle and SIMD on paper; the 32x32 shape reads half as many operand
// registers per FLOP, which is a power argument on a launch that is power-limited.  Every measured launch follows ~70 ms of the
// same stream: the clock takes tens of milliseconds to settle after idle (a 1.3 ms launch from idle reads 1.74-1.85 GHz where
// the settled chip runs 2.0; profiles/r04_mfma_shape_probe_{cold,warm}.txt).  Build: hipcc --offload-arch=gfx950 -O2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int SHAPE>
__global__ void __launch_bounds__(256) k(long long* out, const unsigned* init, int iters) {
  unsigned long long t0 = 0, t1 = 0;
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
  // accumulators start at zero
  asm volatile(".set i, 0\n\t.rept 48\n\tv_accvgpr_write_b32 a[i], 0\n\t.set i, i + 1\n\t.endr" :::
               "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23",
               "a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\t" : "=s"(t0));
  for (int it = 0; it < iters; ++it) {
    if constexpr (SHAPE == 16) {
      // 36 x 16x16x32: three accumulator chains, operands rotate over A (2 tiles) and B (6 tiles)
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
    } else {
      // 18 x 32x32x16 (= the FLOPs of 36 x 16x16x32): three accumulator chains of 16 registers
      asm volatile(
          ".rept 2\n\t"
          "v_mfma_f32_32x32x16_f16 a[0:15], v[0:3], v[8:11], a[0:15]\n\t"
          "v_mfma_f32_32x32x16_f16 a[16:31], v[0:3], v[16:19], a[16:31]\n\t"
          "v_mfma_f32_32x32x16_f16 a[32:47], v[0:3], v[24:27], a[32:47]\n\t"
          "v_mfma_f32_32x32x16_f16 a[0:15], v[0:3], v[12:15], a[0:15]\n\t"
          "v_mfma_f32_32x32x16_f16 a[16:31], v[0:3], v[20:23], a[16:31]\n\t"
          "v_mfma_f32_32x32x16_f16 a[32:47], v[0:3], v[28:31], a[32:47]\n\t"
          "v_mfma_f32_32x32x16_f16 a[0:15], v[4:7], v[8:11], a[0:15]\n\t"
          "v_mfma_f32_32x32x16_f16 a[16:31], v[4:7], v[16:19], a[16:31]\n\t"
          "v_mfma_f32_32x32x16_f16 a[32:47], v[4:7], v[24:27], a[32:47]\n\t"
          ".endr\n\t" :::
          "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23",
          "a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47");
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\t" : "=s"(t1));
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (long long)(t1 - t0);
}

template <int SHAPE>
static void run(long long* dev, const unsigned* init, const char* what, int blocks, int iters) {
  hipMemset(dev, 0, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  // warm-up: the clock ramps for tens of milliseconds after idle (r04: a 1.3 ms launch from idle reads 1.74-1.85 GHz where the
  // settled chip runs 2.13) - so ~80 ms of the same stream first, then the measured launch back to back
  hipLaunchKernelGGL((k<SHAPE>), dim3(blocks), dim3(256), 0, 0, dev, init, 8 * iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<SHAPE>), dim3(blocks), dim3(256), 0, 0, dev, init, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, dev, 8, hipMemcpyDeviceToHost);
  const double flop = 36.0 * 16 * 16 * 32 * 2 * iters * 4.0 * blocks;
  printf("%s operands, %3d CUs, %s: %9lld cycles = %.2f per instruction (%.1f FLOP/cycle/SIMD); launch %.3f ms -> %.2f GHz, %.0f TFLOP/s\n",
         what, blocks, SHAPE == 16 ? "36 x 16x16x32" : "18 x 32x32x16", c, (double)c / ((SHAPE == 16 ? 36.0 : 18.0) * iters),
         36.0 * 16384 * iters / (double)c, ms, c / (ms * 1e6), flop / (ms * 1e-3) / 1e12);
}

int main() {
  long long* dev; hipMalloc(&dev, 64 * 8);
  unsigned* init; hipMalloc(&init, 4096 * 4);
  unsigned h[4096];
  for (int mode = 0; mode < 2; ++mode) {
    srand(1);
    for (int i = 0; i < 4096; ++i) { unsigned a = 0x3800 + (rand() & 0x7ff), b = 0xb800 + (rand() & 0x7ff); h[i] = mode ? (a | (b << 16)) : 0u; }
    hipMemcpy(init, h, sizeof(h), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep)
      for (int blocks : {1, 256}) {
        run<16>(dev, init, mode ? "random" : "zero  ", blocks, 32768);
        run<32>(dev, init, mode ? "random" : "zero  ", blocks, 32768);
      }
  }
  return 0;
}
