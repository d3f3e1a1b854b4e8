// Hardware probe: how many independent VALU instructions issue in the shadow of one
// v_mfma_f32_16x16x32_f16 from the SAME wave (1 wave per SIMD) on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>

#define MF(acc) "v_mfma_f32_16x16x32_f16 a[" acc "], v[0:3], v[4:7], a[" acc "]\n\t"

template <int N, int KIND>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  float r = 0.f;
  for (int it = 0; it < iters; ++it) {
    asm volatile(
        ".rept 16\n\t"
        MF("0:3")
        ".rept %1\n\t"
        ".if %2 == 0\n\t v_fma_f32 v10, v11, v12, v10\n\t"           // independent chain per register
        ".elseif %2 == 1\n\t v_cvt_f16_f32 v13, v14\n\t"
        ".elseif %2 == 2\n\t v_accvgpr_read_b32 v15, a20\n\t"
        ".elseif %2 == 3\n\t v_pk_add_f32 v[16:17], v[18:19], v[20:21]\n\t"
        ".elseif %2 == 4\n\t v_cvt_pk_f16_f32 v13, v14, v22\n\t"
        ".elseif %2 == 5\n\t v_exp_f32 v13, v14\n\t"
        ".elseif %2 == 6\n\t s_mov_b32 s20, s21\n\t"
        ".elseif %2 == 7\n\t v_accvgpr_write_b32 a24, v14\n\t"
        ".endif\n\t"
        ".endr\n\t"
        MF("4:7")
        ".rept %1\n\t"
        ".if %2 == 0\n\t v_fma_f32 v23, v11, v12, v23\n\t"
        ".elseif %2 == 1\n\t v_cvt_f16_f32 v24, v14\n\t"
        ".elseif %2 == 2\n\t v_accvgpr_read_b32 v25, a21\n\t"
        ".elseif %2 == 3\n\t v_pk_add_f32 v[26:27], v[18:19], v[20:21]\n\t"
        ".elseif %2 == 4\n\t v_cvt_pk_f16_f32 v24, v14, v22\n\t"
        ".elseif %2 == 5\n\t v_exp_f32 v24, v14\n\t"
        ".elseif %2 == 6\n\t s_mov_b32 s22, s21\n\t"
        ".elseif %2 == 7\n\t v_accvgpr_write_b32 a25, v14\n\t"
        ".endif\n\t"
        ".endr\n\t"
        MF("8:11")
        ".rept %1\n\t"
        ".if %2 == 0\n\t v_fma_f32 v28, v11, v12, v28\n\t"
        ".elseif %2 == 1\n\t v_cvt_f16_f32 v29, v14\n\t"
        ".elseif %2 == 2\n\t v_accvgpr_read_b32 v30, a22\n\t"
        ".elseif %2 == 3\n\t v_pk_add_f32 v[32:33], v[18:19], v[20:21]\n\t"
        ".elseif %2 == 4\n\t v_cvt_pk_f16_f32 v29, v14, v22\n\t"
        ".elseif %2 == 5\n\t v_exp_f32 v29, v14\n\t"
        ".elseif %2 == 6\n\t s_mov_b32 s23, s21\n\t"
        ".elseif %2 == 7\n\t v_accvgpr_write_b32 a26, v14\n\t"
        ".endif\n\t"
        ".endr\n\t"
        ".endr\n\t"
        : "+v"(r)
        : "n"(N), "n"(KIND)
        : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19",
          "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v32", "v33", "a0", "a1", "a2", "a3",
          "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a20", "a21", "a22", "a24", "a25", "a26", "s20", "s21", "s22", "s23");
  }
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int N, int KIND>
float run(float* dev) {
  const int iters = 2000, blocks = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<N, KIND>), dim3(blocks), dim3(256), 0, 0, dev, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<N, KIND>), dim3(blocks), dim3(256), 0, 0, dev, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6f / (iters * 48.f);  // ns per MFMA group
}
#define ROW(KIND, NAME) printf("%-18s", NAME); \
  printf(" N=0 %5.2f", run<0, KIND>(dev)); printf(" N=1 %5.2f", run<1, KIND>(dev)); printf(" N=2 %5.2f", run<2, KIND>(dev)); \
  printf(" N=3 %5.2f", run<3, KIND>(dev)); printf(" N=4 %5.2f", run<4, KIND>(dev)); printf(" N=5 %5.2f", run<5, KIND>(dev)); \
  printf(" N=6 %5.2f", run<6, KIND>(dev)); printf(" N=8 %5.2f  ns per (MFMA + N ops)\n", run<8, KIND>(dev));
int main() {
  float* dev; hipMalloc(&dev, 256 * 256 * 4);
  ROW(0, "v_fma_f32")
  ROW(1, "v_cvt_f16_f32")
  ROW(4, "v_cvt_pk_f16_f32")
  ROW(2, "v_accvgpr_read")
  ROW(7, "v_accvgpr_write")
  ROW(3, "v_pk_add_f32")
  ROW(5, "v_exp_f32")
  ROW(6, "s_mov_b32")
  return 0;
}
