// Hardware probe: does a VALU / LDS write to an MFMA's SrcB register, issued right after the MFMA,
// corrupt the MFMA on gfx950?  One wave; A = B = 1.0 (fp16) so every D element must be 32 (K=32) or 16 (K=16).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define PRE_MFMA "v_mfma_f32_16x16x32_f16 a[4:7], v[20:23], v[24:27], a[4:7]\n\t"

template <int PRE, int GAP, int MODE>
__global__ void probe(float* out) {
  __shared__ unsigned zero_lds[256];
  zero_lds[threadIdx.x] = 0u;
  __syncthreads();
  const unsigned ones = 0x3c003c00u;
  unsigned ldsaddr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned*)zero_lds + threadIdx.x * 4;
  f4 d;
  asm volatile(
      "v_mov_b32 v0, %1\n\tv_mov_b32 v1, %1\n\tv_mov_b32 v2, %1\n\tv_mov_b32 v3, %1\n\t"
      "v_mov_b32 v4, %1\n\tv_mov_b32 v5, %1\n\tv_mov_b32 v6, %1\n\tv_mov_b32 v7, %1\n\t"
      "v_mov_b32 v20, %1\n\tv_mov_b32 v21, %1\n\tv_mov_b32 v22, %1\n\tv_mov_b32 v23, %1\n\t"
      "v_mov_b32 v24, %1\n\tv_mov_b32 v25, %1\n\tv_mov_b32 v26, %1\n\tv_mov_b32 v27, %1\n\t"
      "v_accvgpr_write_b32 a4, 0\n\tv_accvgpr_write_b32 a5, 0\n\tv_accvgpr_write_b32 a6, 0\n\tv_accvgpr_write_b32 a7, 0\n\t"
      "s_nop 7\n\ts_nop 7\n\t"
      ".rept %3\n\t" PRE_MFMA ".endr\n\t"
      "v_mfma_f32_16x16x32_f16 a[0:3], v[0:3], v[4:7], 0\n\t"
      ".rept %4\n\ts_nop 0\n\t.endr\n\t"
      ".if %5 == 0\n\t"
      "v_mov_b32 v7, 0\n\tv_mov_b32 v4, 0\n\t"
      ".elseif %5 == 1\n\t"
      "ds_read_b32 v7, %2\n\t"
      ".elseif %5 == 2\n\t"
      "v_mov_b32 v3, 0\n\tv_mov_b32 v0, 0\n\t"
      ".endif\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
      "v_accvgpr_read_b32 %0, a0\n\t"
      : "=v"(d[0])
      : "v"(ones), "v"(ldsaddr), "n"(PRE), "n"(GAP), "n"(MODE)
      : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "a0", "a1", "a2",
        "a3", "a4", "a5", "a6", "a7", "memory");
  out[threadIdx.x] = d[0];
}

template <int PRE, int GAP, int MODE>
void run(float* dev, const char* what) {
  std::vector<float> h(64);
  int bad = 0;
  float mn = 1e9f;
  for (int rep = 0; rep < 20; ++rep) {
    hipLaunchKernelGGL((probe<PRE, GAP, MODE>), dim3(1), dim3(64), 0, 0, dev);
    hipMemcpy(h.data(), dev, 64 * sizeof(float), hipMemcpyDeviceToHost);
    for (float v : h) {
      if (v != 32.f) ++bad;
      if (v < mn) mn = v;
    }
  }
  printf("%-10s pre=%d gap=%d : %s (bad lanes %d/1280, min %.0f)\n", what, PRE, GAP, bad ? "CORRUPT" : "ok", bad, mn);
}

int main() {
  float* dev;
  hipMalloc(&dev, 64 * sizeof(float));
#define ROW(MODE, WHAT)                                                                                            \
  run<0, 0, MODE>(dev, WHAT); run<0, 1, MODE>(dev, WHAT); run<0, 2, MODE>(dev, WHAT); run<0, 4, MODE>(dev, WHAT);   \
  run<1, 0, MODE>(dev, WHAT); run<1, 2, MODE>(dev, WHAT); run<1, 4, MODE>(dev, WHAT); run<1, 8, MODE>(dev, WHAT);   \
  run<2, 0, MODE>(dev, WHAT); run<2, 4, MODE>(dev, WHAT); run<2, 8, MODE>(dev, WHAT);                               \
  run<4, 0, MODE>(dev, WHAT); run<4, 4, MODE>(dev, WHAT); run<4, 8, MODE>(dev, WHAT); run<4, 16, MODE>(dev, WHAT);  \
  run<8, 0, MODE>(dev, WHAT); run<8, 8, MODE>(dev, WHAT); run<8, 16, MODE>(dev, WHAT); run<8, 32, MODE>(dev, WHAT);
  ROW(0, "valu->B")
  ROW(2, "valu->A")
  ROW(1, "lds->B")
  return 0;
}
