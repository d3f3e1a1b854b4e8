// Hardware probe (gfx950): the block-scaled fp8 MFMA v_mfma_scale_f32_16x16x128_f8f6f4 as the carrier of the two cross
// terms of a split product (W_lo * x_hi and W_hi * x_lo) next to v_mfma_f32_16x16x32_f16 for W_hi * x_hi:
//   1. operand layout / scale semantics against a host computation, v_cvt_pk_fp8_f32 rounding and saturation,
//   2. issue cost: 12 f16 MFMAs (today's K = 128 unit) against 4 f16 + 2 fp8 (K = 128), one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

static float e4m3(uint8_t b) {
  const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  float v = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.f + m / 8.f, e - 7);
  return s ? -v : v;
}

__global__ void mm(const v8i* a, const v8i* b, f4* c, int sa, int sb) {
  f4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0, sa, 0, sb);
  c[threadIdx.x] = acc;
}

__global__ void cvt(const float* x, int n, uint8_t* o) {
  for (int i = threadIdx.x; i < n / 2; i += 64) {
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(x[2 * i], x[2 * i + 1], 0, false);
    o[2 * i] = w & 255;
    o[2 * i + 1] = (w >> 8) & 255;
  }
}

typedef short v2s __attribute__((ext_vector_type(2)));
__global__ void cvts(const float* x, float scale, uint8_t* o) {
  if (threadIdx.x == 0) {
    v2s q = {0, 0};
    q = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(q, x[0], x[1], scale, false);
    q = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(q, x[2], x[3], scale, true);
    o[0] = q[0] & 255; o[1] = (q[0] >> 8) & 255; o[2] = q[1] & 255; o[3] = (q[1] >> 8) & 255;
  }
}

// ns per [one f16 MFMA + N conversions], one wave per SIMD (compare tools/probe/mfma_valu_overlap.hip)
template <int N, int KIND>
__global__ void __launch_bounds__(256) ov(float* out, int iters) {
  float r = 0.f;
  for (int it = 0; it < iters; ++it) {
    asm volatile(
        ".rept 16\n\t"
        "v_mfma_f32_16x16x32_f16 a[0:3], v[0:3], v[4:7], a[0:3]\n\t"
        ".rept %1\n\t"
        ".if %2 == 0\n\t v_cvt_pk_fp8_f32 v13, v14, v15\n\t"
        ".elseif %2 == 1\n\t v_cvt_scalef32_pk_fp8_f32 v13, v14, v15, s20\n\t"
        ".elseif %2 == 2\n\t v_cvt_pk_fp8_f32 v13, v14, v15 op_sel:[0,0,1]\n\t"
        ".endif\n\t"
        ".endr\n\t"
        "v_mfma_f32_16x16x32_f16 a[4:7], v[0:3], v[4:7], a[4:7]\n\t"
        ".rept %1\n\t"
        ".if %2 == 0\n\t v_cvt_pk_fp8_f32 v16, v14, v15\n\t"
        ".elseif %2 == 1\n\t v_cvt_scalef32_pk_fp8_f32 v16, v14, v15, s20\n\t"
        ".elseif %2 == 2\n\t v_cvt_pk_fp8_f32 v16, v14, v15 op_sel:[0,0,1]\n\t"
        ".endif\n\t"
        ".endr\n\t"
        ".endr\n\t"
        : "+v"(r) : "n"(N), "n"(KIND)
        : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v13", "v14", "v15", "v16", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "s20");
  }
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int N, int KIND>
float runov(float* dev) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((ov<N, KIND>), dim3(256), dim3(256), 0, 0, dev, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((ov<N, KIND>), dim3(256), dim3(256), 0, 0, dev, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6f / (iters * 32.f);
}

#define F16 "v_mfma_f32_16x16x32_f16 "
#define F8 "v_mfma_scale_f32_16x16x128_f8f6f4 "
template <int KIND>
__global__ void __launch_bounds__(256) t(float* out, int iters) {
  float r = 0.f;
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0)
      asm volatile(".rept 8\n\t"
                   F16 "a[0:3], v[0:3], v[4:7], a[0:3]\n\t" F16 "a[4:7], v[0:3], v[4:7], a[4:7]\n\t"
                   F16 "a[8:11], v[0:3], v[4:7], a[8:11]\n\t" F16 "a[12:15], v[0:3], v[4:7], a[12:15]\n\t"
                   F16 "a[0:3], v[0:3], v[4:7], a[0:3]\n\t" F16 "a[4:7], v[0:3], v[4:7], a[4:7]\n\t"
                   F16 "a[8:11], v[0:3], v[4:7], a[8:11]\n\t" F16 "a[12:15], v[0:3], v[4:7], a[12:15]\n\t"
                   F16 "a[0:3], v[0:3], v[4:7], a[0:3]\n\t" F16 "a[4:7], v[0:3], v[4:7], a[4:7]\n\t"
                   F16 "a[8:11], v[0:3], v[4:7], a[8:11]\n\t" F16 "a[12:15], v[0:3], v[4:7], a[12:15]\n\t"
                   ".endr\n\t" : "+v"(r) : : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "a0", "a1", "a2", "a3", "a4", "a5",
                   "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15");
    else if (KIND == 1)
      asm volatile(".rept 8\n\t"
                   F16 "a[0:3], v[0:3], v[4:7], a[0:3]\n\t" F16 "a[4:7], v[0:3], v[4:7], a[4:7]\n\t"
                   F16 "a[8:11], v[0:3], v[4:7], a[8:11]\n\t" F16 "a[12:15], v[0:3], v[4:7], a[12:15]\n\t"
                   F8 "a[0:3], v[8:15], v[16:23], a[0:3], v24, v25 op_sel_hi:[0,0,0]\n\t"
                   F8 "a[4:7], v[8:15], v[16:23], a[4:7], v24, v25 op_sel_hi:[0,0,0]\n\t"
                   ".endr\n\t" : "+v"(r) : : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13",
                   "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "a0", "a1", "a2", "a3", "a4",
                   "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15");
    else if (KIND == 2)
      asm volatile(".rept 8\n\t"
                   F8 "a[0:3], v[8:15], v[16:23], a[0:3], v24, v25 op_sel_hi:[0,0,0]\n\t"
                   F8 "a[4:7], v[8:15], v[16:23], a[4:7], v24, v25 op_sel_hi:[0,0,0]\n\t"
                   F8 "a[8:11], v[8:15], v[16:23], a[8:11], v24, v25 op_sel_hi:[0,0,0]\n\t"
                   F8 "a[12:15], v[8:15], v[16:23], a[12:15], v24, v25 op_sel_hi:[0,0,0]\n\t"
                   F8 "a[0:3], v[8:15], v[16:23], a[0:3], v24, v25 op_sel_hi:[0,0,0]\n\t"
                   F8 "a[4:7], v[8:15], v[16:23], a[4:7], v24, v25 op_sel_hi:[0,0,0]\n\t"
                   ".endr\n\t" : "+v"(r) : : "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20",
                   "v21", "v22", "v23", "v24", "v25", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12",
                   "a13", "a14", "a15");
    else  // f16 and fp8 interleaved: f16 f16 fp8 f16 f16 fp8
      asm volatile(".rept 8\n\t"
                   F16 "a[0:3], v[0:3], v[4:7], a[0:3]\n\t" F16 "a[4:7], v[0:3], v[4:7], a[4:7]\n\t"
                   F8 "a[8:11], v[8:15], v[16:23], a[8:11], v24, v25 op_sel_hi:[0,0,0]\n\t"
                   F16 "a[0:3], v[0:3], v[4:7], a[0:3]\n\t" F16 "a[4:7], v[0:3], v[4:7], a[4:7]\n\t"
                   F8 "a[12:15], v[8:15], v[16:23], a[12:15], v24, v25 op_sel_hi:[0,0,0]\n\t"
                   ".endr\n\t" : "+v"(r) : : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13",
                   "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "a0", "a1", "a2", "a3", "a4",
                   "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15");
  }
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int KIND>
float run(float* dev) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((t<KIND>), dim3(256), dim3(256), 0, 0, dev, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((t<KIND>), dim3(256), dim3(256), 0, 0, dev, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6f / (iters * 8.f);  // ns per group
}

int main() {
  // ---- 1. layout / scale
  uint8_t ha[64 * 32], hb[64 * 32];
  srand(1);
  for (int i = 0; i < 64 * 32; ++i) {
    do ha[i] = rand() & 255; while ((ha[i] & 0x7f) == 0x7f);
    do hb[i] = rand() & 255; while ((hb[i] & 0x7f) == 0x7f);
  }
  v8i *da, *db; f4* dc; float hc[256];
  hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dc, 1024);
  hipMemcpy(da, ha, 2048, hipMemcpyHostToDevice); hipMemcpy(db, hb, 2048, hipMemcpyHostToDevice);
  for (int trial = 0; trial < 3; ++trial) {
    const int sa = trial == 0 ? 127 : (trial == 1 ? 116 : 127), sb = trial == 2 ? 120 : 127;
    hipLaunchKernelGGL(mm, dim3(1), dim3(64), 0, 0, da, db, dc, sa, sb);
    hipMemcpy(hc, dc, 1024, hipMemcpyDeviceToHost);
    double worst = 0, mag = 0;
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 4; ++r) {
        const int n = l & 15, m = 4 * (l >> 4) + r;  // D[m][n]: lane n + 16 (m / 4), register m % 4
        double ref = 0;
        for (int kb = 0; kb < 4; ++kb)
          for (int j = 0; j < 32; ++j) ref += (double)e4m3(ha[(m + 16 * kb) * 32 + j]) * (double)e4m3(hb[(n + 16 * kb) * 32 + j]);
        ref *= ldexp(1.0, sa - 127 + sb - 127);
        worst = fmax(worst, fabs(ref - hc[l * 4 + r]));
        mag = fmax(mag, fabs(ref));
      }
    printf("scaled fp8 MFMA, scale bytes A=%d B=%d: max |gpu - host| = %.3e at magnitude %.3e\n", sa, sb, worst, mag);
  }
  // ---- 1b. conversion
  float hx[16] = {0.3f, -0.3f, 1.0f, 1.0625f, 1.1875f, 447.f, 449.f, 470.f, 1000.f, -1e6f, 0.001f, 0.0009765625f, 0.015625f, 1e-5f, 17.f, 19.f};
  float* dx; uint8_t* dob; uint8_t ho[16];
  hipMalloc(&dx, 64); hipMalloc(&dob, 16);
  hipMemcpy(dx, hx, 64, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(cvt, dim3(1), dim3(64), 0, 0, dx, 16, dob);
  hipMemcpy(ho, dob, 16, hipMemcpyDeviceToHost);
  for (int i = 0; i < 16; ++i) printf("cvt_pk_fp8_f32(%g) = 0x%02x = %g\n", hx[i], ho[i], (ho[i] & 0x7f) == 0x7f ? NAN : e4m3(ho[i]));
  {
    float hy[4] = {1.0f, 3.0f, -0.5f, 100.f};
    hipMemcpy(dx, hy, 16, hipMemcpyHostToDevice);
    for (float sc : {1.0f, 4.0f, 0.25f}) {
      hipLaunchKernelGGL(cvts, dim3(1), dim3(64), 0, 0, dx, sc, dob);
      hipMemcpy(ho, dob, 4, hipMemcpyDeviceToHost);
      printf("cvt_scalef32_pk_fp8_f32(1, 3, -0.5, 100; scale %g) = %g %g %g %g\n", sc, e4m3(ho[0]), e4m3(ho[1]), e4m3(ho[2]),
             (ho[3] & 0x7f) == 0x7f ? NAN : e4m3(ho[3]));
    }
  }
  // ---- 2. timing
  float* dev; hipMalloc(&dev, 256 * 256 * 4);
#define OVROW(KIND, NAME) printf("%-34s N=0 %5.2f N=1 %5.2f N=2 %5.2f N=3 %5.2f N=4 %5.2f N=6 %5.2f ns per (f16 MFMA + N ops)\n", NAME, \
    runov<0, KIND>(dev), runov<1, KIND>(dev), runov<2, KIND>(dev), runov<3, KIND>(dev), runov<4, KIND>(dev), runov<6, KIND>(dev));
  OVROW(0, "v_cvt_pk_fp8_f32")
  OVROW(1, "v_cvt_scalef32_pk_fp8_f32")
  OVROW(2, "v_cvt_pk_fp8_f32 op_sel hi")
  for (int rep = 0; rep < 2; ++rep) {
    printf("12 x f16 16x16x32            : %6.2f ns per K=128 unit\n", run<0>(dev));
    printf("4 x f16 + 2 x fp8 16x16x128  : %6.2f ns per K=128 unit\n", run<1>(dev));
    printf("6 x fp8 16x16x128            : %6.2f ns (per 6)\n", run<2>(dev));
    printf("f16 f16 fp8 f16 f16 fp8      : %6.2f ns per K=128 unit\n", run<3>(dev));
  }
  return 0;
}
