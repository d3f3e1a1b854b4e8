// Hardware probe (gfx950): shader-clock cycles per instruction for the instruction kinds of the encoder-stack glue
// (tools/gen_h3_enc_asm.py), one wave per SIMD, streams of 256 instructions between two s_memtime reads.
#include <hip/hip_runtime.h>
#include <cstdio>

#define BODY(TXT) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\t" : "=s"(t0)); \
  asm volatile(".rept 32\n\t" TXT ".endr\n\t" ::: "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25", \
     "v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","a0","a1","a2","a3","a4","a5","a6","a7","s20","s21","s22","s23","vcc","scc"); \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\t" : "=s"(t1));

template <int KIND>
__global__ void __launch_bounds__(256) k(long long* out) {
  unsigned long long t0 = 0, t1 = 0;
  asm volatile("s_mov_b64 s[20:21], exec\n\ts_mov_b32 s22, 5" ::: "s20", "s21", "s22");
  for (int rep = 0; rep < 3; ++rep) {
    if constexpr (KIND == 0) { BODY("v_pk_fma_f32 v[10:11], v[12:13], v[14:15], v[16:17]\n\tv_pk_fma_f32 v[18:19], v[12:13], v[14:15], v[16:17]\n\tv_pk_fma_f32 v[20:21], v[12:13], v[14:15], v[16:17]\n\tv_pk_fma_f32 v[22:23], v[12:13], v[14:15], v[16:17]\n\tv_pk_fma_f32 v[24:25], v[12:13], v[14:15], v[16:17]\n\tv_pk_fma_f32 v[26:27], v[12:13], v[14:15], v[16:17]\n\tv_pk_fma_f32 v[28:29], v[12:13], v[14:15], v[16:17]\n\tv_pk_fma_f32 v[30:31], v[12:13], v[14:15], v[16:17]\n\t") }
    if constexpr (KIND == 1) { BODY("v_pk_mul_f32 v[10:11], v[12:13], v[14:15]\n\tv_pk_mul_f32 v[18:19], v[12:13], v[14:15]\n\tv_pk_mul_f32 v[20:21], v[12:13], v[14:15]\n\tv_pk_mul_f32 v[22:23], v[12:13], v[14:15]\n\tv_pk_mul_f32 v[24:25], v[12:13], v[14:15]\n\tv_pk_mul_f32 v[26:27], v[12:13], v[14:15]\n\tv_pk_mul_f32 v[28:29], v[12:13], v[14:15]\n\tv_pk_mul_f32 v[30:31], v[12:13], v[14:15]\n\t") }
    if constexpr (KIND == 2) { BODY("v_accvgpr_read_b32 v10, a0\n\tv_accvgpr_read_b32 v11, a1\n\tv_accvgpr_read_b32 v12, a2\n\tv_accvgpr_read_b32 v13, a3\n\tv_accvgpr_read_b32 v14, a4\n\tv_accvgpr_read_b32 v15, a5\n\tv_accvgpr_read_b32 v16, a6\n\tv_accvgpr_read_b32 v17, a7\n\t") }
    if constexpr (KIND == 3) { BODY("v_accvgpr_write_b32 a0, v10\n\tv_accvgpr_write_b32 a1, v11\n\tv_accvgpr_write_b32 a2, v12\n\tv_accvgpr_write_b32 a3, v13\n\tv_accvgpr_write_b32 a4, v14\n\tv_accvgpr_write_b32 a5, v15\n\tv_accvgpr_write_b32 a6, v16\n\tv_accvgpr_write_b32 a7, v17\n\t") }
    if constexpr (KIND == 4) { BODY("v_cvt_pk_f16_f32 v10, v12, v13\n\tv_cvt_pk_f16_f32 v11, v12, v13\n\tv_cvt_pk_f16_f32 v18, v12, v13\n\tv_cvt_pk_f16_f32 v19, v12, v13\n\tv_cvt_pk_f16_f32 v20, v12, v13\n\tv_cvt_pk_f16_f32 v21, v12, v13\n\tv_cvt_pk_f16_f32 v22, v12, v13\n\tv_cvt_pk_f16_f32 v23, v12, v13\n\t") }
    if constexpr (KIND == 5) { BODY("v_fma_mix_f32 v10, v12, -1.0, v13 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 v11, v12, -1.0, v13 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 v18, v12, -1.0, v13 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 v19, v12, -1.0, v13 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 v20, v12, -1.0, v13 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 v21, v12, -1.0, v13 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 v22, v12, -1.0, v13 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 v23, v12, -1.0, v13 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t") }
    if constexpr (KIND == 6) { BODY("v_cndmask_b32_e64 v10, 0, v10, s[20:21]\n\tv_cndmask_b32_e64 v11, 0, v11, s[20:21]\n\tv_cndmask_b32_e64 v12, 0, v12, s[20:21]\n\tv_cndmask_b32_e64 v13, 0, v13, s[20:21]\n\tv_cndmask_b32_e64 v14, 0, v14, s[20:21]\n\tv_cndmask_b32_e64 v15, 0, v15, s[20:21]\n\tv_cndmask_b32_e64 v16, 0, v16, s[20:21]\n\tv_cndmask_b32_e64 v17, 0, v17, s[20:21]\n\t") }
    if constexpr (KIND == 7) { BODY("v_mfma_f32_16x16x16_f16 v[10:13], v[34:35], v[36:37], 0\n\tv_mfma_f32_16x16x16_f16 v[14:17], v[34:35], v[36:37], 0\n\tv_mfma_f32_16x16x16_f16 v[18:21], v[34:35], v[36:37], 0\n\tv_mfma_f32_16x16x16_f16 v[22:25], v[34:35], v[36:37], 0\n\tv_mfma_f32_16x16x16_f16 v[26:29], v[34:35], v[36:37], 0\n\tv_mfma_f32_16x16x16_f16 v[30:33], v[34:35], v[36:37], 0\n\tv_mfma_f32_16x16x16_f16 v[10:13], v[34:35], v[36:37], 0\n\tv_mfma_f32_16x16x16_f16 v[14:17], v[34:35], v[36:37], 0\n\t") }
    if constexpr (KIND == 8) { BODY("v_pk_fma_f32 v[10:11], v[12:13], v[14:15], v[10:11]\n\tv_pk_fma_f32 v[10:11], v[12:13], v[14:15], v[10:11]\n\tv_pk_fma_f32 v[10:11], v[12:13], v[14:15], v[10:11]\n\tv_pk_fma_f32 v[10:11], v[12:13], v[14:15], v[10:11]\n\tv_pk_fma_f32 v[10:11], v[12:13], v[14:15], v[10:11]\n\tv_pk_fma_f32 v[10:11], v[12:13], v[14:15], v[10:11]\n\tv_pk_fma_f32 v[10:11], v[12:13], v[14:15], v[10:11]\n\tv_pk_fma_f32 v[10:11], v[12:13], v[14:15], v[10:11]\n\t") }
    if constexpr (KIND == 9) { BODY("v_accvgpr_read_b32 v10, a0\n\tv_accvgpr_read_b32 v11, a1\n\tv_pk_mul_f32 v[10:11], v[10:11], v[14:15]\n\tv_pk_add_f32 v[16:17], v[16:17], v[10:11]\n\tv_accvgpr_read_b32 v12, a2\n\tv_accvgpr_read_b32 v13, a3\n\tv_pk_mul_f32 v[12:13], v[12:13], v[14:15]\n\tv_pk_add_f32 v[18:19], v[18:19], v[12:13]\n\t") }
    if constexpr (KIND == 10) { BODY("s_bitcmp0_b32 s22, 1\n\ts_cbranch_scc1 1f\n\tv_mov_b32 v10, v11\n\t1:\n\ts_bitcmp0_b32 s22, 1\n\ts_cbranch_scc1 2f\n\tv_mov_b32 v10, v11\n\t2:\n\ts_bitcmp0_b32 s22, 1\n\ts_cbranch_scc1 3f\n\tv_mov_b32 v10, v11\n\t3:\n\ts_bitcmp0_b32 s22, 1\n\ts_cbranch_scc1 4f\n\tv_mov_b32 v10, v11\n\t4:\n\t") }
    if constexpr (KIND == 11) { BODY("v_mov_b32 v10, v20\n\tv_mov_b32 v11, v20\n\tv_mov_b32 v12, v20\n\tv_mov_b32 v13, v20\n\tv_mov_b32 v14, v20\n\tv_mov_b32 v15, v20\n\tv_mov_b32 v16, v20\n\tv_mov_b32 v17, v20\n\t") }
    if constexpr (KIND == 12) { BODY("v_fma_f32 v10, v20, v21, v22\n\tv_fma_f32 v11, v20, v21, v22\n\tv_fma_f32 v12, v20, v21, v22\n\tv_fma_f32 v13, v20, v21, v22\n\tv_fma_f32 v14, v20, v21, v22\n\tv_fma_f32 v15, v20, v21, v22\n\tv_fma_f32 v16, v20, v21, v22\n\tv_fma_f32 v17, v20, v21, v22\n\t") }
    if constexpr (KIND == 13) { BODY("v_pk_add_f32 v[10:11], v[12:13], v[14:15]\n\tv_pk_add_f32 v[18:19], v[12:13], v[14:15]\n\tv_pk_add_f32 v[20:21], v[12:13], v[14:15]\n\tv_pk_add_f32 v[22:23], v[12:13], v[14:15]\n\tv_pk_add_f32 v[24:25], v[12:13], v[14:15]\n\tv_pk_add_f32 v[26:27], v[12:13], v[14:15]\n\tv_pk_add_f32 v[28:29], v[12:13], v[14:15]\n\tv_pk_add_f32 v[30:31], v[12:13], v[14:15]\n\t") }
    if constexpr (KIND == 15) { BODY("v_pk_mul_f32 v[10:11], v[12:13], v[14:15] op_sel_hi:[1,0]\n\tv_pk_mul_f32 v[18:19], v[12:13], v[14:15] op_sel_hi:[1,0]\n\tv_pk_mul_f32 v[20:21], v[12:13], v[14:15] op_sel_hi:[1,0]\n\tv_pk_mul_f32 v[22:23], v[12:13], v[14:15] op_sel_hi:[1,0]\n\tv_pk_mul_f32 v[24:25], v[12:13], v[14:15] op_sel_hi:[1,0]\n\tv_pk_mul_f32 v[26:27], v[12:13], v[14:15] op_sel_hi:[1,0]\n\tv_pk_mul_f32 v[28:29], v[12:13], v[14:15] op_sel_hi:[1,0]\n\tv_pk_mul_f32 v[30:31], v[12:13], v[14:15] op_sel_hi:[1,0]\n\t") }
    if constexpr (KIND == 16) { BODY("v_max_f32 v10, v10, 0\n\tv_max_f32 v11, v11, 0\n\tv_max_f32 v12, v12, 0\n\tv_max_f32 v13, v13, 0\n\tv_max_f32 v14, v14, 0\n\tv_max_f32 v15, v15, 0\n\tv_max_f32 v16, v16, 0\n\tv_max_f32 v17, v17, 0\n\t") }
    if constexpr (KIND == 17) { BODY("v_pk_mul_f32 v[10:11], v[12:13], v[14:15] op_sel_hi:[1,0]\n\tv_pk_mul_f32 v[16:17], v[18:19], v[14:15] op_sel_hi:[1,0]\n\tv_max_f32 v10, v10, 0\n\tv_max_f32 v11, v11, 0\n\tv_max_f32 v16, v16, 0\n\tv_max_f32 v17, v17, 0\n\tv_cvt_pk_f16_f32 v20, v10, v11\n\tv_cvt_pk_f16_f32 v21, v16, v17\n\t") }
    if constexpr (KIND == 18) { BODY("v_fma_f32 v10, v12, s22, v14\n\tv_fma_f32 v11, v13, s22, v15\n\tv_fma_f32 v16, v18, s22, v14\n\tv_fma_f32 v17, v19, s22, v15\n\tv_max_f32 v10, v10, 0\n\tv_max_f32 v11, v11, 0\n\tv_max_f32 v16, v16, 0\n\tv_max_f32 v17, v17, 0\n\t") }
    if constexpr (KIND == 14) { BODY("v_mul_f32 v10, v20, v21\n\tv_mul_f32 v11, v20, v21\n\tv_fma_f32 v12, v20, v21, v22\n\tv_fma_f32 v13, v20, v21, v22\n\tv_add_f32 v14, v20, v21\n\tv_add_f32 v15, v20, v21\n\tv_fma_f32 v16, v20, v21, v22\n\tv_fma_f32 v17, v20, v21, v22\n\t") }
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) out[KIND] = (long long)(t1 - t0);
}

int main() {
  long long* dev; hipMalloc(&dev, 64 * 8); hipMemset(dev, 0, 64 * 8);
#define RUN(K) hipLaunchKernelGGL((k<K>), dim3(256), dim3(256), 0, 0, dev);
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15) RUN(16) RUN(17) RUN(18)
  long long h[64]; hipMemcpy(h, dev, 64 * 8, hipMemcpyDeviceToHost);
  const char* names[] = {"v_pk_fma_f32", "v_pk_mul_f32", "v_accvgpr_read_b32", "v_accvgpr_write_b32", "v_cvt_pk_f16_f32", "v_fma_mix_f32",
                         "v_cndmask_b32_e64 (sgpr mask)", "v_mfma_f32_16x16x16_f16 (0 acc)", "v_pk_fma_f32 dependent chain",
                         "2 acc reads + pk_mul + pk_add (dependent)", "s_bitcmp0 + s_cbranch taken (skipping one v_mov)", "v_mov_b32",
                         "v_fma_f32", "v_pk_add_f32", "v_mul/fma/add_f32 mix", "v_pk_mul_f32 op_sel_hi:[1,0]", "v_max_f32 v, v, 0",
                         "2 pk_mul(bcast) + 4 max + 2 cvt_pk (dependent)", "4 fma(sgpr) + 4 max (dependent)"};
  const int per[] = {8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 4, 8, 8, 8, 8, 8, 8, 8, 8};
  for (int i = 0; i < 19; ++i) printf("%-50s %7lld cycles / %d = %.2f per %s\n", names[i], h[i], 32 * per[i], (double)h[i] / (32 * per[i]), i == 10 ? "branch" : "instruction");
  return 0;
}
