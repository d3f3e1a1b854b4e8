// Hardware probe (gfx950): cost of the non-MFMA instructions of one FFN-like weight stage (36 x v_mfma_f32_16x16x32_f16,
// one wave per SIMD, 4 waves per CU) - what do 8 ds_read_b128, 24 VALU ops, a barrier add to 36 x 16 cycles, and does
// their placement matter?
#include <hip/hip_runtime.h>
#include <cstdio>

#define M(acc, a, b) "v_mfma_f32_16x16x32_f16 a[" acc "], v[" a "], v[" b "], a[" acc "]\n\t"
#define DR(dst, off) "ds_read_b128 v[" dst "], v40 offset:" off "\n\t"
#define VA(d) "v_fma_f32 v" d ", v41, v42, v43\n\t"

template <int KIND>
__global__ void __launch_bounds__(256) k(long long* out, int iters, const char* src) {
  extern __shared__ char lds[];
  unsigned long long t0 = 0, t1 = 0;
  for (int i = threadIdx.x; i < 16384; i += 256) ((float*)lds)[i] = 0.f;
  __syncthreads();
  asm volatile("v_mbcnt_lo_u32_b32 v40, -1, 0\n\tv_mbcnt_hi_u32_b32 v40, -1, v40\n\tv_lshlrev_b32 v40, 4, v40\n\t"
               "v_mov_b32 v41, 1.0\n\tv_mov_b32 v42, 0.5\n\tv_mov_b32 v43, 0.25" ::: "v40", "v41", "v42", "v43");
  // LDS-DMA source: this wave's 2 KiB share of a 9 KiB stage, stages streamed from an 8 MiB buffer (one net's weights)
  const char* gp = src + (threadIdx.x >> 6) * 2048 + (threadIdx.x & 63) * 16;
  const unsigned ring = 32768 + (threadIdx.x >> 6) * 2048;   // LDS byte offset of this wave's share of slot 0
  int slot = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\t" : "=s"(t0));
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND >= 9) {
      const unsigned m0v = __builtin_amdgcn_readfirstlane(ring + slot * 9216);
      slot = slot == 2 ? 0 : slot + 1;
      const char* g = gp + (size_t)(it & 511) * 9216;
      if constexpr (KIND == 9) {  // stage of KIND 6 + two LDS-DMA instructions and a counted wait
        asm volatile(
            M("0:3", "0:3", "8:11") DR("44:47", "0") M("4:7", "0:3", "16:19") DR("48:51", "1024") M("8:11", "0:3", "24:27") DR("52:55", "2048")
            M("0:3", "0:3", "8:11") DR("56:59", "3072") M("4:7", "0:3", "16:19") DR("60:63", "4096") M("8:11", "0:3", "24:27") DR("64:67", "5120")
            M("0:3", "0:3", "8:11") DR("68:71", "6144") M("4:7", "0:3", "16:19") DR("72:75", "7168") M("8:11", "0:3", "24:27")
            ".rept 5\n\t" M("0:3", "0:3", "8:11") VA("76") M("4:7", "0:3", "16:19") VA("77") M("8:11", "0:3", "24:27") VA("78") ".endr\n\t"
            "s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier\n\t"
            M("0:3", "0:3", "8:11") "s_mov_b32 m0, %1\n\t" M("4:7", "0:3", "16:19") "global_load_lds_dwordx4 %0, off\n\t" M("8:11", "0:3", "24:27") "global_load_lds_dwordx4 %0, off offset:1024\n\t"
            ".rept 3\n\t" M("0:3", "0:3", "8:11") VA("76") M("4:7", "0:3", "16:19") VA("77") M("8:11", "0:3", "24:27") VA("78") ".endr\n\t"
            :: "v"(g), "s"(m0v)
            : "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","memory");
      } else {  // the same without the barrier
        asm volatile(
            M("0:3", "0:3", "8:11") DR("44:47", "0") M("4:7", "0:3", "16:19") DR("48:51", "1024") M("8:11", "0:3", "24:27") DR("52:55", "2048")
            M("0:3", "0:3", "8:11") DR("56:59", "3072") M("4:7", "0:3", "16:19") DR("60:63", "4096") M("8:11", "0:3", "24:27") DR("64:67", "5120")
            M("0:3", "0:3", "8:11") DR("68:71", "6144") M("4:7", "0:3", "16:19") DR("72:75", "7168") M("8:11", "0:3", "24:27")
            ".rept 5\n\t" M("0:3", "0:3", "8:11") VA("76") M("4:7", "0:3", "16:19") VA("77") M("8:11", "0:3", "24:27") VA("78") ".endr\n\t"
            "s_waitcnt vmcnt(4) lgkmcnt(0)\n\t"
            M("0:3", "0:3", "8:11") "s_mov_b32 m0, %1\n\t" M("4:7", "0:3", "16:19") "global_load_lds_dwordx4 %0, off\n\t" M("8:11", "0:3", "24:27") "global_load_lds_dwordx4 %0, off offset:1024\n\t"
            ".rept 3\n\t" M("0:3", "0:3", "8:11") VA("76") M("4:7", "0:3", "16:19") VA("77") M("8:11", "0:3", "24:27") VA("78") ".endr\n\t"
            :: "v"(g), "s"(m0v)
            : "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","memory");
      }
    }
    if constexpr (KIND == 0) {
      asm volatile(".rept 12\n\t" M("0:3", "0:3", "8:11") M("4:7", "0:3", "16:19") M("8:11", "0:3", "24:27") ".endr\n\t" ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11");
    } else if constexpr (KIND == 1) {  // 8 reads, one per gap, gaps 1..8
      asm volatile(
          M("0:3", "0:3", "8:11") DR("44:47", "0") M("4:7", "0:3", "16:19") DR("48:51", "1024") M("8:11", "0:3", "24:27") DR("52:55", "2048")
          M("0:3", "0:3", "8:11") DR("56:59", "3072") M("4:7", "0:3", "16:19") DR("60:63", "4096") M("8:11", "0:3", "24:27") DR("64:67", "5120")
          M("0:3", "0:3", "8:11") DR("68:71", "6144") M("4:7", "0:3", "16:19") DR("72:75", "7168") M("8:11", "0:3", "24:27")
          ".rept 9\n\t" M("0:3", "0:3", "8:11") M("4:7", "0:3", "16:19") M("8:11", "0:3", "24:27") ".endr\n\t"
          "s_waitcnt lgkmcnt(0)\n\t" ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75");
    } else if constexpr (KIND == 2) {  // 8 reads, two per gap
      asm volatile(
          M("0:3", "0:3", "8:11") DR("44:47", "0") DR("48:51", "1024") M("4:7", "0:3", "16:19") M("8:11", "0:3", "24:27") DR("52:55", "2048") DR("56:59", "3072")
          M("0:3", "0:3", "8:11") M("4:7", "0:3", "16:19") DR("60:63", "4096") DR("64:67", "5120") M("8:11", "0:3", "24:27")
          M("0:3", "0:3", "8:11") DR("68:71", "6144") DR("72:75", "7168") M("4:7", "0:3", "16:19") M("8:11", "0:3", "24:27")
          ".rept 9\n\t" M("0:3", "0:3", "8:11") M("4:7", "0:3", "16:19") M("8:11", "0:3", "24:27") ".endr\n\t"
          "s_waitcnt lgkmcnt(0)\n\t" ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75");
    } else if constexpr (KIND == 3) {  // 24 VALU, one per gap
      asm volatile(".rept 8\n\t" M("0:3", "0:3", "8:11") VA("44") M("4:7", "0:3", "16:19") VA("45") M("8:11", "0:3", "24:27") VA("46") ".endr\n\t"
                   ".rept 4\n\t" M("0:3", "0:3", "8:11") M("4:7", "0:3", "16:19") M("8:11", "0:3", "24:27") ".endr\n\t" ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","v44","v45","v46");
    } else if constexpr (KIND == 4) {  // 24 VALU, two per gap in 12 gaps
      asm volatile(".rept 4\n\t" M("0:3", "0:3", "8:11") VA("44") VA("47") M("4:7", "0:3", "16:19") VA("45") VA("48") M("8:11", "0:3", "24:27") VA("46") VA("49") ".endr\n\t"
                   ".rept 8\n\t" M("0:3", "0:3", "8:11") M("4:7", "0:3", "16:19") M("8:11", "0:3", "24:27") ".endr\n\t" ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","v44","v45","v46","v47","v48","v49");
    } else if constexpr (KIND == 5) {  // 24 VALU, three per gap in 8 gaps
      asm volatile(".rept 8\n\t" M("0:3", "0:3", "8:11") VA("44") VA("47") VA("45") ".endr\n\t"
                   ".rept 28\n\t" M("4:7", "0:3", "16:19") ".endr\n\t" ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","v44","v45","v46","v47","v48","v49");
    } else if constexpr (KIND == 6) {  // 8 reads (own gaps) + 24 VALU (own gaps) + barrier
      asm volatile(
          M("0:3", "0:3", "8:11") DR("44:47", "0") M("4:7", "0:3", "16:19") DR("48:51", "1024") M("8:11", "0:3", "24:27") DR("52:55", "2048")
          M("0:3", "0:3", "8:11") DR("56:59", "3072") M("4:7", "0:3", "16:19") DR("60:63", "4096") M("8:11", "0:3", "24:27") DR("64:67", "5120")
          M("0:3", "0:3", "8:11") DR("68:71", "6144") M("4:7", "0:3", "16:19") DR("72:75", "7168") M("8:11", "0:3", "24:27")
          ".rept 8\n\t" M("0:3", "0:3", "8:11") VA("76") M("4:7", "0:3", "16:19") VA("77") M("8:11", "0:3", "24:27") VA("78") ".endr\n\t"
          "s_waitcnt lgkmcnt(0)\n\ts_barrier\n\t"
          M("0:3", "0:3", "8:11") M("4:7", "0:3", "16:19") M("8:11", "0:3", "24:27")
          ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78");
    } else if constexpr (KIND == 7) {  // 16 x ds_read_b64 instead of 8 x b128, one per gap
      asm volatile(
          ".rept 16\n\t" M("0:3", "0:3", "8:11") "ds_read_b64 v[44:45], v40 offset:512\n\t" ".endr\n\t"
          ".rept 20\n\t" M("4:7", "0:3", "16:19") ".endr\n\t"
          "s_waitcnt lgkmcnt(0)\n\t" ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","v44","v45");
    } else if constexpr (KIND == 8) {  // the 8 reads before the MFMAs (no overlap at all)
      asm volatile(
          DR("44:47", "0") DR("48:51", "1024") DR("52:55", "2048") DR("56:59", "3072") DR("60:63", "4096") DR("64:67", "5120") DR("68:71", "6144") DR("72:75", "7168")
          ".rept 12\n\t" M("0:3", "0:3", "8:11") M("4:7", "0:3", "16:19") M("8:11", "0:3", "24:27") ".endr\n\t"
          "s_waitcnt lgkmcnt(0)\n\t" ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\t" : "=s"(t1));
  if (threadIdx.x == 0 && blockIdx.x == 0) out[KIND] = (long long)(t1 - t0);
}

int main() {
  long long* dev; (void)hipMalloc(&dev, 64 * 8); (void)hipMemset(dev, 0, 64 * 8);
  const int iters = 256;
char* src; (void)hipMalloc(&src, (size_t)16 << 20); (void)hipMemset(src, 0, (size_t)16 << 20);
#define RUN(K) hipLaunchKernelGGL((k<K>), dim3(256), dim3(256), 65536, 0, dev, iters, src);
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10)
  long long h[64]; (void)hipMemcpy(h, dev, 64 * 8, hipMemcpyDeviceToHost);
  const char* names[] = {"36 MFMA", "+ 8 ds_read_b128, one per gap", "+ 8 ds_read_b128, two per gap", "+ 24 VALU, one per gap", "+ 24 VALU, two per gap",
                         "+ 24 VALU, three per gap", "+ 8 ds_read_b128 + 24 VALU (own gaps) + barrier", "+ 16 ds_read_b64, one per gap", "8 ds_read_b128 in front, then 36 MFMA",
                         "+ 8 ds_read + 24 VALU + barrier + 2 LDS-DMA (9 KiB stage / CU)", "the same without the barrier"};
  for (int i = 0; i < 11; ++i) printf("%-52s %8.1f cycles per stage (36 MFMAs) = %.2f per MFMA\n", names[i], (double)h[i] / iters, (double)h[i] / iters / 36.0);
  return 0;
}
