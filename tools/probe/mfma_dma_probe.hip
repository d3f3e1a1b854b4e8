// Hardware probe (gfx950): one FFN-like weight stage (36 MFMAs, 8 ds_read_b128 of the stage's tiles, 24 VALU ops) with the
// weight stream attached: every wave moves 2 KiB of the next free ring slot by LDS-DMA, a counted vmcnt wait and a
// workgroup barrier per stage.  Which arrangement of wait / barrier / DMA costs least?  (bare stage: 600 cycles)
#include <hip/hip_runtime.h>
#include <cstdio>

#define M(acc, a, b) "v_mfma_f32_16x16x32_f16 a[" acc "], v[" a "], v[" b "], a[" acc "]\n\t"
#define M3 M("0:3", "0:3", "8:11") M("4:7", "0:3", "16:19") M("8:11", "0:3", "24:27")
#define DR(dst, off) "ds_read_b128 v[" dst "], v40 offset:" off "\n\t"
#define VA(d) "v_fma_f32 v" d ", v41, v42, v43\n\t"
#define CLOB "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","memory"
#define READS8 M("0:3", "0:3", "8:11") DR("44:47", "0") M("4:7", "0:3", "16:19") DR("48:51", "1024") M("8:11", "0:3", "24:27") DR("52:55", "2048") \
               M("0:3", "0:3", "8:11") DR("56:59", "3072") M("4:7", "0:3", "16:19") DR("60:63", "4096") M("8:11", "0:3", "24:27") DR("64:67", "5120") \
               M("0:3", "0:3", "8:11") DR("68:71", "6144") M("4:7", "0:3", "16:19") DR("72:75", "7168") M("8:11", "0:3", "24:27")
#define VALU3 M("0:3", "0:3", "8:11") VA("76") M("4:7", "0:3", "16:19") VA("77") M("8:11", "0:3", "24:27") VA("78")

// RING: slots (stage = 9 KiB); INFLIGHT: stages that may still be in flight at the wait (vmcnt = 2 * INFLIGHT)
// PLACE 0: wait+barrier after 24 MFMAs, DMA right behind the barrier (the kernel's arrangement)
//       1: wait+barrier at the END of the stage, DMA at the start of the next
//       2: no barrier at all (incorrect in a real kernel; what the barrier costs)
//       3: as 0, but the two DMA instructions of a wave 6 MFMAs apart
//       4: as 0, one dwordx4 + the second KiB as 4 x dword? (no) -> as 0 with the DMA BEFORE the barrier of the NEXT stage's wait (late)
//       5: as 0 with s_setprio 3 around the MFMAs? -> raise wave priority during the stage
template <int RING, int INFLIGHT, int PLACE>
__global__ void __launch_bounds__(256) k(long long* out, int idx, int iters, const char* src) {
  extern __shared__ char lds[];
  unsigned long long t0 = 0, t1 = 0;
  for (int i = threadIdx.x; i < 16384; i += 256) ((float*)lds)[i] = 0.f;
  __syncthreads();
  asm volatile("v_mbcnt_lo_u32_b32 v40, -1, 0\n\tv_mbcnt_hi_u32_b32 v40, -1, v40\n\tv_lshlrev_b32 v40, 4, v40\n\t"
               "v_mov_b32 v41, 1.0\n\tv_mov_b32 v42, 0.5\n\tv_mov_b32 v43, 0.25" ::: "v40", "v41", "v42", "v43");
  const char* gp = src + (threadIdx.x >> 6) * 2048 + (threadIdx.x & 63) * 16 + (size_t)(blockIdx.x & 1) * ((size_t)8 << 20);
  const unsigned ring = 16384 + (threadIdx.x >> 6) * 2048;
  int slot = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\t" : "=s"(t0));
  for (int it = 0; it < iters; ++it) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane(ring + slot * 9216);
    slot = slot == RING - 1 ? 0 : slot + 1;
    const char* g = gp + (size_t)(it % 900) * 9216;
#define WAITSTR(n) "s_waitcnt vmcnt(" #n ") lgkmcnt(0)\n\t"
#define STAGE(W) \
    if constexpr (PLACE == 0) \
      asm volatile(READS8 VALU3 VALU3 VALU3 VALU3 VALU3 W "s_barrier\n\t" \
                   M("0:3", "0:3", "8:11") "s_mov_b32 m0, %1\n\t" M("4:7", "0:3", "16:19") "global_load_lds_dwordx4 %0, off\n\t" M("8:11", "0:3", "24:27") "global_load_lds_dwordx4 %0, off offset:1024\n\t" \
                   VALU3 VALU3 VALU3 :: "v"(g), "s"(m0v) : CLOB); \
    else if constexpr (PLACE == 1) \
      asm volatile("s_mov_b32 m0, %1\n\t" M("0:3", "0:3", "8:11") "global_load_lds_dwordx4 %0, off\n\t" M("4:7", "0:3", "16:19") "global_load_lds_dwordx4 %0, off offset:1024\n\t" M("8:11", "0:3", "24:27") \
                   READS8 VALU3 VALU3 VALU3 VALU3 VALU3 VALU3 VALU3 W "s_barrier\n\t" :: "v"(g), "s"(m0v) : CLOB); \
    else if constexpr (PLACE == 2) \
      asm volatile(READS8 VALU3 VALU3 VALU3 VALU3 VALU3 W \
                   M("0:3", "0:3", "8:11") "s_mov_b32 m0, %1\n\t" M("4:7", "0:3", "16:19") "global_load_lds_dwordx4 %0, off\n\t" M("8:11", "0:3", "24:27") "global_load_lds_dwordx4 %0, off offset:1024\n\t" \
                   VALU3 VALU3 VALU3 :: "v"(g), "s"(m0v) : CLOB); \
    else if constexpr (PLACE == 3) \
      asm volatile(READS8 VALU3 VALU3 VALU3 VALU3 VALU3 W "s_barrier\n\t" \
                   M("0:3", "0:3", "8:11") "s_mov_b32 m0, %1\n\t" M("4:7", "0:3", "16:19") "global_load_lds_dwordx4 %0, off\n\t" M("8:11", "0:3", "24:27") \
                   VALU3 M("0:3", "0:3", "8:11") "global_load_lds_dwordx4 %0, off offset:1024\n\t" M("4:7", "0:3", "16:19") M("8:11", "0:3", "24:27") VALU3 :: "v"(g), "s"(m0v) : CLOB); \
    else if constexpr (PLACE == 5) \
      asm volatile("s_setprio 3\n\t" READS8 VALU3 VALU3 VALU3 VALU3 VALU3 W "s_barrier\n\t" \
                   M("0:3", "0:3", "8:11") "s_mov_b32 m0, %1\n\t" M("4:7", "0:3", "16:19") "global_load_lds_dwordx4 %0, off\n\t" M("8:11", "0:3", "24:27") "global_load_lds_dwordx4 %0, off offset:1024\n\t" \
                   VALU3 VALU3 VALU3 :: "v"(g), "s"(m0v) : CLOB);
    if constexpr (INFLIGHT == 1) { STAGE(WAITSTR(2)) }
    else if constexpr (INFLIGHT == 2) { STAGE(WAITSTR(4)) }
    else if constexpr (INFLIGHT == 3) { STAGE(WAITSTR(6)) }
    else { STAGE(WAITSTR(8)) }
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\t" : "=s"(t1));
  if (threadIdx.x == 0 && blockIdx.x == 0) out[idx] = (long long)(t1 - t0);
}

int main() {
  long long* dev; (void)hipMalloc(&dev, 64 * 8); (void)hipMemset(dev, 0, 64 * 8);
  char* src; (void)hipMalloc(&src, (size_t)32 << 20); (void)hipMemset(src, 0, (size_t)32 << 20);
  const int iters = 768;
  int n = 0;
  const char* names[32];
#define RUN(R, F, P, NAME) names[n] = NAME; hipLaunchKernelGGL((k<R, F, P>), dim3(256), dim3(256), 16384 + R * 9216, 0, dev, n, iters, src); ++n;
  RUN(5, 3, 0, "ring 5, 3 stages in flight, wait+barrier mid-stage, DMA behind the barrier (kernel)")
  RUN(5, 3, 2, "  ... without the barrier")
  RUN(5, 3, 1, "  ... wait+barrier at the end of the stage")
  RUN(5, 3, 3, "  ... the wave's two DMA instructions 6 MFMAs apart")
  RUN(5, 3, 5, "  ... s_setprio 3")
  RUN(5, 4, 0, "ring 5, 4 stages in flight")
  RUN(5, 2, 0, "ring 5, 2 stages in flight")
  RUN(3, 2, 0, "ring 3, 2 stages in flight")
  RUN(3, 1, 0, "ring 3, 1 stage in flight")
  RUN(7, 5, 0, "ring 7, 5 stages in flight (4 not possible: vmcnt template)")
  long long h[64]; (void)hipMemcpy(h, dev, 64 * 8, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("%-90s %7.1f cycles per stage = %.2f per MFMA\n", names[i], (double)h[i] / iters, (double)h[i] / iters / 36.0);
  return 0;
}
