// Hardware probe (gfx950) for the single-MFMA ("h1") variant of the split-fp16 kernel: a weight stage whose tile pairs
// are fp16 hi tiles only (1 KiB per 16x32 tile) and whose products are ONE v_mfma_f32_16x16x32_f16 per token tile -
// 3 MFMAs per tile against the 9 of the three-term split.  Compute per streamed byte drops 3x / 1.5x, so the question
// is what bounds a stage then: the matrix pipe (16 cycles per MFMA), the LDS-DMA stream, the tile reads (every wave
// reads every tile: 4 KiB of ds_read_b128 per KiB streamed) or the per-stage barrier.
//   NP    tile pairs per stage (4: 12 MFMAs, 8: 24 MFMAs, 16: 48 MFMAs per barrier)
//   MODE  0 DMA + counted wait + barrier in the middle of the stage (the h3 kernel's arrangement)
//         1 no DMA (ring keeps its content)      2 DMA, no barrier (incorrect in a real kernel: what the barrier costs)
//         3 global_load_dwordx4 -> VGPRs -> ds_write_b128 instead of LDS-DMA (+ barrier)
//         4 as 0 with the `nt` policy on the DMA
//   VALU  VALU fillers per stage (the h1 FFN epilogue: 10 ops per unit, 60 per chunk of 4 stages)
// Prints cycles per stage (s_memtime, block 0) and the launch's wall time (all 256 CUs busy, zero weights).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define M(acc, a, b) "v_mfma_f32_16x16x32_f16 a[" acc "], v[" a "], v[" b "], a[" acc "]\n\t"
#define DR(dst, off) "ds_read_b128 v[" dst "], %[tile] offset:" off "\n\t"
#define VA(d) "v_fma_f32 v" d ", v41, v42, v43\n\t"
// one tile pair: its read (into a rotating set of 4 slots) + 3 MFMAs + up to one VALU filler
#define P0(off, va) M("0:3", "44:47", "8:11") DR("44:47", off) M("4:7", "44:47", "16:19") va M("8:11", "44:47", "24:27")
#define P1(off, va) M("12:15", "48:51", "8:11") DR("48:51", off) M("16:19", "48:51", "16:19") va M("20:23", "48:51", "24:27")
#define P2(off, va) M("0:3", "52:55", "8:11") DR("52:55", off) M("4:7", "52:55", "16:19") va M("8:11", "52:55", "24:27")
#define P3(off, va) M("12:15", "56:59", "8:11") DR("56:59", off) M("16:19", "56:59", "16:19") va M("20:23", "56:59", "24:27")
#define CLOB "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23", \
             "v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","memory"

template <int NP, int MODE, int VALU, int RING, int INFLIGHT>
__global__ void __launch_bounds__(256) k(long long* out, int idx, int iters, const char* src) {
  extern __shared__ char lds[];
  unsigned long long t0 = 0, t1 = 0;
  for (int i = threadIdx.x; i < (16384 + RING * NP * 1024) / 4; i += 256) ((float*)lds)[i] = 0.f;
  __syncthreads();
  asm volatile("v_mov_b32 v41, 1.0\n\tv_mov_b32 v42, 0.5\n\tv_mov_b32 v43, 0.25" ::: "v41", "v42", "v43");
  constexpr int STAGE = NP * 1024, SHARE = STAGE / 4;      // bytes per stage / per wave
  constexpr int NDMA = SHARE / 1024;                        // DMA instructions per wave and stage
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const char* gp = src + wave * SHARE + lane * 16 + (size_t)(blockIdx.x & 1) * ((size_t)8 << 20);
  const unsigned ring = 16384;
  const unsigned lane16 = lane * 16;
  int slot = 0, rslot = 0;
  const int wrap = (4 << 20) / STAGE;                       // the hi-only stream of one net: 4.2 MiB
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  u4 srd;                                                    // raw buffer over the whole source (MODE 6)
  srd.x = (unsigned)(size_t)src; srd.y = (unsigned)((size_t)src >> 32); srd.z = 32u << 20; srd.w = 0x00020000;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\t" : "=s"(t0));
  for (int it = 0; it < iters; ++it) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane(ring + slot * STAGE + wave * SHARE);
    slot = slot == RING - 1 ? 0 : slot + 1;
    const unsigned tile = ring + rslot * STAGE + lane16;
    rslot = rslot == RING - 1 ? 0 : rslot + 1;
    const char* g = gp + (size_t)(it % wrap) * STAGE;
#define VF(n) ((VALU) > (n) ? VA("76") : "")
    // first half of the stage's pairs, wait + barrier, DMA, second half
    if constexpr (NP == 4) {
      asm volatile(P0("0", "") P1("1024", "") :: [tile] "v"(tile) : CLOB);
    } else if constexpr (NP == 8) {
      asm volatile(P0("0", "") P1("1024", "") P2("2048", "") P3("3072", "") :: [tile] "v"(tile) : CLOB);
    } else {
      asm volatile(P0("0", "") P1("1024", "") P2("2048", "") P3("3072", "") P0("4096", "") P1("5120", "") P2("6144", "") P3("7168", "") :: [tile] "v"(tile) : CLOB);
    }
    if constexpr (VALU >= 8) asm volatile(VA("76") VA("77") VA("78") VA("79") ::: "v76", "v77", "v78", "v79");
    if constexpr (MODE == 0 || MODE == 4 || MODE == 2 || MODE == 5 || MODE == 6) {
      if constexpr (INFLIGHT * NDMA == 1) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
      else if constexpr (INFLIGHT * NDMA == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
      else if constexpr (INFLIGHT * NDMA == 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
      else if constexpr (INFLIGHT * NDMA == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else if constexpr (INFLIGHT * NDMA == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
      else if constexpr (INFLIGHT * NDMA == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if constexpr (MODE != 2) asm volatile("s_barrier" ::: "memory");
    if constexpr (MODE == 0 || MODE == 2) {
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g), "s"(m0v) : "memory");
      if constexpr (NDMA >= 2) asm volatile("global_load_lds_dwordx4 %0, off offset:1024" :: "v"(g) : "memory");
      if constexpr (NDMA >= 4) asm volatile("global_load_lds_dwordx4 %0, off offset:2048\n\tglobal_load_lds_dwordx4 %0, off offset:3072" :: "v"(g) : "memory");
    } else if constexpr (MODE == 5) {
      // saddr form: uniform 64-bit base in an SGPR pair + one 32-bit lane offset (half the address registers per piece)
      const char* gs = src + (size_t)(blockIdx.x & 1) * ((size_t)8 << 20) + (size_t)(it % wrap) * STAGE;
      const unsigned voff = wave * SHARE + lane * 16;
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" :: "v"(voff), "s"(m0v), "s"(gs) : "memory");
      if constexpr (NDMA >= 2) asm volatile("global_load_lds_dwordx4 %0, %1 offset:1024" :: "v"(voff), "s"(gs) : "memory");
      if constexpr (NDMA >= 4) asm volatile("global_load_lds_dwordx4 %0, %1 offset:2048\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072" :: "v"(voff), "s"(gs) : "memory");
    } else if constexpr (MODE == 6) {
      // buffer form: SRD in four SGPRs, lane offset in a VGPR, stage offset in an SGPR (soffset)
      const unsigned voff = wave * SHARE + lane * 16;
      const unsigned soff = (unsigned)((size_t)(blockIdx.x & 1) * ((size_t)8 << 20) + (size_t)(it % wrap) * STAGE);
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds" :: "v"(voff), "s"(m0v), "s"(srd), "s"(soff) : "memory");
      if constexpr (NDMA >= 2) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:1024 lds" :: "v"(voff), "s"(srd), "s"(soff) : "memory");
    } else if constexpr (MODE == 4) {
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" :: "v"(g), "s"(m0v) : "memory");
      if constexpr (NDMA >= 2) asm volatile("global_load_lds_dwordx4 %0, off offset:1024 nt" :: "v"(g) : "memory");
      if constexpr (NDMA >= 4) asm volatile("global_load_lds_dwordx4 %0, off offset:2048 nt\n\tglobal_load_lds_dwordx4 %0, off offset:3072 nt" :: "v"(g) : "memory");
    }
    if constexpr (NP == 4) {
      asm volatile(P2("2048", "") P3("3072", "") :: [tile] "v"(tile) : CLOB);
    } else if constexpr (NP == 8) {
      asm volatile(P0("4096", "") P1("5120", "") P2("6144", "") P3("7168", "") :: [tile] "v"(tile) : CLOB);
    } else {
      asm volatile(P0("8192", "") P1("9216", "") P2("10240", "") P3("11264", "") P0("12288", "") P1("13312", "") P2("14336", "") P3("15360", "") :: [tile] "v"(tile) : CLOB);
    }
    if constexpr (VALU >= 4) asm volatile(VA("76") VA("77") VA("78") VA("79") ::: "v76", "v77", "v78", "v79");
    if constexpr (VALU >= 16) asm volatile(VA("76") VA("77") VA("78") VA("79") VA("76") VA("77") VA("78") VA("79") ::: "v76", "v77", "v78", "v79");
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\t" : "=s"(t1));
  if (threadIdx.x == 0 && blockIdx.x == 0) out[idx] = (long long)(t1 - t0);
}

// MODE 3: through registers.  Three register buffers (one stage share each, NP / 4 x 4 VGPRs), unrolled by three.
template <int NP, int VALU, int RING>
__global__ void __launch_bounds__(256) kreg(long long* out, int idx, int iters, const char* src) {
  extern __shared__ char lds[];
  unsigned long long t0 = 0, t1 = 0;
  for (int i = threadIdx.x; i < (16384 + RING * NP * 1024) / 4; i += 256) ((float*)lds)[i] = 0.f;
  __syncthreads();
  asm volatile("v_mov_b32 v41, 1.0\n\tv_mov_b32 v42, 0.5\n\tv_mov_b32 v43, 0.25" ::: "v41", "v42", "v43");
  constexpr int STAGE = NP * 1024, SHARE = STAGE / 4, NDMA = SHARE / 1024;
  static_assert(NDMA <= 2, "register buffers sized for <= 2 KiB per wave and stage");
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const char* gp = src + wave * SHARE + lane * 16 + (size_t)(blockIdx.x & 1) * ((size_t)8 << 20);
  const unsigned ring = 16384, lane16 = lane * 16;
  int slot = 0, rslot = 0;
  const int wrap = (4 << 20) / STAGE;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\t" : "=s"(t0));
  for (int it = 0; it < iters; it += 3) {
#define REGSTAGE(B0, B1, NB0, NB1)                                                                                        \
    {                                                                                                                     \
      const unsigned wr = ring + slot * STAGE + wave * SHARE + lane16;                                                    \
      slot = slot == RING - 1 ? 0 : slot + 1;                                                                             \
      const unsigned tile = ring + rslot * STAGE + lane16;                                                                \
      rslot = rslot == RING - 1 ? 0 : rslot + 1;                                                                          \
      const char* g = gp + (size_t)(itv % wrap) * STAGE;                                                                  \
      if constexpr (NP == 4) asm volatile(P0("0", "") P1("1024", "") :: [tile] "v"(tile) : CLOB);                         \
      else asm volatile(P0("0", "") P1("1024", "") P2("2048", "") P3("3072", "") :: [tile] "v"(tile) : CLOB);             \
      /* the share loaded two stages ago has landed: park it in the LDS, then ask for the share three stages ahead */     \
      if constexpr (NDMA == 1) asm volatile("s_waitcnt vmcnt(1)\n\tds_write_b128 %0, v[" B0 "]\n\t" :: "v"(wr) : CLOB); \
      else asm volatile("s_waitcnt vmcnt(2)\n\tds_write_b128 %0, v[" B0 "]\n\tds_write_b128 %0, v[" B1 "] offset:1024\n\t" :: "v"(wr) : CLOB); \
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                                     \
      asm volatile("global_load_dwordx4 v[" NB0 "], %0, off" :: "v"(g) : CLOB);                                       \
      if constexpr (NDMA == 2) asm volatile("global_load_dwordx4 v[" NB1 "], %0, off offset:1024" :: "v"(g) : CLOB);  \
      if constexpr (NP == 4) asm volatile(P2("2048", "") P3("3072", "") :: [tile] "v"(tile) : CLOB);                      \
      else asm volatile(P0("4096", "") P1("5120", "") P2("6144", "") P3("7168", "") :: [tile] "v"(tile) : CLOB);          \
      if constexpr (VALU >= 4) asm volatile(VA("76") VA("77") VA("78") VA("79") ::: "v76", "v77", "v78", "v79");          \
      if constexpr (VALU >= 16) asm volatile(VA("76") VA("77") VA("78") VA("79") VA("76") VA("77") VA("78") VA("79") ::: "v76", "v77", "v78", "v79"); \
    }
    { const int itv = it;     REGSTAGE("80:83", "84:87", "96:99", "100:103") }
    { const int itv = it + 1; REGSTAGE("88:91", "92:95", "80:83", "84:87") }
    { const int itv = it + 2; REGSTAGE("96:99", "100:103", "88:91", "92:95") }
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\t" : "=s"(t1) :: "v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103");
  if (threadIdx.x == 0 && blockIdx.x == 0) out[idx] = (long long)(t1 - t0);
}

int main() {
  long long* dev; (void)hipMalloc(&dev, 64 * 8); (void)hipMemset(dev, 0, 64 * 8);
  char* src; (void)hipMalloc(&src, (size_t)32 << 20); (void)hipMemset(src, 0, (size_t)32 << 20);
  const int iters = 3072;
  int n = 0;
  const char* names[64]; int nmfma[64]; float ms[64];
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
#define RUN(NP, MODE, VALU, RING, FL, NAME) names[n] = NAME; nmfma[n] = 3 * NP; (void)hipEventRecord(e0); \
    hipLaunchKernelGGL((k<NP, MODE, VALU, RING, FL>), dim3(256), dim3(256), 16384 + RING * NP * 1024, 0, dev, n, iters * 4 / NP, src); \
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms[n], e0, e1); ++n;
#define RUNREG(NP, VALU, RING, NAME) names[n] = NAME; nmfma[n] = 3 * NP; (void)hipEventRecord(e0); \
    hipLaunchKernelGGL((kreg<NP, VALU, RING>), dim3(256), dim3(256), 16384 + RING * NP * 1024, 0, dev, n, iters * 4 / NP, src); \
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms[n], e0, e1); ++n;
  RUN(4, 1, 0, 5, 3, "warm-up")
  RUN(4, 1, 0, 5, 3, "4 pairs/stage (12 MFMAs), no DMA, barrier")
  RUN(4, 0, 0, 5, 3, "4 pairs/stage, LDS-DMA 4 KiB/stage, barrier, ring 5, 3 in flight")
  RUN(4, 2, 0, 5, 3, "  ... no barrier")
  RUN(4, 4, 0, 5, 3, "  ... nt policy")
  RUN(4, 0, 4, 5, 3, "  ... + 4 VALU fillers")
  RUN(4, 0, 16, 5, 3, "  ... + 16 VALU fillers")
  RUN(4, 0, 0, 9, 6, "  ... ring 9, 6 in flight")
  RUN(4, 0, 0, 9, 3, "  ... ring 9, 3 in flight")
  RUNREG(4, 0, 5, "4 pairs/stage, global_load -> VGPR -> ds_write, barrier, ring 5")
  RUN(8, 1, 0, 5, 3, "8 pairs/stage (24 MFMAs), no DMA, barrier")
  RUN(8, 0, 0, 5, 3, "8 pairs/stage, LDS-DMA 8 KiB/stage, barrier, ring 5, 3 in flight")
  RUN(8, 2, 0, 5, 3, "  ... no barrier")
  RUN(8, 5, 0, 5, 3, "  ... saddr form (SGPR base + 32-bit lane offset)")
  RUN(8, 6, 0, 5, 3, "  ... buffer_load ... lds (SRD + soffset)")
  RUN(8, 0, 16, 5, 3, "  ... + 16 VALU fillers")
  RUN(8, 0, 0, 5, 2, "  ... 2 in flight")
  RUN(8, 0, 0, 9, 6, "  ... ring 9, 6 in flight")
  RUNREG(8, 0, 5, "8 pairs/stage, global_load -> VGPR -> ds_write, barrier, ring 5")
  RUN(16, 1, 0, 5, 3, "16 pairs/stage (48 MFMAs), no DMA, barrier")
  RUN(16, 0, 0, 5, 3, "16 pairs/stage, LDS-DMA 16 KiB/stage, barrier, ring 5, 3 in flight")
  RUN(16, 0, 0, 5, 2, "  ... 2 in flight")
  long long h[64]; (void)hipMemcpy(h, dev, 64 * 8, hipMemcpyDeviceToHost);
  for (int i = 1; i < n; ++i) {
    const double st = (double)h[i] / (iters * 12 / nmfma[i]);
    printf("%-78s %7.1f cyc/stage  %5.2f per MFMA  %5.2f B/cyc/CU  %6.3f ms  %.2f GHz\n", names[i], st, st / nmfma[i],
           nmfma[i] / 3 * 1024.0 / st, ms[i], (double)h[i] / ms[i] * 1e-6);
  }
  return 0;
}
