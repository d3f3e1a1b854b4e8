// Shared pieces of the f32-MFMA fused net-block kernels (tw_netblock.hip: kernel attention; tw_netblock_dense.hip:
// dense softmax attention): weight-tile packing kernels, the register weight ring, the chained MLP stage and the
// LayerNorm in MFMA D/B layout.  Everything has internal linkage (included by two translation units).
#pragma once
#include "tw_common.h"

namespace tw {
namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

#define XS 144          // LDS row stride (floats): 144 % 32 == 16 -> conflict-free A-fragment reads
#define RING 8          // weight tiles in flight per wave
#define TILE_F 256      // floats per weight tile (64 lanes x float4)

// tiles ordered ot-major: tile (ot,ft) element (lane,r) = src[row0+16ot+(lane&15)][col0+16ft+4(lane>>4)+r]
__global__ void pack_block_kernel(const float* __restrict__ src, int ld, int rows_valid, int cols_valid, int row0,
                                  int col0, int n_ft, float* __restrict__ dst) {
  const int ot = blockIdx.x, ft = blockIdx.y, lane = threadIdx.x;
  const int row = row0 + 16 * ot + (lane & 15);
  float* o = dst + ((int64_t)(ot * n_ft + ft) * 64 + lane) * 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int col = col0 + 16 * ft + 4 * (lane >> 4) + r;
    o[r] = (row < rows_valid && col < cols_valid) ? src[(int64_t)row * ld + col] : 0.f;
  }
}

__global__ void copy_pad_kernel(const float* __restrict__ src, int n, float* __restrict__ dst, int n_pad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_pad) dst[i] = i < n ? src[i] : 0.f;
}

static int pack_block(const float* src, int ld, int rows_valid, int cols_valid, int row0, int col0, int n_ot, int n_ft,
                      float* dst, hipStream_t s) {
  hipLaunchKernelGGL(pack_block_kernel, dim3(n_ot, n_ft), dim3(64), 0, s, src, ld, rows_valid, cols_valid, row0, col0,
                     n_ft, dst);
  TW_LAUNCH_CHECK();
  return TW_OK;
}
static int copy_pad(const float* src, int n, float* dst, int n_pad, hipStream_t s) {
  hipLaunchKernelGGL(copy_pad_kernel, dim3((n_pad + 255) / 256), dim3(256), 0, s, src, n, dst, n_pad);
  TW_LAUNCH_CHECK();
  return TW_OK;
}

__device__ __forceinline__ f4 mfma4(float a, float b, f4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

template <int NT>
__device__ __forceinline__ void tile_mma(const f4 a, const f4 (&b)[NT], f4 (&acc)[NT]) {
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) acc[jt] = mfma4(a[r], b[jt][r], acc[jt]);
}

#define TW_PIN() __builtin_amdgcn_sched_barrier(0)

// consume ring slot (T % RING) for tile T of the current body, then refill it with tile T + RING
#define RING_LOAD(T) ring[(T) % RING] = *(const f4*)(wp + (int64_t)((T) + RING) * TILE_F)

// MLP chain stage:  y[OT_OUT] (+)= W2 . act(W0 . xin + b0)   32 hidden units per chunk
//   xin [FT_IN][NT] B-layout;  yacc pre-initialised by the caller (bias / zero)
template <int NT, int FT_IN, int OT_OUT, int BODY_TILES, bool SILU>
__device__ __forceinline__ void mlp_chain(const f4 (&xin)[FT_IN][NT], f4 (&yacc)[OT_OUT][NT], const float*& wp,
                                          f4 (&ring)[RING], const float* bias_lane, int n_chunks) {
  // bias_lane already points at b0 + 4*g; chunk c, sub-tile o: bias_lane[32c + 16o .. +3]
  f4 bc0 = *(const f4*)(bias_lane);
  f4 bc1 = *(const f4*)(bias_lane + 16);
  for (int c = 0; c < n_chunks; ++c) {
    const f4 bn0 = *(const f4*)(bias_lane + 32 * (c + 1));       // next chunk (side array has slack)
    const f4 bn1 = *(const f4*)(bias_lane + 32 * (c + 1) + 16);
    f4 h[2][NT];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) { h[0][jt] = bc0; h[1][jt] = bc1; }
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int ft = 0; ft < FT_IN; ++ft) {
        const int T = o * FT_IN + ft;
        const f4 a = ring[T % RING];
        tile_mma<NT>(a, xin[ft], h[o]);
        RING_LOAD(T);
        TW_PIN();
      }
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = h[o][jt][r];
          h[o][jt][r] = SILU ? v / (1.f + expf(-v)) : fmaxf(v, 0.f);
        }
#pragma unroll
    for (int ot = 0; ot < OT_OUT; ++ot)
#pragma unroll
      for (int f2 = 0; f2 < 2; ++f2) {
        const int T = 2 * FT_IN + ot * 2 + f2;
        const f4 a = ring[T % RING];
        tile_mma<NT>(a, h[f2], yacc[ot]);
        RING_LOAD(T);
        TW_PIN();
      }
#pragma unroll
    for (int T = 2 * FT_IN + 2 * OT_OUT; T < BODY_TILES; ++T) { RING_LOAD(T); }
    wp += (int64_t)BODY_TILES * TILE_F;
    bc0 = bn0;
    bc1 = bn1;
  }
}

__device__ __forceinline__ float xor16_32_sum(float v) {
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

// x <- LayerNorm(x + y) over the 128 features of every token (T-layout), weights from `lnw`/`lnb`
template <int NT>
__device__ __forceinline__ void add_layernorm(f4 (&x)[8][NT], const f4 (&y)[8][NT], const float* lnw_lane,
                                              const float* lnb_lane, float eps) {
  f4 w[8], b[8];
#pragma unroll
  for (int ft = 0; ft < 8; ++ft) {
    w[ft] = *(const f4*)(lnw_lane + 16 * ft);
    b[ft] = *(const f4*)(lnb_lane + 16 * ft);
  }
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) {
    float s = 0.f;
#pragma unroll
    for (int ft = 0; ft < 8; ++ft) {
      x[ft][jt] = x[ft][jt] + y[ft][jt];
      s += (x[ft][jt][0] + x[ft][jt][1]) + (x[ft][jt][2] + x[ft][jt][3]);
    }
    const float mean = xor16_32_sum(s) * (1.f / 128.f);
    float q = 0.f;
#pragma unroll
    for (int ft = 0; ft < 8; ++ft) {
      x[ft][jt] = x[ft][jt] - mean;
      q += (x[ft][jt][0] * x[ft][jt][0] + x[ft][jt][1] * x[ft][jt][1]) +
           (x[ft][jt][2] * x[ft][jt][2] + x[ft][jt][3] * x[ft][jt][3]);
    }
    const float var = xor16_32_sum(q) * (1.f / 128.f);
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int ft = 0; ft < 8; ++ft) x[ft][jt] = x[ft][jt] * rstd * w[ft] + b[ft];
  }
}

}  // namespace
}  // namespace tw
