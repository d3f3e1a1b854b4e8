// Internal declarations shared by the HIP translation units of libtimewarp_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>

#include "../../include/timewarp_hip.h"

namespace tw {

void set_error(const char* fmt, ...);

#define TW_HIP_CHECK(expr)                                                              \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      tw::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return TW_ERR_HIP;                                                                \
    }                                                                                   \
  } while (0)

#define TW_LAUNCH_CHECK() TW_HIP_CHECK(hipGetLastError())

#define TW_REQUIRE(cond, ...)        \
  do {                               \
    if (!(cond)) {                   \
      tw::set_error(__VA_ARGS__);    \
      return TW_ERR_INVALID;         \
    }                                \
  } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of a kernel ON ONE DEVICE: a process that drives several GPUs
// (tw_mh_iteration keeps per-device state for that) has to raise it on each.  One instance per kernel; bit = device ordinal.
struct LdsLimit {
  std::atomic<unsigned long long> done{0};
  int ensure(const void* fn, int bytes) {
    int dev = 0;
    TW_HIP_CHECK(hipGetDevice(&dev));
    const unsigned long long bit = dev < 64 ? 1ull << dev : 0ull;
    if (done.load(std::memory_order_acquire) & bit) return TW_OK;
    TW_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.fetch_or(bit, std::memory_order_release);
    return TW_OK;
  }
};

// Exclusion matrix of the energy / force kernels: one BIT per ordered atom pair in LDS (V = 691, the reference's test protein,
// is 60 KiB; a byte per pair stopped at ~240 atoms).
__host__ __device__ inline size_t excl_bytes(int V) { return ((size_t)V * V + 31) / 32 * 4; }
#ifdef __HIPCC__
__device__ __forceinline__ bool excl_test(const unsigned* bits, int idx) { return (bits[idx >> 5] >> (idx & 31)) & 1u; }
// all threads of the block (64: one wave, unless `nthreads` says otherwise); leaves the matrix complete after its last
// __syncthreads()
__device__ __forceinline__ void excl_fill(const int* exc_idx, int n_exceptions, int V, unsigned* bits, int lane, int nthreads = 64) {
  const int words = (int)(excl_bytes(V) / 4);
  for (int i = lane; i < words; i += nthreads) bits[i] = 0u;
  __syncthreads();
  for (int e = lane; e < n_exceptions; e += nthreads) {
    const int i = exc_idx[2 * e], j = exc_idx[2 * e + 1];
    atomicOr(&bits[(i * V + j) >> 5], 1u << ((i * V + j) & 31));
    atomicOr(&bits[(j * V + i) >> 5], 1u << ((j * V + i) & 31));
  }
  __syncthreads();
}
#endif

// ------------------------------------------------------------------------------------------
// Raw (canonical row-major) weight layout -- mirrored by timewarp_amd/weights.py
// ------------------------------------------------------------------------------------------
struct LayerOff {  // offsets relative to the start of one encoder layer
  int64_t wv, wo;                     // kernel: values_proj.w [H*d,d], out_projection.w [d,H*d]
  int64_t cheb;                       // kernel with cheb_order > 0: cheb_coeffs [H, order] (else -1)
  int64_t in_w, in_b, out_w, out_b;   // dense : in_proj [3d,d],[3d]; out_proj [d,d],[d]
  int64_t w1, b1, w2, b2, n1w, n1b, n2w, n2b;
  int64_t size;
};

struct NetOff {  // offsets relative to the start of one net (scale_transformer / shift_transformer)
  int64_t in0_w, in0_b, in2_w, in2_b;
  int64_t layers;  // first encoder layer
  int64_t out0_w, out0_b, out2_w, out2_b;
  int64_t size;
};

struct RawLayout {
  int d_in;
  int64_t emb, lengthscales, prior;  // prior: [coords_log_scale, velocs_log_scale]
  int64_t chain;                     // first coupling layer
  int64_t rff;                       // relative to coupling start (dense; [3, d_rff/2])
  int64_t nets;                      // relative to coupling start: scale net, then shift net
  int64_t coupling_size;
  LayerOff layer;
  NetOff net;
  int64_t total;
};

RawLayout raw_layout(const tw_flow_desc& d);

inline int64_t net_base(const RawLayout& L, int c, int net) {
  return L.chain + (int64_t)c * L.coupling_size + L.nets + (int64_t)net * L.net.size;
}

// ------------------------------------------------------------------------------------------
// Fused-path weight stream (tw_netblock.hip)
// ------------------------------------------------------------------------------------------
struct PackedLayout {
  // per net: a stream of 1 KiB tiles in consumption order, then small fp32 side arrays
  int64_t tiles_per_net;       // number of 256-float tiles
  int64_t side_per_net;        // floats: biases, LayerNorm weights
  int64_t net_stride;          // floats (tiles*256 + side, padded to 256)
  int64_t total;
};
PackedLayout packed_layout(const tw_flow_desc& d);

// geometry of the fused kernel for a given atom count
struct FusedGeom {
  int nt;        // 16-token tiles per wave (3 or 4)
  int mpw;       // molecules per wave
  int tile_mask; // bit (jt*nt+mt) set when score tile (jt,mt) can be non-zero
};
bool fused_geom(int n_atoms, FusedGeom* g);
// the layout with exactly `nt` token tiles per wave (the split-fp16 kernels run 48-token waves for every n_atoms <= 48)
bool fused_geom_nt(int n_atoms, int nt, FusedGeom* g);

// launchers implemented in the .hip files ----------------------------------------------------
// coeffs == nullptr / order == 0: Gaussian basis; else rational-Chebyshev basis with coeffs [H, order]
int launch_scores(const float* x, const uint8_t* masked, const float* ls, int H, int64_t B, int V,
                  int normalise, int use_mm, float* out, hipStream_t s, const float* coeffs = nullptr, int order = 0,
                  int force_zero = 0, _Float16* s_hi = nullptr, _Float16* s_lo = nullptr);
bool scores_split_direct(int V);   // launch_scores can write the split fp16 operand of the folded mixing itself (row-wise kernel)
int launch_centre(const float* x, const uint8_t* masked, float* xc, float* com, int64_t n, int V,
                  hipStream_t s);

struct FlowArgs {
  const tw_flow_desc* desc;
  const float* raw;
  const float* packed;
  const int32_t* atom_types;
  const float* x_coords;  // centred
  const float* x_velocs;
  const uint8_t* masked;
  int64_t n_cond;
  float* z_coords;
  float* z_velocs;
  float* delta_logp;
  int64_t n_rows;
  int n_atoms;
  int reverse;
  void* ws;
  int64_t ws_bytes;
  hipStream_t stream;
  int h1 = 0;  // split-fp16 translation unit only: the single-MFMA variant (TW_PATH_FUSED_H1); `packed` is then its stream
  int simple_h3 = 0;  // per-op path only: the linears as split-fp16 MFMA GEMMs (TW_PATH_SIMPLE_H3)
};
// torch.cdist's matmul formulation (what the reference gets above 25 atoms): the row [-2 x, |x|^2, 1] times
// the column [y, 1, |y|^2], clamp_min(0), sqrt.  The result is dominated by fp32 cancellation (|x|^2 + |y|^2 - 2 x.y with
// terms of O(1) for distances of O(0.1)), so it is only reproducible with the SAME sequence of roundings: torch's CPU sgemm
// accumulates the five products in k order with fused multiply-adds from zero, and the norms are x*x + y*y + z*z with every
// product and sum rounded separately (pow(2).sum(-1)) - checked bit for bit against torch.cdist on the 60-atom golden
// geometry (99.4 % of the 3600 entries identical, the rest one ulp).  Contraction is switched OFF for this function: hipcc's
// default (-ffp-contract=fast) fuses a product into a following sum wherever it likes after inlining, which alone moved a
// fifth of the entries by up to 7e-4 nm - and HIP's __fmul_rn / __fadd_rn are plain `*` / `+` (no OCML rounded operations in
// this ROCm), so they do not prevent it: r05 found the same source giving two different sequences in two kernels (36 539 of
// 135 000 scores of a 150-atom molecule differed between scores_kernel and scores_rows_kernel until the pragma went in).
__device__ __forceinline__ float tw_cdist_mm(float qx, float qy, float qz, float mx, float my, float mz) {
#pragma clang fp contract(off)
  const float qn = (qx * qx + qy * qy) + qz * qz;
  const float mn = (mx * mx + my * my) + mz * mz;
  float acc = (-2.f * qx) * mx;
  acc = __builtin_fmaf(-2.f * qy, my, acc);
  acc = __builtin_fmaf(-2.f * qz, mz, acc);
  acc = acc + qn;   // fma(qn, 1, acc)
  acc = acc + mn;   // fma(1, mn, acc)
  return sqrtf(fmaxf(acc, 0.f));
}

// basis value for scaled distance sc: Gaussian exp(-sc^2), or sum_c coeff[c] R_c(sc^2) with the three-term recursion
// of chebyshev_expansion (kernel_attention.py:37-66), evaluated in the same order as the reference's stacked terms
__device__ __forceinline__ float basis_value(float sc, const float* __restrict__ coeff, int order, float coeff_mean) {
  if (order <= 0) return expf(-(sc * sc));
  {
  // torch's sequence of roundings (kernel_attention.py:37-66): every product and difference of the recursion
  // 2.0 * rfactor * rcur - rprev is rounded separately (hipcc would fuse the last two into an FMA), and the contraction with
  // the coefficients (einsum -> sgemm, K = order) is a chain of fused multiply-adds from zero in c order.  The basis values
  // cancel (sum |c_k R_k| >> |sum|), so a different sequence of roundings shows at the 1e-5 level in log p(x~|y~).
  // (contraction off for the whole function, as in tw_cdist_mm: __fmul_rn / __fsub_rn are plain operators in this ROCm)
#pragma clang fp contract(off)
  const float x = sc * sc;
  const float rf = (x - 1.0f) / (x + 1.0f);
  const float rf2 = 2.0f * rf;
  float rprev = 1.0f, rcur = rf;
  float acc = (coeff[0] - coeff_mean) * rprev;
  if (order >= 2) acc = __builtin_fmaf(coeff[1] - coeff_mean, rcur, acc);
  for (int c = 2; c < order; ++c) {
    const float rnext = rf2 * rcur - rprev;
    acc = __builtin_fmaf(coeff[c] - coeff_mean, rnext, acc);
    rprev = rcur;
    rcur = rnext;
  }
  return acc;
  }
}

// Which basis the score-fragment producers of the fused paths evaluate.  order == 0: the Gaussian, one variant.
// chebyshev_kernel: every attention layer of both nets of a coupling layer owns its coefficients, so a coupling
// layer has 2 * n_layers variants; variant v = net * n_layers + layer reads coeff0 + net * net_stride + layer * layer_stride.
struct ScoreBasis {
  const float* coeff0;
  int64_t net_stride, layer_stride;  // floats
  int order, force_zero, n_layers, n_variants;
};
__device__ __forceinline__ const float* basis_coeffs(const ScoreBasis& b, int variant, int head, float* mean) {
  *mean = 0.f;
  if (b.order <= 0) return nullptr;
  const float* cf = b.coeff0 + (variant / b.n_layers) * b.net_stride + (variant % b.n_layers) * b.layer_stride + (int64_t)head * b.order;
  if (b.force_zero) {
    float m = 0.f;
    for (int c = 0; c < b.order; ++c) m += cf[c];
    *mean = m / (float)b.order;
  }
  return cf;
}
ScoreBasis score_basis(const tw_flow_desc& d, const RawLayout& L, const float* raw, int coupling);

int flow_pass_simple(const FlowArgs& a);
int flow_pass_fused(const FlowArgs& a);
int64_t simple_workspace_bytes(const tw_flow_desc& d, int64_t n_rows, int n_atoms);
int64_t fused_workspace_bytes(const tw_flow_desc& d, int64_t n_rows, int n_atoms);
// z_out = affine coupling of z_in (nullptr: in place on z_out)
int launch_coupling(const float* s_raw, const float* t, const uint8_t* masked, int64_t n_cond,
                    float* z_out, float* delta_logp, int64_t n_rows, int V, int reverse, hipStream_t s,
                    const float* z_in = nullptr, int* flag = nullptr);
int* nonfinite_flag_device_ptr();  // address of the per-device sticky non-finite flag (for kernels of other translation units)
// where the flow kernels of this call report a non-finite scale / shift: the descriptor's own word (ABI 7), else the device's
inline int* flow_range_flag(const tw_flow_desc& d) { return d.range_flag ? d.range_flag : nonfinite_flag_device_ptr(); }

// The affine coupling (layers/nvp.py:89-183) of the PREVIOUS coupling layer, applied by the next net-block launch in
// its prologue instead of by a launch of its own: the variable it transforms is exactly the next layer's conditioning
// input z_other, so every workgroup recomputes the values of its own rows from (s, t, z_in); the workgroups of net 0
// also store them (z_out - a different buffer than z_in, which the workgroups of net 1 are still reading) and apply
// the row's log-determinant to delta_logp.
struct PrevCoupling {
  const float* s_raw;  // nullptr: nothing pending, z_other is read from memory
  const float* t;
  const float* z_in;
  float* z_out;
  float* delta_logp;
  int* nonfinite;
  int reverse;
};
int pack_weights(const tw_flow_desc& d, const float* raw, float* packed, hipStream_t s);
int debug_netblock_simple(const FlowArgs& a, int c, int net, const float* z_other, float* dump);
int debug_netblock_fused(const FlowArgs& a, int c, int net, const float* z_other, float* dump);
// fused dense-softmax net-block (tw_netblock_dense.hip); reached through the five functions above when variant == 1
bool dense_fused_supported(const tw_flow_desc& d, int n_atoms);
PackedLayout dense_packed_layout(const tw_flow_desc& d);
int dense_pack_weights(const tw_flow_desc& d, const float* raw, float* packed, hipStream_t s);
int64_t dense_fused_workspace_bytes(const tw_flow_desc& d, int64_t n_rows, int n_atoms);
int flow_pass_fused_dense(const FlowArgs& a);
int debug_netblock_fused_dense(const FlowArgs& a, int c, int net, const float* z_other, float* dump);
// split-fp16 fused path (tw_netblock_h3.hip); FlowArgs::packed points at the h3 stream (bytes)
bool h3_supported(const tw_flow_desc& d, int n_atoms);
bool h1_supported(const tw_flow_desc& d, int n_atoms);  // single-MFMA variant of the same kernel (TW_PATH_FUSED_H1)
int64_t h3_packed_bytes(const tw_flow_desc& d, bool h1 = false);
int64_t h3_workspace_bytes(const tw_flow_desc& d, int64_t n_rows, int n_atoms);
int h3_pack_weights(const tw_flow_desc& d, const float* raw, char* packed, float* scratch, hipStream_t s, bool h1 = false);
int flow_pass_h3(const FlowArgs& a);
int debug_netblock_h3(const FlowArgs& a, int c, int net, const float* z_other, float* dump);
// tw_debug_set_flags: one process-wide word, read once per launch.  The bits that make results WRONG on purpose (timing
// experiments: 1, 2, 64, 128, 2048) exist only in a -DTW_EXPERIMENTS build (TW_EXPERIMENTS=1 python -m timewarp_amd.build);
// the product library refuses to set them and compiles the branches out.
extern std::atomic<int> g_debug_flags;
#define TW_WRONG_RESULT_BITS (1 | 2 | 64 | 128 | 2048)
#ifdef TW_EXPERIMENTS
#define TW_EXPERIMENT(x) (x)
#else
#define TW_EXPERIMENT(x) (false)
#endif
int nonfinite_flag(int reset, int* out);
int profile_mark(hipStream_t s, bool begin);
// the net-block kernel instantiation the calling thread launched last (a string literal; tw_last_netblock_kernel)
void note_netblock_kernel(const char* name);
const char* last_netblock_kernel();
int h3_selected_kernel(const tw_flow_desc& d, int n_atoms, int64_t n_rows, bool h1);  // dry run of the launch code's choice
// TW_PATH_SIMPLE_H3: h <- LayerNorm2(h + FFN(h)) of (coupling, net, layer) on a flat [n_tokens, 128] list, FFN weights from the
// tw_flow_pack_h3 stream (csrc/tw_netblock_h3.hip)
bool h3_ffn_tokens_supported(const tw_flow_desc& d);
int64_t simple_h3_split_offset(const tw_flow_desc& d);  // bytes in front of the split FFN streams in that buffer
int64_t simple_h3_fold_floats(const tw_flow_desc& d);   // tw_flow_pack_simple_h3: Wc of every (coupling, net, layer) behind the stream
int simple_h3_fold(const tw_flow_desc& d, const float* raw, float* out, hipStream_t s);
int h3_ffn_tokens(const tw_flow_desc& d, const void* packed, int coupling, int net, int layer, float* h, int64_t n_tokens,
                  hipStream_t stream, float* scratch, int64_t scratch_floats, const void* split_stages);
// small launches: the hidden layer over four workgroups per token tile - their streams (h3_ffn_split_pack, behind the folded Wc in the
// tw_flow_pack_simple_h3 buffer) and `scratch` for the partial sums
int64_t h3_ffn_split_bytes(const tw_flow_desc& d);
int h3_ffn_split_pack(const tw_flow_desc& d, const void* packed, void* dst, hipStream_t stream);
// ... and the in-MLP (u [n_tokens, d_in <= 64] -> h [n_tokens, 128]) / out-MLP (h -> o [n_tokens, 3]) of (coupling, net)
bool h3_io_tokens_supported(const tw_flow_desc& d);
int h3_io_tokens(const tw_flow_desc& d, const void* packed, int coupling, int net, bool out, const float* in, float* res, int d_in,
                 int64_t n_tokens, hipStream_t stream);
int profile_begin();
int profile_end(double* total_ms, int64_t* launches);

}  // namespace tw
