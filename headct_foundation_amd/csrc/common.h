// Shared device/host helpers for libheadct_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/headct_hip.h"

namespace hct {

typedef __bf16 bf16;
typedef _Float16 f16;  // IEEE half: storage type of cached input volumes only (transforms.py:171-178 cast)
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kWave = 64;

// ---- error plumbing ------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_hip(hipError_t e, const char* what);
#define HCT_CHECK_LAUNCH(what)                         \
  do {                                                 \
    int _rc = ::hct::check_hip(hipGetLastError(), what); \
    if (_rc) return _rc;                               \
  } while (0)
#define HCT_REQUIRE(cond, ...)       \
  do {                               \
    if (!(cond)) {                   \
      ::hct::set_error(__VA_ARGS__); \
      return HCT_E_BADARG;           \
    }                                \
  } while (0)

// masked MSE with a host-side gradient scale folded into dpred (training forward of the plan; elementwise.hip)
int masked_mse_launch(const void* pred, int pred_dtype, const void* x, int x_dtype, const float* mask, int B, int C, int S, int P,
                      int norm_pix, float mask_sum, float* row_loss, float* loss, void* dpred, const float* dpred_scale,
                      float host_scale, hipStream_t s, const int32_t* masked_ids = nullptr, int K = 0);  // masked_ids: compact-row form
// first decoder block on the "cat" rows (kept tokens + one table row per patch position; elementwise.hip, used by mae_plan.hip)
int dec0_index(const int32_t* ids_restore, int B, int L, int K, int32_t* kept_rows, int32_t* cat_idx, hipStream_t s);
int dec0_table(const float* mask_token, const float* pos, int L, int D, float* out, hipStream_t s);
int dec0_aggregate(const void* g, int dtype, const int32_t* kept_rows, const int32_t* ids_restore, int B, int L, int K, int W, void* out, hipStream_t s);
int dec0_token_grads(const float* dh, const int32_t* ids_shuffle, const float* dcat, int B, int L, int K, int D, float* dmask, float* dcls,
                     void* workspace, size_t workspace_bytes, hipStream_t s);
// host-side test: can this product join a grouped weight-gradient launch (hct_gemm_tn_group_*; gemm.hip)
bool tn_group_ok(const hct_gemm_args* a);
// buf[i] *= *scale unless *scale == 1 (every block then leaves after one scalar load)
int scale_unless_one(void* buf, int dtype, int64_t n, const float* scale, hipStream_t s);
// out[d] = sum_{b < nblk} partial[b*D + d]  (fixed order; elementwise.hip)
int fold_rows(const float* partial, int nblk, int D, float* out, hipStream_t s);

// Deferred folds.  The fixed-order folds of per-block partial sums (LayerNorm gamma / beta / column sums, fused bias column
// sums) are launches of a few dozen workgroups that the NEXT kernel on the stream would have to wait for although it does not
// read their result.  While a sink is installed (the model driver does so around a block's backward, with a partial buffer of
// its own for every producer) these folds are recorded instead of launched, and fold_flush() runs them all in ONE launch, each
// with exactly the summation order of the separate launch.
struct FoldJob {
  const float* partial;  // [nblk][nq_stride][D]
  int nblk, nq_stride, D, nq;
  float* out[3];         // per quantity q < nq (a null output is skipped)
};
struct FoldSink {
  static constexpr int kMax = 8;
  FoldJob jobs[kMax];
  int n = 0;
};
extern thread_local FoldSink* g_fold_sink;
int fold_partials(const FoldJob& job, hipStream_t s);  // into the sink if one is installed (and has room), else launched now
int fold_flush(FoldSink& sink, hipStream_t s);

inline size_t dtype_size(int dt) { return dt == HCT_BF16 ? 2 : 4; }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- scalar load/store by storage type -----------------------------------------------------
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }  // RNE, NaN-preserving cvt

// 4-wide vector access (16 B fp32 / 8 B bf16); pointers must be suitably aligned.
template <typename T> struct Vec4;
template <> struct Vec4<float> {
  static __device__ __forceinline__ f32x4 load(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
  static __device__ __forceinline__ void store(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
  // streamed once: non-temporal (does not displace what the neighbouring GEMMs re-read through L2)
  static __device__ __forceinline__ f32x4 load_nt(const float* p) { return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)); }
  static __device__ __forceinline__ void store_nt(float* p, f32x4 v) { __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p)); }
};
template <> struct Vec4<bf16> {
  static __device__ __forceinline__ f32x4 load(const bf16* p) {
    bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  }
  static __device__ __forceinline__ void store(bf16* p, f32x4 v) {
    bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
    *reinterpret_cast<bf16x4*>(p) = o;
  }
  static __device__ __forceinline__ f32x4 load_nt(const bf16* p) {
    bf16x4 v = __builtin_nontemporal_load(reinterpret_cast<const bf16x4*>(p));
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  }
  static __device__ __forceinline__ void store_nt(bf16* p, f32x4 v) {
    bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
    __builtin_nontemporal_store(o, reinterpret_cast<bf16x4*>(p));
  }
};

template <> struct Vec4<f16> {
  static __device__ __forceinline__ f32x4 load(const f16* p) {
    f16x4 v = *reinterpret_cast<const f16x4*>(p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  }
  static __device__ __forceinline__ void store(f16* p, f32x4 v) {
    f16x4 o = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
    *reinterpret_cast<f16x4*>(p) = o;
  }
  static __device__ __forceinline__ f32x4 load_nt(const f16* p) {
    f16x4 v = __builtin_nontemporal_load(reinterpret_cast<const f16x4*>(p));
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  }
};
__device__ __forceinline__ float to_f32(f16 v) { return (float)v; }

// ---- wave / block reductions ---------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// exact-erf GELU and its derivative (nn.GELU() default; MONAI MLPBlock)
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float dgelu_erf(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// Fast GELU for bf16-stored outputs (the MFMA GEMM epilogues; the fp32 parity mode keeps gelu_erf).  The epilogue of the GELU /
// GELU' GEMM instances is VALU-issue bound (PMC: 3.5-4x the VALU instructions of the plain instance), so the normal CDF is a
// clamped odd polynomial with no transcendental:  Phi(x) = 0.5 + xc P(xc^2),  xc = clamp(x, -4, 4),  P of degree 7 fitted with
// P(16) = 1/8 so that Phi(-4) = 0 and Phi(4) = 1 (gelu(x) = 0 resp. x beyond the clamp, to rounding).  max |Phi error| 3.5e-5
// (mostly the 3.2e-5 mass beyond |x| = 4), max |gelu error| 1.4e-4, against the 2^-9 relative rounding of the bf16 store.
// (Round 1 used Abramowitz-Stegun 7.1.26: |error| 3e-7 for a v_rcp + v_exp + 12 VALU per element; this form is 11 VALU.)
__device__ __forceinline__ float gelu_cdf_poly(float x) {
  const float xc = __builtin_amdgcn_fmed3f(x, -4.0f, 4.0f);
  const float s = xc * xc;
  float p = fmaf(-1.4272797388414915e-09f, s, 1.1287725243391833e-07f);
  p = fmaf(p, s, -3.897907390637556e-06f);
  p = fmaf(p, s, 7.829771493561566e-05f);
  p = fmaf(p, s, -0.0010334221879020333f);
  p = fmaf(p, s, 0.009617664851248264f);
  p = fmaf(p, s, -0.06610910594463348f);
  p = fmaf(p, s, 0.398820698261261f);
  return fmaf(xc, p, 0.5f);
}
#ifndef HCT_GELU_AS
__device__ __forceinline__ float gelu_fast(float x) { return x * gelu_cdf_poly(x); }
// gelu'(x) = Phi(x) + x phi(x): the density keeps its exponential (of the unclamped x: it is what dies out beyond the clamp)
__device__ __forceinline__ float dgelu_fast(float x) {
  const float E = __builtin_amdgcn_exp2f(x * x * (-0.5f * 1.44269504088896340736f));
  return fmaf(x * 0.39894228040143267794f, E, gelu_cdf_poly(x));
}
// both at once (the forward epilogue that saves gelu' instead of the pre-activation): one CDF polynomial, one exponential
__device__ __forceinline__ void gelu_both(float x, float& g, float& dg) {
  const float cdf = gelu_cdf_poly(x);
  const float E = __builtin_amdgcn_exp2f(x * x * (-0.5f * 1.44269504088896340736f));
  g = x * cdf;
  dg = fmaf(x * 0.39894228040143267794f, E, cdf);
}
// Eight elements at once, the Horner steps of the two 4-vectors interleaved: the same operations on the same values in the same
// order per element as gelu_both (bit-identical), but a step's four packed instructions (v_pk_fma_f32 on two elements each) are
// independent of one another, so hipcc no longer puts a wait state between every pair of them (one dependent chain at a time cost
// ~590 s_nop per wave and tile in the GELU epilogue, ~2 us of its 8).
__device__ __forceinline__ void gelu_both8(const f32x4& xa, const f32x4& xb, f32x4& ga, f32x4& da, f32x4& gb, f32x4& db) {
  auto med = [](f32x4 v) {
    return f32x4{__builtin_amdgcn_fmed3f(v[0], -4.0f, 4.0f), __builtin_amdgcn_fmed3f(v[1], -4.0f, 4.0f), __builtin_amdgcn_fmed3f(v[2], -4.0f, 4.0f),
                 __builtin_amdgcn_fmed3f(v[3], -4.0f, 4.0f)};
  };
  auto splat = [](float c) { return f32x4{c, c, c, c}; };
  const f32x4 ca = med(xa), cb = med(xb);
  const f32x4 sa = ca * ca, sb = cb * cb;
  f32x4 pa = __builtin_elementwise_fma(splat(-1.4272797388414915e-09f), sa, splat(1.1287725243391833e-07f));
  f32x4 pb = __builtin_elementwise_fma(splat(-1.4272797388414915e-09f), sb, splat(1.1287725243391833e-07f));
#define HCT_GELU_STEP(c_)                                   \
  pa = __builtin_elementwise_fma(pa, sa, splat(c_));        \
  pb = __builtin_elementwise_fma(pb, sb, splat(c_));
  HCT_GELU_STEP(-3.897907390637556e-06f)
  HCT_GELU_STEP(7.829771493561566e-05f)
  HCT_GELU_STEP(-0.0010334221879020333f)
  HCT_GELU_STEP(0.009617664851248264f)
  HCT_GELU_STEP(-0.06610910594463348f)
  HCT_GELU_STEP(0.398820698261261f)
#undef HCT_GELU_STEP
  const f32x4 cdfa = __builtin_elementwise_fma(ca, pa, splat(0.5f)), cdfb = __builtin_elementwise_fma(cb, pb, splat(0.5f));
  const f32x4 ta = xa * xa * (-0.5f * 1.44269504088896340736f), tb = xb * xb * (-0.5f * 1.44269504088896340736f);
  f32x4 ea, eb;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    ea[q] = __builtin_amdgcn_exp2f(ta[q]);
    eb[q] = __builtin_amdgcn_exp2f(tb[q]);
  }
  ga = xa * cdfa;
  gb = xb * cdfb;
  da = __builtin_elementwise_fma(xa * 0.39894228040143267794f, ea, cdfa);
  db = __builtin_elementwise_fma(xb * 0.39894228040143267794f, eb, cdfb);
}
#else  // diagnostic build (A/B of the two forms): erf by Abramowitz-Stegun 7.1.26, gelu and gelu' sharing the exponential
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& pdf) {
  const float t = __builtin_amdgcn_rcpf(fmaf(fabsf(x), 0.3275911f * 0.70710678118654752440f, 1.0f));
  const float E = __builtin_amdgcn_exp2f(x * x * (-0.5f * 1.44269504088896340736f));
  float p = fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
  p = fmaf(p, t, 0.5f * 1.421413741f);
  p = fmaf(p, t, 0.5f * -0.284496736f);
  p = fmaf(p, t, 0.5f * 0.254829592f);
  const float half_tail = p * t * E;  // 0.5 * (1 - erf(|x| / sqrt 2)), |abs err| <= 3e-7
  cdf = x >= 0.f ? 1.0f - half_tail : half_tail;
  pdf = 0.39894228040143267794f * E;
}
__device__ __forceinline__ float gelu_fast(float x) { float c, p; gelu_parts(x, c, p); return x * c; }
__device__ __forceinline__ float dgelu_fast(float x) { float c, p; gelu_parts(x, c, p); return fmaf(x, p, c); }
__device__ __forceinline__ void gelu_both(float x, float& g, float& dg) { float c, p; gelu_parts(x, c, p); g = x * c; dg = fmaf(x, p, c); }
__device__ __forceinline__ void gelu_both8(const f32x4& xa, const f32x4& xb, f32x4& ga, f32x4& da, f32x4& gb, f32x4& db) {
  for (int q = 0; q < 4; ++q) { gelu_both(xa[q], ga[q], da[q]); gelu_both(xb[q], gb[q], db[q]); }
}
#endif

// dispatch a storage dtype code to a template parameter
#define HCT_DISPATCH_DTYPE(dt, T, ...)                 \
  do {                                                 \
    if ((dt) == HCT_BF16) {                            \
      using T = ::hct::bf16;                           \
      __VA_ARGS__;                                     \
    } else {                                           \
      using T = float;                                 \
      __VA_ARGS__;                                     \
    }                                                  \
  } while (0)

}  // namespace hct
