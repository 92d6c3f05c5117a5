// Per-parameter gradient clip + AdamW over one flat fp32 buffer (HBM-bound, one pass each).
//   reference: src/utils/misc.py:374-383 (per-tensor clip), src/utils/optimizers.py:354-360 (torch AdamW).
// The flat buffer is cut into 1024-element units (one float4 per thread of a 256-thread block); every
// segment (= parameter tensor) starts on a unit boundary, so a unit belongs to exactly one segment.
#include "common.h"

namespace hct {

constexpr int kUnit = 1024;

__device__ __forceinline__ float block_sum_256(float v, float* s_tmp) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_tmp[threadIdx.x >> 6] = v;
  __syncthreads();
  return (s_tmp[0] + s_tmp[1]) + (s_tmp[2] + s_tmp[3]);
}

__global__ void __launch_bounds__(256) sumsq_units_kernel(const float* __restrict__ g, int64_t total,
                                                          float* __restrict__ unit_sumsq) {
  __shared__ float s_tmp[4];
  const int64_t i = (int64_t)blockIdx.x * kUnit + threadIdx.x * 4;
  float s = 0.f;
  if (i < total) {
    const f32x4 v = Vec4<float>::load(g + i);
    s = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  }
  s = block_sum_256(s, s_tmp);
  if (threadIdx.x == 0) unit_sumsq[blockIdx.x] = s;
}

// one block per segment: fixed-order fold of its units -> norm, coef
__global__ void __launch_bounds__(256) seg_norm_kernel(const float* __restrict__ unit_sumsq, const int64_t* __restrict__ seg_off,
                                                       float clip, float* __restrict__ norms, float* __restrict__ coef) {
  __shared__ float s_tmp[4];
  const int sgi = blockIdx.x;
  const int64_t u0 = seg_off[sgi] / kUnit, u1 = seg_off[sgi + 1] / kUnit;
  float s = 0.f;
  for (int64_t u = u0 + threadIdx.x; u < u1; u += 256) s += unit_sumsq[u];
  s = block_sum_256(s, s_tmp);
  if (threadIdx.x == 0) {
    const float n = sqrtf(s);
    norms[sgi] = n;
    float c = 1.0f;
    if (clip > 0.f) {
      const float cc = clip / (n + 1e-6f);  // misc.py:380
      if (cc < 1.0f) c = cc;                // misc.py:381
    }
    coef[sgi] = c;
  }
}

__device__ __forceinline__ int find_segment(const int64_t* __restrict__ seg_off, int nseg, int64_t pos) {
  int lo = 0, hi = nseg;  // seg_off[lo] <= pos < seg_off[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (seg_off[mid] <= pos) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ void __launch_bounds__(256) scale_units_kernel(float* __restrict__ g, const int64_t* __restrict__ seg_off,
                                                          const float* __restrict__ coef, int nseg, int64_t total) {
  const int64_t base = (int64_t)blockIdx.x * kUnit;
  const float c = coef[find_segment(seg_off, nseg, base)];
  if (c == 1.0f) return;
  const int64_t i = base + threadIdx.x * 4;
  if (i < total) Vec4<float>::store(g + i, Vec4<float>::load(g + i) * c);
}

struct AdamArgs {
  float lr_wd_keep;   // 1 - lr*wd
  float one_m_b1, b2, one_m_b2;
  float step_size;    // lr / (1 - b1^t)
  float inv_bc2_sqrt; // 1 / sqrt(1 - b2^t)
  float eps;
};

__global__ void __launch_bounds__(256) adamw_units_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                          float* __restrict__ v, const int64_t* __restrict__ seg_off,
                                                          const float* __restrict__ coef, const uint8_t* __restrict__ skip,
                                                          int nseg, int64_t total, AdamArgs a, bf16* __restrict__ p_bf16) {
  const int64_t base = (int64_t)blockIdx.x * kUnit;
  const int sgi = find_segment(seg_off, nseg, base);
  const int64_t i = base + threadIdx.x * 4;
  if (i >= total) return;
  if (skip && skip[sgi]) return;
  const float c = coef ? coef[sgi] : 1.0f;
#ifndef HCT_ADAM_NT  // A/B builds: 1 = the fp32 parameter / gradient / moment streams non-temporal (the bf16 weight copies, which the next step's first GEMMs read, stay cacheable)
#define HCT_ADAM_NT 1  /* measured in the step: -0.17 ms */
#endif
#define HCT_ADAM_LD(ptr) (HCT_ADAM_NT ? Vec4<float>::load_nt(ptr) : Vec4<float>::load(ptr))
#define HCT_ADAM_ST(ptr, val) do { if (HCT_ADAM_NT) Vec4<float>::store_nt(ptr, val); else Vec4<float>::store(ptr, val); } while (0)
  f32x4 gv = HCT_ADAM_LD(g + i);
  if (c != 1.0f) {
    gv = gv * c;
    Vec4<float>::store(g + i, gv);  // leave the clipped gradient in .grad like the reference does
  }
  f32x4 pv = HCT_ADAM_LD(p + i) * a.lr_wd_keep;          // param.mul_(1 - lr*wd)
  f32x4 mv = HCT_ADAM_LD(m + i);
  mv = mv + (gv - mv) * a.one_m_b1;                            // exp_avg.lerp_(grad, 1-beta1)
  f32x4 vv = HCT_ADAM_LD(v + i) * a.b2 + gv * gv * a.one_m_b2;
  f32x4 den;
#pragma unroll
  for (int e = 0; e < 4; ++e) den[e] = sqrtf(vv[e]) * a.inv_bc2_sqrt + a.eps;
#pragma unroll
  for (int e = 0; e < 4; ++e) pv[e] = pv[e] - a.step_size * (mv[e] / den[e]);
  HCT_ADAM_ST(p + i, pv);
  HCT_ADAM_ST(m + i, mv);
  HCT_ADAM_ST(v + i, vv);
  if (p_bf16) Vec4<bf16>::store(p_bf16 + i, pv);
}

}  // namespace hct

using namespace hct;

extern "C" {

size_t hct_grad_norms_workspace_bytes(int64_t total) { return (size_t)((total + kUnit - 1) / kUnit) * sizeof(float); }

int hct_grad_norms(float* grads, const int64_t* seg_off, int nseg, int64_t total, float clip, int scale_in_place,
                   float* norms, float* coef, void* workspace, size_t workspace_bytes, void* stream) {
  HCT_REQUIRE(total > 0 && total % kUnit == 0 && nseg > 0, "hct_grad_norms: total must be a positive multiple of %d", kUnit);
  if (workspace_bytes < hct_grad_norms_workspace_bytes(total)) {
    set_error("hct_grad_norms: workspace too small");
    return HCT_E_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const int units = (int)(total / kUnit);
  hipLaunchKernelGGL(sumsq_units_kernel, dim3(units), dim3(256), 0, s, grads, total, (float*)workspace);
  hipLaunchKernelGGL(seg_norm_kernel, dim3(nseg), dim3(256), 0, s, (const float*)workspace, seg_off, clip, norms, coef);
  if (scale_in_place && clip > 0.f)
    hipLaunchKernelGGL(scale_units_kernel, dim3(units), dim3(256), 0, s, grads, seg_off, coef, nseg, total);
  HCT_CHECK_LAUNCH("hct_grad_norms");
  return 0;
}

int hct_adamw_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, const int64_t* seg_off,
                   const float* coef, const uint8_t* skip, int nseg, int64_t total, float lr, float beta1, float beta2,
                   float eps, float weight_decay, int step, void* params_bf16, void* stream) {
  HCT_REQUIRE(total > 0 && total % kUnit == 0 && nseg > 0 && step >= 1, "hct_adamw_step: bad arguments");
  AdamArgs a;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  a.lr_wd_keep = (float)(1.0 - (double)lr * (double)weight_decay);
  a.one_m_b1 = (float)(1.0 - (double)beta1);
  a.b2 = beta2;
  a.one_m_b2 = (float)(1.0 - (double)beta2);
  a.step_size = (float)((double)lr / bc1);
  a.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  a.eps = eps;
  hipLaunchKernelGGL(adamw_units_kernel, dim3((int)(total / kUnit)), dim3(256), 0, (hipStream_t)stream, params, grads,
                     exp_avg, exp_avg_sq, seg_off, coef, skip, nseg, total, a, (bf16*)params_bf16);
  HCT_CHECK_LAUNCH("hct_adamw_step");
  return 0;
}

}  // extern "C"
