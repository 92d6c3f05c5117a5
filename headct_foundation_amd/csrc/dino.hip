// DINO self-distillation loss over K prototypes (reference: src/losses/losses.py:46-102) and the momentum-teacher update
// (src/utils/misc.py:386-397).  HBM-bound streaming kernels: every logit is read twice (row statistics, then loss + gradient).
//
//   teacher_out [2B, K], student_out [V*B, K] (V crops; crop v of sample b is row v*B + b), center [K]:
//     q_i   = softmax((teacher_i - center) / T_t)                 i = 0, 1 (the two global crops), detached
//     logp_v = log_softmax(student_v / T_s)                        v = 0 .. V-1
//     loss  = 1/n  sum_{i} sum_{v != i} mean_b( - sum_k q_i[b,k] logp_v[b,k] ),   n = 2 (V - 1)
//     d loss / d student_v[b,k] = (c_v * softmax(student_v / T_s)[b,k] - sum_{i != v} q_i[b,k]) / (n B T_s),
//                                 c_v = number of teacher views paired with v (1 for v < 2, else 2)
//     batch_center_sum[k] = sum over the 2B teacher rows (the caller all-reduces it and calls hct_dino_center_update).
#include "common.h"

#include <algorithm>

namespace hct {
namespace {

// one block per row: max and sum of exp of (x - c) * inv_t over K columns; stats[row] = {max, log(sum)} in units of the
// scaled logit (so log p = z - max - log(sum))
template <typename T>
__global__ void __launch_bounds__(256) dino_row_stats_kernel(const T* __restrict__ x, const float* __restrict__ center, int K,
                                                             float inv_t, float2* __restrict__ stats) {
  __shared__ float s_red[8];
  const T* row = x + (size_t)blockIdx.x * K;
  float m = -INFINITY;
  for (int k = threadIdx.x * 4; k < K; k += 1024) {
    f32x4 v = Vec4<T>::load(row + k);
    if (center) v -= Vec4<float>::load(center + k);
    m = fmaxf(m, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3])) * inv_t;  // inv_t > 0
  __syncthreads();
  float s = 0.f;
  for (int k = threadIdx.x * 4; k < K; k += 1024) {
    f32x4 v = Vec4<T>::load(row + k);
    if (center) v -= Vec4<float>::load(center + k);
#pragma unroll
    for (int q = 0; q < 4; ++q) s += __expf(v[q] * inv_t - m);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) s_red[4 + (threadIdx.x >> 6)] = s;
  __syncthreads();
  if (threadIdx.x == 0) stats[blockIdx.x] = make_float2(m, __logf((s_red[4] + s_red[5]) + (s_red[6] + s_red[7])));
}

// grid (K / 1024 chunks, B): one block per (sample, 1024-column chunk); loops over the V student crops.
// partial[(b * nchunk + chunk)] = sum over the chunk of  - q_i * logp_v  over all pairs (divided later); fixed-order fold.
template <typename T>
__global__ void __launch_bounds__(256) dino_loss_grad_kernel(const T* __restrict__ student, const T* __restrict__ teacher,
                                                             const float* __restrict__ center, int V, int B, int K, float inv_ts,
                                                             float inv_tt, const float2* __restrict__ st_stats,
                                                             const float2* __restrict__ te_stats, float* __restrict__ partial,
                                                             T* __restrict__ dstudent, const float* __restrict__ dloss, float gcoef,
                                                             float* __restrict__ center_partial) {
  __shared__ float s_red[4];
  const int b = blockIdx.y, k = blockIdx.x * 1024 + threadIdx.x * 4;
  const bool ok = k < K;
  f32x4 q[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
  f32x4 tsum = {0, 0, 0, 0};
  if (ok) {
    const f32x4 c = Vec4<float>::load(center + k);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const f32x4 t = Vec4<T>::load(teacher + ((size_t)i * B + b) * K + k);
      tsum += t;
      const float2 s = te_stats[i * B + b];
#pragma unroll
      for (int e = 0; e < 4; ++e) q[i][e] = __expf((t[e] - c[e]) * inv_tt - s.x - s.y);
    }
  }
  // column sums of the teacher logits over this sample's two rows (summed over samples by the caller's fold): the centre update
  if (center_partial && ok) Vec4<float>::store(center_partial + (size_t)b * K + k, tsum);
  const float g = gcoef * (dloss ? *dloss : 1.0f);
  float acc = 0.f;
  for (int v = 0; v < V; ++v) {
    if (!ok) break;
    const size_t off = ((size_t)v * B + b) * K + k;
    const f32x4 sv = Vec4<T>::load(student + off);
    const float2 s = st_stats[v * B + b];
    f32x4 qs;  // sum of the teacher distributions paired with crop v
    float cv;
    if (v == 0) { qs = q[1]; cv = 1.f; }
    else if (v == 1) { qs = q[0]; cv = 1.f; }
    else { qs = q[0] + q[1]; cv = 2.f; }
    f32x4 gr;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float logp = sv[e] * inv_ts - s.x - s.y;
      acc -= qs[e] * logp;
      gr[e] = (cv * __expf(logp) - qs[e]) * g;
    }
    if (dstudent) Vec4<T>::store(dstudent + off, gr);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[(size_t)b * gridDim.x + blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

__global__ void __launch_bounds__(256) dino_fold_kernel(const float* __restrict__ partial, int n, float scale, float* __restrict__ loss) {
  __shared__ float s_red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *loss = ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) * scale;
}

// batch_center_sum[k] = sum_b center_partial[b, k]  (fixed order)
__global__ void __launch_bounds__(256) dino_center_fold_kernel(const float* __restrict__ cp, int B, int K, float* __restrict__ out) {
  const int k = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (k >= K) return;
  f32x4 s = {0, 0, 0, 0};
  for (int b = 0; b < B; ++b) s += Vec4<float>::load(cp + (size_t)b * K + k);
  Vec4<float>::store(out + k, s);
}

// center = center * m + (batch_center_sum / count) * (1 - m)   (losses.py:95-102, torch's operation order)
__global__ void __launch_bounds__(256) dino_center_update_kernel(float* __restrict__ center, const float* __restrict__ sum, int K,
                                                                 float m, float one_minus_m, float count) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  const float bc = sum[k] / count;
  center[k] = center[k] * m + bc * one_minus_m;
}

// param_k = param_k * m + (1 - m) * param_q   (misc.py:396-397: mul_ then add_ of the scaled student: two roundings + one)
__global__ void __launch_bounds__(256) ema_update_kernel(float* __restrict__ k, const float* __restrict__ q, int64_t n4, float m,
                                                         float one_minus_m) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const f32x4 a = Vec4<float>::load(k + i * 4), b = Vec4<float>::load(q + i * 4);
    Vec4<float>::store(k + i * 4, a * m + b * one_minus_m);
  }
}

// ---- projection-head pieces (src/models/dino_head.py:37-41): one wave per row, n a multiple of 4 --------------------------
// F.normalize(z, dim=-1, p=2): zn = z / max(||z||, 1e-12)
template <typename T>
__global__ void __launch_bounds__(256) l2norm_rows_fwd_kernel(const float* __restrict__ z, int M, int n, T* __restrict__ zn, float* __restrict__ inv_norm) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* src = z + (size_t)row * n;
  float ss = 0.f;
  for (int k = lane * 4; k < n; k += 256) {
    const f32x4 v = Vec4<float>::load(src + k);
    ss += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  }
  ss = wave_sum(ss);
  const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
  for (int k = lane * 4; k < n; k += 256) Vec4<T>::store(zn + (size_t)row * n + k, Vec4<float>::load(src + k) * inv);
  if (lane == 0) inv_norm[row] = inv;
}
// dz = (dzn - zn (zn . dzn)) * inv_norm   (rows whose norm was clamped are not treated specially: ||z|| >= 1e-12 in practice)
template <typename T>
__global__ void __launch_bounds__(256) l2norm_rows_bwd_kernel(const float* __restrict__ dzn, const T* __restrict__ zn, const float* __restrict__ inv_norm,
                                                               int M, int n, float* __restrict__ dz) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  float dot = 0.f;
  for (int k = lane * 4; k < n; k += 256) {
    const f32x4 a = Vec4<float>::load(dzn + (size_t)row * n + k), b = Vec4<T>::load(zn + (size_t)row * n + k);
    dot += (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]);
  }
  dot = wave_sum(dot);
  const float inv = inv_norm[row];
  for (int k = lane * 4; k < n; k += 256) {
    const f32x4 a = Vec4<float>::load(dzn + (size_t)row * n + k), b = Vec4<T>::load(zn + (size_t)row * n + k);
    Vec4<float>::store(dz + (size_t)row * n + k, (a - b * dot) * inv);
  }
}
// torch.nn.utils.weight_norm (dim 0): W[k,:] = g[k] * v[k,:] / ||v[k,:]||
template <typename T>
__global__ void __launch_bounds__(256) weight_norm_fwd_kernel(const float* __restrict__ v, const float* __restrict__ g, int K, int n, T* __restrict__ w,
                                                               float* __restrict__ inv_norm) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= K) return;
  const float* src = v + (size_t)row * n;
  float ss = 0.f;
  for (int k = lane * 4; k < n; k += 256) {
    const f32x4 x = Vec4<float>::load(src + k);
    ss += (x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3]);
  }
  ss = wave_sum(ss);
  const float inv = 1.0f / sqrtf(ss), sc = g[row] * inv;
  for (int k = lane * 4; k < n; k += 256) Vec4<T>::store(w + (size_t)row * n + k, Vec4<float>::load(src + k) * sc);
  if (lane == 0) inv_norm[row] = inv;
}
// dv = g / ||v|| * (dW - (dW . vhat) vhat),  dg = dW . vhat,  vhat = v / ||v||
__global__ void __launch_bounds__(256) weight_norm_bwd_kernel(const float* __restrict__ dw, const float* __restrict__ v, const float* __restrict__ g,
                                                               const float* __restrict__ inv_norm, int K, int n, float* __restrict__ dv,
                                                               float* __restrict__ dg) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= K) return;
  const float inv = inv_norm[row];
  float dot = 0.f;
  for (int k = lane * 4; k < n; k += 256) {
    const f32x4 a = Vec4<float>::load(dw + (size_t)row * n + k), b = Vec4<float>::load(v + (size_t)row * n + k) * inv;
    dot += (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]);
  }
  dot = wave_sum(dot);
  const float sc = g[row] * inv;
  for (int k = lane * 4; k < n; k += 256) {
    const f32x4 a = Vec4<float>::load(dw + (size_t)row * n + k), b = Vec4<float>::load(v + (size_t)row * n + k) * inv;
    Vec4<float>::store(dv + (size_t)row * n + k, (a - b * dot) * sc);
  }
  if (dg && lane == 0) dg[row] = dot;
}

// ---- BatchNorm1d (affine) + GELU of the projection head with use_bn (dino_head.py:15-21: Linear -> BatchNorm1d -> GELU) ---------
// u [M, D] fp32 = the Linear's output; mean / var per feature: the batch statistics in training (hct_batchnorm_stats; all-reduced by
// the caller under data parallelism, as SyncBatchNorm does, main_pretrain_dino.py:183-185) or the running ones in eval.
//   xhat = (u - mean) * rsqrt(var + eps);  y = gamma * xhat + beta;  h = gelu(y) (exact erf);  dact = gelu'(y)
template <typename T>
__global__ void __launch_bounds__(256) bn_gelu_fwd_kernel(const float* __restrict__ u, const float* __restrict__ mean, const float* __restrict__ var,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int64_t n4, int D,
                                                          T* __restrict__ h, float* __restrict__ xhat, float* __restrict__ dact) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const int k = (int)((i * 4) % D);
  const f32x4 uv = Vec4<float>::load(u + i * 4), mu = Vec4<float>::load(mean + k), vr = Vec4<float>::load(var + k);
  const f32x4 g = Vec4<float>::load(gamma + k), b = Vec4<float>::load(beta + k);
  f32x4 xh, hv, dv;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    xh[e] = (uv[e] - mu[e]) * (1.0f / sqrtf(vr[e] + eps));
    const float y = g[e] * xh[e] + b[e];
    hv[e] = gelu_erf(y);
    dv[e] = dgelu_erf(y);
  }
  Vec4<T>::store(h + i * 4, hv);
  if (xhat) Vec4<float>::store(xhat + i * 4, xh);
  if (dact) Vec4<float>::store(dact + i * 4, dv);
}

// sums[0][k] = sum_rows dy, sums[1][k] = sum_rows dy * xhat, dy = dh * dact  (= dbeta, dgamma of this rank's rows); rows in order
template <typename T>
__global__ void __launch_bounds__(256) bn_gelu_bwd_sums_kernel(const T* __restrict__ dh, const float* __restrict__ dact, const float* __restrict__ xhat,
                                                               int M, int D, float* __restrict__ sums) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= D) return;
  float sb = 0.f, sg = 0.f;
  for (int r = 0; r < M; ++r) {
    const size_t o = (size_t)r * D + k;
    const float dy = to_f32(dh[o]) * dact[o];
    sb += dy;
    sg += dy * xhat[o];
  }
  sums[k] = sb;
  sums[D + k] = sg;
}

// du = gamma * rstd * (dy - sums[0] / count - xhat * sums[1] / count): gradient wrt the Linear's output (count = rows of ALL ranks
// that shared the statistics; sums all-reduced by the caller then)
template <typename T, typename TO>
__global__ void __launch_bounds__(256) bn_gelu_bwd_apply_kernel(const T* __restrict__ dh, const float* __restrict__ dact, const float* __restrict__ xhat,
                                                                const float* __restrict__ gamma, const float* __restrict__ var, float eps,
                                                                const float* __restrict__ sums, float inv_count, int64_t n4, int D, TO* __restrict__ du) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const int k = (int)((i * 4) % D);
  const f32x4 dhv = Vec4<T>::load(dh + i * 4), da = Vec4<float>::load(dact + i * 4), xh = Vec4<float>::load(xhat + i * 4);
  const f32x4 g = Vec4<float>::load(gamma + k), vr = Vec4<float>::load(var + k), sb = Vec4<float>::load(sums + k), sg = Vec4<float>::load(sums + D + k);
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float dy = dhv[e] * da[e];
    o[e] = g[e] * (1.0f / sqrtf(vr[e] + eps)) * (dy - sb[e] * inv_count - xh[e] * sg[e] * inv_count);
  }
  Vec4<TO>::store(du + i * 4, o);
}

}  // namespace
}  // namespace hct

using namespace hct;

extern "C" {

size_t hct_dino_loss_workspace_bytes(int V, int B, int K) {
  const size_t nchunk = (K + 1023) / 1024;
  return align_up((size_t)(V + 2) * B * sizeof(float2), 256) + align_up((size_t)B * nchunk * sizeof(float), 256) + (size_t)B * K * sizeof(float);
}

int hct_dino_loss(const void* student, const void* teacher, int dtype, int V, int B, int K, const float* center, float student_temp,
                  float teacher_temp, float* loss, void* dstudent, const float* dloss, float* batch_center_sum, void* workspace,
                  size_t workspace_bytes, void* stream) {
  HCT_REQUIRE(student && teacher && center && loss && V >= 2 && B > 0 && K > 0 && K % 4 == 0, "hct_dino_loss: bad arguments (K %% 4 == 0, >= 2 crops)");
  HCT_REQUIRE(student_temp > 0.f && teacher_temp > 0.f, "hct_dino_loss: temperatures must be positive");
  if (workspace_bytes < hct_dino_loss_workspace_bytes(V, B, K) || !workspace) { set_error("hct_dino_loss: workspace too small"); return HCT_E_WORKSPACE; }
  hipStream_t s = (hipStream_t)stream;
  unsigned char* ws = (unsigned char*)workspace;
  float2* st_stats = (float2*)ws;
  float2* te_stats = st_stats + (size_t)V * B;
  const int nchunk = (K + 1023) / 1024;
  float* partial = (float*)(ws + align_up((size_t)(V + 2) * B * sizeof(float2), 256));
  float* cpart = (float*)((unsigned char*)partial + align_up((size_t)B * nchunk * sizeof(float), 256));
  const float inv_ts = 1.0f / student_temp, inv_tt = 1.0f / teacher_temp;
  const int nterms = 2 * (V - 1);
  const float gcoef = inv_ts / ((float)nterms * (float)B);
  HCT_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL(dino_row_stats_kernel<T>, dim3(V * B), dim3(256), 0, s, (const T*)student, (const float*)nullptr, K, inv_ts, st_stats);
    hipLaunchKernelGGL(dino_row_stats_kernel<T>, dim3(2 * B), dim3(256), 0, s, (const T*)teacher, center, K, inv_tt, te_stats);
    hipLaunchKernelGGL(dino_loss_grad_kernel<T>, dim3(nchunk, B), dim3(256), 0, s, (const T*)student, (const T*)teacher, center, V, B, K, inv_ts,
                       inv_tt, st_stats, te_stats, partial, (T*)dstudent, dloss, gcoef, batch_center_sum ? cpart : nullptr);
  });
  hipLaunchKernelGGL(dino_fold_kernel, dim3(1), dim3(256), 0, s, partial, B * nchunk, 1.0f / ((float)nterms * (float)B), loss);
  if (batch_center_sum) hipLaunchKernelGGL(dino_center_fold_kernel, dim3((K / 4 + 255) / 256), dim3(256), 0, s, cpart, B, K, batch_center_sum);
  HCT_CHECK_LAUNCH("hct_dino_loss");
  return 0;
}

int hct_l2norm_rows_fwd(const float* z, int M, int n, void* zn, int zn_dtype, float* inv_norm, void* stream) {
  HCT_REQUIRE(z && zn && inv_norm && M >= 0 && n > 0 && n % 4 == 0, "hct_l2norm_rows_fwd: bad arguments");
  if (M == 0) return 0;
  HCT_DISPATCH_DTYPE(zn_dtype, T, hipLaunchKernelGGL(l2norm_rows_fwd_kernel<T>, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, z, M, n, (T*)zn, inv_norm));
  HCT_CHECK_LAUNCH("hct_l2norm_rows_fwd");
  return 0;
}
int hct_l2norm_rows_bwd(const float* dzn, const void* zn, int zn_dtype, const float* inv_norm, int M, int n, float* dz, void* stream) {
  HCT_REQUIRE(dzn && zn && inv_norm && dz && M >= 0 && n > 0 && n % 4 == 0, "hct_l2norm_rows_bwd: bad arguments");
  if (M == 0) return 0;
  HCT_DISPATCH_DTYPE(zn_dtype, T, hipLaunchKernelGGL(l2norm_rows_bwd_kernel<T>, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, dzn, (const T*)zn, inv_norm, M, n, dz));
  HCT_CHECK_LAUNCH("hct_l2norm_rows_bwd");
  return 0;
}
int hct_weight_norm_fwd(const float* v, const float* g, int K, int n, void* w, int w_dtype, float* inv_norm, void* stream) {
  HCT_REQUIRE(v && g && w && inv_norm && K > 0 && n > 0 && n % 4 == 0, "hct_weight_norm_fwd: bad arguments");
  HCT_DISPATCH_DTYPE(w_dtype, T, hipLaunchKernelGGL(weight_norm_fwd_kernel<T>, dim3((K + 3) / 4), dim3(256), 0, (hipStream_t)stream, v, g, K, n, (T*)w, inv_norm));
  HCT_CHECK_LAUNCH("hct_weight_norm_fwd");
  return 0;
}
int hct_weight_norm_bwd(const float* dw, const float* v, const float* g, const float* inv_norm, int K, int n, float* dv, float* dg, void* stream) {
  HCT_REQUIRE(dw && v && g && inv_norm && dv && K > 0 && n > 0 && n % 4 == 0, "hct_weight_norm_bwd: bad arguments");
  hipLaunchKernelGGL(weight_norm_bwd_kernel, dim3((K + 3) / 4), dim3(256), 0, (hipStream_t)stream, dw, v, g, inv_norm, K, n, dv, dg);
  HCT_CHECK_LAUNCH("hct_weight_norm_bwd");
  return 0;
}

int hct_dino_center_update(float* center, const float* batch_center_sum, int K, double momentum, double count, void* stream) {
  HCT_REQUIRE(center && batch_center_sum && K > 0 && count > 0, "hct_dino_center_update: bad arguments");
  // 1 - momentum is formed in double like Python does before torch narrows the scalar to fp32
  hipLaunchKernelGGL(dino_center_update_kernel, dim3((K + 255) / 256), dim3(256), 0, (hipStream_t)stream, center, batch_center_sum, K,
                     (float)momentum, (float)(1.0 - momentum), (float)count);
  HCT_CHECK_LAUNCH("hct_dino_center_update");
  return 0;
}

int hct_ema_update(float* momentum_params, const float* params, int64_t n, double m, void* stream) {
  HCT_REQUIRE(momentum_params && params && n >= 0 && n % 4 == 0, "hct_ema_update: n %% 4 != 0 or null buffer");
  if (n == 0) return 0;
  const int blocks = (int)std::min<int64_t>(2048, (n / 4 + 255) / 256);
  hipLaunchKernelGGL(ema_update_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, momentum_params, params, n / 4, (float)m, (float)(1.0 - m));
  HCT_CHECK_LAUNCH("hct_ema_update");
  return 0;
}

int hct_bn_gelu_fwd(const float* u, const float* mean, const float* var, const float* gamma, const float* beta, float eps, int M, int D, void* h,
                    int h_dtype, float* xhat, float* dact, void* stream) {
  HCT_REQUIRE(u && mean && var && gamma && beta && h && M > 0 && D > 0 && D % 4 == 0, "hct_bn_gelu_fwd: bad arguments (D must be a multiple of 4)");
  const int64_t n4 = (int64_t)M * D / 4;
  HCT_DISPATCH_DTYPE(h_dtype, T, hipLaunchKernelGGL(bn_gelu_fwd_kernel<T>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, u, mean, var,
                                                     gamma, beta, eps, n4, D, (T*)h, xhat, dact));
  HCT_CHECK_LAUNCH("hct_bn_gelu_fwd");
  return 0;
}

int hct_bn_gelu_bwd_sums(const void* dh, int dh_dtype, const float* dact, const float* xhat, int M, int D, float* sums, void* stream) {
  HCT_REQUIRE(dh && dact && xhat && sums && M > 0 && D > 0, "hct_bn_gelu_bwd_sums: bad arguments");
  HCT_DISPATCH_DTYPE(dh_dtype, T, hipLaunchKernelGGL(bn_gelu_bwd_sums_kernel<T>, dim3((D + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const T*)dh, dact,
                                                      xhat, M, D, sums));
  HCT_CHECK_LAUNCH("hct_bn_gelu_bwd_sums");
  return 0;
}

int hct_bn_gelu_bwd_apply(const void* dh, int dh_dtype, const float* dact, const float* xhat, const float* gamma, const float* var, float eps,
                          const float* sums, double count, int M, int D, void* du, int du_dtype, void* stream) {
  HCT_REQUIRE(dh && dact && xhat && gamma && var && sums && du && M > 0 && D > 0 && D % 4 == 0 && count >= 1.0, "hct_bn_gelu_bwd_apply: bad arguments");
  HCT_REQUIRE(dh_dtype == du_dtype || du_dtype == HCT_F32, "hct_bn_gelu_bwd_apply: du is written in the dtype of dh, or fp32");
  const int64_t n4 = (int64_t)M * D / 4;
  const float inv = (float)(1.0 / count);
  const dim3 grid((unsigned)((n4 + 255) / 256));
  hipStream_t s = (hipStream_t)stream;
  if (du_dtype == HCT_F32 && dh_dtype == HCT_BF16)
    hipLaunchKernelGGL((bn_gelu_bwd_apply_kernel<bf16, float>), grid, dim3(256), 0, s, (const bf16*)dh, dact, xhat, gamma, var, eps, sums, inv, n4, D, (float*)du);
  else
    HCT_DISPATCH_DTYPE(dh_dtype, T, hipLaunchKernelGGL((bn_gelu_bwd_apply_kernel<T, T>), grid, dim3(256), 0, s, (const T*)dh, dact, xhat, gamma, var, eps, sums,
                                                        inv, n4, D, (T*)du));
  HCT_CHECK_LAUNCH("hct_bn_gelu_bwd_apply");
  return 0;
}

}  // extern "C"

