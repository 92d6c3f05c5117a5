// Classification heads on top of the ViT features (forward only): the eval-mode arithmetic of LinearClassifier /
// AttentionClassifier (src/models/classifier.py:7-99) and of ViT's own `classification_head` (src/models/vit.py:133-137,
// :170-171).  Tiny, latency-bound kernels: B x num_classes dot products and a few learnt queries against all tokens.
#include "common.h"

namespace hct {

// BatchNorm1d(affine=False) in eval mode over the channel axis of [rows, C]: (x - mean) / sqrt(var + eps)
template <typename T>
__global__ void __launch_bounds__(256) channel_norm_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ var, float eps, T* __restrict__ out, int C,
                                                           int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const int c = (int)(i % (C / 4)) * 4;
  const f32x4 v = Vec4<float>::load(x + i * 4), m = Vec4<float>::load(mean + c), s = Vec4<float>::load(var + c);
  f32x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = (v[k] - m[k]) * (1.0f / sqrtf(s[k] + eps));
  Vec4<T>::store(out + i * 4, o);
}

// out[b, h, q, :] = softmax_n(logit_scale * <qv[q, h, :], K[b, n, h, :]>) @ V[b, :, h, :]     kv: [B, N, 2, H, dh]
// One workgroup of 4 waves per (h, b).  Dynamic LDS: Q*N scores + 4*Q*dh partial outputs + Q reciprocals.
template <typename T>
__global__ void __launch_bounds__(256) query_attention_kernel(const float* __restrict__ qv, int Q, const T* __restrict__ kv, int N,
                                                              int H, int dh, float logit_scale, float* __restrict__ out) {
  extern __shared__ float lds[];
  float* sc = lds;                      // [Q][N]
  float* part = lds + (size_t)Q * N;    // [4][Q*dh]
  float* inv = part + 4 * (size_t)Q * dh;  // [Q]
  const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Cc = H * dh;
  const T* base = kv + (size_t)b * N * 2 * Cc + (size_t)h * dh;
  for (int idx = tid; idx < Q * N; idx += 256) {
    const int q = idx / N, n = idx - q * N;
    const T* kp = base + (size_t)n * 2 * Cc;
    const float* qp = qv + (size_t)q * Cc + (size_t)h * dh;
    float s = 0.f;
    for (int d = 0; d < dh; ++d) s = fmaf(qp[d], to_f32(kp[d]), s);
    sc[idx] = s * logit_scale;
  }
  __syncthreads();
  for (int q = wave; q < Q; q += 4) {
    float m = -INFINITY;
    for (int n = lane; n < N; n += 64) m = fmaxf(m, sc[q * N + n]);
    m = wave_max(m);
    float z = 0.f;
    for (int n = lane; n < N; n += 64) {
      const float p = __expf(sc[q * N + n] - m);
      sc[q * N + n] = p;
      z += p;
    }
    z = wave_sum(z);
    if (lane == 0) inv[q] = 1.0f / z;
  }
  __syncthreads();
  const int total = Q * dh;
  for (int idx = lane; idx < total; idx += 64) {
    const int q = idx / dh, d = idx - q * dh;
    float acc = 0.f;
    for (int n = wave; n < N; n += 4) acc = fmaf(sc[q * N + n], to_f32(base[(size_t)n * 2 * Cc + Cc + d]), acc);
    part[wave * total + idx] = acc;
  }
  __syncthreads();
  for (int idx = tid; idx < total; idx += 256) {
    const int q = idx / dh, d = idx - q * dh;
    const float acc = (part[idx] + part[total + idx]) + (part[2 * total + idx] + part[3 * total + idx]);
    out[(((size_t)b * H + h) * Q + q) * dh + d] = acc * inv[q];
  }
}

// out[r, c] = act( <mean_q norm(x[r*ldx + q*D + :]), W[c, :]> + bias[c] ),  norm = eval-mode BatchNorm1d (if mean given).
// One wave per output element.
__global__ void __launch_bounds__(256) head_linear_kernel(const float* __restrict__ x, int64_t ldx, int nq,
                                                          const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                                          const float* __restrict__ W, const float* __restrict__ bias, int act,
                                                          float* __restrict__ out, int rows, int D, int n_out) {
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (w >= (int64_t)rows * n_out) return;
  const int r = (int)(w / n_out), c = (int)(w - (int64_t)r * n_out);
  const float* xr = x + (int64_t)r * ldx;
  const float* wr = W + (size_t)c * D;
  float acc = 0.f;
  for (int k = lane; k < D; k += 64) {
    float v = 0.f;
    const float m = mean ? mean[k] : 0.f, is = mean ? 1.0f / sqrtf(var[k] + eps) : 1.0f;
    for (int q = 0; q < nq; ++q) v += (xr[(size_t)q * D + k] - m) * is;
    if (nq > 1) v = v / (float)nq;
    acc = fmaf(v, wr[k], acc);
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    acc += bias ? bias[c] : 0.f;
    out[w] = act == HCT_ACT_TANH ? tanhf(acc) : acc;
  }
}

// ---- linear probing (LinearClassifier in training mode on frozen features, engine_downstream.py:70-117) ----------------

// nn.BatchNorm1d training statistics over the batch axis of [B, D]: mean, biased variance (used to normalise) and the
// running-statistics update with the unbiased variance (momentum 0.1).  One thread per channel, rows in index order.
__global__ void __launch_bounds__(256) batch_stats_kernel(const float* __restrict__ x, int B, int D, float momentum,
                                                          float* __restrict__ mean, float* __restrict__ var,
                                                          float* __restrict__ rmean, float* __restrict__ rvar) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= D) return;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += x[(size_t)b * D + k];
  const float m = s / (float)B;
  float q = 0.f;
  for (int b = 0; b < B; ++b) {
    const float d = x[(size_t)b * D + k] - m;
    q = fmaf(d, d, q);
  }
  mean[k] = m;
  var[k] = q / (float)B;
  if (rmean) {
    rmean[k] = (1.0f - momentum) * rmean[k] + momentum * m;
    rvar[k] = (1.0f - momentum) * rvar[k] + momentum * (q / (float)(B - 1));
  }
}

// nn.CrossEntropyLoss() (mean reduction, class-index targets): loss = mean_b( logsumexp(l_b) - l_b[t_b] ),
// dlogits[b, c] = (softmax(l_b)[c] - [c == t_b]) * dloss / B.  One workgroup; rows folded in a fixed order.
__global__ void __launch_bounds__(256) softmax_xent_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target, int B,
                                                           int Cn, const float* __restrict__ dloss, float* __restrict__ loss,
                                                           float* __restrict__ dlogits) {
  __shared__ float red[256];
  const int tid = threadIdx.x;
  const float g = (dloss ? dloss[0] : 1.0f) / (float)B;
  float acc = 0.f;
  for (int b = tid; b < B; b += 256) {
    const float* l = logits + (size_t)b * Cn;
    const int t = (int)target[b];
    float m = -INFINITY;
    for (int c = 0; c < Cn; ++c) m = fmaxf(m, l[c]);
    float z = 0.f;
    for (int c = 0; c < Cn; ++c) z += expf(l[c] - m);
    acc += (logf(z) + m) - (t >= 0 && t < Cn ? l[t] : NAN);  // a target outside [0, C) poisons the loss instead of reading out of bounds
    if (dlogits) {
      const float iz = 1.0f / z;
      for (int c = 0; c < Cn; ++c) dlogits[(size_t)b * Cn + c] = (expf(l[c] - m) * iz - (c == t ? 1.0f : 0.0f)) * g;
    }
  }
  red[tid] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0 && loss) loss[0] = red[0] / (float)B;
}

// parameter gradients of Linear(BatchNorm(x)):  dW[c, k] = sum_b dl[b, c] * (x[b, k] - mean[k]) / sqrt(var[k] + eps),
// db[c] = sum_b dl[b, c];  one thread per (c, k), rows in index order.
__global__ void __launch_bounds__(256) head_linear_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                                const float* __restrict__ var, float eps,
                                                                const float* __restrict__ dl, int B, int D, int Cn,
                                                                float* __restrict__ dW, float* __restrict__ db) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)Cn * D) return;
  const int c = (int)(i / D), k = (int)(i - (int64_t)c * D);
  const float m = mean ? mean[k] : 0.f, is = mean ? 1.0f / sqrtf(var[k] + eps) : 1.0f;
  float acc = 0.f, bs = 0.f;
  for (int b = 0; b < B; ++b) {
    const float d = dl[(size_t)b * Cn + c];
    acc = fmaf(d, (x[(size_t)b * D + k] - m) * is, acc);
    bs += d;
  }
  dW[i] = acc;
  if (k == 0 && db) db[c] = bs;
}

}  // namespace hct

extern "C" {

int hct_batchnorm_stats(const float* x, int B, int D, float momentum, float* mean, float* var, float* running_mean,
                        float* running_var, void* stream) {
  HCT_REQUIRE(x && mean && var && B > 1 && D > 0 && (!running_mean == !running_var),
              "hct_batchnorm_stats: bad arguments (training-mode BatchNorm needs more than one row)");
  hipLaunchKernelGGL(hct::batch_stats_kernel, dim3((D + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, B, D, momentum, mean, var,
                     running_mean, running_var);
  HCT_CHECK_LAUNCH("hct_batchnorm_stats");
  return 0;
}

int hct_softmax_xent(const float* logits, const int64_t* target, int B, int n_classes, const float* dloss, float* loss,
                     float* dlogits, void* stream) {
  HCT_REQUIRE(logits && target && B > 0 && n_classes > 0 && (loss || dlogits), "hct_softmax_xent: bad arguments");
  hipLaunchKernelGGL(hct::softmax_xent_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, target, B, n_classes, dloss, loss,
                     dlogits);
  HCT_CHECK_LAUNCH("hct_softmax_xent");
  return 0;
}

int hct_head_linear_wgrad(const float* x, const float* mean, const float* var, float eps, const float* dlogits, int B, int D,
                          int n_out, float* dW, float* db, void* stream) {
  HCT_REQUIRE(x && dlogits && dW && B > 0 && D > 0 && n_out > 0 && (!mean == !var), "hct_head_linear_wgrad: bad arguments");
  const int64_t n = (int64_t)n_out * D;
  hipLaunchKernelGGL(hct::head_linear_wgrad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, mean, var,
                     eps, dlogits, B, D, n_out, dW, db);
  HCT_CHECK_LAUNCH("hct_head_linear_wgrad");
  return 0;
}

int hct_channel_norm(const float* x, const float* mean, const float* var, float eps, void* out, int out_dtype, int64_t rows,
                     int C, void* stream) {
  HCT_REQUIRE(x && mean && var && out && rows > 0 && C > 0 && C % 4 == 0, "hct_channel_norm: bad arguments (C must be a multiple of 4)");
  HCT_REQUIRE(out_dtype == HCT_F32 || out_dtype == HCT_BF16, "hct_channel_norm: unsupported output dtype %d", out_dtype);
  const int64_t n4 = rows * (C / 4);
  HCT_DISPATCH_DTYPE(out_dtype, T,
                     hipLaunchKernelGGL(hct::channel_norm_kernel<T>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0,
                                        (hipStream_t)stream, x, mean, var, eps, (T*)out, C, n4));
  HCT_CHECK_LAUNCH("hct_channel_norm");
  return 0;
}

int hct_query_attention(const float* q, int Q, const void* kv, int kv_dtype, int B, int N, int H, int dh, float logit_scale,
                        float* out, void* stream) {
  HCT_REQUIRE(q && kv && out && Q > 0 && B > 0 && N > 0 && H > 0 && dh > 0, "hct_query_attention: bad arguments");
  HCT_REQUIRE(kv_dtype == HCT_F32 || kv_dtype == HCT_BF16, "hct_query_attention: unsupported kv dtype %d", kv_dtype);
  const size_t lds = ((size_t)Q * N + 4 * (size_t)Q * dh + Q) * sizeof(float);
  if (lds > 64 * 1024 || B > 65535) {
    hct::set_error("hct_query_attention: Q*N + 4*Q*dh + Q = %zu floats exceed the 64 KiB of LDS this kernel uses (or B > 65535)",
                   lds / sizeof(float));
    return HCT_E_UNSUPPORTED;
  }
  HCT_DISPATCH_DTYPE(kv_dtype, T,
                     hipLaunchKernelGGL(hct::query_attention_kernel<T>, dim3(H, B), dim3(256), lds, (hipStream_t)stream, q, Q,
                                        (const T*)kv, N, H, dh, logit_scale, out));
  HCT_CHECK_LAUNCH("hct_query_attention");
  return 0;
}

int hct_head_linear(const float* x, int64_t ldx, int nq, const float* mean, const float* var, float eps, const float* W,
                    const float* bias, int act, float* out, int rows, int D, int n_out, void* stream) {
  HCT_REQUIRE(x && W && out && rows > 0 && D > 0 && n_out > 0 && nq > 0 && ldx >= (int64_t)nq * D && (!mean == !var),
              "hct_head_linear: bad arguments");
  HCT_REQUIRE(act == HCT_ACT_NONE || act == HCT_ACT_TANH, "hct_head_linear: act must be HCT_ACT_NONE or HCT_ACT_TANH");
  const int64_t waves = (int64_t)rows * n_out;
  hipLaunchKernelGGL(hct::head_linear_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, nq, mean,
                     var, eps, W, bias, act, out, rows, D, n_out);
  HCT_CHECK_LAUNCH("hct_head_linear");
  return 0;
}

}  // extern "C"
