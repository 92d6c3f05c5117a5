// HBM-bound kernels of the MAE step: masking rank, patch gather, token assembly, LayerNorm fwd/bwd,
// masked-MSE loss (+ patchify fused), column sums, casts.  All are streaming kernels: one pass over
// their operands, 16-byte (fp32) / 8-byte (bf16) accesses per lane, wave-per-row where a row reduction
// is needed (64 lanes x 4 elements = one 1 KiB fp32 request per step).
#include "common.h"

// A/B builds (-DHCT_MISC_NT=n): bit 0 = the loss kernel's volume / prediction loads non-temporal, bit 1 = the patch gather's volume loads
#ifndef HCT_MISC_NT
#define HCT_MISC_NT 0
#endif

#include <stdarg.h>
#include <algorithm>

namespace hct {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return 0;
  set_error("%s: %s", what, hipGetErrorString(e));
  return (int)e;
}

// =============================================================================================
// masking: rank of each noise value inside its row (stable) -- mae.py:204-216
// =============================================================================================
__global__ void mask_rank_kernel(const float* __restrict__ noise, int L, int K, int32_t* __restrict__ ids_restore,
                                 int32_t* __restrict__ ids_shuffle, float* __restrict__ mask) {
  extern __shared__ float s_noise[];
  const int b = blockIdx.x;
  const float* row = noise + (size_t)b * L;
  for (int i = threadIdx.x; i < L; i += blockDim.x) s_noise[i] = row[i];
  __syncthreads();
  for (int l = threadIdx.x; l < L; l += blockDim.x) {
    const float v = s_noise[l];
    int r = 0;
    for (int j = 0; j < L; ++j) {
      const float w = s_noise[j];
      r += (w < v) || (w == v && j < l);
    }
    ids_restore[(size_t)b * L + l] = r;
    ids_shuffle[(size_t)b * L + r] = l;
    mask[(size_t)b * L + l] = r < K ? 0.0f : 1.0f;
  }
}

// =============================================================================================
// patch gather: kept tokens only, Conv3d weight order (c, ph, pw, pd) -- patch_embedding.py:149
// =============================================================================================
template <typename TX, typename T>
__global__ void patch_gather_kernel(const TX* __restrict__ x, const int32_t* __restrict__ ids_shuffle, int C, int S,
                                    int P, int L, int K, T* __restrict__ rows) {
  const int r = blockIdx.x;  // b*K + j
  const int b = r / K, j = r - b * K;
  const int g = S / P;
  const int l = ids_shuffle ? ids_shuffle[(size_t)b * L + j] : j;  // NULL: every patch in grid order (plain ViT)
  const int gh = l / (g * g), gw = (l / g) % g, gd = l % g;
  const int P4 = P >> 2;
  const int nvec = C * P * P * P4;
  T* out = rows + (size_t)r * (C * P * P * P);
  for (int v = threadIdx.x; v < nvec; v += blockDim.x) {
    int t = v;
    const int q = t % P4; t /= P4;
    const int pw = t % P; t /= P;
    const int ph = t % P; t /= P;
    const int c = t;
    const TX* src = x + ((((size_t)b * C + c) * S + (gh * P + ph)) * S + (gw * P + pw)) * S + gd * P + q * 4;
    Vec4<T>::store(out + (size_t)v * 4, (HCT_MISC_NT & 2) ? Vec4<TX>::load_nt(src) : Vec4<TX>::load(src));
  }
}

// Every patch in grid order (plain ViT / DINO: ids_shuffle == NULL): one workgroup per (volume, channel, gh, gw) takes the PENCIL of
// S/P patches along the contiguous axis -- P*P runs of S contiguous voxels in, one contiguous P^3 segment of each of the S/P patch rows
// out -- through LDS, so both sides move whole lines.  (The per-patch kernel above reads P-voxel runs: 48 B of every 128-B line at
// P = 12, 2.2 TB/s on DINO's 3 x 96^3 crops.)
template <typename TX, typename T>
__global__ void __launch_bounds__(256) patch_gather_pencil_kernel(const TX* __restrict__ x, int C, int S, int P, T* __restrict__ rows) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pg_smem[];
  T* lds = reinterpret_cast<T*>(pg_smem);  // [S/P][P*P][P]
  const int g = S / P, P2 = P * P, P3 = P2 * P;
  int w = blockIdx.x;
  const int gw = w % g; w /= g;
  const int gh = w % g; w /= g;
  const int c = w % C;
  const int b = w / C;
  const TX* src = x + ((((size_t)b * C + c) * S + gh * P) * S + gw * P) * S;
  const int S4 = S >> 2;
  for (int v = threadIdx.x; v < P2 * S4; v += blockDim.x) {
    const int r = v / S4, e = (v - r * S4) * 4;  // run r = ph * P + pw, voxel e of the run (4 voxels never straddle a patch: P % 4 == 0)
    const int ph = r / P, pw = r - ph * P;
    const f32x4 val = Vec4<TX>::load(src + ((size_t)ph * S + pw) * S + e);
    const int gd = e / P, pd = e - gd * P;
    Vec4<T>::store(lds + gd * P3 + r * P + pd, val);
  }
  __syncthreads();
  const int L = g * g * g;
  const size_t row0 = (size_t)b * L + ((size_t)gh * g + gw) * g;
  const int P34 = P3 >> 2;
  for (int v = threadIdx.x; v < g * P34; v += blockDim.x) {
    const int gd = v / P34, k = (v - gd * P34) * 4;
    Vec4<T>::store(rows + (row0 + gd) * ((size_t)C * P3) + (size_t)c * P3 + k, Vec4<T>::load(lds + gd * P3 + k));
  }
}

// =============================================================================================
// encoder input assembly  -- patch_embedding.py:155-156, mae.py:212,233-234
// =============================================================================================
template <typename T>
__global__ void encoder_assemble_fwd_kernel(const T* __restrict__ tok, const float* __restrict__ cls,
                                            const float* __restrict__ pos, const int32_t* __restrict__ ids_shuffle,
                                            int L, int K, int D, float* __restrict__ h0) {
  const int r = blockIdx.x;  // b*(K+1) + t
  const int b = r / (K + 1), t = r - b * (K + 1);
  float* out = h0 + (size_t)r * D;
  if (t == 0) {
    for (int d = threadIdx.x * 4; d < D; d += blockDim.x * 4) Vec4<float>::store(out + d, Vec4<float>::load(cls + d));
    return;
  }
  const int j = t - 1;
  const int l = ids_shuffle[(size_t)b * L + j];
  const T* src = tok + ((size_t)b * K + j) * D;
  for (int d = threadIdx.x * 4; d < D; d += blockDim.x * 4) {
    f32x4 v = Vec4<T>::load(src + d);
    if (pos) v += Vec4<float>::load(pos + (size_t)l * D + d);
    Vec4<float>::store(out + d, v);
  }
}

template <typename T>
__global__ void encoder_assemble_bwd_tok_kernel(const float* __restrict__ dh0, int K, int D, T* __restrict__ dtok) {
  const int r = blockIdx.x;  // b*K + j
  const int b = r / K, j = r - b * K;
  const float* src = dh0 + ((size_t)b * (K + 1) + 1 + j) * D;
  T* out = dtok + (size_t)r * D;
  for (int d = threadIdx.x * 4; d < D; d += blockDim.x * 4) Vec4<T>::store(out + d, Vec4<float>::load(src + d));
}

// dpos[l,:] = sum over b with l kept of dh0[b, 1+ids_restore[b,l], :]; fixed b order => deterministic
__global__ void encoder_assemble_bwd_pos_kernel(const float* __restrict__ dh0, const int32_t* __restrict__ ids_restore,
                                                int B, int L, int K, int D, float* __restrict__ dpos) {
  const int l = blockIdx.x;
  for (int d = threadIdx.x * 4; d < D; d += blockDim.x * 4) {
    f32x4 acc = {0, 0, 0, 0};
    // batches of 8 independent (index, row) loads; the additions keep the b order (bit-identical to the serial loop)
    for (int b0 = 0; b0 < B; b0 += 8) {
      int r[8];
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) r[u] = b0 + u < B ? ids_restore[(size_t)(b0 + u) * L + l] : K;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        v[u] = r[u] < K ? Vec4<float>::load(dh0 + ((size_t)(b0 + u) * (K + 1) + 1 + r[u]) * D + d) : f32x4{0, 0, 0, 0};
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (r[u] < K) acc += v[u];
    }
    Vec4<float>::store(dpos + (size_t)l * D + d, acc);
  }
}

template <typename T>
__global__ void vit_assemble_bwd_tok_kernel(const float* __restrict__ dh0, int L, int R, int D, T* __restrict__ dtok) {
  const int r = blockIdx.x;  // b*L + l
  const int b = r / L, l = r - b * L;
  const float* src = dh0 + ((size_t)b * (1 + R + L) + 1 + R + l) * D;
  T* out = dtok + (size_t)r * D;
  for (int d = threadIdx.x * 4; d < D; d += blockDim.x * 4) Vec4<T>::store(out + d, Vec4<float>::load(src + d));
}

// out[d] = sum_b src[b*stride + d]   (row 0 of each volume: cls-token gradients)
__global__ void strided_rowsum_kernel(const float* __restrict__ src, int B, size_t stride, int D, float* __restrict__ out) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= D) return;
  float acc = 0.f;
  for (int b0 = 0; b0 < B; b0 += 16) {  // 16 loads in flight, additions in b order
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = b0 + u < B ? src[(size_t)(b0 + u) * stride + d] : 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (b0 + u < B) acc += v[u];
  }
  out[d] = acc;
}

// =============================================================================================
// LayerNorm forward: one wave per row, two-pass statistics (mean, then centred variance)
// =============================================================================================
template <typename T>
__global__ void __launch_bounds__(256) layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int rows, int D, float eps,
                                                            T* __restrict__ y, float* __restrict__ mean,
                                                            float* __restrict__ rstd) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (size_t)row * D;
  float s = 0.f;
  for (int d = lane * 4; d < D; d += 256) {
    f32x4 v = Vec4<float>::load(xr + d);
    s += (v[0] + v[1]) + (v[2] + v[3]);
  }
  const float mu = wave_sum(s) / (float)D;
  float q = 0.f;
  for (int d = lane * 4; d < D; d += 256) {
    f32x4 v = Vec4<float>::load(xr + d) - mu;
    q += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  }
  const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
  if (lane == 0) {
    mean[row] = mu;
    rstd[row] = rs;
  }
  T* yr = y + (size_t)row * D;
  for (int d = lane * 4; d < D; d += 256) {
    f32x4 v = (Vec4<float>::load(xr + d) - mu) * rs;
    v = v * Vec4<float>::load(gamma + d) + Vec4<float>::load(beta + d);
    Vec4<T>::store(yr + d, v);
  }
}

#ifdef HCT_LN_NT
#define HCT_LN_NT_FWD HCT_LN_NT
#else
#define HCT_LN_NT_FWD 6  /* = the default of HCT_LN_NT below (bit 2: the forward's x load) */
#endif
// Register-resident variant (D <= 256 * NV): the row is loaded once (the kernel above reads it three times, one memory round
// trip per pass), each wave walks rows w, w+W, ... with gamma / beta held in registers.  Same arithmetic in the same order.
template <typename T, int NV>
__global__ void __launch_bounds__(256) layernorm_fwd_reg_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, int rows, int D, float eps,
                                                                T* __restrict__ y, float* __restrict__ mean,
                                                                float* __restrict__ rstd) {
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
  f32x4 g[NV], bt[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int d = lane * 4 + 256 * i;
    g[i] = d < D ? Vec4<float>::load(gamma + d) : f32x4{0, 0, 0, 0};
    bt[i] = d < D ? Vec4<float>::load(beta + d) : f32x4{0, 0, 0, 0};
  }
  for (int row = wid; row < rows; row += nw) {
    const float* xr = x + (size_t)row * D;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int d = lane * 4 + 256 * i;
      if (d < D) {
        v[i] = (HCT_LN_NT_FWD & 4) ? Vec4<float>::load_nt(xr + d) : Vec4<float>::load(xr + d);
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
      }
    }
    const float mu = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int d = lane * 4 + 256 * i;
      if (d < D) {
        const f32x4 c = v[i] - mu;
        q += (c[0] * c[0] + c[1] * c[1]) + (c[2] * c[2] + c[3] * c[3]);
      }
    }
    const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
    if (lane == 0) {
      mean[row] = mu;
      rstd[row] = rs;
    }
    T* yr = y + (size_t)row * D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int d = lane * 4 + 256 * i;
      if (d < D) {
        f32x4 o = (v[i] - mu) * rs;
        o = o * g[i] + bt[i];
        if (HCT_LN_NT_FWD & 8) Vec4<T>::store_nt(yr + d, o);
        else Vec4<T>::store(yr + d, o);
      }
    }
  }
}

// =============================================================================================
// LayerNorm backward + residual-gradient add + column partials (dgamma, dbeta, colsum(dx_total)).
//   dx = rstd * (dy*g - mean_d(dy*g) - xhat * mean_d(dy*g*xhat));   dx_total = dres + dx
// Each wave walks rows  w, w+W, w+2W ...; lane owns columns {lane*4 + 256*i}; column partials stay in
// registers (NV compile-time) and are reduced over the block's 4 waves through LDS, then written to
// partial[block][3][D]; a second kernel folds the partials in fixed order (deterministic).
// =============================================================================================
constexpr int kLnBwdBlocks = 1024;  // 4 workgroups = 16 waves per CU (36 KB of LDS each): the row loop is a load -> reduce -> store chain per wave

// Cache policy of the LayerNorm kernels' streams (A/B builds: -DHCT_LN_NT=n): bit 0 = the backward's fp32 dx store non-temporal (it
// is next read three GEMMs later), bit 1 = the backward's dy / x / residual-gradient loads non-temporal (last use), bit 2 = the
// forward's x load non-temporal, bit 3 = the forward's y store, bit 4 = the backward's shadow (GEMM operand) store.
// Measured inside the step (scripts/ab_step.py, variant libraries, two boxes): 4 -> -0.30 ms, 6 -> -0.44, 7 -> -0.45, 1 -> 0, 3 -> -0.1:
// the forward's pass over the fp32 residual stream was evicting what the GEMMs around it re-read.  The OUTPUTS the next GEMM reads must
// stay cacheable: 6 + 8 (forward's y store non-temporal) -> +0.5 ms, 6 + 16 (backward's shadow store) -> +0.2 ms.
#ifndef HCT_LN_NT
#define HCT_LN_NT 6
#endif
template <typename T> __device__ __forceinline__ f32x4 ln_load_nt(const T* p) { return Vec4<T>::load_nt(p); }

template <typename T, typename TS, int NV>
__global__ void __launch_bounds__(256) layernorm_bwd_kernel(const T* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* dres,
                                                            const int32_t* __restrict__ dres_rows,
                                                            int rows, int D, float* dx, TS* __restrict__ shadow,
                                                            float* __restrict__ partial, int want_colsum) {
  __shared__ float s_red[4][3][NV * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wid = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
  f32x4 g[NV], pg[NV], pb[NV], pc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int d = lane * 4 + 256 * i;
    g[i] = d < D ? Vec4<float>::load(gamma + d) : f32x4{0, 0, 0, 0};
    pg[i] = pb[i] = pc[i] = f32x4{0, 0, 0, 0};
  }
  for (int row = wid; row < rows; row += nw) {
    const float mu = mean[row], rs = rstd[row];
    // residual gradient of this row: row `row` of dres, or -- with a row map (the compact tail of the MAE decoder) -- row
    // dres_rows[row] of a compact matrix, nothing where the map says -1
    const int rrow = dres_rows ? dres_rows[row] : row;
    f32x4 dyv[NV], xh[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int d = lane * 4 + 256 * i;
      if (d < D) {
        dyv[i] = (HCT_LN_NT & 2) ? ln_load_nt(dy + (size_t)row * D + d) : Vec4<T>::load(dy + (size_t)row * D + d);
        xh[i] = (((HCT_LN_NT & 2) ? Vec4<float>::load_nt(x + (size_t)row * D + d) : Vec4<float>::load(x + (size_t)row * D + d)) - mu) * rs;
      } else {
        dyv[i] = xh[i] = f32x4{0, 0, 0, 0};
      }
      const f32x4 a = dyv[i] * g[i];
      const f32x4 bq = a * xh[i];
      s1 += (a[0] + a[1]) + (a[2] + a[3]);
      s2 += (bq[0] + bq[1]) + (bq[2] + bq[3]);
      pg[i] += dyv[i] * xh[i];
      pb[i] += dyv[i];
    }
    s1 = wave_sum(s1) / (float)D;
    s2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int d = lane * 4 + 256 * i;
      if (d < D) {
        f32x4 v = (dyv[i] * g[i] - s1 - xh[i] * s2) * rs;
        if (dres && rrow >= 0) v += (HCT_LN_NT & 2) ? Vec4<float>::load_nt(dres + (size_t)rrow * D + d) : Vec4<float>::load(dres + (size_t)rrow * D + d);
        if (HCT_LN_NT & 1) Vec4<float>::store_nt(dx + (size_t)row * D + d, v);
        else Vec4<float>::store(dx + (size_t)row * D + d, v);
        if (shadow) {
          if (HCT_LN_NT & 16) Vec4<TS>::store_nt(shadow + (size_t)row * D + d, v);
          else Vec4<TS>::store(shadow + (size_t)row * D + d, v);
        }
        pc[i] += v;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s_red[wave][0][i * 256 + lane * 4 + e] = pg[i][e];
      s_red[wave][1][i * 256 + lane * 4 + e] = pb[i][e];
      s_red[wave][2][i * 256 + lane * 4 + e] = pc[i][e];
    }
  }
  __syncthreads();
  const int nq = want_colsum ? 3 : 2;
  for (int idx = threadIdx.x; idx < nq * NV * 256; idx += 256) {
    const int qn = idx / (NV * 256), c = idx - qn * (NV * 256);
    if (c < D) {
      const float v = (s_red[0][qn][c] + s_red[1][qn][c]) + (s_red[2][qn][c] + s_red[3][qn][c]);
      partial[((size_t)blockIdx.x * 3 + qn) * D + c] = v;
    }
  }
}

// out_q[d] = sum_blocks partial[blk][q][d]
// 512 threads = 32 columns x 16 row groups; fixed summation order => deterministic
constexpr int kFoldThreads = 512;
__global__ void __launch_bounds__(512) fold_partials_kernel(const float* __restrict__ partial, int nblk, int nq_stride, int D,
                                                            float* o0, float* o1, float* o2) {
  __shared__ float s_red[16][32];
  const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int d = blockIdx.x * 32 + c;
  const int q = blockIdx.y;
  float* out = q == 0 ? o0 : (q == 1 ? o1 : o2);
  if (out == nullptr) return;
  float acc = 0.f;
  if (d < D)
    for (int b0 = rg; b0 < nblk; b0 += 16 * 8) {  // 8 loads in flight per thread, additions in block order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = b0 + 16 * u < nblk ? partial[((size_t)(b0 + 16 * u) * nq_stride + q) * D + d] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (b0 + 16 * u < nblk) acc += v[u];
    }
  s_red[rg][c] = acc;
  __syncthreads();
  if (rg == 0 && d < D) {
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) v += s_red[i][c];
    out[d] = v;
  }
}

// =============================================================================================
// decoder input assembly -- mae.py:257-265
// =============================================================================================
template <typename T>
__global__ void decoder_assemble_fwd_kernel(const T* __restrict__ e, const float* __restrict__ mask_token,
                                            const float* __restrict__ dec_cls, const float* __restrict__ dec_pos,
                                            const int32_t* __restrict__ ids_restore, int L, int K, int D,
                                            float* __restrict__ y) {
  const int r = blockIdx.x;  // b*(L+1) + t
  const int b = r / (L + 1), t = r - b * (L + 1);
  float* out = y + (size_t)r * D;
  if (t == 0) {
    const T* src = e + (size_t)b * (K + 1) * D;
    for (int d = threadIdx.x * 4; d < D; d += blockDim.x * 4)
      Vec4<float>::store(out + d, Vec4<T>::load(src + d) + Vec4<float>::load(dec_cls + d));
    return;
  }
  const int l = t - 1;
  const int rk = ids_restore[(size_t)b * L + l];
  const T* src = e + ((size_t)b * (K + 1) + 1 + rk) * D;
  for (int d = threadIdx.x * 4; d < D; d += blockDim.x * 4) {
    f32x4 v = rk < K ? Vec4<T>::load(src + d) : Vec4<float>::load(mask_token + d);
    Vec4<float>::store(out + d, v + Vec4<float>::load(dec_pos + (size_t)l * D + d));
  }
}

template <typename T>
__global__ void decoder_assemble_bwd_kernel(const float* __restrict__ dy, const int32_t* __restrict__ ids_shuffle, int L,
                                            int K, int D, T* __restrict__ de) {
  const int r = blockIdx.x;  // b*(K+1) + t
  const int b = r / (K + 1), t = r - b * (K + 1);
  const int srow = t == 0 ? 0 : 1 + ids_shuffle[(size_t)b * L + (t - 1)];
  const float* src = dy + ((size_t)b * (L + 1) + srow) * D;
  T* out = de + (size_t)r * D;
  for (int d = threadIdx.x * 4; d < D; d += blockDim.x * 4) Vec4<T>::store(out + d, Vec4<float>::load(src + d));
}

// partial[blk][0][d] = sum over the block's volumes of the masked rows of dy; partial[blk][1][d] = cls rows
constexpr int kAsmBlocks = 256;
__global__ void decoder_assemble_bwd_reduce_kernel(const float* __restrict__ dy, const int32_t* __restrict__ ids_shuffle,
                                                   int B, int L, int K, int D, float* __restrict__ partial) {
  for (int d = threadIdx.x * 4; d < D; d += blockDim.x * 4) {
    f32x4 am = {0, 0, 0, 0}, ac = {0, 0, 0, 0};
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
      const float* base = dy + (size_t)b * (L + 1) * D;
      ac += Vec4<float>::load(base + d);
      for (int j0 = K; j0 < L; j0 += 8) {  // 8 independent (index, row) loads; additions in j order
        int l[8];
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) l[u] = j0 + u < L ? ids_shuffle[(size_t)b * L + j0 + u] : 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = Vec4<float>::load(base + (size_t)(1 + l[u]) * D + d);
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (j0 + u < L) am += v[u];
      }
    }
    Vec4<float>::store(partial + ((size_t)blockIdx.x * 2 + 0) * D + d, am);
    Vec4<float>::store(partial + ((size_t)blockIdx.x * 2 + 1) * D + d, ac);
  }
}

// =============================================================================================
// masked MSE fused with patchify (+ optional per-patch target normalisation) -- mae.py:277-301
// one block per prediction row (b, t); t == 0 is the cls row (dropped, mae.py:273).
// =============================================================================================
__device__ __forceinline__ float block_sum_256(float v, float* s_tmp) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_tmp[threadIdx.x >> 6] = v;
  __syncthreads();
  return (s_tmp[0] + s_tmp[1]) + (s_tmp[2] + s_tmp[3]);
}

template <typename TX, typename T>
__global__ void __launch_bounds__(256) masked_mse_kernel(const T* __restrict__ pred, const TX* __restrict__ x,
                                                         const float* __restrict__ mask, int C, int S, int P, int L,
                                                         int norm_pix, float inv_masksum, float* __restrict__ row_loss,
                                                         T* __restrict__ dpred, const float* __restrict__ dscale, float hscale,
                                                         const int32_t* __restrict__ masked_ids, int K) {
  __shared__ float s_tmp[4];
  const int r = blockIdx.x;
  const int pd = P * P * P * C;
  const T* prow = pred + (size_t)r * pd;
  T* drow = dpred ? dpred + (size_t)r * pd : nullptr;
  int b, l;
  size_t lrow;  // slot of this row's loss
  if (masked_ids) {
    // compact form: pred / dpred hold ONLY the masked patches' rows, row r = (volume b, j-th masked patch in shuffle order),
    // masked_ids = ids_shuffle (mae.py:209: its entries K.. are the removed patches); the loss never sees another row (mae.py:298-299)
    const int Lm = L - K;
    b = r / Lm;
    l = masked_ids[(size_t)b * L + K + (r - b * Lm)];
    lrow = r;
  } else {
    b = r / (L + 1);
    const int t = r - b * (L + 1);
    if (t == 0) {
      if (drow)
        for (int k = threadIdx.x * 4; k < pd; k += 1024) Vec4<T>::store(drow + k, f32x4{0, 0, 0, 0});
      return;
    }
    l = t - 1;
    lrow = (size_t)b * L + l;
    const float m = mask[lrow];
    if (m == 0.f) {  // kept token: contributes nothing (mask = 0) and has zero gradient
      if (threadIdx.x == 0 && row_loss) row_loss[lrow] = 0.f;
      if (drow)
        for (int k = threadIdx.x * 4; k < pd; k += 1024) Vec4<T>::store(drow + k, f32x4{0, 0, 0, 0});
      return;
    }
  }
  const int g = S / P;
  const int gh = l / (g * g), gw = (l / g) % g, gd = l % g;
  const TX* vol = x + (size_t)b * C * S * S * S;
  auto tgt_at = [&](int k) -> float {  // patchify order (ph, pw, pd, c), c fastest -- mae.py:166-168
    const int c = k % C;
    int u = k / C;
    const int pz = u % P; u /= P;
    const int pw = u % P; u /= P;
    const int ph = u;
    return to_f32(vol[(((size_t)c * S + (gh * P + ph)) * S + (gw * P + pw)) * S + gd * P + pz]);
  };
  float mu = 0.f, rsd = 1.f;
  if (norm_pix) {
    float s = 0.f;
    for (int k = threadIdx.x; k < pd; k += 256) s += tgt_at(k);
    mu = block_sum_256(s, s_tmp) / (float)pd;
    float q = 0.f;
    for (int k = threadIdx.x; k < pd; k += 256) {
      const float dlt = tgt_at(k) - mu;
      q += dlt * dlt;
    }
    const float var = block_sum_256(q, s_tmp) / (float)(pd - 1);  // unbiased, mae.py:292
    rsd = 1.0f / sqrtf(var + 1.0e-6f);
  }
  const float gscale = 2.0f * inv_masksum / (float)pd * (dscale ? *dscale : 1.0f) * hscale;
  float sse = 0.f;
  if (C == 1) {
    for (int k = threadIdx.x * 4; k < pd; k += 1024) {
      int u = k;
      const int pz = u % P; u /= P;
      const int pw = u % P; u /= P;
      const int ph = u;
      const TX* tp = vol + ((size_t)(gh * P + ph) * S + (gw * P + pw)) * S + gd * P + pz;
      f32x4 tv = (HCT_MISC_NT & 1) ? Vec4<TX>::load_nt(tp) : Vec4<TX>::load(tp);
      tv = (tv - mu) * rsd;
      const f32x4 df = ((HCT_MISC_NT & 1) ? Vec4<T>::load_nt(prow + k) : Vec4<T>::load(prow + k)) - tv;
      sse += (df[0] * df[0] + df[1] * df[1]) + (df[2] * df[2] + df[3] * df[3]);
      if (drow) Vec4<T>::store(drow + k, df * gscale);
    }
  } else {
    for (int k = threadIdx.x; k < pd; k += 256) {
      const float df = to_f32(prow[k]) - (tgt_at(k) - mu) * rsd;
      sse += df * df;
      if (drow) drow[k] = from_f32<T>(df * gscale);
    }
  }
  sse = block_sum_256(sse, s_tmp);
  if (threadIdx.x == 0 && row_loss) row_loss[lrow] = sse / (float)pd;
}

__global__ void __launch_bounds__(256) loss_fold_kernel(const float* __restrict__ row_loss, int n, float inv_masksum,
                                                        float* __restrict__ loss) {
  __shared__ float s_tmp[4];
  float s = 0.f;
  for (int i0 = threadIdx.x; i0 < n; i0 += 256 * 8) {  // 8 loads in flight per thread, additions in index order
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = i0 + u * 256 < n ? row_loss[i0 + u * 256] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + u * 256 < n) s += v[u];
  }
  s = block_sum_256(s, s_tmp);
  if (threadIdx.x == 0) *loss = s * inv_masksum;
}

template <typename T>
__global__ void unpatchify_kernel(const T* __restrict__ pred, int has_cls, int C, int S, int P, float* __restrict__ vol) {
  // one block per (b, l); writes the patch's voxels -- mae.py:188-190
  const int g = S / P, L = g * g * g;
  const int r = blockIdx.x;
  const int b = r / L, l = r - b * L;
  const int gh = l / (g * g), gw = (l / g) % g, gd = l % g;
  const int pd = P * P * P * C;
  const T* prow = pred + ((size_t)b * (L + has_cls) + has_cls + l) * pd;
  for (int k = threadIdx.x; k < pd; k += blockDim.x) {
    const int c = k % C;
    int u = k / C;
    const int pz = u % P; u /= P;
    const int pw = u % P; u /= P;
    const int ph = u;
    vol[((((size_t)b * C + c) * S + (gh * P + ph)) * S + (gw * P + pw)) * S + gd * P + pz] = to_f32(prow[k]);
  }
}

// =============================================================================================
// column sums (bias gradients): grid (col blocks of 256, row chunks) -> partial -> fold
// =============================================================================================
template <typename T>
__global__ void __launch_bounds__(256) colsum_partial_kernel(const T* __restrict__ x, int rows, int cols, int64_t ld,
                                                             int rows_per_chunk, float* __restrict__ partial) {
  __shared__ float s_red[4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = blockIdx.x * 256 + lane * 4;
  const int r0 = blockIdx.y * rows_per_chunk;
  const int r1 = min(rows, r0 + rows_per_chunk);
  f32x4 acc = {0, 0, 0, 0};
  if (c0 < cols)
    for (int r = r0 + wave; r < r1; r += 4 * 8) {  // 8 row loads in flight per lane; added in row order (same sums as one by one)
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = r + 4 * u < r1 ? Vec4<T>::load(x + (size_t)(r + 4 * u) * ld + c0) : f32x4{0, 0, 0, 0};
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (r + 4 * u < r1) acc += v[u];
    }
#pragma unroll
  for (int e = 0; e < 4; ++e) s_red[wave][lane * 4 + e] = acc[e];
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < cols)
    partial[(size_t)blockIdx.y * cols + c] =
        (s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + (s_red[2][threadIdx.x] + s_red[3][threadIdx.x]);
}

// =============================================================================================
// casts
// =============================================================================================
template <typename TI, typename TO>
__global__ void cast_kernel(const TI* __restrict__ src, TO* __restrict__ dst, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n) {
      Vec4<TO>::store(dst + i, Vec4<TI>::load(src + i));
    } else {
      for (int64_t k = i; k < n; ++k) dst[k] = from_f32<TO>(to_f32(src[k]));
    }
  }
}

template <typename TI, typename TO>
__global__ void __launch_bounds__(256) transpose_cast_kernel(const TI* __restrict__ src, TO* __restrict__ dst, int rows,
                                                             int cols) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? to_f32(src[(size_t)r * cols + c]) : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;  // dst[c][r]
    if (c < cols && r < rows) dst[(size_t)c * rows + r] = from_f32<TO>(tile[tx][i]);
  }
}

// several folds in one launch: blockIdx.y walks the (job, quantity) pairs, blockIdx.x the 32-column groups of the widest job
struct FoldBatch {
  FoldJob jobs[FoldSink::kMax];
  int first_y[FoldSink::kMax + 1];
  int n;
};
__global__ void __launch_bounds__(512) fold_batch_kernel(FoldBatch b) {
  __shared__ float s_red[16][32];
  int j = 0;
  while (j + 1 < b.n && (int)blockIdx.y >= b.first_y[j + 1]) ++j;
  const FoldJob& job = b.jobs[j];
  const int q = blockIdx.y - b.first_y[j];
  const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int d = blockIdx.x * 32 + c;
  float* out = job.out[q];
  if (out == nullptr || (int)blockIdx.x * 32 >= job.D) return;  // (uniform over the workgroup)
  const float* partial = job.partial;
  const int nblk = job.nblk, nq_stride = job.nq_stride, D = job.D;
  float acc = 0.f;
  if (d < D)
    for (int b0 = rg; b0 < nblk; b0 += 16 * 8) {  // the loop of fold_partials_kernel: same order of additions, same result
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = b0 + 16 * u < nblk ? partial[((size_t)(b0 + 16 * u) * nq_stride + q) * D + d] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (b0 + 16 * u < nblk) acc += v[u];
    }
  s_red[rg][c] = acc;
  __syncthreads();
  if (rg == 0 && d < D) {
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) v += s_red[i][c];
    out[d] = v;
  }
}

thread_local FoldSink* g_fold_sink = nullptr;

int fold_partials(const FoldJob& job, hipStream_t s) {
  if (g_fold_sink && g_fold_sink->n < FoldSink::kMax) {
    g_fold_sink->jobs[g_fold_sink->n++] = job;
    return 0;
  }
  hipLaunchKernelGGL(fold_partials_kernel, dim3((job.D + 31) / 32, job.nq), dim3(kFoldThreads), 0, s, job.partial, job.nblk, job.nq_stride,
                     job.D, job.out[0], job.nq > 1 ? job.out[1] : (float*)nullptr, job.nq > 2 ? job.out[2] : (float*)nullptr);
  return check_hip(hipGetLastError(), "fold_partials");
}

int fold_flush(FoldSink& sink, hipStream_t s) {
  if (sink.n == 0) return 0;
  FoldBatch b;
  int y = 0, maxd = 0;
  for (int i = 0; i < sink.n; ++i) {
    b.jobs[i] = sink.jobs[i];
    b.first_y[i] = y;
    y += sink.jobs[i].nq;
    maxd = sink.jobs[i].D > maxd ? sink.jobs[i].D : maxd;
  }
  for (int i = sink.n; i <= FoldSink::kMax; ++i) b.first_y[i] = y;
  for (int i = sink.n; i < FoldSink::kMax; ++i) b.jobs[i] = FoldJob{nullptr, 0, 0, 0, 0, {nullptr, nullptr, nullptr}};
  b.n = sink.n;
  sink.n = 0;
  hipLaunchKernelGGL(fold_batch_kernel, dim3((maxd + 31) / 32, y), dim3(kFoldThreads), 0, s, b);
  return check_hip(hipGetLastError(), "fold_flush");
}

int fold_rows(const float* partial, int nblk, int D, float* out, hipStream_t s) {
  return fold_partials(FoldJob{partial, nblk, 1, D, 1, {out, nullptr, nullptr}}, s);
}


// =============================================================================================
// Separable 3-D Gaussian smoothing (RandGaussianSmoothd of mae3d_transforms, src/data/transforms.py:230-238): one axis per
// launch, 4 consecutive voxels of the fastest axis per thread, zero padding outside the volume (MONAI's separable_filtering
// default).  taps [B][3][kGaussTaps]: the sample's centred 1-D kernels (host-computed: erf differences, truncated at 4 sigma ->
// at most 9 taps for sigma <= 1.06), zero beyond the tail.  Samples whose transform did not fire: the first pass copies them,
// the other two skip them.
// =============================================================================================
constexpr int kGaussTaps = 9;
__global__ void __launch_bounds__(256) gaussian_pass_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int S, int axis,
                                                            const float* __restrict__ taps, const unsigned char* __restrict__ apply,
                                                            int copy_skipped, int64_t n4) {
  const int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i4 >= n4) return;
  const int s4 = S >> 2;
  const int z4 = (int)(i4 % s4);
  int64_t t = i4 / s4;
  const int y = (int)(t % S); t /= S;
  const int x = (int)(t % S); t /= S;
  const int b = (int)(t / C);
  const int64_t base = i4 * 4;
  if (!apply[b]) {
    if (copy_skipped) Vec4<float>::store(out + base, Vec4<float>::load(in + base));
    return;
  }
  const float* k = taps + ((int64_t)b * 3 + axis) * kGaussTaps;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (axis == 2) {  // along the contiguous axis: 4 outputs share a 12-voxel window
    float w[4 + kGaussTaps - 1];
#pragma unroll
    for (int j = 0; j < 4 + kGaussTaps - 1; ++j) {
      const int z = z4 * 4 + j - kGaussTaps / 2;
      w[j] = (z >= 0 && z < S) ? in[base - z4 * 4 + z] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < kGaussTaps; ++u) {
      const float kv = k[u];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] += kv * w[q + u];
    }
  } else {
    const int pos = axis == 0 ? x : y;
    const int64_t stride = axis == 0 ? (int64_t)S * S : S;
#pragma unroll
    for (int u = 0; u < kGaussTaps; ++u) {
      const int p = pos + u - kGaussTaps / 2;
      if (p >= 0 && p < S) acc += Vec4<float>::load(in + base + (int64_t)(u - kGaussTaps / 2) * stride) * k[u];
    }
  }
  Vec4<float>::store(out + base, acc);
}

}  // namespace hct

// =================================================================================================
// C ABI
// =================================================================================================
using namespace hct;

template <typename T, typename TS>
static int launch_ln_bwd(const void* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                         const float* dres, const int32_t* dres_rows, int rows, int D, float* dx, void* shadow, float* partial,
                         int want_colsum, int nblk, hipStream_t s) {
#define HCT_LN_CASE(NV)                                                                                               \
  hipLaunchKernelGGL((layernorm_bwd_kernel<T, TS, NV>), dim3(nblk), dim3(256), 0, s, (const T*)dy, x, mean, rstd, gamma, \
                     dres, dres_rows, rows, D, dx, (TS*)shadow, partial, want_colsum)
  const int nv = (D + 255) / 256;
  switch (nv) {
    case 1: HCT_LN_CASE(1); break;
    case 2: HCT_LN_CASE(2); break;
    case 3: HCT_LN_CASE(3); break;
    case 4: HCT_LN_CASE(4); break;
    default: set_error("hct_layernorm_bwd: D=%d > 1024 unsupported", D); return HCT_E_UNSUPPORTED;
  }
#undef HCT_LN_CASE
  return 0;
}

// ViT encoder input (vit.py:144-162): class token, register tokens, position-embedded patch tokens
namespace hct {
template <typename T>
__global__ void vit_assemble_fwd_kernel(const T* __restrict__ tok, const float* __restrict__ cls, const float* __restrict__ reg,
                                        const float* __restrict__ pos, int L, int R, int D, float* __restrict__ h, int64_t total4) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total4) return;
  const int d4 = D >> 2, T1 = 1 + R + L;
  const int c = (int)(gid % d4) * 4;
  const int64_t row = gid / d4;
  const int t = (int)(row % T1);
  const int64_t b = row / T1;
  f32x4 v;
  if (t == 0) v = Vec4<float>::load(cls + c);
  else if (t <= R) v = Vec4<float>::load(reg + (int64_t)(t - 1) * D + c);
  else {
    const int l = t - 1 - R;
    v = Vec4<T>::load(tok + (b * L + l) * D + c);
    if (pos) v += Vec4<float>::load(pos + (int64_t)l * D + c);
  }
  Vec4<float>::store(h + row * D + c, v);
}
}  // namespace hct

namespace hct {
// HU windows (transforms.py:119-133 via MONAI ScaleIntensityRange, b_min 0, b_max 1, clip): W output channels per volume,
//   out[b, w, v] = clip((hu[b, v] - a_min[w]) / (a_max[w] - a_min[w]), 0, 1)
// fp32 subtract, IEEE divide, clip -- the operation order of the numpy / torch expression, so the fp32 result is the same
// bit pattern; one pass over the HU volume feeds all W windows.  TOut = float or IEEE half (the persistent cache's type).
template <typename TIn, typename TOut>
__global__ void __launch_bounds__(256) hu_window_kernel(const TIn* __restrict__ hu, TOut* __restrict__ out, int W, int64_t vox4,
                                                         const float* __restrict__ a_min, const float* __restrict__ a_max) {
  const int b = blockIdx.y;
  const TIn* src = hu + (int64_t)b * vox4 * 4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < vox4; i += (int64_t)gridDim.x * 256) {
    const f32x4 x = Vec4<TIn>::load(src + i * 4);
    for (int w = 0; w < W; ++w) {
      const float lo = a_min[w], range = a_max[w] - a_min[w];
      f32x4 y;
#pragma unroll
      for (int q = 0; q < 4; ++q) y[q] = fminf(fmaxf((x[q] - lo) / range, 0.0f), 1.0f);
      Vec4<TOut>::store(out + (((int64_t)b * W + w) * vox4 + i) * 4, y);
    }
  }
}
}  // namespace hct


// input transforms (transforms.py:193-228): cast + per-sample axis flips + intensity shift; one thread per 4 voxels of the
// innermost axis (a flipped innermost axis is read as a reversed group of 4)
namespace hct {
template <typename TIn>
__global__ void augment_volume_kernel(const TIn* __restrict__ in, float* __restrict__ out, int C, int S, const unsigned char* __restrict__ flip,
                                      const float* __restrict__ shift, int64_t total4) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total4) return;
  const int s4 = S >> 2;
  const int k4 = (int)(gid % s4);
  int64_t r = gid / s4;
  const int j = (int)(r % S); r /= S;
  const int i = (int)(r % S); r /= S;  // r = b * C + c
  const int b = (int)(r / C);
  const unsigned f = flip ? flip[b] : 0u;
  const float sh = shift ? shift[b] : 0.f;
  const int si = (f & 1u) ? S - 1 - i : i, sj = (f & 2u) ? S - 1 - j : j;
  const TIn* row = in + ((r * S + si) * S + sj) * S;
  f32x4 v;
  if (f & 4u) {
    const int k0 = S - 4 - k4 * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = (float)row[k0 + 3 - q] + sh;
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = (float)row[k4 * 4 + q] + sh;
  }
  Vec4<float>::store(out + ((r * S + i) * S + j) * S + k4 * 4, v);
}
}  // namespace hct

// trilinear resize of the position table (pos_embed.py:102-153): one thread per (output token, 4 channels); source index
// arithmetic in fp32 exactly as ATen's area_pixel_compute_source_index (align_corners = false, negative clamped to 0)
namespace hct {
__global__ void pos_embed_interp3d_kernel(const float* __restrict__ src, int gs, float* __restrict__ dst, int gd, int D, int extra) {
  const int d4 = D >> 2;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t ntok = (int64_t)extra + (int64_t)gd * gd * gd;
  if (gid >= ntok * d4) return;
  const int tok = (int)(gid / d4), c = (int)(gid - (int64_t)tok * d4) * 4;
  if (tok < extra) {
    Vec4<float>::store(dst + (int64_t)tok * D + c, Vec4<float>::load(src + (int64_t)tok * D + c));
    return;
  }
  const int t = tok - extra;
  const int oz = t % gd, oy = (t / gd) % gd, ox = t / (gd * gd);
  const float scale = (float)gs / (float)gd;
  int i0[3], i1[3];
  float w1[3];
  const int o[3] = {ox, oy, oz};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float cs = fmaxf(0.f, ((float)o[a] + 0.5f) * scale - 0.5f);
    i0[a] = (int)cs;
    i1[a] = min(i0[a] + 1, gs - 1);
    w1[a] = cs - (float)i0[a];
  }
  f32x4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float w = (a ? w1[0] : 1.f - w1[0]) * (b ? w1[1] : 1.f - w1[1]) * (e ? w1[2] : 1.f - w1[2]);
        const int64_t row = extra + ((int64_t)(a ? i1[0] : i0[0]) * gs + (b ? i1[1] : i0[1])) * gs + (e ? i1[2] : i0[2]);
        acc += w * Vec4<float>::load(src + row * D + c);
      }
  Vec4<float>::store(dst + (int64_t)tok * D + c, acc);
}
}  // namespace hct

template <typename T>
__global__ void __launch_bounds__(256) scale_unless_one_kernel(T* __restrict__ buf, int64_t n4, const float* __restrict__ scale) {
  const float sc = *scale;
  if (sc == 1.0f) return;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
    Vec4<T>::store(buf + i * 4, Vec4<T>::load(buf + i * 4) * sc);
}

namespace hct {
__global__ void __launch_bounds__(256) tail_rows_kernel(const int32_t* __restrict__ ids_restore, int B, int L, int K,
                                                        int32_t* __restrict__ tail_rows, int32_t* __restrict__ tail_inv) {
  const int r = blockIdx.x * 256 + threadIdx.x;  // decoder row b * (L + 1) + t
  if (r >= B * (L + 1)) return;
  const int b = r / (L + 1), t = r - b * (L + 1);
  int c = -1;
  if (t > 0) {
    const int rank = ids_restore[(size_t)b * L + (t - 1)];  // position of patch t-1 in the shuffle (mae.py:210)
    if (rank >= K) {
      c = b * (L - K) + (rank - K);
      tail_rows[c] = r;
    }
  }
  tail_inv[r] = c;
}

__global__ void __launch_bounds__(256) gather_rows16_kernel(const uint4* __restrict__ src, const int32_t* __restrict__ idx, int n_rows,
                                                            int chunks, uint4* __restrict__ dst) {
  const int64_t total = (int64_t)n_rows * chunks;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / chunks), c = (int)(i - (int64_t)r * chunks);
    const int sr = idx[r];
    uint4 v = {0u, 0u, 0u, 0u};
    if (sr >= 0) v = src[(size_t)sr * chunks + c];
    dst[i] = v;
  }
}

// ---- first decoder block: the masked tokens' input is mask_token + pos[l] for every volume (mae.py:259-265), so LayerNorm and
// the qkv Linear of THAT block need them once per position, not once per (volume, position) ----------------------------------
// rows of the "cat" matrices: c < Nc = B (K+1): class token + kept patches of volume b in shuffle order (the layout of the
// decoder_embed output); c = Nc + l: the table row of patch position l.
//   kept_rows[c]  (length Nc + L): decoder row b (L+1) + t of compact row c, -1 for the table rows
//   cat_idx[r]    (length B (L+1)): cat row that holds decoder row r's LayerNorm / qkv values
__global__ void __launch_bounds__(256) dec0_index_kernel(const int32_t* __restrict__ ids_restore, int B, int L, int K,
                                                         int32_t* __restrict__ kept_rows, int32_t* __restrict__ cat_idx) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  const int Nc = B * (K + 1);
  if (r < L) kept_rows[Nc + r] = -1;
  if (r >= B * (L + 1)) return;
  const int b = r / (L + 1), t = r - b * (L + 1);
  int c;
  if (t == 0) c = b * (K + 1);
  else {
    const int rank = ids_restore[(size_t)b * L + (t - 1)];
    c = rank < K ? b * (K + 1) + 1 + rank : Nc + (t - 1);
  }
  if (c < Nc) kept_rows[c] = r;
  cat_idx[r] = c;
}

__global__ void __launch_bounds__(256) dec0_table_kernel(const float* __restrict__ mask_token, const float* __restrict__ pos, int D,
                                                         float* __restrict__ out) {  // out[l] = mask_token + pos[l]
  const int l = blockIdx.x;
  for (int d = threadIdx.x * 4; d < D; d += 1024)
    Vec4<float>::store(out + (size_t)l * D + d, Vec4<float>::load(mask_token + d) + Vec4<float>::load(pos + (size_t)l * D + d));
}

// table rows of the cat gradient from the full [B (L+1), W] gradient: out[l] = sum over the volumes that masked position l of
// g[b (L+1) + 1 + l].  One block per (position, 256 columns): 64 column quads x 4 volume groups, 8 row loads in flight per thread,
// the four groups' sums added in group order (fixed summation order).  (The kept rows are a plain row gather.)
template <typename T>
__global__ void __launch_bounds__(256) dec0_table_sum_kernel(const T* __restrict__ g, const int32_t* __restrict__ ids_restore, int B, int L, int K,
                                                             int W, T* __restrict__ out) {
  __shared__ f32x4 s_part[4][64];
  const int l = blockIdx.x, q = threadIdx.x & 63, vg = threadIdx.x >> 6;
  const int col = blockIdx.y * 256 + q * 4;
  f32x4 acc = {0, 0, 0, 0};
  if (col < W) {
    for (int b0 = vg; b0 < B; b0 += 32) {
      int rk[8];
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) rk[u] = b0 + 4 * u < B ? ids_restore[(size_t)(b0 + 4 * u) * L + l] : -1;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        v[u] = rk[u] >= K ? Vec4<T>::load(g + ((size_t)(b0 + 4 * u) * (L + 1) + 1 + l) * W + col) : f32x4{0, 0, 0, 0};
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
  }
  s_part[vg][q] = acc;
  __syncthreads();
  if (vg == 0 && col < W) Vec4<T>::store(out + (size_t)l * W + col, (s_part[0][q] + s_part[1][q]) + (s_part[2][q] + s_part[3][q]));
}

// dmask_token[d] = sum_blk partial[blk][0][d] (masked rows of the residual gradient, decoder_assemble_bwd_reduce_kernel)
//                + sum_l dcat[Nc + l][d] (the table rows' LayerNorm gradient);   ddec_cls[d] = sum_b dcat[b (K+1)][d]
// 512 threads = 32 columns x 16 row groups, 8 loads in flight per thread, fixed order.
__global__ void __launch_bounds__(512) dec0_token_grads_kernel(const float* __restrict__ partial, int nblk, const float* __restrict__ dcat,
                                                               int B, int L, int K, int D, float* __restrict__ dmask, float* __restrict__ dcls) {
  __shared__ float s_m[16][32], s_c[16][32];
  const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int d = blockIdx.x * 32 + c;
  const size_t Nc = (size_t)B * (K + 1);
  auto sum_rows = [&](const float* base, int n, size_t stride) {
    float acc = 0.f;
    for (int r0 = rg; r0 < n; r0 += 16 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = r0 + 16 * u < n ? base[(size_t)(r0 + 16 * u) * stride + d] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    return acc;
  };
  float m = 0.f, cl = 0.f;
  if (d < D) {
    m = sum_rows(partial, nblk, (size_t)2 * D) + sum_rows(dcat + Nc * D, L, D);
    cl = sum_rows(dcat, B, (size_t)(K + 1) * D);
  }
  s_m[rg][c] = m;
  s_c[rg][c] = cl;
  __syncthreads();
  if (rg == 0 && d < D) {
    float a = 0.f, bsum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { a += s_m[i][c]; bsum += s_c[i][c]; }
    dmask[d] = a;
    dcls[d] = bsum;
  }
}

int scale_unless_one(void* buf, int dtype, int64_t n, const float* scale, hipStream_t s) {
  HCT_REQUIRE(n % 4 == 0 && scale, "scale_unless_one: n %% 4 != 0 or null scale");
  const int blocks = (int)std::min<int64_t>(2048, (n / 4 + 255) / 256);
  if (blocks == 0) return 0;
  HCT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(scale_unless_one_kernel<T>, dim3(blocks), dim3(256), 0, s, (T*)buf, n / 4, scale));
  HCT_CHECK_LAUNCH("scale_unless_one");
  return 0;
}

int masked_mse_launch(const void* pred, int pred_dtype, const void* x, int x_dtype, const float* mask, int B, int C, int S, int P,
                      int norm_pix, float mask_sum, float* row_loss, float* loss, void* dpred, const float* dpred_scale,
                      float host_scale, hipStream_t s, const int32_t* masked_ids, int K) {
  HCT_REQUIRE(P % 4 == 0 && S % P == 0 && mask_sum > 0.f, "hct_masked_mse: bad geometry S=%d P=%d mask_sum=%f", S, P, mask_sum);
  HCT_REQUIRE(x_dtype == HCT_F32 || x_dtype == HCT_F16, "hct_masked_mse: volumes are fp32 or fp16");
  const int g = S / P, L = g * g * g;
  const int pd = P * P * P * C;
  HCT_REQUIRE(pd % 4 == 0, "hct_masked_mse: patch dim %% 4 != 0");
  const float inv = 1.0f / mask_sum;
  HCT_REQUIRE(!masked_ids || (K >= 0 && K < L), "hct_masked_mse: compact form needs 0 <= K < L");
  const int nrows = masked_ids ? B * (L - K) : B * (L + 1);  // prediction rows visited = blocks
  if (x_dtype == HCT_F16) {
    HCT_DISPATCH_DTYPE(pred_dtype, T,
                       hipLaunchKernelGGL((masked_mse_kernel<f16, T>), dim3(nrows), dim3(256), 0, s, (const T*)pred, (const f16*)x, mask,
                                          C, S, P, L, norm_pix, inv, loss ? row_loss : nullptr, (T*)dpred, dpred_scale, host_scale, masked_ids, K));
  } else
  HCT_DISPATCH_DTYPE(pred_dtype, T,
                     hipLaunchKernelGGL((masked_mse_kernel<float, T>), dim3(nrows), dim3(256), 0, s, (const T*)pred, (const float*)x, mask,
                                        C, S, P, L, norm_pix, inv, loss ? row_loss : nullptr, (T*)dpred, dpred_scale, host_scale, masked_ids, K));
  if (loss) hipLaunchKernelGGL(loss_fold_kernel, dim3(1), dim3(256), 0, s, row_loss, masked_ids ? B * (L - K) : B * L, inv, loss);
  HCT_CHECK_LAUNCH("hct_masked_mse");
  return 0;
}

int dec0_index(const int32_t* ids_restore, int B, int L, int K, int32_t* kept_rows, int32_t* cat_idx, hipStream_t s) {
  const int n = std::max(B * (L + 1), L);
  hipLaunchKernelGGL(dec0_index_kernel, dim3((n + 255) / 256), dim3(256), 0, s, ids_restore, B, L, K, kept_rows, cat_idx);
  HCT_CHECK_LAUNCH("dec0_index");
  return 0;
}
int dec0_table(const float* mask_token, const float* pos, int L, int D, float* out, hipStream_t s) {
  hipLaunchKernelGGL(dec0_table_kernel, dim3(L), dim3(256), 0, s, mask_token, pos, D, out);
  HCT_CHECK_LAUNCH("dec0_table");
  return 0;
}
int dec0_aggregate(const void* g, int dtype, const int32_t* kept_rows, const int32_t* ids_restore, int B, int L, int K, int W, void* out, hipStream_t s) {
  HCT_REQUIRE(W % 4 == 0 && (W * dtype_size(dtype)) % 16 == 0, "dec0_aggregate: row width must be a multiple of 16 bytes");
  const int Nc = B * (K + 1);
  if (int rc = hct_gather_rows(g, kept_rows, Nc, (int)(W * dtype_size(dtype)), out, s)) return rc;  // kept / class rows: copies
  const dim3 grid(L, (W + 255) / 256);
  HCT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(dec0_table_sum_kernel<T>, grid, dim3(256), 0, s, (const T*)g, ids_restore, B, L, K, W, (T*)out + (size_t)Nc * W));
  HCT_CHECK_LAUNCH("dec0_aggregate");
  return 0;
}
// masked-token / class-token gradients of the decoder input when the first block ran through the cat matrices: the masked rows of
// the residual gradient `dh` are reduced as in hct_decoder_assemble_bwd, the LayerNorm part comes from `dcat`
int dec0_token_grads(const float* dh, const int32_t* ids_shuffle, const float* dcat, int B, int L, int K, int D, float* dmask, float* dcls,
                     void* workspace, size_t workspace_bytes, hipStream_t s) {
  if (workspace_bytes < hct_assemble_bwd_workspace_bytes(D)) {
    set_error("dec0_token_grads: workspace too small");
    return HCT_E_WORKSPACE;
  }
  const int threads = D / 4 >= 256 ? 256 : (D / 4 > 64 ? 128 : 64);
  const int nblk = std::min(B, kAsmBlocks);
  float* partial = (float*)workspace;
  hipLaunchKernelGGL(decoder_assemble_bwd_reduce_kernel, dim3(nblk), dim3(threads), 0, s, dh, ids_shuffle, B, L, K, D, partial);
  hipLaunchKernelGGL(dec0_token_grads_kernel, dim3((D + 31) / 32), dim3(512), 0, s, partial, nblk, dcat, B, L, K, D, dmask, dcls);
  HCT_CHECK_LAUNCH("dec0_token_grads");
  return 0;
}
}  // namespace hct

extern "C" {

const char* hct_last_error_string(void) { return g_err; }
int hct_version(void) { return 100; }
int hct_has_mfma_kernels(void) { return 1; }

int hct_mask_rank(const float* noise, int B, int L, int K, int32_t* ids_restore, int32_t* ids_shuffle, float* mask,
                  void* stream) {
  HCT_REQUIRE(B > 0 && L > 0 && K >= 0 && K <= L && L <= 12288, "hct_mask_rank: bad shape B=%d L=%d K=%d", B, L, K);
  hipLaunchKernelGGL(mask_rank_kernel, dim3(B), dim3(256), L * sizeof(float), (hipStream_t)stream, noise, L, K,
                     ids_restore, ids_shuffle, mask);
  HCT_CHECK_LAUNCH("hct_mask_rank");
  return 0;
}

int hct_patch_gather(const void* x, int x_dtype, const int32_t* ids_shuffle, int B, int C, int S, int P, int L, int K, void* rows,
                     int rows_dtype, void* stream) {
  HCT_REQUIRE(P % 4 == 0 && S % P == 0 && (S / P) * (S / P) * (S / P) == L, "hct_patch_gather: bad geometry S=%d P=%d L=%d", S, P, L);
  HCT_REQUIRE(x_dtype == HCT_F32 || x_dtype == HCT_F16, "hct_patch_gather: volumes are fp32 or fp16");
  if (B * K == 0) return 0;
  const size_t pencil_lds = (size_t)P * P * S * dtype_size(rows_dtype);
  if (!ids_shuffle && K == L && S % 4 == 0 && pencil_lds <= 64 * 1024) {  // every patch, grid order: pencils of S / P patches
    const int g = S / P;
    if (x_dtype == HCT_F16) {
      HCT_DISPATCH_DTYPE(rows_dtype, T,
                         hipLaunchKernelGGL((patch_gather_pencil_kernel<f16, T>), dim3(B * C * g * g), dim3(256), pencil_lds, (hipStream_t)stream,
                                            (const f16*)x, C, S, P, (T*)rows));
    } else {
      HCT_DISPATCH_DTYPE(rows_dtype, T,
                         hipLaunchKernelGGL((patch_gather_pencil_kernel<float, T>), dim3(B * C * g * g), dim3(256), pencil_lds, (hipStream_t)stream,
                                            (const float*)x, C, S, P, (T*)rows));
    }
    HCT_CHECK_LAUNCH("hct_patch_gather(pencil)");
    return 0;
  }
  if (x_dtype == HCT_F16) {
    HCT_DISPATCH_DTYPE(rows_dtype, T,
                       hipLaunchKernelGGL((patch_gather_kernel<f16, T>), dim3(B * K), dim3(256), 0, (hipStream_t)stream, (const f16*)x,
                                          ids_shuffle, C, S, P, L, K, (T*)rows));
  } else
  HCT_DISPATCH_DTYPE(rows_dtype, T,
                     hipLaunchKernelGGL((patch_gather_kernel<float, T>), dim3(B * K), dim3(256), 0, (hipStream_t)stream, (const float*)x,
                                        ids_shuffle, C, S, P, L, K, (T*)rows));
  HCT_CHECK_LAUNCH("hct_patch_gather");
  return 0;
}

int hct_encoder_assemble_fwd(const void* tok, int tok_dtype, const float* cls, const float* pos,
                             const int32_t* ids_shuffle, int B, int L, int K, int D, float* h0, void* stream) {
  HCT_REQUIRE(D % 4 == 0, "hct_encoder_assemble_fwd: D %% 4 != 0 (%d)", D);
  const int threads = D / 4 >= 256 ? 256 : (D / 4 > 64 ? 128 : 64);
  HCT_DISPATCH_DTYPE(tok_dtype, T,
                     hipLaunchKernelGGL(encoder_assemble_fwd_kernel<T>, dim3(B * (K + 1)), dim3(threads), 0,
                                        (hipStream_t)stream, (const T*)tok, cls, pos, ids_shuffle, L, K, D, h0));
  HCT_CHECK_LAUNCH("hct_encoder_assemble_fwd");
  return 0;
}

size_t hct_assemble_bwd_workspace_bytes(int D) { return (size_t)kAsmBlocks * 2 * D * sizeof(float); }

int hct_encoder_assemble_bwd(const float* dh0, const int32_t* ids_restore, int B, int L, int K, int D, void* dtok,
                             int dtok_dtype, float* dcls, float* dpos, void* workspace, size_t workspace_bytes,
                             void* stream) {
  (void)workspace; (void)workspace_bytes;
  HCT_REQUIRE(D % 4 == 0, "hct_encoder_assemble_bwd: D %% 4 != 0 (%d)", D);
  const int threads = D / 4 >= 256 ? 256 : (D / 4 > 64 ? 128 : 64);
  hipStream_t s = (hipStream_t)stream;
  if (dtok && B * K > 0)
    HCT_DISPATCH_DTYPE(dtok_dtype, T,
                       hipLaunchKernelGGL(encoder_assemble_bwd_tok_kernel<T>, dim3(B * K), dim3(threads), 0, s, dh0, K, D,
                                          (T*)dtok));
  if (dcls) hipLaunchKernelGGL(strided_rowsum_kernel, dim3((D + 255) / 256), dim3(256), 0, s, dh0, B, (size_t)(K + 1) * D, D, dcls);
  if (dpos) hipLaunchKernelGGL(encoder_assemble_bwd_pos_kernel, dim3(L), dim3(threads), 0, s, dh0, ids_restore, B, L, K, D, dpos);
  HCT_CHECK_LAUNCH("hct_encoder_assemble_bwd");
  return 0;
}

// backward of hct_vit_assemble_fwd: dtok[b*L+l] = dh0[b, 1+R+l] (cast); dcls = sum_b dh0[b,0]; dreg[r] = sum_b dh0[b,1+r];
// dpos[l] = sum_b dh0[b,1+R+l]  (sums in batch order)
int hct_vit_assemble_bwd(const float* dh0, int B, int L, int R, int D, void* dtok, int dtok_dtype, float* dcls, float* dreg, float* dpos,
                         void* stream) {
  HCT_REQUIRE(dh0 && B > 0 && L > 0 && R >= 0 && D > 0 && D % 4 == 0, "hct_vit_assemble_bwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const size_t stride = (size_t)(1 + R + L) * D;
  if (dtok) {
    HCT_DISPATCH_DTYPE(dtok_dtype, T, hipLaunchKernelGGL(hct::vit_assemble_bwd_tok_kernel<T>, dim3(B * L), dim3(D / 4 >= 256 ? 256 : 64), 0, s, dh0, L,
                                                        R, D, (T*)dtok));
  }
  if (dcls) hipLaunchKernelGGL(strided_rowsum_kernel, dim3((D + 255) / 256), dim3(256), 0, s, dh0, B, stride, D, dcls);
  // registers and position rows: the rows 1 .. R+L of every volume, summed over the batch -> one launch over (R + L) * D columns
  if (dreg && R > 0) hipLaunchKernelGGL(strided_rowsum_kernel, dim3((R * D + 255) / 256), dim3(256), 0, s, dh0 + D, B, stride, R * D, dreg);
  if (dpos) hipLaunchKernelGGL(strided_rowsum_kernel, dim3((L * D + 255) / 256), dim3(256), 0, s, dh0 + (size_t)(1 + R) * D, B, stride, L * D, dpos);
  HCT_CHECK_LAUNCH("hct_vit_assemble_bwd");
  return 0;
}

int hct_layernorm_fwd(const float* x, const float* gamma, const float* beta, int rows, int D, float eps, void* y,
                      int y_dtype, float* mean, float* rstd, void* stream) {
  HCT_REQUIRE(D % 4 == 0 && rows >= 0, "hct_layernorm_fwd: bad shape rows=%d D=%d", rows, D);
  if (rows == 0) return 0;
  const int nblk = min((rows + 3) / 4, 2048);  // 8 workgroups (32 waves) per CU, each wave strides over rows
#define HCT_LN_FWD(NV_)                                                                                                 \
  HCT_DISPATCH_DTYPE(y_dtype, T,                                                                                       \
                     hipLaunchKernelGGL((layernorm_fwd_reg_kernel<T, NV_>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, x, \
                                        gamma, beta, rows, D, eps, (T*)y, mean, rstd))
  if (D <= 256) HCT_LN_FWD(1);
  else if (D <= 512) HCT_LN_FWD(2);
  else if (D <= 768) HCT_LN_FWD(3);
  else if (D <= 1024) HCT_LN_FWD(4);
  else
    HCT_DISPATCH_DTYPE(y_dtype, T,
                       hipLaunchKernelGGL(layernorm_fwd_kernel<T>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                                          x, gamma, beta, rows, D, eps, (T*)y, mean, rstd));
#undef HCT_LN_FWD
  HCT_CHECK_LAUNCH("hct_layernorm_fwd");
  return 0;
}

size_t hct_layernorm_bwd_workspace_bytes(int rows, int D) {
  (void)rows;
  return (size_t)kLnBwdBlocks * 3 * D * sizeof(float);
}

int hct_layernorm_bwd(const void* dy, int dy_dtype, const float* x, const float* mean, const float* rstd,
                      const float* gamma, const float* dres, int rows, int D, float* dx, void* dx_shadow,
                      int shadow_dtype, float* dgamma, float* dbeta, float* dcolsum, void* workspace,
                      size_t workspace_bytes, void* stream) {
  return hct_layernorm_bwd_mapped(dy, dy_dtype, x, mean, rstd, gamma, dres, nullptr, rows, D, dx, dx_shadow, shadow_dtype, dgamma, dbeta,
                                  dcolsum, workspace, workspace_bytes, stream);
}

int hct_layernorm_bwd_mapped(const void* dy, int dy_dtype, const float* x, const float* mean, const float* rstd,
                             const float* gamma, const float* dres, const int32_t* dres_rows, int rows, int D, float* dx,
                             void* dx_shadow, int shadow_dtype, float* dgamma, float* dbeta, float* dcolsum, void* workspace,
                             size_t workspace_bytes, void* stream) {
  HCT_REQUIRE(D % 4 == 0 && rows > 0, "hct_layernorm_bwd: bad shape rows=%d D=%d", rows, D);
  HCT_REQUIRE(!dres_rows || (dres && (const void*)dres != (const void*)dx), "hct_layernorm_bwd_mapped: a mapped residual gradient cannot alias dx");
  if (workspace_bytes < hct_layernorm_bwd_workspace_bytes(rows, D)) {
    set_error("hct_layernorm_bwd: workspace too small");
    return HCT_E_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const int nblk = min(kLnBwdBlocks, (rows + 3) / 4);
  float* partial = (float*)workspace;
  int rc = 0;
  if (dy_dtype == HCT_BF16) {
    if (dx_shadow && shadow_dtype == HCT_F32) rc = launch_ln_bwd<bf16, float>(dy, x, mean, rstd, gamma, dres, dres_rows, rows, D, dx, dx_shadow, partial, dcolsum != nullptr, nblk, s);
    else rc = launch_ln_bwd<bf16, bf16>(dy, x, mean, rstd, gamma, dres, dres_rows, rows, D, dx, dx_shadow, partial, dcolsum != nullptr, nblk, s);
  } else {
    if (dx_shadow && shadow_dtype == HCT_BF16) rc = launch_ln_bwd<float, bf16>(dy, x, mean, rstd, gamma, dres, dres_rows, rows, D, dx, dx_shadow, partial, dcolsum != nullptr, nblk, s);
    else rc = launch_ln_bwd<float, float>(dy, x, mean, rstd, gamma, dres, dres_rows, rows, D, dx, dx_shadow, partial, dcolsum != nullptr, nblk, s);
  }
  if (rc) return rc;
  HCT_CHECK_LAUNCH("hct_layernorm_bwd");
  return fold_partials(FoldJob{partial, nblk, 3, D, dcolsum ? 3 : 2, {dgamma, dbeta, dcolsum}}, s);
}

int hct_decoder_assemble_fwd(const void* e, int e_dtype, const float* mask_token, const float* dec_cls,
                             const float* dec_pos, const int32_t* ids_restore, int B, int L, int K, int D, float* y,
                             void* stream) {
  HCT_REQUIRE(D % 4 == 0, "hct_decoder_assemble_fwd: D %% 4 != 0 (%d)", D);
  const int threads = D / 4 >= 256 ? 256 : (D / 4 > 64 ? 128 : 64);
  HCT_DISPATCH_DTYPE(e_dtype, T,
                     hipLaunchKernelGGL(decoder_assemble_fwd_kernel<T>, dim3(B * (L + 1)), dim3(threads), 0,
                                        (hipStream_t)stream, (const T*)e, mask_token, dec_cls, dec_pos, ids_restore, L, K,
                                        D, y));
  HCT_CHECK_LAUNCH("hct_decoder_assemble_fwd");
  return 0;
}

int hct_decoder_assemble_bwd(const float* dy, const int32_t* ids_restore, const int32_t* ids_shuffle, int B, int L,
                             int K, int D, void* de, int de_dtype, float* dmask_token, float* ddec_cls,
                             void* workspace, size_t workspace_bytes, void* stream) {
  (void)ids_restore;
  HCT_REQUIRE(D % 4 == 0, "hct_decoder_assemble_bwd: D %% 4 != 0 (%d)", D);
  if (workspace_bytes < hct_assemble_bwd_workspace_bytes(D)) {
    set_error("hct_decoder_assemble_bwd: workspace too small");
    return HCT_E_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const int threads = D / 4 >= 256 ? 256 : (D / 4 > 64 ? 128 : 64);
  HCT_DISPATCH_DTYPE(de_dtype, T,
                     hipLaunchKernelGGL(decoder_assemble_bwd_kernel<T>, dim3(B * (K + 1)), dim3(threads), 0, s, dy,
                                        ids_shuffle, L, K, D, (T*)de));
  const int nblk = min(B, kAsmBlocks);
  float* partial = (float*)workspace;
  hipLaunchKernelGGL(decoder_assemble_bwd_reduce_kernel, dim3(nblk), dim3(threads), 0, s, dy, ids_shuffle, B, L, K, D, partial);
  hipLaunchKernelGGL(fold_partials_kernel, dim3((D + 31) / 32, 2), dim3(kFoldThreads), 0, s, partial, nblk, 2, D, dmask_token,
                     ddec_cls, (float*)nullptr);
  HCT_CHECK_LAUNCH("hct_decoder_assemble_bwd");
  return 0;
}

int hct_masked_mse(const void* pred, int pred_dtype, const void* x, int x_dtype, const float* mask, int B, int C, int S, int P,
                   int norm_pix, float mask_sum, float* row_loss, float* loss, void* dpred, const float* dpred_scale,
                   void* stream) {
  return masked_mse_launch(pred, pred_dtype, x, x_dtype, mask, B, C, S, P, norm_pix, mask_sum, row_loss, loss, dpred, dpred_scale, 1.0f,
                           (hipStream_t)stream, nullptr, 0);
}

// Rows the loss sees (mae.py:298-299 keeps only the removed patches): for every volume the decoder rows of its L - K masked
// patches, in shuffle order.  tail_rows[b * (L-K) + j] = row of the [B, L+1] decoder layout; tail_inv = the inverse, -1 for
// the class token and the kept patches.
int hct_tail_rows(const int32_t* ids_restore, int B, int L, int K, int32_t* tail_rows, int32_t* tail_inv, void* stream) {
  HCT_REQUIRE(ids_restore && tail_rows && tail_inv && B > 0 && L > 0 && K >= 0 && K < L, "hct_tail_rows: bad arguments");
  const int n = B * (L + 1);
  hipLaunchKernelGGL(hct::tail_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, ids_restore, B, L, K, tail_rows, tail_inv);
  HCT_CHECK_LAUNCH("hct_tail_rows");
  return 0;
}

// dst row r = src row idx[r] (any order, repeats allowed), zeros where idx[r] < 0.  Rows are row_bytes long (a multiple of 16,
// both bases 16-byte aligned): the gather of the decoder's masked rows and, with the inverse map, the scatter back.
int hct_gather_rows(const void* src, const int32_t* idx, int n_rows, int row_bytes, void* dst, void* stream) {
  HCT_REQUIRE(src && idx && dst && n_rows >= 0 && row_bytes > 0 && row_bytes % 16 == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0,
              "hct_gather_rows: bad arguments (rows of a multiple of 16 bytes, 16-byte aligned)");
  if (n_rows == 0) return 0;
  const int64_t total = (int64_t)n_rows * (row_bytes / 16);
  const int blocks = (int)std::min<int64_t>(8192, (total + 255) / 256);
  hipLaunchKernelGGL(hct::gather_rows16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, idx, n_rows, row_bytes / 16, (uint4*)dst);
  HCT_CHECK_LAUNCH("hct_gather_rows");
  return 0;
}

int hct_unpatchify(const void* pred, int pred_dtype, int has_cls_row, int B, int C, int S, int P, float* vol,
                   void* stream) {
  HCT_REQUIRE(S % P == 0, "hct_unpatchify: bad geometry");
  const int g = S / P, L = g * g * g;
  HCT_DISPATCH_DTYPE(pred_dtype, T,
                     hipLaunchKernelGGL(unpatchify_kernel<T>, dim3(B * L), dim3(256), 0, (hipStream_t)stream,
                                        (const T*)pred, has_cls_row ? 1 : 0, C, S, P, vol));
  HCT_CHECK_LAUNCH("hct_unpatchify");
  return 0;
}

static int colsum_chunks(int rows, int cols) {
  const int colblk = (cols + 255) / 256;
  int chunks = (1024 + colblk - 1) / colblk;
  if (chunks > (rows + 15) / 16) chunks = (rows + 15) / 16;
  return chunks < 1 ? 1 : chunks;
}
size_t hct_colsum_workspace_bytes(int rows, int cols) { return (size_t)colsum_chunks(rows, cols) * cols * sizeof(float); }

int hct_colsum(const void* x, int dtype, int rows, int cols, int64_t ld, float* out, void* workspace,
               size_t workspace_bytes, void* stream) {
  HCT_REQUIRE(cols % 4 == 0 && ld % 4 == 0 && rows > 0, "hct_colsum: cols/ld must be multiples of 4");
  if (workspace_bytes < hct_colsum_workspace_bytes(rows, cols)) {
    set_error("hct_colsum: workspace too small");
    return HCT_E_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const int chunks = colsum_chunks(rows, cols);
  const int rpc = (rows + chunks - 1) / chunks;
  float* partial = (float*)workspace;
  HCT_DISPATCH_DTYPE(dtype, T,
                     hipLaunchKernelGGL(colsum_partial_kernel<T>, dim3((cols + 255) / 256, chunks), dim3(256), 0, s,
                                        (const T*)x, rows, cols, ld, rpc, partial));
  hipLaunchKernelGGL(fold_partials_kernel, dim3((cols + 31) / 32, 1), dim3(kFoldThreads), 0, s, partial, chunks, 1, cols, out,
                     (float*)nullptr, (float*)nullptr);
  HCT_CHECK_LAUNCH("hct_colsum");
  return 0;
}

int hct_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream) {
  if (n <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int blocks = (int)std::min<int64_t>(2048, (n + 1023) / 1024);
  if (src_dtype == HCT_F32 && dst_dtype == HCT_BF16)
    hipLaunchKernelGGL((cast_kernel<float, bf16>), dim3(blocks), dim3(256), 0, s, (const float*)src, (bf16*)dst, n);
  else if (src_dtype == HCT_BF16 && dst_dtype == HCT_F32)
    hipLaunchKernelGGL((cast_kernel<bf16, float>), dim3(blocks), dim3(256), 0, s, (const bf16*)src, (float*)dst, n);
  else if (src_dtype == HCT_F32)
    hipLaunchKernelGGL((cast_kernel<float, float>), dim3(blocks), dim3(256), 0, s, (const float*)src, (float*)dst, n);
  else
    hipLaunchKernelGGL((cast_kernel<bf16, bf16>), dim3(blocks), dim3(256), 0, s, (const bf16*)src, (bf16*)dst, n);
  HCT_CHECK_LAUNCH("hct_cast");
  return 0;
}

int hct_transpose_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int rows, int cols, void* stream) {
  if (rows <= 0 || cols <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((cols + 63) / 64, (rows + 63) / 64);
  if (src_dtype == HCT_F32 && dst_dtype == HCT_BF16)
    hipLaunchKernelGGL((transpose_cast_kernel<float, bf16>), grid, dim3(256), 0, s, (const float*)src, (bf16*)dst, rows, cols);
  else if (src_dtype == HCT_F32 && dst_dtype == HCT_F32)
    hipLaunchKernelGGL((transpose_cast_kernel<float, float>), grid, dim3(256), 0, s, (const float*)src, (float*)dst, rows, cols);
  else if (src_dtype == HCT_BF16 && dst_dtype == HCT_BF16)
    hipLaunchKernelGGL((transpose_cast_kernel<bf16, bf16>), grid, dim3(256), 0, s, (const bf16*)src, (bf16*)dst, rows, cols);
  else
    hipLaunchKernelGGL((transpose_cast_kernel<bf16, float>), grid, dim3(256), 0, s, (const bf16*)src, (float*)dst, rows, cols);
  HCT_CHECK_LAUNCH("hct_transpose_cast");
  return 0;
}


int hct_pos_embed_interp3d(const float* src, int g_src, float* dst, int g_dst, int D, int extra, void* stream) {
  HCT_REQUIRE(src && dst && g_src > 0 && g_dst > 0 && D > 0 && D % 4 == 0 && extra >= 0, "hct_pos_embed_interp3d: bad arguments (D must be a multiple of 4)");
  const int64_t n = ((int64_t)extra + (int64_t)g_dst * g_dst * g_dst) * (D / 4);
  hipLaunchKernelGGL(hct::pos_embed_interp3d_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, g_src, dst,
                     g_dst, D, extra);
  HCT_CHECK_LAUNCH("hct_pos_embed_interp3d");
  return 0;
}

int hct_hu_window(const void* hu, int in_dtype, void* out, int out_dtype, int B, int64_t voxels, int n_windows, const float* a_min,
                  const float* a_max, void* stream) {
  HCT_REQUIRE(hu && out && a_min && a_max && B > 0 && voxels > 0 && voxels % 4 == 0 && n_windows > 0 && n_windows <= 8,
              "hct_hu_window: bad arguments (voxels per volume must be a multiple of 4, 1..8 windows)");
  HCT_REQUIRE((in_dtype == HCT_F32 || in_dtype == HCT_F16) && (out_dtype == HCT_F32 || out_dtype == HCT_F16), "hct_hu_window: dtypes are fp32 or fp16");
  const int64_t vox4 = voxels / 4;
  const dim3 grid((unsigned)std::min<int64_t>(1024, (vox4 + 255) / 256), (unsigned)B), block(256);
  hipStream_t s = (hipStream_t)stream;
#define HCT_HUW(TI, TO) hipLaunchKernelGGL((hct::hu_window_kernel<TI, TO>), grid, block, 0, s, (const TI*)hu, (TO*)out, n_windows, vox4, a_min, a_max)
  if (in_dtype == HCT_F32 && out_dtype == HCT_F32) HCT_HUW(float, float);
  else if (in_dtype == HCT_F32) HCT_HUW(float, hct::f16);
  else if (out_dtype == HCT_F32) HCT_HUW(hct::f16, float);
  else HCT_HUW(hct::f16, hct::f16);
#undef HCT_HUW
  HCT_CHECK_LAUNCH("hct_hu_window");
  return 0;
}

int hct_augment_volume(const void* in, int in_dtype, float* out, int B, int C, int S, const unsigned char* flip,
                       const float* shift, void* stream) {
  HCT_REQUIRE(in && out && B > 0 && C > 0 && S > 0 && S % 4 == 0 && (const void*)out != in, "hct_augment_volume: bad arguments (S must be a multiple of 4)");
  HCT_REQUIRE(in_dtype == HCT_F32 || in_dtype == HCT_BF16 || in_dtype == HCT_F16, "hct_augment_volume: unsupported input dtype %d", in_dtype);
  const int64_t n4 = (int64_t)B * C * S * S * (S / 4);
  const dim3 grid((unsigned)((n4 + 255) / 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (in_dtype == HCT_F32) hipLaunchKernelGGL(hct::augment_volume_kernel<float>, grid, block, 0, s, (const float*)in, out, C, S, flip, shift, n4);
  else if (in_dtype == HCT_BF16) hipLaunchKernelGGL(hct::augment_volume_kernel<hct::bf16>, grid, block, 0, s, (const hct::bf16*)in, out, C, S, flip, shift, n4);
  else hipLaunchKernelGGL(hct::augment_volume_kernel<_Float16>, grid, block, 0, s, (const _Float16*)in, out, C, S, flip, shift, n4);
  HCT_CHECK_LAUNCH("hct_augment_volume");
  return 0;
}

int hct_gaussian_smooth3d(const float* in, float* out, float* tmp, int B, int C, int S, const float* taps, const unsigned char* apply,
                          void* stream) {
  HCT_REQUIRE(in && out && tmp && taps && apply && B > 0 && C > 0 && S > 0 && S % 4 == 0, "hct_gaussian_smooth3d: bad arguments (S must be a multiple of 4)");
  HCT_REQUIRE(in != out && in != tmp && out != tmp, "hct_gaussian_smooth3d: in, out and tmp must be three different buffers");
  const int64_t n4 = (int64_t)B * C * S * S * (S / 4);
  const dim3 grid((unsigned)((n4 + 255) / 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(hct::gaussian_pass_kernel, grid, block, 0, s, in, out, C, S, 0, taps, apply, 1, n4);                 // first spatial axis
  hipLaunchKernelGGL(hct::gaussian_pass_kernel, grid, block, 0, s, (const float*)out, tmp, C, S, 1, taps, apply, 0, n4);  // second
  hipLaunchKernelGGL(hct::gaussian_pass_kernel, grid, block, 0, s, (const float*)tmp, out, C, S, 2, taps, apply, 0, n4);  // third (contiguous)
  HCT_CHECK_LAUNCH("hct_gaussian_smooth3d");
  return 0;
}

int hct_vit_assemble_fwd(const void* tok, int tok_dtype, const float* cls, const float* reg, const float* pos, int B,
                         int L, int R, int D, float* h, void* stream) {
  HCT_REQUIRE(tok && cls && h && B > 0 && L > 0 && R >= 0 && D > 0 && D % 4 == 0 && (R == 0 || reg), "hct_vit_assemble_fwd: bad arguments");
  const int64_t n4 = (int64_t)B * (1 + R + L) * (D / 4);
  HCT_DISPATCH_DTYPE(tok_dtype, T,
                     hipLaunchKernelGGL(hct::vit_assemble_fwd_kernel<T>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0,
                                        (hipStream_t)stream, (const T*)tok, cls, reg, pos, L, R, D, h, n4));
  HCT_CHECK_LAUNCH("hct_vit_assemble_fwd");
  return 0;
}
}  // extern "C"
