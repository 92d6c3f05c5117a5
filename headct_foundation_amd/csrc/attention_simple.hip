// Reference-precision attention (fp32 math, any storage dtype): one query (or key) row per lane, the
// other operand streamed through LDS tiles.  Used by the fp32 parity mode and for head sizes the MFMA
// kernel does not cover.  attentionblock.py:54-62 (fused qkv view [B,N,3,H,dh], SDPA scale dh^-1/2, no mask).
#include "common.h"

namespace hct {

constexpr int kTile = 32;

template <typename T, int DH>
__global__ void __launch_bounds__(64) attn_fwd_simple_kernel(const T* __restrict__ qkv, int N, int H, T* __restrict__ o,
                                                             float* __restrict__ lse) {
  __shared__ float sK[kTile][DH];
  __shared__ float sV[kTile][DH];
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
  const int qi = blockIdx.y * 64 + threadIdx.x;
  const int64_t rs = (int64_t)3 * H * DH;  // token row stride in qkv
  const T* base = qkv + (int64_t)b * N * rs + h * DH;
  const float scale = rsqrtf((float)DH);
  float q[DH], acc[DH];
  float m = -INFINITY, l = 0.f;
  const bool live = qi < N;
#pragma unroll
  for (int d = 0; d < DH; ++d) {
    q[d] = live ? to_f32(base[(int64_t)qi * rs + d]) * scale : 0.f;
    acc[d] = 0.f;
  }
  for (int k0 = 0; k0 < N; k0 += kTile) {
    __syncthreads();
    for (int i = threadIdx.x; i < kTile * DH; i += 64) {
      const int r = i / DH, d = i - r * DH;
      const int kj = k0 + r;
      sK[r][d] = kj < N ? to_f32(base[(int64_t)kj * rs + H * DH + d]) : 0.f;
      sV[r][d] = kj < N ? to_f32(base[(int64_t)kj * rs + 2 * H * DH + d]) : 0.f;
    }
    __syncthreads();
    const int kn = min(kTile, N - k0);
    for (int r = 0; r < kn; ++r) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < DH; ++d) s = fmaf(q[d], sK[r][d], s);
      const float mn = fmaxf(m, s);
      const float corr = __expf(m - mn);
      const float p = __expf(s - mn);
      l = l * corr + p;
#pragma unroll
      for (int d = 0; d < DH; ++d) acc[d] = fmaf(acc[d], corr, p * sV[r][d]);
      m = mn;
    }
  }
  if (live) {
    const float inv = 1.0f / l;
    T* orow = o + ((int64_t)b * N + qi) * (H * DH) + h * DH;
#pragma unroll
    for (int d = 0; d < DH; ++d) orow[d] = from_f32<T>(acc[d] * inv);
    lse[(int64_t)bh * N + qi] = m + __logf(l);
  }
}

// dQ: lane per query row
template <typename T, int DH>
__global__ void __launch_bounds__(64) attn_bwd_dq_simple_kernel(const T* __restrict__ qkv, const T* __restrict__ o,
                                                                const T* __restrict__ d_o, const float* __restrict__ lse,
                                                                int N, int H, T* __restrict__ dqkv) {
  __shared__ float sK[kTile][DH];
  __shared__ float sV[kTile][DH];
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
  const int qi = blockIdx.y * 64 + threadIdx.x;
  const int64_t rs = (int64_t)3 * H * DH;
  const T* base = qkv + (int64_t)b * N * rs + h * DH;
  const float scale = rsqrtf((float)DH);
  const bool live = qi < N;
  float q[DH], dov[DH], dq[DH];
  float delta = 0.f;
  const int64_t orow = ((int64_t)b * N + qi) * (H * DH) + h * DH;
#pragma unroll
  for (int d = 0; d < DH; ++d) {
    q[d] = live ? to_f32(base[(int64_t)qi * rs + d]) * scale : 0.f;
    dov[d] = live ? to_f32(d_o[orow + d]) : 0.f;
    delta += live ? dov[d] * to_f32(o[orow + d]) : 0.f;
    dq[d] = 0.f;
  }
  const float L = live ? lse[(int64_t)bh * N + qi] : 0.f;
  for (int k0 = 0; k0 < N; k0 += kTile) {
    __syncthreads();
    for (int i = threadIdx.x; i < kTile * DH; i += 64) {
      const int r = i / DH, d = i - r * DH;
      const int kj = k0 + r;
      sK[r][d] = kj < N ? to_f32(base[(int64_t)kj * rs + H * DH + d]) : 0.f;
      sV[r][d] = kj < N ? to_f32(base[(int64_t)kj * rs + 2 * H * DH + d]) : 0.f;
    }
    __syncthreads();
    const int kn = min(kTile, N - k0);
    for (int r = 0; r < kn; ++r) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < DH; ++d) {
        s = fmaf(q[d], sK[r][d], s);
        dp = fmaf(dov[d], sV[r][d], dp);
      }
      const float p = __expf(s - L);
      const float ds = p * (dp - delta) * scale;
#pragma unroll
      for (int d = 0; d < DH; ++d) dq[d] = fmaf(ds, sK[r][d], dq[d]);
    }
  }
  if (live) {
    T* out = dqkv + ((int64_t)b * N + qi) * rs + h * DH;
#pragma unroll
    for (int d = 0; d < DH; ++d) out[d] = from_f32<T>(dq[d]);
  }
}

// dK, dV: lane per key row; queries streamed through LDS (with their lse and delta)
template <typename T, int DH>
__global__ void __launch_bounds__(64) attn_bwd_dkv_simple_kernel(const T* __restrict__ qkv, const T* __restrict__ o,
                                                                 const T* __restrict__ d_o, const float* __restrict__ lse,
                                                                 int N, int H, T* __restrict__ dqkv) {
  __shared__ float sQ[kTile][DH];
  __shared__ float sdO[kTile][DH];
  __shared__ float sL[kTile], sD[kTile];
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
  const int kj = blockIdx.y * 64 + threadIdx.x;
  const int64_t rs = (int64_t)3 * H * DH;
  const T* base = qkv + (int64_t)b * N * rs + h * DH;
  const float scale = rsqrtf((float)DH);
  const bool live = kj < N;
  float kk[DH], vv[DH], dk[DH], dv[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) {
    kk[d] = live ? to_f32(base[(int64_t)kj * rs + H * DH + d]) : 0.f;
    vv[d] = live ? to_f32(base[(int64_t)kj * rs + 2 * H * DH + d]) : 0.f;
    dk[d] = dv[d] = 0.f;
  }
  for (int q0 = 0; q0 < N; q0 += kTile) {
    __syncthreads();
    for (int i = threadIdx.x; i < kTile * DH; i += 64) {
      const int r = i / DH, d = i - r * DH;
      const int qi = q0 + r;
      sQ[r][d] = qi < N ? to_f32(base[(int64_t)qi * rs + d]) * scale : 0.f;
      sdO[r][d] = qi < N ? to_f32(d_o[((int64_t)b * N + qi) * (H * DH) + h * DH + d]) : 0.f;
    }
    if (threadIdx.x < kTile) {
      const int qi = q0 + threadIdx.x;
      float dl = 0.f;
      if (qi < N) {
        const int64_t orow = ((int64_t)b * N + qi) * (H * DH) + h * DH;
        for (int d = 0; d < DH; ++d) dl += to_f32(d_o[orow + d]) * to_f32(o[orow + d]);
        sL[threadIdx.x] = lse[(int64_t)bh * N + qi];
      } else {
        sL[threadIdx.x] = 0.f;
      }
      sD[threadIdx.x] = dl;
    }
    __syncthreads();
    const int qn = min(kTile, N - q0);
    for (int r = 0; r < qn; ++r) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < DH; ++d) {
        s = fmaf(sQ[r][d], kk[d], s);
        dp = fmaf(sdO[r][d], vv[d], dp);
      }
      const float p = __expf(s - sL[r]);
      const float ds = p * (dp - sD[r]);  // sQ already carries the 1/sqrt(dh) factor
#pragma unroll
      for (int d = 0; d < DH; ++d) {
        dv[d] = fmaf(p, sdO[r][d], dv[d]);
        dk[d] = fmaf(ds, sQ[r][d], dk[d]);
      }
    }
  }
  if (live) {
    T* outk = dqkv + ((int64_t)b * N + kj) * rs + H * DH + h * DH;
    T* outv = outk + H * DH;
#pragma unroll
    for (int d = 0; d < DH; ++d) {
      outk[d] = from_f32<T>(dk[d]);
      outv[d] = from_f32<T>(dv[d]);
    }
  }
}

template <typename T, int DH>
static int launch_simple_fwd(const void* qkv, int B, int N, int H, void* o, float* lse, hipStream_t s) {
  hipLaunchKernelGGL((attn_fwd_simple_kernel<T, DH>), dim3(B * H, (N + 63) / 64), dim3(64), 0, s, (const T*)qkv, N, H,
                     (T*)o, lse);
  return 0;
}
template <typename T, int DH>
static int launch_simple_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, int B, int N, int H,
                             void* dqkv, hipStream_t s) {
  dim3 grid(B * H, (N + 63) / 64);
  hipLaunchKernelGGL((attn_bwd_dq_simple_kernel<T, DH>), grid, dim3(64), 0, s, (const T*)qkv, (const T*)o, (const T*)d_o,
                     lse, N, H, (T*)dqkv);
  hipLaunchKernelGGL((attn_bwd_dkv_simple_kernel<T, DH>), grid, dim3(64), 0, s, (const T*)qkv, (const T*)o, (const T*)d_o,
                     lse, N, H, (T*)dqkv);
  return 0;
}

#define HCT_DH_SWITCH(dh, CALL)                 \
  switch (dh) {                                 \
    case 16: { constexpr int DHc = 16; CALL; } break;   \
    case 32: { constexpr int DHc = 32; CALL; } break;   \
    case 48: { constexpr int DHc = 48; CALL; } break;   \
    case 64: { constexpr int DHc = 64; CALL; } break;   \
    case 96: { constexpr int DHc = 96; CALL; } break;   \
    case 128: { constexpr int DHc = 128; CALL; } break; \
    default: set_error("attention: head dim %d unsupported (16/32/48/64/96/128)", dh); return HCT_E_UNSUPPORTED; \
  }

int attention_fwd_simple(const void* qkv, int B, int N, int H, int dh, int dtype, void* o, float* lse, hipStream_t s) {
  if (dtype == HCT_BF16) { HCT_DH_SWITCH(dh, (launch_simple_fwd<bf16, DHc>(qkv, B, N, H, o, lse, s))); }
  else { HCT_DH_SWITCH(dh, (launch_simple_fwd<float, DHc>(qkv, B, N, H, o, lse, s))); }
  return check_hip(hipGetLastError(), "attention_fwd_simple");
}
int attention_bwd_simple(const void* qkv, const void* o, const void* d_o, const float* lse, int B, int N, int H, int dh,
                         int dtype, void* dqkv, hipStream_t s) {
  if (dtype == HCT_BF16) { HCT_DH_SWITCH(dh, (launch_simple_bwd<bf16, DHc>(qkv, o, d_o, lse, B, N, H, dqkv, s))); }
  else { HCT_DH_SWITCH(dh, (launch_simple_bwd<float, DHc>(qkv, o, d_o, lse, B, N, H, dqkv, s))); }
  return check_hip(hipGetLastError(), "attention_bwd_simple");
}

}  // namespace hct
