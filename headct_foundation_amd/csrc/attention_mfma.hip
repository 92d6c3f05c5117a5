// bf16 MFMA attention for the MAE token counts (55 / 217 / 129 / 513 tokens, head dim 48 / 64): the whole K and V
// (forward) or Q, K, V, dO (backward) of one (batch, head) live in LDS, so there is no K/V re-streaming.
//
// LDS image (shared by all four operands): row = token, 128 B = 64 bf16 per row (head dim zero-padded to 64),
// 16-B chunk c of row r stored at chunk  c ^ ((r>>1)&7)  -> the b128 row reads of the MFMA A/B operands are
// bank-conflict-free; column ("transposed") operands come from the same image with ds_read_b64_tr_b16.
//
// forward  (4 waves, wave = 16-query tile, online softmax over 32-key steps):
//     S^T = K.Q^T  (key on the accumulator row, query on the lane)  -> P^T feeds O^T = V^T.P^T directly as the
//     B operand (accumulator-as-operand, no LDS round trip); V^T fragments by transposed reads.
// backward (8 waves, no atomics, deterministic): every wave first owns 16-key tiles (dK, dV: loops query pairs,
//     S = Q.K^T, dP = dO.V^T, dV^T += dO^T.P, dK^T += Q^T.dS), then owns 16-query tiles (dQ: loops key pairs,
//     S^T, dP^T recomputed, dQ^T += K^T.dS^T).  7 MFMA products instead of 5; attention is ~3 % of the step's
//     FLOPs, so the recompute is cheaper than cross-wave reductions.
#include "common.h"

namespace hct {

int g_attn_dbg = 0;  // timing experiments on the backward kernel: bit0 skip key-owner pass, bit1 skip query-owner pass

namespace {

constexpr int kRowBytes = 128;

__device__ __forceinline__ int img_off(int r, int c) { return r * kRowBytes + ((c ^ ((r >> 1) & 7)) << 4); }

// cooperative load of one operand image: rows [0,N) x DH columns from global (row stride rs elements), zero pad.
// Loads are issued in batches of 8 per thread BEFORE any LDS write: a plain load->write loop costs one global
// round trip per iteration (4-7 serialized trips per image, which was ~40 % of the backward kernel).
template <int DH>
__device__ __forceinline__ void load_image(unsigned char* img, const bf16* __restrict__ g, int64_t rs, int N, int Npad,
                                           int nthreads) {
  constexpr int U = 8;
  const int total = Npad * 8;
  for (int base = threadIdx.x; base < total; base += nthreads * U) {
    bf16x8 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base + u * nthreads;
      const int r = idx >> 3, c = idx & 7;
      v[u] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      if (idx < total && r < N && c * 8 < DH) v[u] = *reinterpret_cast<const bf16x8*>(g + (int64_t)r * rs + c * 8);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base + u * nthreads;
      if (idx < total) *reinterpret_cast<bf16x8*>(img + img_off(idx >> 3, idx & 7)) = v[u];
    }
  }
}

// delta[r] = sum_d dO[r,d] * O[r,d] and lse[r] into LDS.  8 lanes per row (one 16-B chunk each, coalesced 96/128-B row
// reads, 3 shuffle steps) with all loads of a thread issued before their use; the old one-thread-per-row form touched 64
// different cache lines per wave instruction and serialised 12 of them per thread.
template <int DH>
__device__ __forceinline__ void compute_delta(const bf16* __restrict__ ob, const bf16* __restrict__ dob, int64_t os,
                                              const float* __restrict__ lse_row, int N, int Npad, float* sLse, float* sDel,
                                              int nthreads) {
  constexpr int U = 4;
  const int total = Npad * 8;
  for (int base = threadIdx.x; base < total; base += nthreads * U) {
    bf16x8 a[U], d[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base + u * nthreads;
      const int r = idx >> 3, c = idx & 7;
      a[u] = d[u] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      if (idx < total && r < N && c * 8 < DH) {
        a[u] = *reinterpret_cast<const bf16x8*>(ob + (int64_t)r * os + c * 8);
        d[u] = *reinterpret_cast<const bf16x8*>(dob + (int64_t)r * os + c * 8);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base + u * nthreads;
      float part = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) part += (float)a[u][e] * (float)d[u][e];
      part += __shfl_xor(part, 1, 64);
      part += __shfl_xor(part, 2, 64);
      part += __shfl_xor(part, 4, 64);
      if (idx < total && (idx & 7) == 0) {
        const int r = idx >> 3;
        sDel[r] = r < N ? part : 0.f;
        sLse[r] = r < N ? lse_row[r] * 1.44269504088896340736f : INFINITY;  // base-2 (see exp2 below); padded rows: p = 0
      }
    }
  }
}

// operand with its row index on lane&15 and 8 consecutive d (k-step ks) per lane
__device__ __forceinline__ bf16x8 frag_row(const unsigned char* img, int r0, int ks, int lane) {
  const int r = r0 + (lane & 15), c = ks * 4 + (lane >> 4);
  return *reinterpret_cast<const bf16x8*>(img + img_off(r, c));
}

// operand with column d0 + (lane&15) on the lane and 8 ROWS per lane: rows rA + 4g + {0..3}, rB + 4g + {0..3}
__device__ __forceinline__ bf16x8 frag_tr(const unsigned char* img, int rA, int rB, int d0, int lane) {
  const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
  const int c = (d0 >> 3) + (pp >> 1), sub = (pp & 1) * 8;
  const unsigned char* pa = img + img_off(rA + 4 * g + qq, c) + sub;
  const unsigned char* pb = img + img_off(rB + 4 * g + qq, c) + sub;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pa));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pb));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ bf16x8 pack8(f32x4 a, f32x4 b) {
  bf16x8 v = {(bf16)a[0], (bf16)a[1], (bf16)a[2], (bf16)a[3], (bf16)b[0], (bf16)b[1], (bf16)b[2], (bf16)b[3]};
  return v;
}

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

// (batch, head) of this workgroup.  Workgroups are dealt round-robin over the 8 XCDs; the heads of one volume interleave
// inside the same 128-B lines of the [B,N,3,H,dh] rows, so give every XCD a CONTIGUOUS run of (b,h) indices: the
// workgroups resident on an XCD at one time then work on neighbouring heads and share those lines in its L2 instead of
// each L2 fetching them again (measured: the backward's load phase ran at ~2x the algorithmic bytes).
__device__ __forceinline__ int xcd_bh() {
  const int n = gridDim.x, i = blockIdx.x;
  const int xcd = i & 7, q = n >> 3, r = n & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (i >> 3);
}

// ============================================================================================================
template <int DH>
__global__ void __launch_bounds__(256) attn_fwd_mfma_kernel(const bf16* __restrict__ qkv, int N, int H, int Npad,
                                                            bf16* __restrict__ o, float* __restrict__ lse) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Kimg = smem;
  unsigned char* Vimg = smem + Npad * kRowBytes;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bh = xcd_bh(), b = bh / H, h = bh - b * H;
  const int64_t rs = (int64_t)3 * H * DH;
  const bf16* qb = qkv + (int64_t)b * N * rs + h * DH;
  load_image<DH>(Kimg, qb + H * DH, rs, N, Npad, 256);
  load_image<DH>(Vimg, qb + 2 * H * DH, rs, N, Npad, 256);
  __syncthreads();
  const float scale = rsqrtf((float)DH);
  const float scale2 = scale * 1.44269504088896340736f;  // softmax probabilities are rebuilt in base 2 (lse is stored * log2 e)
  const int g = lane >> 4;
  constexpr int ND = DH / 16;
  const int nqt = (N + 15) >> 4;
  for (int qt = wave; qt < nqt; qt += 4) {
    const int q = qt * 16 + (lane & 15);
    bf16x8 qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int d = ks * 32 + 8 * g;
      bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (q < N && d < DH) v = *reinterpret_cast<const bf16x8*>(qb + (int64_t)q * rs + d);
      qf[ks] = v;
    }
    float m = -INFINITY, lsum = 0.f;
    f32x4 oacc[ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) oacc[dt] = f32x4{0, 0, 0, 0};
    for (int s = 0; s < (Npad >> 5); ++s) {
      const int k0 = s * 32;
      f32x4 st0 = {0, 0, 0, 0}, st1 = {0, 0, 0, 0};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        st0 = MFMA(frag_row(Kimg, k0, ks, lane), qf[ks], st0);
        st1 = MFMA(frag_row(Kimg, k0 + 16, ks, lane), qf[ks], st1);
      }
      float mx = -INFINITY;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        st0[r] = (k0 + 4 * g + r < N) ? st0[r] * scale : -INFINITY;
        st1[r] = (k0 + 16 + 4 * g + r < N) ? st1[r] * scale : -INFINITY;
        mx = fmaxf(mx, fmaxf(st0[r], st1[r]));
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mnew = fmaxf(m, mx);
      const float corr = __expf(m - mnew);
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        st0[r] = __expf(st0[r] - mnew);
        st1[r] = __expf(st1[r] - mnew);
        ps += st0[r] + st1[r];
      }
      lsum = lsum * corr + ps;
      const bf16x8 pb = pack8(st0, st1);
#pragma unroll
      for (int dt = 0; dt < ND; ++dt) {
        oacc[dt] = oacc[dt] * corr;
        oacc[dt] = MFMA(frag_tr(Vimg, k0, k0 + 16, dt * 16, lane), pb, oacc[dt]);
      }
      m = mnew;
    }
    lsum += __shfl_xor(lsum, 16, 64);
    lsum += __shfl_xor(lsum, 32, 64);
    if (q < N) {
      const float inv = 1.0f / lsum;
      bf16* orow = o + ((int64_t)b * N + q) * (H * DH) + h * DH + 4 * g;
#pragma unroll
      for (int dt = 0; dt < ND; ++dt) Vec4<bf16>::store(orow + dt * 16, oacc[dt] * inv);
      if (g == 0) lse[(int64_t)bh * N + q] = m + __logf(lsum);
    }
  }
}

// ============================================================================================================
// Forward, full-row form for short sequences (NT = Npad/16 key tiles known at compile time): per 16-query tile all
// 2*NT Q.K^T MFMAs issue back-to-back into NT accumulators, ONE softmax over the whole row (max / exp / sum: two
// cross-lane steps in total instead of per 32-key step), then the (NT/2)*ND P.V MFMAs.  8 waves per (batch, head).
template <int DH, int NT>
__global__ void __launch_bounds__(512) attn_fwd_row_kernel(const bf16* __restrict__ qkv, int N, int H, bf16* __restrict__ o,
                                                           float* __restrict__ lse) {
  constexpr int Npad = NT * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Kimg = smem;
  unsigned char* Vimg = smem + Npad * kRowBytes;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bh = xcd_bh(), b = bh / H, h = bh - b * H;
  const int64_t rs = (int64_t)3 * H * DH;
  const bf16* qb = qkv + (int64_t)b * N * rs + h * DH;
  const int g = lane >> 4;
  constexpr int ND = DH / 16;
  constexpr int MAXT = (NT + 7) / 8;  // query tiles per wave
  const int nqt = (N + 15) >> 4;
  // Q fragments of all of this wave's query tiles are fetched up front, together with the K/V images, so their HBM
  // latency is paid once per workgroup instead of once per query tile
  bf16x8 qall[MAXT][2];
#pragma unroll
  for (int it = 0; it < MAXT; ++it) {
    const int q = (wave + it * 8) * 16 + (lane & 15);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int d = ks * 32 + 8 * g;
      bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (q < N && d < DH) v = *reinterpret_cast<const bf16x8*>(qb + (int64_t)q * rs + d);
      qall[it][ks] = v;
    }
  }
  load_image<DH>(Kimg, qb + H * DH, rs, N, Npad, 512);
  load_image<DH>(Vimg, qb + 2 * H * DH, rs, N, Npad, 512);
  __syncthreads();
  const float scale = rsqrtf((float)DH) * 1.44269504088896340736f;  // fold log2(e): softmax in base 2
#pragma unroll
  for (int it = 0; it < MAXT; ++it) {
    const int qt = wave + it * 8;
    if (qt >= nqt) break;
    const int q = qt * 16 + (lane & 15);
    bf16x8 qf[2] = {qall[it][0], qall[it][1]};
    f32x4 st[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      st[t] = f32x4{0, 0, 0, 0};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) st[t] = MFMA(frag_row(Kimg, t * 16, ks, lane), qf[ks], st[t]);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        st[t][r] = (t * 16 + 4 * g + r < N) ? st[t][r] * scale : -INFINITY;
        mx = fmaxf(mx, st[t][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float ps = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        st[t][r] = __builtin_amdgcn_exp2f(st[t][r] - mx);
        ps += st[t][r];
      }
    ps += __shfl_xor(ps, 16, 64);
    ps += __shfl_xor(ps, 32, 64);
    f32x4 oacc[ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) oacc[dt] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int s2 = 0; s2 < NT / 2; ++s2) {
      const bf16x8 pb = pack8(st[2 * s2], st[2 * s2 + 1]);
#pragma unroll
      for (int dt = 0; dt < ND; ++dt) oacc[dt] = MFMA(frag_tr(Vimg, s2 * 32, s2 * 32 + 16, dt * 16, lane), pb, oacc[dt]);
    }
    if (q < N) {
      const float inv = 1.0f / ps;
      bf16* orow = o + ((int64_t)b * N + q) * (H * DH) + h * DH + 4 * g;
#pragma unroll
      for (int dt = 0; dt < ND; ++dt) Vec4<bf16>::store(orow + dt * 16, oacc[dt] * inv);
      if (g == 0) lse[(int64_t)bh * N + q] = (mx + __builtin_amdgcn_logf(ps)) * 0.69314718055994530942f;  // back to natural log
    }
  }
}

// ============================================================================================================
template <int DH>
__global__ void __launch_bounds__(512) attn_bwd_mfma_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ o,
                                                            const bf16* __restrict__ d_o, const float* __restrict__ lse,
                                                            int N, int H, int Npad, bf16* __restrict__ dqkv, int dbg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Qimg = smem;
  unsigned char* Kimg = Qimg + Npad * kRowBytes;
  unsigned char* Vimg = Kimg + Npad * kRowBytes;
  unsigned char* Dimg = Vimg + Npad * kRowBytes;  // dO
  float* sLse = reinterpret_cast<float*>(Dimg + Npad * kRowBytes);
  float* sDel = sLse + Npad;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bh = xcd_bh(), b = bh / H, h = bh - b * H;
  const int64_t rs = (int64_t)3 * H * DH;
  const int64_t os = (int64_t)H * DH;
  const bf16* qb = qkv + (int64_t)b * N * rs + h * DH;
  const bf16* ob = o + (int64_t)b * N * os + h * DH;
  const bf16* dob = d_o + (int64_t)b * N * os + h * DH;
  load_image<DH>(Qimg, qb, rs, N, Npad, 512);
  load_image<DH>(Kimg, qb + H * DH, rs, N, Npad, 512);
  load_image<DH>(Vimg, qb + 2 * H * DH, rs, N, Npad, 512);
  load_image<DH>(Dimg, dob, os, N, Npad, 512);
  compute_delta<DH>(ob, dob, os, lse + (int64_t)bh * N, N, Npad, sLse, sDel, 512);
  __syncthreads();
  const float scale = rsqrtf((float)DH);
  const float scale2 = scale * 1.44269504088896340736f;  // softmax probabilities are rebuilt in base 2 (lse is stored * log2 e)
  const int g = lane >> 4;
  constexpr int ND = DH / 16;
  const int ntile = Npad >> 4, npair = Npad >> 5;

  // ---- role 1: own a 16-key tile -> dK, dV -------------------------------------------------------------
  for (int kt = wave; kt * 16 < N && !(dbg & 1); kt += 8) {
    const int key0 = kt * 16;
    bf16x8 kf[2], vf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      kf[ks] = frag_row(Kimg, key0, ks, lane);
      vf[ks] = frag_row(Vimg, key0, ks, lane);
    }
    const bool key_ok = key0 + (lane & 15) < N;
    f32x4 dKt[ND], dVt[ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) dKt[dt] = dVt[dt] = f32x4{0, 0, 0, 0};
    for (int qp = 0; qp < npair; ++qp) {
      f32x4 P[2], dS[2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int q0 = qp * 32 + hh * 16;
        f32x4 s = {0, 0, 0, 0}, dp = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          s = MFMA(frag_row(Qimg, q0, ks, lane), kf[ks], s);     // S[q = 4g+r][key = lane&15]
          dp = MFMA(frag_row(Dimg, q0, ks, lane), vf[ks], dp);   // dP[q][key]
        }
        const f32x4 L4 = *reinterpret_cast<const f32x4*>(sLse + q0 + 4 * g);
        const f32x4 D4 = *reinterpret_cast<const f32x4*>(sDel + q0 + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = key_ok ? __builtin_amdgcn_exp2f(s[r] * scale2 - L4[r]) : 0.f;
          P[hh][r] = p;
          dS[hh][r] = p * (dp[r] - D4[r]) * scale;
        }
      }
      const bf16x8 pa = pack8(P[0], P[1]);
      const bf16x8 dsa = pack8(dS[0], dS[1]);
#pragma unroll
      for (int dt = 0; dt < ND; ++dt) {
        dVt[dt] = MFMA(frag_tr(Dimg, qp * 32, qp * 32 + 16, dt * 16, lane), pa, dVt[dt]);   // dV^T[d][key] += dO^T.P
        dKt[dt] = MFMA(frag_tr(Qimg, qp * 32, qp * 32 + 16, dt * 16, lane), dsa, dKt[dt]);  // dK^T[d][key] += Q^T.dS
      }
    }
    if (key_ok) {
      bf16* outk = dqkv + ((int64_t)b * N + key0 + (lane & 15)) * rs + H * DH + h * DH + 4 * g;
      bf16* outv = outk + H * DH;
#pragma unroll
      for (int dt = 0; dt < ND; ++dt) {
        Vec4<bf16>::store(outk + dt * 16, dKt[dt]);
        Vec4<bf16>::store(outv + dt * 16, dVt[dt]);
      }
    }
  }

  // ---- role 2: own a 16-query tile -> dQ -----------------------------------------------------------------
  for (int qt = wave; qt * 16 < N && !(dbg & 2); qt += 8) {
    const int q0 = qt * 16;
    bf16x8 qf[2], dof[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qf[ks] = frag_row(Qimg, q0, ks, lane);
      dof[ks] = frag_row(Dimg, q0, ks, lane);
    }
    const float Lq = sLse[q0 + (lane & 15)], Dq = sDel[q0 + (lane & 15)];
    f32x4 dQt[ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) dQt[dt] = f32x4{0, 0, 0, 0};
    for (int kp = 0; kp < npair; ++kp) {
      f32x4 dS[2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int k0 = kp * 32 + hh * 16;
        f32x4 s = {0, 0, 0, 0}, dp = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          s = MFMA(frag_row(Kimg, k0, ks, lane), qf[ks], s);     // S^T[key = 4g+r][q = lane&15]
          dp = MFMA(frag_row(Vimg, k0, ks, lane), dof[ks], dp);  // dP^T[key][q]
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = (k0 + 4 * g + r < N) ? __builtin_amdgcn_exp2f(s[r] * scale2 - Lq) : 0.f;
          dS[hh][r] = p * (dp[r] - Dq) * scale;
        }
      }
      const bf16x8 dsb = pack8(dS[0], dS[1]);
#pragma unroll
      for (int dt = 0; dt < ND; ++dt)
        dQt[dt] = MFMA(frag_tr(Kimg, kp * 32, kp * 32 + 16, dt * 16, lane), dsb, dQt[dt]);  // dQ^T[d][q] += K^T.dS^T
    }
    const int q = q0 + (lane & 15);
    if (q < N) {
      bf16* outq = dqkv + ((int64_t)b * N + q) * rs + h * DH + 4 * g;
#pragma unroll
      for (int dt = 0; dt < ND; ++dt) Vec4<bf16>::store(outq + dt * 16, dQt[dt]);
    }
  }
  (void)ntile;
}

// own-tile operand straight from global memory (row index on lane&15, 8 consecutive d of k-step ks), zero padded
template <int DH>
__device__ __forceinline__ bf16x8 gfrag(const bf16* __restrict__ base, int64_t rs, int r0, int ks, int lane, int N) {
  const int r = r0 + (lane & 15), d = ks * 32 + 8 * (lane >> 4);
  bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
  if (r < N && d < DH) v = *reinterpret_cast<const bf16x8*>(base + (int64_t)r * rs + d);
  return v;
}

// ============================================================================================================
// Backward, two-phase LDS use: only the two STREAMED operands of a pass live in LDS (pass 1: Q and dO; pass 2: K and V),
// the tile a wave owns comes from global memory (prefetched one tile ahead).  2 images + lse/delta = 58 KiB for 224
// tokens, 4 waves per workgroup -> two (or three) workgroups per CU, so one workgroup's load phases (40 % of the old
// single-phase kernel, which needed 112 KiB and ran alone on its CU) hide behind another's MFMA passes.
#ifndef HCT_BWD2_WPE
#define HCT_BWD2_WPE 4
#endif
template <int DH, int NW>
__global__ void __launch_bounds__(NW * 64, (NW == 8 ? HCT_BWD2_WPE : 2)) attn_bwd2_mfma_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ o,
                                                                const bf16* __restrict__ d_o, const float* __restrict__ lse,
                                                                int N, int H, int Npad, bf16* __restrict__ dqkv) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* img0 = smem;
  unsigned char* img1 = smem + Npad * kRowBytes;
  float* sLse = reinterpret_cast<float*>(img1 + Npad * kRowBytes);
  float* sDel = sLse + Npad;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bh = xcd_bh(), b = bh / H, h = bh - b * H;
  const int64_t rs = (int64_t)3 * H * DH;
  const int64_t os = (int64_t)H * DH;
  const bf16* qb = qkv + (int64_t)b * N * rs + h * DH;
  const bf16* kb = qb + H * DH;
  const bf16* vb = qb + 2 * H * DH;
  const bf16* ob = o + (int64_t)b * N * os + h * DH;
  const bf16* dob = d_o + (int64_t)b * N * os + h * DH;
  const float scale = rsqrtf((float)DH);
  const float scale2 = scale * 1.44269504088896340736f;  // softmax probabilities are rebuilt in base 2 (lse is stored * log2 e)
  const int g = lane >> 4;
  constexpr int ND = DH / 16;
  const int npair = Npad >> 5;

  // ---- pass 1: Q, dO streamed from LDS; the wave owns 16-key tiles (dK, dV) --------------------------------
  bf16x8 kf[2], vf[2], kn[2], vn[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    kf[ks] = gfrag<DH>(kb, rs, wave * 16, ks, lane, N);
    vf[ks] = gfrag<DH>(vb, rs, wave * 16, ks, lane, N);
  }
  load_image<DH>(img0, qb, rs, N, Npad, NW * 64);
  load_image<DH>(img1, dob, os, N, Npad, NW * 64);
  compute_delta<DH>(ob, dob, os, lse + (int64_t)bh * N, N, Npad, sLse, sDel, NW * 64);
  __syncthreads();
  for (int kt = wave; kt * 16 < N; kt += NW) {
    const int key0 = kt * 16;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {  // prefetch the next owned tile (zero beyond N)
      kn[ks] = gfrag<DH>(kb, rs, key0 + NW * 16, ks, lane, N);
      vn[ks] = gfrag<DH>(vb, rs, key0 + NW * 16, ks, lane, N);
    }
    const bool key_ok = key0 + (lane & 15) < N;
    f32x4 dKt[ND], dVt[ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) dKt[dt] = dVt[dt] = f32x4{0, 0, 0, 0};
    for (int qp = 0; qp < npair; ++qp) {
      f32x4 P[2], dS[2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int q0 = qp * 32 + hh * 16;
        f32x4 sacc = {0, 0, 0, 0}, dp = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          sacc = MFMA(frag_row(img0, q0, ks, lane), kf[ks], sacc);  // S[q = 4g+r][key = lane&15]
          dp = MFMA(frag_row(img1, q0, ks, lane), vf[ks], dp);      // dP[q][key]
        }
        const f32x4 L4 = *reinterpret_cast<const f32x4*>(sLse + q0 + 4 * g);
        const f32x4 D4 = *reinterpret_cast<const f32x4*>(sDel + q0 + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = key_ok ? __builtin_amdgcn_exp2f(sacc[r] * scale2 - L4[r]) : 0.f;
          P[hh][r] = pv;
          dS[hh][r] = pv * (dp[r] - D4[r]) * scale;
        }
      }
      const bf16x8 pa = pack8(P[0], P[1]);
      const bf16x8 dsa = pack8(dS[0], dS[1]);
#pragma unroll
      for (int dt = 0; dt < ND; ++dt) {
        dVt[dt] = MFMA(frag_tr(img1, qp * 32, qp * 32 + 16, dt * 16, lane), pa, dVt[dt]);   // dV^T[d][key] += dO^T.P
        dKt[dt] = MFMA(frag_tr(img0, qp * 32, qp * 32 + 16, dt * 16, lane), dsa, dKt[dt]);  // dK^T[d][key] += Q^T.dS
      }
    }
    if (key_ok) {
      bf16* outk = dqkv + ((int64_t)b * N + key0 + (lane & 15)) * rs + H * DH + h * DH + 4 * g;
      bf16* outv = outk + H * DH;
#pragma unroll
      for (int dt = 0; dt < ND; ++dt) {
        Vec4<bf16>::store(outk + dt * 16, dKt[dt]);
        Vec4<bf16>::store(outv + dt * 16, dVt[dt]);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) { kf[ks] = kn[ks]; vf[ks] = vn[ks]; }
  }

  // ---- pass 2: K, V streamed from LDS; the wave owns 16-query tiles (dQ) -----------------------------------
  bf16x8 qf[2], dof[2], qn[2], dn[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    qf[ks] = gfrag<DH>(qb, rs, wave * 16, ks, lane, N);
    dof[ks] = gfrag<DH>(dob, os, wave * 16, ks, lane, N);
  }
  __syncthreads();  // every wave is done with the Q / dO images
  load_image<DH>(img0, kb, rs, N, Npad, NW * 64);
  load_image<DH>(img1, vb, rs, N, Npad, NW * 64);
  __syncthreads();
  for (int qt = wave; qt * 16 < N; qt += NW) {
    const int q0 = qt * 16;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qn[ks] = gfrag<DH>(qb, rs, q0 + NW * 16, ks, lane, N);
      dn[ks] = gfrag<DH>(dob, os, q0 + NW * 16, ks, lane, N);
    }
    const float Lq = sLse[q0 + (lane & 15)], Dq = sDel[q0 + (lane & 15)];
    f32x4 dQt[ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) dQt[dt] = f32x4{0, 0, 0, 0};
    for (int kp = 0; kp < npair; ++kp) {
      f32x4 dS[2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int k0 = kp * 32 + hh * 16;
        f32x4 sacc = {0, 0, 0, 0}, dp = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          sacc = MFMA(frag_row(img0, k0, ks, lane), qf[ks], sacc);  // S^T[key = 4g+r][q = lane&15]
          dp = MFMA(frag_row(img1, k0, ks, lane), dof[ks], dp);     // dP^T[key][q]
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = (k0 + 4 * g + r < N) ? __builtin_amdgcn_exp2f(sacc[r] * scale2 - Lq) : 0.f;
          dS[hh][r] = pv * (dp[r] - Dq) * scale;
        }
      }
      const bf16x8 dsb = pack8(dS[0], dS[1]);
#pragma unroll
      for (int dt = 0; dt < ND; ++dt)
        dQt[dt] = MFMA(frag_tr(img0, kp * 32, kp * 32 + 16, dt * 16, lane), dsb, dQt[dt]);  // dQ^T[d][q] += K^T.dS^T
    }
    const int q = q0 + (lane & 15);
    if (q < N) {
      bf16* outq = dqkv + ((int64_t)b * N + q) * rs + h * DH + 4 * g;
#pragma unroll
      for (int dt = 0; dt < ND; ++dt) Vec4<bf16>::store(outq + dt * 16, dQt[dt]);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) { qf[ks] = qn[ks]; dof[ks] = dn[ks]; }
  }
}

inline size_t bwd2_lds(int N) { return (size_t)2 * ((N + 31) / 32 * 32) * kRowBytes + (size_t)2 * ((N + 31) / 32 * 32) * sizeof(float); }

constexpr int kMaxLds = 160 * 1024;
inline int npad_of(int N) { return (N + 31) / 32 * 32; }
inline size_t fwd_lds(int N) { return (size_t)2 * npad_of(N) * kRowBytes; }
inline size_t bwd_lds(int N) { return (size_t)4 * npad_of(N) * kRowBytes + (size_t)2 * npad_of(N) * sizeof(float); }

template <typename F>
int set_lds(F func, size_t bytes) {
  if (bytes <= 64 * 1024) return 0;
  return check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(func), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes),
                   "hipFuncSetAttribute(LDS)");
}

}  // namespace

bool attention_mfma_supported(int N, int H, int dh) {
  (void)H;
  // forward: K and V images; default backward: two phases of two images each (Q/dO, then K/V) -> N <= 608 tokens, which
  // covers the ViT-L/128^3 decoder (513 tokens)
  return (dh == 48 || dh == 64) && fwd_lds(N) <= (size_t)kMaxLds && bwd2_lds(N) <= (size_t)kMaxLds;
}

template <int DH, int NT>
static int launch_fwd_row(const void* qkv, int B, int N, int H, void* o, float* lse, hipStream_t s) {
  const size_t lds = (size_t)2 * NT * 16 * kRowBytes;
  if (int rc = set_lds(attn_fwd_row_kernel<DH, NT>, lds)) return rc;
  hipLaunchKernelGGL((attn_fwd_row_kernel<DH, NT>), dim3(B * H), dim3(512), lds, s, (const bf16*)qkv, N, H, (bf16*)o, lse);
  return check_hip(hipGetLastError(), "attention_fwd_row");
}

int g_attn_row = 1;  // testing hook: 0 = always the online-softmax kernel

int attention_fwd_mfma(const void* qkv, int B, int N, int H, int dh, void* o, float* lse, hipStream_t s) {
  const int Npad = npad_of(N);
  const size_t lds = fwd_lds(N);
  if (g_attn_row) {  // full-row kernels for the MAE token counts (<= 64, <= 160, <= 224, <= 288 keys)
    const int nt = Npad / 16;
#define HCT_ROW(DH_, NT_) if (dh == DH_ && nt <= NT_) return launch_fwd_row<DH_, NT_>(qkv, B, N, H, o, lse, s)
    HCT_ROW(64, 4); HCT_ROW(48, 4); HCT_ROW(64, 10); HCT_ROW(48, 10); HCT_ROW(64, 14); HCT_ROW(48, 14); HCT_ROW(64, 18); HCT_ROW(48, 18);
#undef HCT_ROW
  }
  if (dh == 48) {
    if (int rc = set_lds(attn_fwd_mfma_kernel<48>, lds)) return rc;
    hipLaunchKernelGGL(attn_fwd_mfma_kernel<48>, dim3(B * H), dim3(256), lds, s, (const bf16*)qkv, N, H, Npad, (bf16*)o, lse);
  } else {
    if (int rc = set_lds(attn_fwd_mfma_kernel<64>, lds)) return rc;
    hipLaunchKernelGGL(attn_fwd_mfma_kernel<64>, dim3(B * H), dim3(256), lds, s, (const bf16*)qkv, N, H, Npad, (bf16*)o, lse);
  }
  return check_hip(hipGetLastError(), "attention_fwd_mfma");
}

int attention_bwd_mfma(const void* qkv, const void* o, const void* d_o, const float* lse, int B, int N, int H, int dh,
                       void* dqkv, hipStream_t s) {
  const int Npad = npad_of(N);
  if (!(g_attn_dbg & 4) || bwd_lds(N) > (size_t)kMaxLds) {  // single-phase variant (testing hook) only where its 4 images fit
    const size_t l2 = bwd2_lds(N);
#define HCT_BWD2(DH_, NW_)                                                                                           \
  do {                                                                                                               \
    if (int rc = set_lds(attn_bwd2_mfma_kernel<DH_, NW_>, l2)) return rc;                                            \
    hipLaunchKernelGGL((attn_bwd2_mfma_kernel<DH_, NW_>), dim3(B * H), dim3(NW_ * 64), l2, s, (const bf16*)qkv,       \
                       (const bf16*)o, (const bf16*)d_o, lse, N, H, Npad, (bf16*)dqkv);                              \
  } while (0)
    const bool w8 = (g_attn_dbg & 8) ? false : N > 64;  // 8 waves (16 per CU) once there are enough tiles to share
    if (dh == 48) { if (w8) HCT_BWD2(48, 8); else HCT_BWD2(48, 4); }
    else { if (w8) HCT_BWD2(64, 8); else HCT_BWD2(64, 4); }
#undef HCT_BWD2
    return check_hip(hipGetLastError(), "attention_bwd2_mfma");
  }
  const size_t lds = bwd_lds(N);
  if (dh == 48) {
    if (int rc = set_lds(attn_bwd_mfma_kernel<48>, lds)) return rc;
    hipLaunchKernelGGL(attn_bwd_mfma_kernel<48>, dim3(B * H), dim3(512), lds, s, (const bf16*)qkv, (const bf16*)o,
                       (const bf16*)d_o, lse, N, H, Npad, (bf16*)dqkv, g_attn_dbg);
  } else {
    if (int rc = set_lds(attn_bwd_mfma_kernel<64>, lds)) return rc;
    hipLaunchKernelGGL(attn_bwd_mfma_kernel<64>, dim3(B * H), dim3(512), lds, s, (const bf16*)qkv, (const bf16*)o,
                       (const bf16*)d_o, lse, N, H, Npad, (bf16*)dqkv, g_attn_dbg);
  }
  return check_hip(hipGetLastError(), "attention_bwd_mfma");
}

}  // namespace hct
