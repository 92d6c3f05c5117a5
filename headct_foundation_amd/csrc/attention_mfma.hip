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

#include <type_traits>

namespace hct {

int g_attn_bwd3 = 54 + 128 + 1024;  // bit10: the persistent key-owner kernel (bwd4) also for head dim 64 with 129 .. 160 tokens (the ViT-L encoder's 129: 79.6 vs 87.0 us for the two-phase kernel).  bit8 (opt-in, slower): bwd5 for head dim 64 too.   // bit7: long sequences (225 .. 576 tokens) on the five-product one-wave-per-SIMD kernel (bwd5) instead of the two-phase one.   // bit5: the encoder's bwd3 instance is four waves x one key tile (44.7 us) instead of two x two (53.5 us).  bit4: bwd4 as 16 waves x one key tile (128 registers, four waves per SIMD: 287 vs 305 us, -0.09 ms per step) instead of 8 x two.  bit3 (opt-in: measured equal to the kernel it would replace, 116 vs 117 us): persistent forward (fwd4) for head dim 48 with 193 .. 224 tokens.  Which shapes use the key-owner five-product backward: bit0 head dim 48 (<= 256 tokens, bwd3), bit1 head dim 64 (<= 192 tokens, bwd3), bit2 head dim 48 with 193 .. 224 tokens (persistent bwd4)
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

// LDS-DMA fill of a 128-byte-row image (img_off: 8 chunks per row, XOR swizzle) whatever the head dim, 1 KiB pieces dealt to the waves:
// for head dim 48 the two chunks behind the head's 96 bytes are the next head's (or the next token's) data -- finite bf16 that
// only ever meets the zero-padded Q fragment -- or zero where they fall outside the tensor (the range check uses the true width).
__device__ __forceinline__ void dma_image128(unsigned char* img, const bf16* __restrict__ g, int64_t rs, int N, int Npad, int valid_cols,
                                             int wave, int nwaves, int lane) {
  const int64_t bytes = ((int64_t)(N - 1) * rs + valid_cols) * 2;
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, bytes > 0xFFFFFFFFll ? 0xFFFFFFFFu : (uint32_t)bytes, 0x00020000);
  const int pieces = Npad * 8 / 64;
  for (int p = wave; p < pieces; p += nwaves) {
    const int ci = p * 64 + lane;
    const int row = ci >> 3, slot = ci & 7;
    const uint32_t voff = (uint32_t)(row * (int)rs * 2 + ((slot ^ ((row >> 1) & 7)) << 4));
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(img + p * 1024), 16, voff, 0, 0, 0);
  }
}

// delta[r] = sum_d dO[r,d] * O[r,d] and lse[r] into LDS.  8 lanes per row (one 16-B chunk each, coalesced 96/128-B row
// reads, 3 shuffle steps) with all loads of a thread issued before their use; the old one-thread-per-row form touched 64
// different cache lines per wave instruction and serialised 12 of them per thread.
template <int DH>
__device__ __forceinline__ void compute_delta(const bf16* __restrict__ ob, const bf16* __restrict__ dob, int64_t os,
                                              const float* __restrict__ lse_row, int N, int Npad, float* sLse, float* sDel,
                                              int nthreads, float lse_mul = 1.44269504088896340736f, float del_mul = 1.0f) {
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
        sDel[r] = r < N ? part * del_mul : 0.f;
        // default: lse in base 2 (see exp2 below), +inf on padded rows so that p = 0; the key-owner kernel stores -lse / scale
        // (an accumulator input) and gets -inf there
        sLse[r] = r < N ? lse_row[r] * lse_mul : (lse_mul < 0.f ? -INFINITY : INFINITY);
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
// cross-lane steps in total instead of per 32-key step), then the (NT/2)*ND P.V MFMAs.
// NW waves per (batch, head): 8, or 4 when there are at most 4 query tiles (the encoder's 55 tokens), so that every wave owns rows.
template <int DH, int NT, int NW = (NT <= 4 ? 4 : 8)>
__global__ void __launch_bounds__(NW * 64) attn_fwd_row_kernel(const bf16* __restrict__ qkv, int N, int H, bf16* __restrict__ o,
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
  constexpr int MAXT = (NT + NW - 1) / NW;  // query tiles per wave
  const int nqt = (N + 15) >> 4;
  // Q fragments of all of this wave's query tiles are fetched up front, together with the K/V images, so their HBM
  // latency is paid once per workgroup instead of once per query tile
  // (kept to two tiles ahead: with the long score rows of the 34-tile instances more of them cost an accumulator its registers)
  constexpr int QA = MAXT <= 2 ? MAXT : 1;
  bf16x8 qall[QA][2];
  auto load_q = [&](int it, bf16x8* dst) {
    const int q = (wave + it * NW) * 16 + (lane & 15);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int d = ks * 32 + 8 * g;
      bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (q < N && d < DH) v = *reinterpret_cast<const bf16x8*>(qb + (int64_t)q * rs + d);
      dst[ks] = v;
    }
  };
#pragma unroll
  for (int it = 0; it < QA; ++it) load_q(it, qall[it]);
  dma_image128(Kimg, qb + H * DH, rs, N, Npad, DH, wave, NW, lane);  // (no registers, no VALU, no LDS stores on the way)
  dma_image128(Vimg, qb + 2 * H * DH, rs, N, Npad, DH, wave, NW, lane);
  __syncthreads();  // (with a DMA in flight hipcc's barrier waits vmcnt(0) first)
  const float scale = rsqrtf((float)DH) * 1.44269504088896340736f;  // fold log2(e): softmax in base 2
#pragma unroll
  for (int it = 0; it < MAXT; ++it) {
    const int qt = wave + it * NW;
    if (qt >= nqt) break;
    const int q = qt * 16 + (lane & 15);
    bf16x8 qf[2];
    if constexpr (MAXT <= 2) { qf[0] = qall[it][0]; qf[1] = qall[it][1]; }
    else if (it == 0) { qf[0] = qall[0][0]; qf[1] = qall[0][1]; }
    else load_q(it, qf);
    f32x4 st[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {  // (a literal zero accumulator input: no register zeroing)
      st[t] = MFMA(frag_row(Kimg, t * 16, 0, lane), qf[0], (f32x4{0, 0, 0, 0}));
      st[t] = MFMA(frag_row(Kimg, t * 16, 1, lane), qf[1], st[t]);
    }
    // row maximum of the raw scores (scale > 0), keys >= N masked; only the tiles that can hold such keys carry the mask: the
    // dispatch (attention_fwd_mfma: instances NT = 4, 10, 14, 18, 34, the smallest that fits) uses this instance only for sequences
    // longer than the previous instance's 16 * kPrevNT keys
    constexpr int kPrevNT = NT <= 4 ? 0 : NT <= 10 ? 4 : NT <= 14 ? 10 : NT <= 18 ? 14 : 18;
    float mraw = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (t >= kPrevNT && t * 16 + 4 * g + r >= N) st[t][r] = -INFINITY;
        mraw = fmaxf(mraw, st[t][r]);
      }
    mraw = fmaxf(mraw, __shfl_xor(mraw, 16, 64));
    mraw = fmaxf(mraw, __shfl_xor(mraw, 32, 64));
    const float mx = mraw * scale;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) st[t][r] = __builtin_amdgcn_exp2f(fmaf(st[t][r], scale, -mx));  // one fma + exp per score
    // The row sum rides on the matrix pipe (the kernel is VALU-issue bound, the MFMAs are 13 % busy): one more "d-tile" whose V^T
    // rows 4g are all ones gives sum_key P[key][q] in element 0 of every lane -- of the bf16-rounded P, as the numerator uses.
    const bf16 one = (bf16)1.0f, zero = (bf16)0.0f, ov = (lane & 3) == 0 ? one : zero;
    const bf16x8 ones_rows = {ov, ov, ov, ov, ov, ov, ov, ov};
    f32x4 oacc[ND], psum = {0, 0, 0, 0};
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) oacc[dt] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int s2 = 0; s2 < NT / 2; ++s2) {
      const bf16x8 pb = pack8(st[2 * s2], st[2 * s2 + 1]);
#pragma unroll
      for (int dt = 0; dt < ND; ++dt) oacc[dt] = MFMA(frag_tr(Vimg, s2 * 32, s2 * 32 + 16, dt * 16, lane), pb, oacc[dt]);
      psum = MFMA(ones_rows, pb, psum);
    }
    const float ps = psum[0];
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
__global__ void __launch_bounds__(NW * 64, (NW >= 8 ? HCT_BWD2_WPE : 2)) attn_bwd2_mfma_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ o,
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

  // The next owned tile's fragments are prefetched one tile ahead -- except for head dim 64, where those 16 registers are
  // what the 128-register budget lacks (14 spilled registers, reloaded in the inner loop); there the tile's fragments are
  // fetched at its start, once per 17+ query pairs of work.
  constexpr bool kAhead = DH != 64;
  // ---- pass 1: Q, dO streamed from LDS; the wave owns 16-key tiles (dK, dV) --------------------------------
  bf16x8 kf[2], vf[2], kn[2], vn[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    kf[ks] = gfrag<DH>(kb, rs, wave * 16, ks, lane, N);
    vf[ks] = gfrag<DH>(vb, rs, wave * 16, ks, lane, N);
  }
  load_image<DH>(img0, qb, rs, N, Npad, NW * 64);
  load_image<DH>(img1, dob, os, N, Npad, NW * 64);
  compute_delta<DH>(ob, dob, os, lse + (int64_t)bh * N, N, Npad, sLse, sDel, NW * 64, 1.44269504088896340736f, scale);  // delta stored times the softmax scale
  __syncthreads();
  for (int kt = wave; kt * 16 < N; kt += NW) {
    const int key0 = kt * 16;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {  // prefetch the next owned tile (zero beyond N)
      if (kAhead) {
        kn[ks] = gfrag<DH>(kb, rs, key0 + NW * 16, ks, lane, N);
        vn[ks] = gfrag<DH>(vb, rs, key0 + NW * 16, ks, lane, N);
      } else if (kt != wave) {
        kf[ks] = gfrag<DH>(kb, rs, key0, ks, lane, N);
        vf[ks] = gfrag<DH>(vb, rs, key0, ks, lane, N);
      }
    }
    const bool key_ok = key0 + (lane & 15) < N;
    const bool tail_tile = key0 + 16 > N;  // wave-uniform: only the tile that straddles N masks its probabilities
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
          float pv = __builtin_amdgcn_exp2f(fmaf(sacc[r], scale2, -L4[r]));  // (padded query rows: L = +inf -> 0)
          if (tail_tile && !key_ok) pv = 0.f;
          P[hh][r] = pv;
          dS[hh][r] = pv * fmaf(dp[r], scale, -D4[r]);
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
    if (kAhead) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) { kf[ks] = kn[ks]; vf[ks] = vn[ks]; }
    }
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
      if (kAhead) {
        qn[ks] = gfrag<DH>(qb, rs, q0 + NW * 16, ks, lane, N);
        dn[ks] = gfrag<DH>(dob, os, q0 + NW * 16, ks, lane, N);
      } else if (qt != wave) {
        qf[ks] = gfrag<DH>(qb, rs, q0, ks, lane, N);
        dof[ks] = gfrag<DH>(dob, os, q0, ks, lane, N);
      }
    }
    const float Lq = sLse[q0 + (lane & 15)], Dq = sDel[q0 + (lane & 15)];  // (Dq = delta * scale)
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
          float pv = __builtin_amdgcn_exp2f(fmaf(sacc[r], scale2, -Lq));
          if (k0 + 16 > N && k0 + 4 * g + r >= N) pv = 0.f;  // (the first test is wave-uniform: only the last key tile pays for the mask)
          dS[hh][r] = pv * fmaf(dp[r], scale, -Dq);
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
    if (kAhead) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) { qf[ks] = qn[ks]; dof[ks] = dn[ks]; }
    }
  }
}


// Key-owner backward kernels (bwd3 / bwd4 / bwd5), keys past the last token: their K and V rows are ZERO in the images (range-checked
// fill), so S = dP = 0 there; their dK / dV rows are never stored, and their dS only ever meets those zero K rows in the dQ product.
// All that is needed of them is FINITE P and dS -- exp2(0 - lse) overflows when every real score of a row is very negative -- and a
// probability clamped to [0, 1] is that: the clamp is an output modifier of v_exp_f32 (hipcc folds the med3), so the per-element
// multiply + select of the explicit key mask (two of the ~7 VALU operations per score in these VALU-issue-bound loops) goes away.
// A real probability is <= 1 up to rounding, so the clamp changes nothing else.  HCT_ATTN_CLAMP_MASK=0 restores the explicit mask (A/B).
#ifndef HCT_ATTN_CLAMP_MASK
#define HCT_ATTN_CLAMP_MASK 1
#endif
__device__ __forceinline__ float hct_prob(float p) {
#if HCT_ATTN_CLAMP_MASK
  return __builtin_amdgcn_fmed3f(p, 0.f, 1.f);
#else
  return p;
#endif
}

// ============================================================================================================
// Backward, key-owner form with FIVE products (bwd3): every wave owns KT consecutive 16-key tiles of one (batch, head)
// and keeps their dK^T / dV^T in accumulators while the workgroup sweeps the queries in blocks of 32.
//   per block and own key tile:  S = Q.K^T and dP = dO.V^T with the KEY on the lane  ->  P, dS in registers are already
//   the B operands of dV^T += dO^T.P and dK^T += Q^T.dS (accumulator-as-operand, no LDS trip);
//   dS crosses LDS once, as 8-byte (key, 4 queries) granules, and dQ^T = K^T.dS^T of the block is formed by the waves
//   from that image with transposed reads on both operands -- nothing is recomputed (the two-phase kernel above pays
//   7 products), no atomics, fixed summation order.
// LDS: Q, dO, K images of the head only (V never enters LDS: a wave needs just its own keys' rows, held in registers),
// with 96-B rows for head dim 48 (conflict-free without a swizzle: 24-dword row stride) and swizzled 128-B rows for 64,
// filled by LDS-DMA (buffer_load ... lds, zero fill past the last token through the descriptor's range check).
// 80 KiB per 217-token head -> two 4-wave workgroups per CU: one loads while the other computes.
// A wave streams 14 KiB of fragments per query block whatever KT is, so KT = 4 quarters the LDS read traffic of the
// 16-key-per-wave kernels above (their LDS port was as busy as their matrix pipes).
// Head dim 48 = one 32-deep + one 16-deep MFMA (v_mfma_f32_16x16x16_bf16) instead of zero-padding to 64.
template <int DH> struct HeadImg {
  static constexpr int kRow = DH == 48 ? 96 : 128;
  static constexpr int kChunks = DH / 8;
  static __device__ __forceinline__ int off(int r, int c) {
    return DH == 48 ? r * 96 + c * 16 : r * 128 + ((c ^ ((r >> 1) & 7)) << 4);
  }
};

// 16 rows x DH of an operand, row on lane&15: two 32-deep MFMA steps.  Head dim 48: the second step's upper 16 columns
// are padding -- read as whatever follows in the row-major image (finite bf16: every byte of the images is written) on the
// streamed side and as ZERO on the hoisted side (`zero_pad`), so the products vanish.
// (A 16x16x16 MFMA for the 16-column tail, chained behind the 16x16x32 one through its accumulator input, gave wrong sums on
//  gfx950 / ROCm 7.2 whenever hipcc scheduled the pair back to back -- it pads nothing between the two shapes; run as a
//  chain of its own plus a VALU add it was correct, but the adds cost more than the padded MFMA.)
template <int DH> struct RowFrag { bf16x8 lo, hi; };

template <int DH>
__device__ __forceinline__ f32x4 mma_rows(const RowFrag<DH>& a, const RowFrag<DH>& b, f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.lo, b.lo, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.hi, c, 0, 0, 0);
}

template <int DH>
__device__ __forceinline__ RowFrag<DH> rows_lds(const unsigned char* img, int r0, int lane, bool zero_pad) {
  const int r = r0 + (lane & 15), g = lane >> 4;
  RowFrag<DH> f;
  f.lo = *reinterpret_cast<const bf16x8*>(img + HeadImg<DH>::off(r, g));
  if constexpr (DH == 48) {
    f.hi = *reinterpret_cast<const bf16x8*>(img + r * 96 + 64 + 16 * g);  // g >= 2: the next row's first 32 bytes
    if (zero_pad && g >= 2) f.hi = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  } else {
    f.hi = *reinterpret_cast<const bf16x8*>(img + HeadImg<DH>::off(r, 4 + g));
  }
  return f;
}
template <int DH>
__device__ __forceinline__ RowFrag<DH> rows_global(const bf16* __restrict__ base, int64_t rs, int r0, int lane, int N) {
  const int r = r0 + (lane & 15), g = lane >> 4;
  RowFrag<DH> f;
  f.lo = f.hi = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  if (r < N) {
    const bf16* p = base + (int64_t)r * rs;
    f.lo = *reinterpret_cast<const bf16x8*>(p + 8 * g);
    if (DH == 64 || g < 2) f.hi = *reinterpret_cast<const bf16x8*>(p + 32 + 8 * g);
  }
  return f;
}
// column operand (d0 + lane&15 on the lane, 8 tokens per lane: rA + 4g + {0..3}, rB + 4g + {0..3}) of a head image
template <int DH>
__device__ __forceinline__ bf16x8 cols_lds(const unsigned char* img, int rA, int rB, int d0, int lane) {
  const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
  const int c = (d0 >> 3) + (pp >> 1), sub = (pp & 1) * 8;
  const unsigned char* pa = img + HeadImg<DH>::off(rA + 4 * g + qq, c) + sub;
  const unsigned char* pb = img + HeadImg<DH>::off(rB + 4 * g + qq, c) + sub;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pa));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pb));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// LDS-DMA fill of one head image: rows [0, Npad) x DH bf16 from global rows of stride rs elements; rows >= N read as zero
// (buffer range check).  Pieces of 1 KiB (64 lanes x 16 B) in image order, dealt round-robin to the waves.
template <int DH>
__device__ __forceinline__ void dma_image(unsigned char* img, const bf16* __restrict__ g, int64_t rs, int N, int Npad, int wave,
                                          int nwaves, int lane) {
  constexpr int CH = HeadImg<DH>::kChunks;
  const int64_t bytes = ((int64_t)(N - 1) * rs + DH) * 2;
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, bytes > 0xFFFFFFFFll ? 0xFFFFFFFFu : (uint32_t)bytes, 0x00020000);
  const int pieces = Npad * CH / 64;
  for (int p = wave; p < pieces; p += nwaves) {
    const int ci = p * 64 + lane;
    const int row = ci / CH, slot = ci - row * CH;
    const int src = DH == 48 ? slot : (slot ^ ((row >> 1) & 7));
    const uint32_t voff = (uint32_t)(row * (int)rs * 2 + src * 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(img + p * 1024), 16, voff, 0, 0, 0);
  }
}

// dS^T granule (key, 4 consecutive queries 4p..4p+3 of the block's 16-query half hh): 8 bytes at
//   hh * Npad * 32 + key * 32 + ((p ^ ((key >> 2) & 3)) << 3)
// 8 keys x 4 granules = one 256-B bank row: the transposed reads of the dQ product are conflict-free, and so are the
// accumulator-shaped writes (16 keys x one p per 16-lane group; the XOR spreads keys 0,4,8,12 over the row's four slots).
__device__ __forceinline__ int dst_off(int Npad, int hh, int key, int p) { return (hh * Npad + key) * 32 + ((p ^ ((key >> 2) & 3)) << 3); }

template <int DH, int GS, int KT, int WPS>
__global__ void __launch_bounds__(GS * 64, WPS) attn_bwd3_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ o,
                                                               const bf16* __restrict__ d_o, const float* __restrict__ lse,
                                                               int N, int H, int Npad, bf16* __restrict__ dqkv, int dbg, int stagger_from,
                                                               int stagger_sleeps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int ROW = HeadImg<DH>::kRow;
  constexpr int ND = DH / 16;
  unsigned char* Qimg = smem;
  unsigned char* Dimg = Qimg + Npad * ROW;
  unsigned char* Kimg = Dimg + Npad * ROW;
  unsigned char* dsT = Kimg + Npad * ROW;                 // [2][Npad][32 B]
  float* sLse = reinterpret_cast<float*>(dsT + 2 * Npad * 32);
  float* sDel = sLse + Npad;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bh = xcd_bh(), b = bh / H, h = bh - b * H;
  const int64_t rs = (int64_t)3 * H * DH;
  const int64_t os = (int64_t)H * DH;
  const bf16* qb = qkv + (int64_t)b * N * rs + h * DH;
  const bf16* kb = qb + H * DH;
  const bf16* vb = qb + 2 * H * DH;
  const bf16* ob = o + (int64_t)b * N * os + h * DH;
  const bf16* dob = d_o + (int64_t)b * N * os + h * DH;
  const float scale = rsqrtf((float)DH);
  const float scale2 = scale * 1.44269504088896340736f;
  const int g = lane >> 4;
  const int key_base = wave * (KT * 16);

  unsigned long long tstamp[12];
  int nstamp = 0;
#define BWD3_STAMP() do { if (dbg & 0x80) { __builtin_amdgcn_sched_barrier(0); tstamp[nstamp < 11 ? nstamp : 11] = __builtin_amdgcn_s_memrealtime(); ++nstamp; __builtin_amdgcn_sched_barrier(0); } } while (0)
  BWD3_STAMP();
  // Two workgroups share a CU and do identical work: dispatched together they would run their load, compute and store
  // phases in lockstep and never overlap.  The second one of every CU (blocks CUs .. 2*CUs-1 of the first dispatch round)
  // starts half a workgroup lifetime late; later workgroups take the slot of one that exits, which keeps the offset.
  if ((int)blockIdx.x >= stagger_from && (int)blockIdx.x < 2 * stagger_from)
    for (int i = 0; i < stagger_sleeps; ++i) __builtin_amdgcn_s_sleep(127);

  if (!(dbg & 0x800)) {
  dma_image<DH>(Qimg, qb, rs, N, Npad, wave, GS, lane);
  dma_image<DH>(Dimg, dob, os, N, Npad, wave, GS, lane);
  dma_image<DH>(Kimg, kb, rs, N, Npad, wave, GS, lane);
  }
  RowFrag<DH> vf[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) vf[t] = rows_global<DH>(vb, rs, key_base + t * 16, lane, N);
  BWD3_STAMP();  // 1: DMA + V loads issued
  if (!(dbg & 0x400)) compute_delta<DH>(ob, dob, os, lse + (int64_t)bh * N, N, Npad, sLse, sDel, GS * 64, -1.0f / scale, -1.0f);
  BWD3_STAMP();  // 2: delta done
  // key tiles past the last token are never written by an owner: their dS^T rows must read as zero in the dQ product
  for (int i = threadIdx.x * 16; i < 2 * Npad * 32; i += GS * 64 * 16) *reinterpret_cast<f32x4*>(dsT + i) = f32x4{0, 0, 0, 0};
  f32x4 dKt[KT][ND], dVt[KT][ND];
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) dKt[t][dt] = dVt[t][dt] = f32x4{0, 0, 0, 0};
  __syncthreads();  // (drains the LDS-DMA: hipcc waits vmcnt(0) in front of the barrier)
  BWD3_STAMP();  // 3: images landed

  const int nqb = Npad >> 5;
  for (int qbk = 0; qbk < nqb; ++qbk) {
    const int q0 = qbk * 32;
    // ---- main: own key tiles against the block's 32 queries -------------------------------------------------
    // Row constants ride in the accumulator inputs: S' = Q.K^T - lse / scale2 and dP' = dO.V^T - delta leave the MFMA
    // chains ready, so p = exp2(scale2 * S') and dS / scale = p * dP' are three VALU operations per element; the factor
    // `scale` of dS is applied once to the finished dK and dQ.
    if (key_base < N && !(dbg & 0x100)) {
      RowFrag<DH> qr[2], dr[2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        qr[hh] = rows_lds<DH>(Qimg, q0 + 16 * hh, lane, true);
        dr[hh] = rows_lds<DH>(Dimg, q0 + 16 * hh, lane, true);
      }
      // one wave per SIMD (WPS == 1): the whole 512-register file is this wave's, so every streamed fragment of the block
      // is fetched once and reused by all key tiles; with two waves per SIMD the column fragments and accumulator inputs
      // are re-read per key tile instead (the kernel then sits at the 256-register cap)
      constexpr bool kHoist = WPS == 1;
      bf16x8 qT[ND], dT[ND];
      f32x4 Lh[2], Dh[2];
      if constexpr (kHoist) {
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) {
          qT[dt] = cols_lds<DH>(Qimg, q0, q0 + 16, dt * 16, lane);
          dT[dt] = cols_lds<DH>(Dimg, q0, q0 + 16, dt * 16, lane);
        }
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          Lh[hh] = *reinterpret_cast<const f32x4*>(sLse + q0 + 16 * hh + 4 * g);
          Dh[hh] = *reinterpret_cast<const f32x4*>(sDel + q0 + 16 * hh + 4 * g);
        }
      }
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        const int key0 = key_base + t * 16;
        {  // straight-line over the KT tiles (no per-tile branch: hipcc schedules across tiles only inside one basic block);
           // a tile past the last token works on zero K / V rows and is masked like the straddling one
          const RowFrag<DH> kf = rows_lds<DH>(Kimg, key0 < Npad ? key0 : 0, lane, false);
          const int key = key0 + (lane & 15);
          f32x4 P[2], dS[2];
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            // accumulator inputs re-read per key tile (two broadcast reads) rather than held across the tile loop: registers
            f32x4 Linit, Dinit;  // -lse / scale (-inf on pad rows), -delta
            if constexpr (kHoist) { Linit = Lh[hh]; Dinit = Dh[hh]; }
            else {
              Linit = *reinterpret_cast<const f32x4*>(sLse + q0 + 16 * hh + 4 * g);
              Dinit = *reinterpret_cast<const f32x4*>(sDel + q0 + 16 * hh + 4 * g);
            }
            const f32x4 sacc = mma_rows<DH>(qr[hh], kf, Linit);    // S'[q = 4g+r][key = lane&15]
            const f32x4 dp = mma_rows<DH>(dr[hh], vf[t], Dinit);   // dP'[q][key]
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              P[hh][r] = hct_prob(__builtin_amdgcn_exp2f(sacc[r] * scale2));
              dS[hh][r] = P[hh][r] * dp[r];
            }
          }
#if !HCT_ATTN_CLAMP_MASK
          const float keep = key < N ? 1.f : 0.f;  // keys past the last token contribute nothing
          P[0] *= keep; P[1] *= keep; dS[0] *= keep; dS[1] *= keep;
#endif
          const bf16x8 pa = pack8(P[0], P[1]);
          const bf16x8 dsa = pack8(dS[0], dS[1]);
          const bf16x4 d0 = {dsa[0], dsa[1], dsa[2], dsa[3]}, d1 = {dsa[4], dsa[5], dsa[6], dsa[7]};
          if (key0 < Npad) {
            *reinterpret_cast<bf16x4*>(dsT + dst_off(Npad, 0, key, g)) = d0;
            *reinterpret_cast<bf16x4*>(dsT + dst_off(Npad, 1, key, g)) = d1;
          }
#pragma unroll
          for (int dt = 0; dt < ND; ++dt) {
            if constexpr (kHoist) {
              dVt[t][dt] = MFMA(dT[dt], pa, dVt[t][dt]);   // dV^T[d][key] += dO^T.P
              dKt[t][dt] = MFMA(qT[dt], dsa, dKt[t][dt]);  // dK^T[d][key] += Q^T.(dS / scale)
            } else {
              dVt[t][dt] = MFMA(cols_lds<DH>(Dimg, q0, q0 + 16, dt * 16, lane), pa, dVt[t][dt]);
              dKt[t][dt] = MFMA(cols_lds<DH>(Qimg, q0, q0 + 16, dt * 16, lane), dsa, dKt[t][dt]);
            }
          }
        }
      }
    }
    if (qbk < 2) BWD3_STAMP();  // 4, 8: main of block 0 / 1
    __syncthreads();  // the block's dS^T image is complete
    if (qbk < 2) BWD3_STAMP();  // 5, 9
    // ---- dQ^T[d][q] = sum_key K^T[d][key] . dS^T[key][q] for the block: output tiles (d-tile, query half) over the waves --
    for (int dt = wave; dt < ND && !(dbg & 0x200); dt += GS) {
      f32x4 dq0 = {0, 0, 0, 0}, dq1 = {0, 0, 0, 0};
      const int G = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
      auto ds_frag = [&](int hh, int k0) -> bf16x8 {
        const int ka = k0 + 4 * G + qq;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(dsT + dst_off(Npad, hh, ka, pp)));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(dsT + dst_off(Npad, hh, ka + 16, pp)));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
      };
      // two-deep: the fragments of key step k+1 are in flight while the MFMAs of step k run
      bf16x8 kT = cols_lds<DH>(Kimg, 0, 16, dt * 16, lane), b0 = ds_frag(0, 0), b1 = ds_frag(1, 0);
      for (int k0 = 32; k0 < Npad; k0 += 32) {
        const bf16x8 kTn = cols_lds<DH>(Kimg, k0, k0 + 16, dt * 16, lane), b0n = ds_frag(0, k0), b1n = ds_frag(1, k0);
        dq0 = MFMA(kT, b0, dq0);
        dq1 = MFMA(kT, b1, dq1);
        kT = kTn; b0 = b0n; b1 = b1n;
      }
      dq0 = MFMA(kT, b0, dq0);
      dq1 = MFMA(kT, b1, dq1);
      const int q = q0 + (lane & 15);
      if (q < N) Vec4<bf16>::store(dqkv + ((int64_t)b * N + q) * rs + h * DH + dt * 16 + 4 * g, dq0 * scale);
      if (q + 16 < N) Vec4<bf16>::store(dqkv + ((int64_t)b * N + q + 16) * rs + h * DH + dt * 16 + 4 * g, dq1 * scale);
    }
    if (qbk < 2) BWD3_STAMP();  // 6, 10: dQ of the block
    __syncthreads();  // dS^T may be overwritten by the next block
    if (qbk < 1) BWD3_STAMP();  // 7
  }
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    const int key = key_base + t * 16 + (lane & 15);
    if (key < N) {
      bf16* outk = dqkv + ((int64_t)b * N + key) * rs + H * DH + h * DH + 4 * g;
      bf16* outv = outk + H * DH;
#pragma unroll
      for (int dt = 0; dt < ND; ++dt) {
        Vec4<bf16>::store(outk + dt * 16, dKt[t][dt] * scale);
        Vec4<bf16>::store(outv + dt * 16, dVt[t][dt]);
      }
    }
  }
  if ((dbg & 0x80) && blockIdx.x == 0 && threadIdx.x == 0) {
    tstamp[11] = __builtin_amdgcn_s_memrealtime();
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(dqkv);
    for (int i = 0; i < 12; ++i) dst[i] = tstamp[i];
  }
}

#undef BWD3_STAMP
template <int DH>
inline size_t bwd3_lds(int Npad) { return (size_t)3 * Npad * HeadImg<DH>::kRow + (size_t)2 * Npad * 32 + (size_t)2 * Npad * sizeof(float); }


// ============================================================================================================
// Backward, PERSISTENT key-owner form (bwd4): the five-product algorithm of bwd3 with the three things its phase stamps
// asked for (attention):
//   * one 8-wave workgroup per CU walks the (batch, head) items; the NEXT item's Q / dO / K images (LDS-DMA into the other
//     LDS buffer), its own V rows, its O rows and lse values (registers) are in flight while the current item is computed,
//     so no workgroup ever sits in a load phase;
//   * each wave owns two 16-key tiles (256 registers per wave: every streamed fragment of a query block is fetched once,
//     nothing spills); delta = rowsum(dO . O) is formed from the dO image already in LDS and the prefetched O registers;
//   * dS^T is double-buffered, so the dQ product of query block j-1 runs in the same barrier interval as the main part of
//     block j (one barrier per block), and the two waves of a SIMD take the two parts in opposite order: one is in the
//     VALU-heavy S / P / dS part while its partner streams the MFMA-only dQ product.
// The operand stream is inline-asm LDS-DMA (hipcc would drain vmcnt before the first LDS read behind a builtin DMA).
typedef __attribute__((ext_vector_type(4))) int attn_i32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2a;
__device__ __forceinline__ attn_i32x4 attn_srd(const void* base, int64_t bytes) {
  const uint64_t pa = (uint64_t)base;
  const uint32_t rec = bytes > 0xFFFFFFFFll ? 0xFFFFFFFFu : (bytes < 0 ? 0u : (uint32_t)bytes);
  return attn_i32x4{(int)__builtin_amdgcn_readfirstlane((uint32_t)pa), (int)(__builtin_amdgcn_readfirstlane((uint32_t)(pa >> 32)) & 0xFFFF),
                    (int)__builtin_amdgcn_readfirstlane(rec), 0x00020000};
}
#ifndef HCT_ATTN_OUT_NT  // measured: +0.6 ms per step (the wgrad / dgrad GEMMs that read dqkv next want it cacheable)
#define HCT_ATTN_OUT_NT 0
#endif
#ifndef HCT_ATTN_DMA_NT  // A/B builds: 1 = the persistent kernels' Q / K / V / dO image loads non-temporal (read once per launch); measured: within noise (-0.08 ms)
#define HCT_ATTN_DMA_NT 0
#endif
__device__ __forceinline__ void attn_dma16(attn_i32x4 rsrc, uint32_t lds_base, uint32_t voff) {
#if HCT_ATTN_DMA_NT
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen nt lds" ::"s"(lds_base), "v"(voff), "s"(rsrc) : "memory", "m0");
#else
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_base), "v"(voff), "s"(rsrc) : "memory", "m0");
#endif
}

template <int DH, int Npad, int KT = 2>
__global__ void __launch_bounds__(1024 / KT, 2 * 2 / KT) attn_bwd4_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ o,
                                                           const bf16* __restrict__ d_o, const float* __restrict__ lse, int N, int H,
                                                           bf16* __restrict__ dqkv, int nbh, int dbg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int ROW = HeadImg<DH>::kRow, CH = HeadImg<DH>::kChunks, ND = DH / 16, NW = 16 / KT;  // KT key tiles per wave, NW waves
  constexpr int img = Npad * ROW;                       // one image
  constexpr int bufsz = 3 * img + 32 + 2 * Npad * 4;    // Q | dO | K | 32 zero bytes (the streamed K fragment of the last row reads them) | lse | delta
  unsigned char* const dsT0 = smem + 2 * bufsz;     // dS^T, two buffers of 2 * Npad * 32
  constexpr int dstsz = 2 * Npad * 32;
  const uint32_t lds0 = (uint32_t)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t rs = (int64_t)3 * H * DH, os = (int64_t)H * DH;
  const float scale = rsqrtf((float)DH);
  const float scale2 = scale * 1.44269504088896340736f;
  const int key_base = wave * (KT * 16);
  constexpr int nqb = Npad >> 5;
  constexpr int pieces = Npad * CH / 64;                // 1-KiB DMA pieces per image
  constexpr int MAXP = (pieces + NW - 1) / NW;
  // Per-lane offsets used only between two items (DMA sources, register prefetch, dK / dV rows) are recomputed there from an
  // opaque copy of the lane id: hoisted out of the item loop they were spilled to scratch, and every reload is a vector-memory
  // operation whose wait also drains the stores in flight.
  auto opaque = [](int v) { asm volatile("" : "+v"(v)); return v; };
  if (threadIdx.x < 16) {  // the two 32-byte pads
    *reinterpret_cast<uint32_t*>(smem + (threadIdx.x >> 3) * bufsz + 3 * img + (threadIdx.x & 7) * 4) = 0u;
  }
  auto item_bh = [&](int item) {  // XCD-contiguous walk: the workgroups of one XCD work on neighbouring heads at any time
    const int xcd = item & 7, q = nbh >> 3, r = nbh & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (item >> 3);
  };
  // dQ tiles (query half hh, d-tile dt) of a query block go to waves 4 .. 7, which run them BEFORE their main part: the two waves
  // of a SIMD are then half a block apart (one reads fragments / runs the MFMA-only dQ product while the other is in the
  // VALU-heavy P / dS part) and the fragment reads after a barrier come in two bursts.  With 14 key tiles (Npad = 224) wave 7
  // owns no keys and takes one query half whole.
  int dq_hh, dq_dt0, dq_n;
  if (KT == 1) {  // 16 waves, 14 key tiles: the two waves without keys take one query half each
    dq_hh = wave - (NW - 2); dq_dt0 = 0; dq_n = wave >= NW - 2 ? ND : 0;
  } else if (key_base >= Npad) { dq_hh = 0; dq_dt0 = 0; dq_n = ND; }
  else if (8 * KT * 16 > Npad) { dq_hh = 1; dq_dt0 = wave - 4; dq_n = (wave >= 4 && wave < 4 + ND) ? 1 : 0; }
  else { dq_hh = (wave - 2) / ND; dq_dt0 = (wave - 2) - dq_hh * ND; dq_n = wave >= 2 ? 1 : 0; }

  auto prefetch_images = [&](int item, int buf) {
    const int bh = item_bh(item), b = bh / H, h = bh - b * H;
    const bf16* qb = qkv + (int64_t)b * N * rs + h * DH;
    const bf16* dob = d_o + (int64_t)b * N * os + h * DH;
    const attn_i32x4 rq = attn_srd(qb, ((int64_t)(N - 1) * rs + DH) * 2);
    const attn_i32x4 rk = attn_srd(qb + H * DH, ((int64_t)(N - 1) * rs + DH) * 2);
    const attn_i32x4 rd = attn_srd(dob, ((int64_t)(N - 1) * os + DH) * 2);
    const uint32_t base = lds0 + buf * bufsz;
    const int ln = opaque(lane);
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      const int p = wave + NW * i;
      if (p < pieces) {  // wave-uniform; piece p = 64 consecutive 16-byte chunks of the image, lane -> (row, slot)
        const int ci = p * 64 + ln, row = ci / CH, slot = ci - row * CH;
        const uint32_t src = (uint32_t)((DH == 48 ? slot : (slot ^ ((row >> 1) & 7))) * 16);
        const uint32_t vq = (uint32_t)row * (uint32_t)(rs * 2) + src, vd = (uint32_t)row * (uint32_t)(os * 2) + src;
        attn_dma16(rq, base + p * 1024, vq);
        attn_dma16(rd, base + img + p * 1024, vd);
        attn_dma16(rk, base + 2 * img + p * 1024, vq);
      }
    }
  };
  // own V rows, O chunks (for delta) and lse values of an item, into registers
  RowFrag<DH> vfN[KT];
  constexpr int NU = (Npad * 8 + NW * 64 - 1) / (NW * 64);  // 16-byte O chunks per thread
  bf16x8 oN[NU];
  float lN = 0.f;
  auto prefetch_regs = [&](int item) {
    const int bh = item_bh(item), b = bh / H, h = bh - b * H;
    const bf16* vb = qkv + (int64_t)b * N * rs + 2 * H * DH + h * DH;
    const int ln = opaque(lane), tid = wave * 64 + ln;
#pragma unroll
    for (int t = 0; t < KT; ++t) vfN[t] = rows_global<DH>(vb, rs, key_base + t * 16, ln, N);
    const bf16* ob = o + (int64_t)b * N * os + h * DH;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int idx = tid + u * NW * 64, r = idx >> 3, c = idx & 7;
      oN[u] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      if (r < N && c * 8 < DH) oN[u] = *reinterpret_cast<const bf16x8*>(ob + (int64_t)r * os + c * 8);
    }
    lN = tid < N ? lse[(int64_t)bh * N + tid] : 0.f;
  };

  // (testing, dbg & 0x80) phase stamps of workgroup 0, waves 0 / NW/2 / NW-1, first four items: 10 per item, kept in the spare LDS
  // behind the dS^T buffers and dumped one per token row into the dQ slice of the workgroup's last item
  unsigned long long* const stamps = reinterpret_cast<unsigned long long*>(dsT0 + 2 * dstsz);
  const int swave = wave == 0 ? 0 : wave == NW / 2 ? 1 : wave == NW - 1 ? 2 : -1;
#define BWD4_STAMP(k_) do { if ((dbg & 0x80) && blockIdx.x == 0 && swave >= 0 && it < 4 && lane == 0) { __builtin_amdgcn_sched_barrier(0); stamps[(swave * 4 + it) * 10 + (k_)] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
  int item = blockIdx.x;
  if (item >= nbh) return;
  prefetch_images(item, 0);
  prefetch_regs(item);
  for (int it = 0; item < nbh; item += gridDim.x, ++it) {
    const int cur = it & 1;
    unsigned char* Qimg = smem + cur * bufsz;
    unsigned char* Dimg = Qimg + img;
    unsigned char* Kimg = Dimg + img;
    float* sLse = reinterpret_cast<float*>(Kimg + img + 32);
    float* sDel = sLse + Npad;
    const int bh = item_bh(item), b = bh / H, h = bh - b * H;
    // this item's prefetch has landed: own DMA retired, then every wave's
    BWD4_STAMP(0);
    // A use of the YOUNGEST prefetched value: the compiler's counted vmcnt wait for it also covers the older loads and DMA pieces
    // (vector-memory operations retire in issue order) but leaves the previous item's dK / dV stores, issued after it, in flight.
    asm volatile("" ::"v"(lN) : "memory");
    BWD4_STAMP(1);
    __builtin_amdgcn_s_barrier();
    BWD4_STAMP(2);
    RowFrag<DH> vf[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) vf[t] = vfN[t];
    RowFrag<DH> kfh = vf[0];  // (KT == 1 only) this wave's K rows, zero-padded: read once per item
    if constexpr (KT == 1) kfh = rows_lds<DH>(Kimg, key_base < Npad ? key_base : 0, opaque(lane), true);
    // delta[r] = sum_d dO[r,d] O[r,d] from the dO image and the prefetched O registers (8 lanes per row), stored times the
    // softmax scale: dS = P (dP scale - delta scale)
    const int tid_top = wave * 64 + opaque(lane);  // (opaque: the offsets below are not worth registers across the query loop)
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int idx = tid_top + u * NW * 64, r = idx >> 3, c = idx & 7;
      float part = 0.f;
      if (idx < Npad * 8 && c * 8 < DH) {
        const bf16x8 dv = *reinterpret_cast<const bf16x8*>(Dimg + HeadImg<DH>::off(r, c));
#pragma unroll
        for (int e = 0; e < 8; ++e) part += (float)oN[u][e] * (float)dv[e];
      }
      part += __shfl_xor(part, 1, 64);
      part += __shfl_xor(part, 2, 64);
      part += __shfl_xor(part, 4, 64);
      if (idx < Npad * 8 && c == 0) sDel[r] = r < N ? part * scale : 0.f;
    }
    if (tid_top < Npad) sLse[tid_top] = tid_top < N ? lN * 1.44269504088896340736f : INFINITY;  // base 2; +inf: p = 0 on padded query rows
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    BWD4_STAMP(3);
    const int next = item + gridDim.x;
    if (next < nbh) prefetch_images(next, cur ^ 1);  // in flight during the whole compute below
    BWD4_STAMP(4);

    f32x4 dKt[KT][ND], dVt[KT][ND];
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
      for (int dt = 0; dt < ND; ++dt) dKt[t][dt] = dVt[t][dt] = f32x4{0, 0, 0, 0};

    auto main_part = [&](int qbk) {
      const int q0 = qbk * 32;
      unsigned char* dsT = dsT0 + (qbk & 1) * dstsz;
      // 128-register variant: per-lane LDS offsets recomputed per block (kept across the loop they were spilled, two scratch reloads per block)
      const int ln = KT == 1 ? opaque(lane) : lane, g = ln >> 4;
      RowFrag<DH> qr[2], dr[2];
      f32x4 L4[2], D4[2];
      if constexpr (KT > 1) {  // shared by the wave's key tiles: fetched once per block
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          qr[hh] = rows_lds<DH>(Qimg, q0 + 16 * hh, ln, true);
          dr[hh] = rows_lds<DH>(Dimg, q0 + 16 * hh, ln, false);  // (its partner, the V fragment, is the zero-padded side)
          L4[hh] = *reinterpret_cast<const f32x4*>(sLse + q0 + 16 * hh + 4 * g);
          D4[hh] = *reinterpret_cast<const f32x4*>(sDel + q0 + 16 * hh + 4 * g);
        }
      }
      bf16x8 qT[ND], dT[ND];
#pragma unroll
      for (int dt = 0; dt < ND; ++dt) {
        qT[dt] = cols_lds<DH>(Qimg, q0, q0 + 16, dt * 16, ln);
        dT[dt] = cols_lds<DH>(Dimg, q0, q0 + 16, dt * 16, ln);
      }
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        const int key0 = key_base + t * 16;  // < Npad: Npad is a multiple of 32
        // KT == 1: the wave's K fragment is fetched (and zero-padded) once per item, so the Q / dO fragments stream unpadded
        const RowFrag<DH> kf = KT == 1 ? kfh : rows_lds<DH>(Kimg, key0, ln, false);
        const int key = key0 + (ln & 15);
#if !HCT_ATTN_CLAMP_MASK
        const float keep = key < N ? 1.f : 0.f;
        const bool tail_tile = key0 + 16 > N;  // wave-uniform: only the tile that straddles N masks its probabilities
#endif
        f32x4 P[2], dS[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          if constexpr (KT == 1) {  // one key tile per wave (128 registers): the two query halves one after the other
            qr[hh] = rows_lds<DH>(Qimg, q0 + 16 * hh, ln, false);
            dr[hh] = rows_lds<DH>(Dimg, q0 + 16 * hh, ln, false);
            L4[hh] = *reinterpret_cast<const f32x4*>(sLse + q0 + 16 * hh + 4 * g);
            D4[hh] = *reinterpret_cast<const f32x4*>(sDel + q0 + 16 * hh + 4 * g);
          }
          const f32x4 sacc = mma_rows<DH>(qr[hh], kf, f32x4{0, 0, 0, 0});    // S[q = 4g+r][key = lane&15]
          const f32x4 dp = mma_rows<DH>(dr[hh], vf[t], f32x4{0, 0, 0, 0});   // dP[q][key]
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            P[hh][r] = hct_prob(__builtin_amdgcn_exp2f(fmaf(sacc[r], scale2, -L4[hh][r])));
#if !HCT_ATTN_CLAMP_MASK
            if (tail_tile) P[hh][r] *= keep;
#endif
            dS[hh][r] = P[hh][r] * fmaf(dp[r], scale, -D4[hh][r]);
          }
        }
        const bf16x8 pa = pack8(P[0], P[1]);
        const bf16x8 dsa = pack8(dS[0], dS[1]);
        const bf16x4 d0 = {dsa[0], dsa[1], dsa[2], dsa[3]}, d1 = {dsa[4], dsa[5], dsa[6], dsa[7]};
        *reinterpret_cast<bf16x4*>(dsT + dst_off(Npad, 0, key, g)) = d0;
        *reinterpret_cast<bf16x4*>(dsT + dst_off(Npad, 1, key, g)) = d1;
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) {
          dVt[t][dt] = MFMA(dT[dt], pa, dVt[t][dt]);   // dV^T[d][key] += dO^T.P
          dKt[t][dt] = MFMA(qT[dt], dsa, dKt[t][dt]);  // dK^T[d][key] += Q^T.dS
        }
#ifdef HCT_BWD4_SEQ_TILES  // (experiment) keep the two key tiles strictly one after the other in the instruction stream
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
    };
    // dQ^T[d][q] = sum_key K^T[d][key] dS^T[key][q] of block qbk, tiles (hh, dt0 .. dt0 + NT - 1): the dS^T fragment is shared
    auto dq_tiles = [&](int qbk, int hh, int dt0, auto nt_tag, int ln) {
      constexpr int NT = decltype(nt_tag)::value;
      const unsigned char* dsT = dsT0 + (qbk & 1) * dstsz;
      const int qq = (ln & 15) >> 2, pp = ln & 3, g = ln >> 4;
      auto ds_frag = [&](int k0) -> bf16x8 {
        const int ka = k0 + 4 * g + qq;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(dsT + dst_off(Npad, hh, ka, pp)));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(dsT + dst_off(Npad, hh, ka + 16, pp)));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
      };
      // the fragments of the product are requested ahead of the MFMAs that use them (the registers of the main part are free
      // here), in one go where they fit and in halves of the key range otherwise: the steps then run back to back instead of
      // one LDS round trip each
      f32x4 dq[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) dq[j] = f32x4{0, 0, 0, 0};
      constexpr int kHalf = (KT == 1 && NT > 1) ? (nqb + 1) / 2 : nqb;
#pragma unroll
      for (int k0 = 0; k0 < nqb; k0 += kHalf) {
        bf16x8 kT[kHalf][NT], bq[kHalf];
#pragma unroll
        for (int kb = 0; kb < kHalf; ++kb)
          if (k0 + kb < nqb) {
            bq[kb] = ds_frag((k0 + kb) * 32);
#pragma unroll
            for (int j = 0; j < NT; ++j) kT[kb][j] = cols_lds<DH>(Kimg, (k0 + kb) * 32, (k0 + kb) * 32 + 16, (dt0 + j) * 16, ln);
          }
#pragma unroll
        for (int kb = 0; kb < kHalf; ++kb)
          if (k0 + kb < nqb) {
#pragma unroll
            for (int j = 0; j < NT; ++j) dq[j] = MFMA(kT[kb][j], bq[kb], dq[j]);
          }
      }
      const int q = qbk * 32 + 16 * hh + (ln & 15);
#pragma unroll
      for (int j = 0; j < NT; ++j)
#if HCT_ATTN_OUT_NT
        if (q < N) Vec4<bf16>::store_nt(dqkv + ((int64_t)b * N + q) * rs + h * DH + (dt0 + j) * 16 + 4 * g, dq[j]);
#else
        if (q < N) Vec4<bf16>::store(dqkv + ((int64_t)b * N + q) * rs + h * DH + (dt0 + j) * 16 + 4 * g, dq[j]);
#endif
    };
    auto dq_part = [&](int qbk, int ln) {
      if (dq_n == 1) dq_tiles(qbk, dq_hh, dq_dt0, std::integral_constant<int, 1>{}, ln);
      else if (dq_n == ND) dq_tiles(qbk, dq_hh, dq_dt0, std::integral_constant<int, ND>{}, ln);
    };
    const bool owns_keys = key_base < Npad;
    for (int qbk = 0; qbk < nqb; ++qbk) {
      if (wave < NW / 2) {
        if (owns_keys && !(dbg & 0x100)) main_part(qbk);
        if (qbk > 0 && !(dbg & 0x200)) dq_part(qbk - 1, KT == 1 ? opaque(lane) : lane);  // (128-register variant: nothing hoisted out of the item loop)
      } else {
        if (qbk > 0 && !(dbg & 0x200)) dq_part(qbk - 1, KT == 1 ? opaque(lane) : lane);
        if (owns_keys && !(dbg & 0x100)) main_part(qbk);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (qbk == 0) BWD4_STAMP(5);
      if (qbk == 1) BWD4_STAMP(6);
    }
    BWD4_STAMP(7);
    if (next < nbh) prefetch_regs(next);  // consumed at the top of the next item: the dQ tail and the dK / dV stores cover the latency
    if (!(dbg & 0x200)) dq_part(nqb - 1, opaque(lane));
    BWD4_STAMP(8);
    if (owns_keys) {
      const int ln = opaque(lane);
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        const int key = key_base + t * 16 + (ln & 15);
        if (key < N) {
          bf16* outk = dqkv + ((int64_t)b * N + key) * rs + H * DH + h * DH + 4 * (ln >> 4);
          bf16* outv = outk + H * DH;
#pragma unroll
          for (int dt = 0; dt < ND; ++dt) {
#if HCT_ATTN_OUT_NT  // A/B builds: the persistent backward's dQ / dK / dV stores non-temporal (the GEMM epilogues' bf16 outputs do best that way)
            Vec4<bf16>::store_nt(outk + dt * 16, dKt[t][dt]);
            Vec4<bf16>::store_nt(outv + dt * 16, dVt[t][dt]);
#else
            Vec4<bf16>::store(outk + dt * 16, dKt[t][dt]);
            Vec4<bf16>::store(outv + dt * 16, dVt[t][dt]);
#endif
          }
        }
      }
    }
    BWD4_STAMP(9);
    if ((dbg & 0x80) && blockIdx.x == 0 && next >= nbh) {
      __syncthreads();
      if (threadIdx.x < 120) *reinterpret_cast<unsigned long long*>(dqkv + ((int64_t)b * N + threadIdx.x) * rs + h * DH) = stamps[threadIdx.x];
    }
  }
#undef BWD4_STAMP
}

// ============================================================================================================
// Backward for LONG sequences (bwd5: 225 .. 576 tokens -- the ViT-L/128^3 decoder's 513 at head dim 48, DINO's 517 at head dim 64):
// the five-product key-owner algorithm of bwd3 / bwd4 where the Q / dO / K images of a head no longer fit LDS together (the
// two-phase kernel these lengths ran on pays seven products and two passes over the images).
//   * ONE wave per SIMD (four waves, the whole 512-register file each): a wave owns KT = 9 consecutive 16-key tiles, their
//     dK^T / dV^T accumulators (up to 288 registers) and V rows (72) stay in registers for the whole head; every streamed
//     fragment of a query block is fetched once and shared by the nine tiles;
//   * only the K image is resident in LDS (rows for S = Q.K^T, transposed reads for dQ^T = K^T.dS^T); Q and dO stream through a
//     two-slot ring of 32-row blocks: the rows of block j+1 (and the matching O rows, for delta = rowsum(dO . O)) are requested
//     into registers at the top of block j and written to the other slot behind its arithmetic -- plain loads the compiler
//     counts, no LDS-DMA in the loop;
//   * dS^T is double-buffered, so the dQ product of block j-1 runs in the same barrier interval as the main part of block j:
//     one barrier per block;  no atomics, fixed summation order.
// LDS at 544 padded tokens, head dim 64: K 68 KiB + ring 16 KiB + dS^T 2 x 34 KiB = 153 KiB.
template <int DH, int KT>
__global__ void __launch_bounds__(256, 1) attn_bwd5_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ o,
                                                           const bf16* __restrict__ d_o, const float* __restrict__ lse, int N, int H,
                                                           int Npad, bf16* __restrict__ dqkv) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int ROW = HeadImg<DH>::kRow, CH = HeadImg<DH>::kChunks, ND = DH / 16;
  constexpr int blk = 32 * ROW;                      // one 32-row block image
  unsigned char* const Kimg = smem;                  // Npad rows, then 32 zero bytes (the last row's padded fragment reads them)
  unsigned char* const ring = Kimg + Npad * ROW + 32;  // [2][Q block | dO block], then 32 zero bytes
  unsigned char* const dsT0 = ring + 4 * blk + 32;   // [2][2 halves][Npad][32 B]
  constexpr int kRows = 4 * KT * 16;                  // dS^T rows per query half: every tile slot of every wave has its own (no branch around the writes)
  constexpr int dstsz = 2 * kRows * 32;
  float* const sLse = reinterpret_cast<float*>(dsT0 + 2 * dstsz);  // [2][32]
  float* const sDel = sLse + 64;                                   // [2][32]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bh = xcd_bh(), b = bh / H, h = bh - b * H;
  const int64_t rs = (int64_t)3 * H * DH, os = (int64_t)H * DH;
  const bf16* qb = qkv + (int64_t)b * N * rs + h * DH;
  const bf16* kb = qb + H * DH;
  const bf16* vb = qb + 2 * H * DH;
  const bf16* ob = o + (int64_t)b * N * os + h * DH;
  const bf16* dob = d_o + (int64_t)b * N * os + h * DH;
  const float* lse_row = lse + (int64_t)bh * N;
  const float scale = rsqrtf((float)DH);
  const float scale2 = scale * 1.44269504088896340736f;
  const int g = lane >> 4;
  const int key_base = wave * (KT * 16);
  const int nqb = Npad >> 5;

  // every byte a padded fragment may read must be finite (0 x NaN = NaN on the matrix pipe): ring, its pad and K's pad
  for (int i = threadIdx.x * 16; i < 4 * blk + 32; i += 256 * 16) *reinterpret_cast<f32x4*>(ring + i) = f32x4{0, 0, 0, 0};
  if (threadIdx.x < 2) *reinterpret_cast<f32x4*>(Kimg + Npad * ROW + threadIdx.x * 16) = f32x4{0, 0, 0, 0};
  dma_image<DH>(Kimg, kb, rs, N, Npad, wave, 4, lane);
  RowFrag<DH> vf[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) vf[t] = rows_global<DH>(vb, rs, key_base + t * 16, lane, N);

  // staging of one query block: thread -> (row tid >> 3, 16-byte chunk tid & 7) of Q, dO and O; thread < 32 -> the row's lse
  const int srow = threadIdx.x >> 3, sc = threadIdx.x & 7;
  bf16x8 sq, sd, so;
  float sl = 0.f;
  auto stage_load = [&](int j) {
    const int r = j * 32 + srow;
    sq = sd = so = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (r < N && sc < CH) {
      sq = *reinterpret_cast<const bf16x8*>(qb + (int64_t)r * rs + sc * 8);
      sd = *reinterpret_cast<const bf16x8*>(dob + (int64_t)r * os + sc * 8);
      so = *reinterpret_cast<const bf16x8*>(ob + (int64_t)r * os + sc * 8);
    }
    const int rl = j * 32 + (int)threadIdx.x;
    sl = (threadIdx.x < 32 && rl < N) ? lse_row[rl] : 0.f;
  };
  auto stage_write = [&](int j) {
    unsigned char* Qb = ring + (j & 1) * 2 * blk;
    if (sc < CH) {
      *reinterpret_cast<bf16x8*>(Qb + HeadImg<DH>::off(srow, sc)) = sq;
      *reinterpret_cast<bf16x8*>(Qb + blk + HeadImg<DH>::off(srow, sc)) = sd;
    }
    float part = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) part += (float)so[e] * (float)sd[e];
    part += __shfl_xor(part, 1, 64);
    part += __shfl_xor(part, 2, 64);
    part += __shfl_xor(part, 4, 64);
    // delta times the softmax scale: dS = P (dP scale - delta scale); lse in base 2, +inf on padded query rows (p = 0)
    if (sc == 0) sDel[(j & 1) * 32 + srow] = j * 32 + srow < N ? part * scale : 0.f;
    if (threadIdx.x < 32) sLse[(j & 1) * 32 + threadIdx.x] = j * 32 + (int)threadIdx.x < N ? sl * 1.44269504088896340736f : INFINITY;
  };
  stage_load(0);
  stage_write(0);

  f32x4 dKt[KT][ND], dVt[KT][ND];
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) dKt[t][dt] = dVt[t][dt] = f32x4{0, 0, 0, 0};
  __syncthreads();  // K image landed (hipcc drains the LDS-DMA in front of the barrier), block 0 staged

  auto main_part = [&](int j) {
    const unsigned char* Qb = ring + (j & 1) * 2 * blk;
    const unsigned char* Db = Qb + blk;
    unsigned char* dsT = dsT0 + (j & 1) * dstsz;
    RowFrag<DH> qr[2], dr[2];
    f32x4 L4[2], D4[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      qr[hh] = rows_lds<DH>(Qb, 16 * hh, lane, true);
      dr[hh] = rows_lds<DH>(Db, 16 * hh, lane, false);  // (its partner, the V fragment, is the zero-padded side)
      L4[hh] = *reinterpret_cast<const f32x4*>(sLse + (j & 1) * 32 + 16 * hh + 4 * g);
      D4[hh] = *reinterpret_cast<const f32x4*>(sDel + (j & 1) * 32 + 16 * hh + 4 * g);
    }
    bf16x8 qT[ND], dT[ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) {
      qT[dt] = cols_lds<DH>(Qb, 0, 16, dt * 16, lane);
      dT[dt] = cols_lds<DH>(Db, 0, 16, dt * 16, lane);
    }
    // One wave per SIMD has nobody to cover its latencies, so the tile loop is software-pipelined by hand: the S / dP products of
    // tile t+1 are ISSUED before the exponentials of tile t (they run on the matrix pipe under that arithmetic), the K rows of
    // tile t+2 are requested before them, and the dV / dK products of tile t go out behind its arithmetic, under tile t+1's.
    // (tiles at or past Npad -- the last wave's spare ones -- run on K row 0 and zero V rows: finite, never stored)
    auto krows = [&](int t) { const int key0 = key_base + t * 16; return rows_lds<DH>(Kimg, key0 < Npad ? key0 : 0, lane, false); };
    RowFrag<DH> kfA = krows(0), kfB = krows(KT > 1 ? 1 : 0);
    f32x4 sc[2], pc[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      sc[hh] = mma_rows<DH>(qr[hh], kfA, f32x4{0, 0, 0, 0});    // S[q = 4g+r][key = lane&15]
      pc[hh] = mma_rows<DH>(dr[hh], vf[0], f32x4{0, 0, 0, 0});  // dP[q][key]
    }
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      const int key = key_base + t * 16 + (lane & 15);
      f32x4 sn[2] = {sc[0], sc[1]}, pn[2] = {pc[0], pc[1]};
      kfA = kfB;
      if (t + 2 < KT) kfB = krows(t + 2);
      if (t + 1 < KT) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          sn[hh] = mma_rows<DH>(qr[hh], kfA, f32x4{0, 0, 0, 0});
          pn[hh] = mma_rows<DH>(dr[hh], vf[t + 1], f32x4{0, 0, 0, 0});
        }
      }
      // No key mask: a probability is clamped to [0, 1] (free: an output modifier of v_exp_f32), so the keys past the last token --
      // zero K and V rows, S = dP = 0 -- give finite P and dS whatever lse is; their dK / dV rows are never stored and their dS
      // meets zero K rows in the dQ product.
      f32x4 P[2], dS[2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          P[hh][r] = __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(fmaf(sc[hh][r], scale2, -L4[hh][r])), 0.f, 1.f);
          dS[hh][r] = P[hh][r] * fmaf(pc[hh][r], scale, -D4[hh][r]);
        }
      }
      const bf16x8 pa = pack8(P[0], P[1]);
      const bf16x8 dsa = pack8(dS[0], dS[1]);
      const bf16x4 d0 = {dsa[0], dsa[1], dsa[2], dsa[3]}, d1 = {dsa[4], dsa[5], dsa[6], dsa[7]};
      *reinterpret_cast<bf16x4*>(dsT + dst_off(kRows, 0, key, g)) = d0;
      *reinterpret_cast<bf16x4*>(dsT + dst_off(kRows, 1, key, g)) = d1;
#pragma unroll
      for (int dt = 0; dt < ND; ++dt) {
        dVt[t][dt] = MFMA(dT[dt], pa, dVt[t][dt]);   // dV^T[d][key] += dO^T.P
        dKt[t][dt] = MFMA(qT[dt], dsa, dKt[t][dt]);  // dK^T[d][key] += Q^T.dS
      }
      sc[0] = sn[0]; sc[1] = sn[1]; pc[0] = pn[0]; pc[1] = pn[1];
    }
  };
  // dQ^T[d][q] = sum_key K^T[d][key] dS^T[key][q] of block j: d-tile = wave, both query halves share the K^T fragment
  auto dq_part = [&](int j) {
    const unsigned char* dsT = dsT0 + (j & 1) * dstsz;
    for (int dt = wave; dt < ND; dt += 4) {
      f32x4 dq0 = {0, 0, 0, 0}, dq1 = {0, 0, 0, 0};
      const int qq = (lane & 15) >> 2, pp = lane & 3;
      auto ds_frag = [&](int hh, int k0) -> bf16x8 {
        const int ka = k0 + 4 * g + qq;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(dsT + dst_off(kRows, hh, ka, pp)));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(dsT + dst_off(kRows, hh, ka + 16, pp)));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
      };
      // four key steps of fragments in flight ahead of their MFMAs (the registers are there)
      constexpr int D = 4;
      bf16x8 kT[D], b0[D], b1[D];
#pragma unroll
      for (int u = 0; u < D; ++u) {
        const int k0 = u * 32 < Npad ? u * 32 : 0;
        kT[u] = cols_lds<DH>(Kimg, k0, k0 + 16, dt * 16, lane);
        b0[u] = ds_frag(0, k0);
        b1[u] = ds_frag(1, k0);
      }
      for (int k0 = 0; k0 < Npad; k0 += 32 * D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
          if (k0 + u * 32 < Npad) {  // wave-uniform
            dq0 = MFMA(kT[u], b0[u], dq0);
            dq1 = MFMA(kT[u], b1[u], dq1);
          }
          const int kn = k0 + (u + D) * 32;
          if (kn < Npad) {
            kT[u] = cols_lds<DH>(Kimg, kn, kn + 16, dt * 16, lane);
            b0[u] = ds_frag(0, kn);
            b1[u] = ds_frag(1, kn);
          }
        }
      }
      const int q = j * 32 + (lane & 15);
      if (q < N) Vec4<bf16>::store(dqkv + ((int64_t)b * N + q) * rs + h * DH + dt * 16 + 4 * g, dq0);
      if (q + 16 < N) Vec4<bf16>::store(dqkv + ((int64_t)b * N + q + 16) * rs + h * DH + dt * 16 + 4 * g, dq1);
    }
  };

  for (int j = 0; j < nqb; ++j) {
    if (j + 1 < nqb) stage_load(j + 1);  // in flight under this block's arithmetic
    main_part(j);
    if (j > 0) dq_part(j - 1);
    if (j + 1 < nqb) stage_write(j + 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  dq_part(nqb - 1);
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    const int key = key_base + t * 16 + (lane & 15);
    if (key < N) {
      bf16* outk = dqkv + ((int64_t)b * N + key) * rs + H * DH + h * DH + 4 * g;
      bf16* outv = outk + H * DH;
#pragma unroll
      for (int dt = 0; dt < ND; ++dt) {
        Vec4<bf16>::store(outk + dt * 16, dKt[t][dt]);
        Vec4<bf16>::store(outv + dt * 16, dVt[t][dt]);
      }
    }
  }
}
template <int DH>
inline size_t bwd5_lds(int Npad, int KT) { return (size_t)Npad * HeadImg<DH>::kRow + 32 + (size_t)4 * 32 * HeadImg<DH>::kRow + 32 + (size_t)2 * 2 * (4 * KT * 16) * 32 + 512; }

// ============================================================================================================
// Forward, PERSISTENT form (fwd4) for the decoder shape (head dim 48, 193 .. 224 tokens): the structure of bwd4 without any
// exchange between waves.  One 16-wave workgroup per CU walks the (batch, head) items; the next item's Q / K / V images land by
// LDS-DMA in the other LDS buffer while the current one is computed; one barrier per item.  A wave owns one 16-query tile, keeps
// the whole score row in registers (S^T = K Q^T: 14 key tiles = 56 registers), takes the row maximum and sum across its four
// 16-lane groups, and feeds P as the B operand of O^T = V^T P.  Four waves per SIMD (128 registers each) drift apart between two
// item barriers, so one wave's softmax VALU runs under another's MFMAs -- with 8 waves of two tiles each (256 registers, half
// the K / V fragment reads) the two waves of a SIMD stayed in step and the kernel was slower than the one it replaces (128 vs
// 116 us).  MEASURED (scripts/dbg/attn_fwd4.py, B = 256, N = 217, H = 16): 117 us against 116 us for attn_fwd_row_kernel, so it
// is NOT the default (g_attn_bwd3 bit 3 selects it); image traffic alone 55 us, compute + stores on stale images 78 us: the two
// overlap only partly, and the compute part is VALU-issue bound (per 16-query tile 290 VALU of which 56 v_exp_f32, 49 MFMA).  The O / lse stores are buffer stores with out-of-range offsets on masked lanes, so that every wave issues the same
// number of them and the next item's "images have landed" wait can leave exactly those in flight.
template <int DH, int Npad>
__global__ void __launch_bounds__(1024) attn_fwd4_kernel(const bf16* __restrict__ qkv, int N, int H, bf16* __restrict__ o,
                                                         float* __restrict__ lse, int nbh, int dbg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int ROW = HeadImg<DH>::kRow, CH = HeadImg<DH>::kChunks, ND = DH / 16, NT = Npad / 16, NW = 16;
  constexpr int img = Npad * ROW;
  constexpr int bufsz = 3 * img;  // Q | K | V (the streamed K fragment of the last row reads on into V: finite bf16)
  constexpr int pieces = Npad * CH / 64, MAXP = (pieces + NW - 1) / NW;
  static_assert(NT <= NW, "fwd4: one query tile per wave");
  const uint32_t lds0 = (uint32_t)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4;
  const int64_t rs = (int64_t)3 * H * DH, os = (int64_t)H * DH;
  const float scale = rsqrtf((float)DH) * 1.44269504088896340736f;  // softmax in base 2
  auto opaque = [](int v) { asm volatile("" : "+v"(v)); return v; };
  auto item_bh = [&](int item) {
    const int xcd = item & 7, q = nbh >> 3, r = nbh & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (item >> 3);
  };
  auto prefetch_images = [&](int item, int buf) {
    const int bh = item_bh(item), b = bh / H, h = bh - b * H;
    const bf16* qb = qkv + (int64_t)b * N * rs + h * DH;
    const int64_t bytes = ((int64_t)(N - 1) * rs + DH) * 2;
    const attn_i32x4 rq = attn_srd(qb, bytes), rk = attn_srd(qb + H * DH, bytes), rv = attn_srd(qb + 2 * H * DH, bytes);
    const uint32_t base = lds0 + buf * bufsz;
    const int ln = opaque(lane);
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      const int p = wave + NW * i;
      if (p < pieces) {
        const int ci = p * 64 + ln, row = ci / CH, slot = ci - row * CH;
        const uint32_t src = (uint32_t)((DH == 48 ? slot : (slot ^ ((row >> 1) & 7))) * 16);
        const uint32_t vq = (uint32_t)row * (uint32_t)(rs * 2) + src;
        attn_dma16(rq, base + p * 1024, vq);
        attn_dma16(rk, base + img + p * 1024, vq);
        attn_dma16(rv, base + 2 * img + p * 1024, vq);
      }
    }
  };
  int item = blockIdx.x;
  if (item >= nbh) return;
  prefetch_images(item, 0);
  const bool has_rows = wave < NT;    // (Npad = 224: waves 14 and 15 own no queries; they still move their share of the images)
  constexpr int kStores = ND + 1;     // buffer stores a query-owning wave issues per item
  for (int it = 0; item < nbh; item += gridDim.x, ++it) {
    const int cur = it & 1;
    const unsigned char* Qimg = smem + cur * bufsz;
    const unsigned char* Kimg = Qimg + img;
    const unsigned char* Vimg = Kimg + img;
    const int bh = item_bh(item), b = bh / H, h = bh - b * H;
    // images landed: the previous item's stores (younger than this item's DMA pieces) may stay in flight
    if (it == 0 || !has_rows) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kStores) : "memory");
    __builtin_amdgcn_s_barrier();
    const int next = item + gridDim.x;
    if (next < nbh && !(dbg & 0x200)) prefetch_images(next, cur ^ 1);  // (0x200, testing: compute on stale images)
    if (has_rows && !(dbg & 0x100)) {  // (0x100, testing: image traffic only)
      const int q0 = wave * 16;
      const RowFrag<DH> qf = rows_lds<DH>(Qimg, q0, lane, true);
      f32x4 st[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) st[t] = mma_rows<DH>(rows_lds<DH>(Kimg, t * 16, lane, false), qf, f32x4{0, 0, 0, 0});  // S^T[key = 16 t + 4 g + r][q = lane & 15]
      float m = -INFINITY;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (t * 16 + 15 >= 192 && t * 16 + 4 * g + r >= N) st[t][r] = -INFINITY;  // (only the tiles that can hold keys >= N: N > 192)
          m = fmaxf(m, st[t][r]);
        }
      m = fmaxf(m, __shfl_xor(m, 16, 64));
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      const float mx = m * scale;
      float ps = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          st[t][r] = __builtin_amdgcn_exp2f(fmaf(st[t][r], scale, -mx));
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
        for (int dt = 0; dt < ND; ++dt) oacc[dt] = MFMA(cols_lds<DH>(Vimg, s2 * 32, s2 * 32 + 16, dt * 16, lane), pb, oacc[dt]);  // O^T[d = 16 dt + 4 g + r][q]
      }
      // stores: O [B, N, H dh] bf16 (4 consecutive d per lane) and lse [B, H, N] in natural-log units; masked lanes out of range
      const int ln = opaque(lane);
      __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(o + (int64_t)b * N * os + h * DH), 0,
                                                                    (uint32_t)(((int64_t)(N - 1) * os + DH) * 2), 0x00020000);
      __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void*)(lse + (int64_t)bh * N), 0, (uint32_t)(N * 4), 0x00020000);
      const int q = q0 + (ln & 15);
      const float inv = 1.0f / ps;
      const uint32_t voff = q < N ? (uint32_t)(q * (int)os * 2 + (ln >> 4) * 8) : 0xFFFFFFF0u;
#pragma unroll
      for (int dt = 0; dt < ND; ++dt) {
        const f32x4 v = oacc[dt] * inv;
        const bf16x4 ob = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        // d-tile in the scalar offset: it is not part of the range check, so a masked lane's offset cannot wrap back into the buffer
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2a, ob), ro, voff, dt * 32, 0);
      }
      const float l = (mx + __builtin_amdgcn_logf(ps)) * 0.69314718055994530942f;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, l), rl, (q < N && (ln >> 4) == 0) ? (uint32_t)(q * 4) : 0xFFFFFFF0u, 0, 0);
    }
  }
}

template <int DH>
constexpr size_t fwd4_lds(int Npad) { return (size_t)2 * (3 * Npad * HeadImg<DH>::kRow); }

template <int DH>
constexpr size_t bwd4_lds(int Npad) { return (size_t)2 * (3 * Npad * HeadImg<DH>::kRow + 32 + 2 * Npad * 4) + (size_t)2 * (2 * Npad * 32) + 1024; }  // + stamps (testing)

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
  hipLaunchKernelGGL((attn_fwd_row_kernel<DH, NT>), dim3(B * H), dim3(NT <= 4 ? 256 : 512), lds, s, (const bf16*)qkv, N, H, (bf16*)o, lse);
  return check_hip(hipGetLastError(), "attention_fwd_row");
}

int g_attn_row = 1;  // testing hook: 0 = always the online-softmax kernel

static int num_cus_cached() {  // (hipGetDeviceProperties costs tens of microseconds of host time per call)
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  return n;
}

int attention_fwd_mfma(const void* qkv, int B, int N, int H, int dh, void* o, float* lse, hipStream_t s) {
  const int Npad = npad_of(N);
  const size_t lds = fwd_lds(N);
  if (g_attn_row && (g_attn_bwd3 & 8) && dh == 48 && Npad == 224) {  // persistent prefetching kernel for the decoder shape
    constexpr size_t l4 = fwd4_lds<48>(224);
    static_assert(l4 <= (size_t)kMaxLds, "fwd4 LDS");
    const int nbh = B * H, ncu = num_cus_cached(), grid = nbh < ncu ? nbh : ncu;
    if (int rc = set_lds(attn_fwd4_kernel<48, 224>, l4)) return rc;
    hipLaunchKernelGGL((attn_fwd4_kernel<48, 224>), dim3(grid), dim3(1024), l4, s, (const bf16*)qkv, N, H, (bf16*)o, lse, nbh, g_attn_dbg & 0xF00);
    return check_hip(hipGetLastError(), "attention_fwd4");
  }
  if (g_attn_row) {  // full-row kernels (<= 64, <= 160, <= 224, <= 288, <= 544 keys)
    const int nt = Npad / 16;
#define HCT_ROW(DH_, NT_) if (dh == DH_ && nt <= NT_) return launch_fwd_row<DH_, NT_>(qkv, B, N, H, o, lse, s)  /* ascending NT: the kernel's kPrevNT relies on it */
    HCT_ROW(64, 4); HCT_ROW(48, 4); HCT_ROW(64, 10); HCT_ROW(48, 10); HCT_ROW(64, 14); HCT_ROW(48, 14); HCT_ROW(64, 18); HCT_ROW(48, 18);
    HCT_ROW(64, 34); HCT_ROW(48, 34);  // 289 .. 544 keys (DINO: 517 tokens, ViT-L decoder: 513): 139 KB of K / V images, one workgroup per CU, 256-register waves
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
  const int ncu = num_cus_cached();
  if (!(g_attn_dbg & (4 | 8 | 32)) && (g_attn_bwd3 & 1024) && dh == 64 && Npad == 160) {
    // (bit 10, default) the persistent key-owner kernel for the ViT-L encoder's 129 tokens at head dim 64: 16 waves x one key tile,
    // 79.6 us against 87.0 for the two-phase kernel at B * H = 1 536 (scripts/dbg/attn_enc_vitl.py)
    constexpr size_t l4 = bwd4_lds<64>(160);
    static_assert(l4 <= (size_t)kMaxLds, "bwd4 LDS (64, 160)");
    const int nbh = B * H, grid = nbh < ncu ? nbh : ncu;
    if (int rc = set_lds(attn_bwd4_kernel<64, 160, 1>, l4)) return rc;
    hipLaunchKernelGGL((attn_bwd4_kernel<64, 160, 1>), dim3(grid), dim3(1024), l4, s, (const bf16*)qkv, (const bf16*)o, (const bf16*)d_o, lse, N, H, (bf16*)dqkv, nbh, g_attn_dbg & 0xF80);
    return check_hip(hipGetLastError(), "attention_bwd4(64,160)");
  }
  if (!(g_attn_dbg & (4 | 8 | 32)) && (g_attn_bwd3 & 4) && dh == 48 && Npad == 224) {
    // persistent prefetching key-owner kernel, one workgroup per CU: 8 waves x 2 key tiles, 193 .. 224 tokens (the MAE decoder: 217)
    constexpr size_t l4 = bwd4_lds<48>(224);
    static_assert(l4 <= (size_t)kMaxLds, "bwd4 LDS");
    const int nbh = B * H, grid = nbh < ncu ? nbh : ncu;
    if (g_attn_bwd3 & 16) {  // 16 waves x one key tile, four waves per SIMD (default)
      if (int rc = set_lds(attn_bwd4_kernel<48, 224, 1>, l4)) return rc;
      hipLaunchKernelGGL((attn_bwd4_kernel<48, 224, 1>), dim3(grid), dim3(1024), l4, s, (const bf16*)qkv, (const bf16*)o, (const bf16*)d_o, lse, N, H, (bf16*)dqkv, nbh, g_attn_dbg & 0xF80);
      return check_hip(hipGetLastError(), "attention_bwd4");
    }
    if (int rc = set_lds(attn_bwd4_kernel<48, 224>, l4)) return rc;
    hipLaunchKernelGGL((attn_bwd4_kernel<48, 224>), dim3(grid), dim3(512), l4, s, (const bf16*)qkv, (const bf16*)o, (const bf16*)d_o, lse, N, H, (bf16*)dqkv, nbh, g_attn_dbg & 0xF80);
    return check_hip(hipGetLastError(), "attention_bwd4");
  }
  if (!(g_attn_dbg & (4 | 8 | 32))) {  // (testing hooks 4 / 8 / 32 select the older kernels)  five-product key-owner kernel for the sequence lengths whose Q / dO / K images leave room for two workgroups per CU
#define HCT_BWD3(DH_, GS_, KT_, WPS_)                                                                                        \
  do {                                                                                                                 \
    const size_t l3 = bwd3_lds<DH_>(Npad);                                                                             \
    if (int rc = set_lds(attn_bwd3_kernel<DH_, GS_, KT_, WPS_>, l3)) return rc;                                              \
    hipLaunchKernelGGL((attn_bwd3_kernel<DH_, GS_, KT_, WPS_>), dim3(B * H), dim3(GS_ * 64), l3, s, (const bf16*)qkv,        \
                       (const bf16*)o, (const bf16*)d_o, lse, N, H, Npad, (bf16*)dqkv, g_attn_dbg & 0xF80, ncu,    \
                       (g_attn_dbg >> 12) ? (g_attn_dbg >> 12) - 1 : (int)(Npad * Npad / 4096));                      \
    return check_hip(hipGetLastError(), "attention_bwd3");                                                             \
  } while (0)
    if (dh == 48 && Npad <= 256 && (g_attn_bwd3 & 1)) { if (g_attn_dbg & 64) HCT_BWD3(48, 4, 4, 1); else HCT_BWD3(48, 4, 4, 2); }  // 64: one wave per SIMD (testing)
    if (dh == 64 && Npad <= 64 && (g_attn_bwd3 & 2) && (g_attn_bwd3 & 32)) HCT_BWD3(64, 4, 1, 4);  // four waves x one key tile per head, 128 registers (default)
    if (dh == 64 && Npad <= 64 && (g_attn_bwd3 & 2)) HCT_BWD3(64, 2, 2, 2);
    // 65 .. 192 tokens at head dim 64 (ViT-L encoder: 129): the two-phase kernel is faster (58.9 us against 84.8 for this instance
    // and 83.6 for twelve waves x one key tile), so bwd3 takes them only when forced onto every shape it covers (bits 0 and 1)
    if (dh == 64 && Npad <= 192 && (g_attn_bwd3 & 3) == 3) HCT_BWD3(64, 4, 3, 2);
#undef HCT_BWD3
  }
  // (every wave runs its nine tile slots whatever the length: below ~450 tokens the spare slots make it slower than the two-phase
  //  kernel -- 276 vs 239 us at 289 tokens, 363 vs 356 at 385, 507 vs 580 at 513, 502 vs 600 at 576; bit 9 forces it from 225 tokens on)
  if (!(g_attn_dbg & (4 | 8 | 32)) && (g_attn_bwd3 & 128) && Npad > ((g_attn_bwd3 & 512) ? 224 : 448) && Npad <= 576) {
    // long sequences (ViT-L decoder: 513 tokens at head dim 48; DINO: 517 at 64): five-product key-owner kernel, one wave per SIMD
#define HCT_BWD5(DH_, KT_)                                                                                                       \
  do {                                                                                                                           \
    const size_t l5 = bwd5_lds<DH_>(Npad, KT_);                                                                                      \
    if (int rc = set_lds(attn_bwd5_kernel<DH_, KT_>, l5)) return rc;                                                             \
    hipLaunchKernelGGL((attn_bwd5_kernel<DH_, KT_>), dim3(B * H), dim3(256), l5, s, (const bf16*)qkv, (const bf16*)o,            \
                       (const bf16*)d_o, lse, N, H, Npad, (bf16*)dqkv);                                                          \
    return check_hip(hipGetLastError(), "attention_bwd5");                                                                       \
  } while (0)
    if (dh == 48) HCT_BWD5(48, 9);
    if (dh == 64 && (g_attn_bwd3 & 256)) HCT_BWD5(64, 9);  // (opt-in: 288 accumulator registers + 72 of V rows spill -- also with the block's column fragments re-read per tile --, 914 us against the two-phase kernel's 605 on DINO's 517 tokens)
#undef HCT_BWD5
  }
  if (!(g_attn_dbg & 4) || bwd_lds(N) > (size_t)kMaxLds) {  // single-phase variant (testing hook) only where its 4 images fit
    const size_t l2 = bwd2_lds(N);
#define HCT_BWD2(DH_, NW_)                                                                                           \
  do {                                                                                                               \
    if (int rc = set_lds(attn_bwd2_mfma_kernel<DH_, NW_>, l2)) return rc;                                            \
    hipLaunchKernelGGL((attn_bwd2_mfma_kernel<DH_, NW_>), dim3(B * H), dim3(NW_ * 64), l2, s, (const bf16*)qkv,       \
                       (const bf16*)o, (const bf16*)d_o, lse, N, H, Npad, (bf16*)dqkv);                              \
  } while (0)
    const bool w8 = (g_attn_dbg & 8) ? false : N > 64;  // 8 waves (16 per CU) once there are enough tiles to share
    // beyond ~80 KB of images only one workgroup fits a CU: 16 waves (four per SIMD at the same 128 registers) instead of 8
    const bool w16 = w8 && l2 > (size_t)81408 && !(g_attn_bwd3 & 64);
    if (dh == 48) { if (w16) HCT_BWD2(48, 16); else if (w8) HCT_BWD2(48, 8); else HCT_BWD2(48, 4); }
    else { if (w16) HCT_BWD2(64, 16); else if (w8) HCT_BWD2(64, 8); else HCT_BWD2(64, 4); }
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
