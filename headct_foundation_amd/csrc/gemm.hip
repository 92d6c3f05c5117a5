// GEMM with fused epilogue for the MAE step (MFMA-bound: ~97 % of the step's FLOPs).
//
//   gemm_bf16_nt_kernel : C[M,N] = A[M,K] . B[N,K]^T    bf16 operands, v_mfma_f32_16x16x32_bf16, fp32 accumulate.
//                         forward Linears (x.W^T) and dgrad (dY.(W^T)^T with the transposed bf16 weight copy).
//   gemm_bf16_tn_kernel : C[M,N] = A[R,M]^T . B[R,N]    wgrad (dW = dY^T.X), reduction R = token rows, operands read
//                         from row-major LDS tiles with ds_read_b64_tr_b16; split over R with fp32 partial slabs
//                         and a fixed-order fold (deterministic, no atomics).
//   gemm_generic_kernel : any strides / dtypes, fp32 FMA accumulation (the fp32 parity mode and odd shapes).
//
// Tuned kernels: 128x128 output tile, 64-deep reduction step, 4 waves (each 64x64 = 4x4 MFMA tiles), operands
// staged HBM -> LDS with buffer_load_dwordx4 ... lds (16 B per lane, zero-fill past the matrix end through the
// buffer descriptor's num_records), two LDS buffers, XOR-swizzled so every fragment read is bank-conflict-free,
// XCD-aware tile order (tiles that share an A row-panel run back-to-back on one XCD's L2).
#include "common.h"
#include "prof.h"

#include <cstdlib>
#include <cstring>
#include <vector>

namespace hct {

static int g_w4_auto = 0;   // auto-dispatch of the 2-WG/CU variant: faster in isolation on the decoder's GELU / +residual
                            // GEMMs (362 vs 395 us, 126 vs 142 us) but 1 % slower inside the step -> off
static int g_stagger = -1;  // -1 auto, >= 0 forced (testing)
static int g_tn_separate_fold = 1;  // 1 = split partials folded by gemm_fold_kernel; 0 = inside the wgrad launch (measured 0.36 ms per step SLOWER:
                                     // DESIGN.md section 5; kept selectable and tested, -6 / -7 of hct_debug_set_gemm_variant)
static int g_nt_variant = 0;  // 0 auto; 128 / 256 / 4 force one NT kernel (tests cover every instance)
static int g_w4_small = 0;    // testing (-10 / -11): the two-workgroups-per-CU variant for single-round shapes (tiles < CUs < 2 x tiles)
static int g_mt3 = 1;         // (-14 / -15: on / off) 192-row tiles for single-round plain / +residual shapes
static int g_even_rounds = 1; // (-12 / -13: on / off): whole-tile NT launches on ceil(tiles / rounds) workgroups instead of all CUs
static int g_sk_drop = 0;     // testing (hct_debug_set_gemm_variant(-8 / -9)): stream-K followers publish a wrong sequence number -> every owner times out

struct Epilogue {
  const float* bias;
  const float* residual;
  int64_t ldr;
  int act;
  void* aux; int aux_dtype; int64_t ldaux;
  void* C; int c_dtype; int64_t ldc;
  void* C2; int c2_dtype; int64_t ldc2;
  float alpha;
  float* colsum_partial;  // [ceil(M/256)*4][N] per-(tile-row, wave-row) column sums of the output (DGELU mode), or null
  int aux_deriv;          // HCT_ACT_GELU_D / HCT_ACT_MULAUX: aux holds gelu'(pre-activation) instead of the pre-activation
};

__device__ __forceinline__ void store4(void* base, int dtype, int64_t off, f32x4 v) {
  if (dtype == HCT_BF16) Vec4<bf16>::store((bf16*)base + off, v);
  else Vec4<float>::store((float*)base + off, v);
}
__device__ __forceinline__ f32x4 load4(const void* base, int dtype, int64_t off) {
  return dtype == HCT_BF16 ? Vec4<bf16>::load((const bf16*)base + off) : Vec4<float>::load((const float*)base + off);
}
__device__ __forceinline__ void store1(void* base, int dtype, int64_t off, float v) {
  if (dtype == HCT_BF16) ((bf16*)base)[off] = (bf16)v; else ((float*)base)[off] = v;
}
__device__ __forceinline__ float load1(const void* base, int dtype, int64_t off) {
  return dtype == HCT_BF16 ? (float)((const bf16*)base)[off] : ((const float*)base)[off];
}

// 4 consecutive columns n..n+3 of row m (all leading dims and n multiples of 4)
__device__ __forceinline__ void epilogue4(const Epilogue& e, int m, int n, f32x4 acc) {
  f32x4 v = acc * e.alpha;
  if (e.bias) v += Vec4<float>::load(e.bias + n);
  if (e.act == HCT_ACT_GELU) {
    if (e.aux) {
      f32x4 sv = v;
      if (e.aux_deriv) {
#pragma unroll
        for (int i = 0; i < 4; ++i) sv[i] = dgelu_erf(v[i]);
      }
      store4(e.aux, e.aux_dtype, (int64_t)m * e.ldaux + n, sv);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = gelu_erf(v[i]);
  } else if (e.act == HCT_ACT_DGELU) {
    const f32x4 u = load4(e.aux, e.aux_dtype, (int64_t)m * e.ldaux + n);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] *= e.aux_deriv ? u[i] : dgelu_erf(u[i]);
  }
  if (e.residual) v += Vec4<float>::load(e.residual + (int64_t)m * e.ldr + n);
  store4(e.C, e.c_dtype, (int64_t)m * e.ldc + n, v);
  if (e.C2) store4(e.C2, e.c2_dtype, (int64_t)m * e.ldc2 + n, v);
}
// NJ vectors of one output row m at columns n0 + 16*j (the MFMA accumulator row of a wave): all loads first, then
// math + stores, so the residual / aux reads of a tile are not serialised behind its stores.
template <int NJ>
__device__ __forceinline__ void epilogue_row(const Epilogue& e, int m, int n0, int N, const f32x4* acc) {
  f32x4 r[NJ], u[NJ], bv[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int n = n0 + 16 * j;
    const bool ok = n < N;
    r[j] = (e.residual && ok) ? Vec4<float>::load(e.residual + (int64_t)m * e.ldr + n) : f32x4{0, 0, 0, 0};
    u[j] = (e.act == HCT_ACT_DGELU && ok) ? load4(e.aux, e.aux_dtype, (int64_t)m * e.ldaux + n) : f32x4{0, 0, 0, 0};
    bv[j] = (e.bias && ok) ? Vec4<float>::load(e.bias + n) : f32x4{0, 0, 0, 0};
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int n = n0 + 16 * j;
    if (n >= N) continue;
    f32x4 v = acc[j] * e.alpha + bv[j];
    if (e.act == HCT_ACT_GELU) {
      if (e.aux) {
      f32x4 sv = v;
      if (e.aux_deriv) {
#pragma unroll
        for (int i = 0; i < 4; ++i) sv[i] = dgelu_erf(v[i]);
      }
      store4(e.aux, e.aux_dtype, (int64_t)m * e.ldaux + n, sv);
    }
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = gelu_erf(v[i]);
    } else if (e.act == HCT_ACT_DGELU) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] *= e.aux_deriv ? u[j][i] : dgelu_erf(u[j][i]);
    }
    v += r[j];
    store4(e.C, e.c_dtype, (int64_t)m * e.ldc + n, v);
    if (e.C2) store4(e.C2, e.c2_dtype, (int64_t)m * e.ldc2 + n, v);
  }
}

// Coalesced tile epilogue.  A wave's accumulator row-tile i holds C[16 rows][64 cols] as acc[j][e] =
// C[row = lane&15][col = 16*j + 4*(lane>>4) + e].  Written as-is that is 16 rows x 32/64-B pieces per store instruction
// (store-issue-bound: ~24 us per 256x256 tile).  Instead each wave bounces the row-tile through a private 16 x 64 fp32
// LDS patch (272-B rows) and re-reads it with 16 lanes per row, so every global load/store instruction of the epilogue
// touches 4 rows x 256 contiguous bytes (fp32) / 128 bytes (bf16).
constexpr int kStageRow = 272;
constexpr int kStageBytes = 16 * kStageRow;

__device__ __forceinline__ void epilogue_tile16x64(const Epilogue& e, unsigned char* patch, int lane, int m_base, int n_base,
                                                   int M, int N, const f32x4* acc) {
  const int frow = lane & 15, fchk = lane >> 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(patch + frow * kStageRow + j * 64 + fchk * 16) = acc[j];
  const int rr = lane >> 4, cc = lane & 15;
  const int n = n_base + cc * 4;
  f32x4 v[4], r[4], u[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) v[it] = *reinterpret_cast<const f32x4*>(patch + (it * 4 + rr) * kStageRow + cc * 16);
  const bool nok = n < N;
  f32x4 bv = (e.bias && nok) ? Vec4<float>::load(e.bias + n) : f32x4{0, 0, 0, 0};
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int m = m_base + it * 4 + rr;
    const bool ok = nok && m < M;
    r[it] = (e.residual && ok) ? Vec4<float>::load(e.residual + (int64_t)m * e.ldr + n) : f32x4{0, 0, 0, 0};
    u[it] = (e.act == HCT_ACT_DGELU && ok) ? load4(e.aux, e.aux_dtype, (int64_t)m * e.ldaux + n) : f32x4{0, 0, 0, 0};
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int m = m_base + it * 4 + rr;
    if (!(nok && m < M)) continue;
    f32x4 x = v[it] * e.alpha + bv;
    if (e.act == HCT_ACT_GELU) {
      if (e.aux) {
        f32x4 sv = x;
        if (e.aux_deriv) {
#pragma unroll
          for (int i = 0; i < 4; ++i) sv[i] = dgelu_erf(x[i]);
        }
        store4(e.aux, e.aux_dtype, (int64_t)m * e.ldaux + n, sv);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) x[i] = gelu_erf(x[i]);
    } else if (e.act == HCT_ACT_DGELU) {
#pragma unroll
      for (int i = 0; i < 4; ++i) x[i] *= e.aux_deriv ? u[it][i] : dgelu_erf(u[it][i]);
    }
    x += r[it];
    store4(e.C, e.c_dtype, (int64_t)m * e.ldc + n, x);
    if (e.C2) store4(e.C2, e.c2_dtype, (int64_t)m * e.ldc2 + n, x);
  }
}

__device__ __forceinline__ void epilogue1(const Epilogue& e, int m, int n, float acc) {
  float v = acc * e.alpha;
  if (e.bias) v += e.bias[n];
  if (e.act == HCT_ACT_GELU) {
    if (e.aux) store1(e.aux, e.aux_dtype, (int64_t)m * e.ldaux + n, e.aux_deriv ? dgelu_erf(v) : v);
    v = gelu_erf(v);
  } else if (e.act == HCT_ACT_DGELU) {
    {
      const float u = load1(e.aux, e.aux_dtype, (int64_t)m * e.ldaux + n);
      v *= e.aux_deriv ? u : dgelu_erf(u);
    }
  }
  if (e.residual) v += e.residual[(int64_t)m * e.ldr + n];
  store1(e.C, e.c_dtype, (int64_t)m * e.ldc + n, v);
  if (e.C2) store1(e.C2, e.c2_dtype, (int64_t)m * e.ldc2 + n, v);
}

// =============================================================================================
// generic strided kernel: 64x64 tile, 16-deep, 256 threads x (4x4) outputs, fp32 FMA in k order
// =============================================================================================
template <typename TA, typename TB>
__global__ void __launch_bounds__(256) gemm_generic_kernel(int M, int N, int K, const TA* __restrict__ A, int64_t sam,
                                                           int64_t sak, const TB* __restrict__ B, int64_t sbk, int64_t sbn,
                                                           Epilogue e, int vec_ok) {
  __shared__ float As[16][64 + 4];
  __shared__ float Bs[16][64 + 4];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  float acc[4][4] = {};
  for (int k0 = 0; k0 < K; k0 += 16) {
    for (int i = threadIdx.x; i < 64 * 16; i += 256) {
      int mm, kk;
      if (sak == 1) { kk = i & 15; mm = i >> 4; } else { mm = i & 63; kk = i >> 6; }
      const int gm = m0 + mm, gk = k0 + kk;
      As[kk][mm] = (gm < M && gk < K) ? to_f32(A[(int64_t)gm * sam + (int64_t)gk * sak]) : 0.f;
    }
    for (int i = threadIdx.x; i < 64 * 16; i += 256) {
      int nn, kk;
      if (sbk == 1) { kk = i & 15; nn = i >> 4; } else { nn = i & 63; kk = i >> 6; }
      const int gn = n0 + nn, gk = k0 + kk;
      Bs[kk][nn] = (gn < N && gk < K) ? to_f32(B[(int64_t)gk * sbk + (int64_t)gn * sbn]) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs[kk][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i, n = n0 + tx * 4;
    if (m >= M || n >= N) continue;
    if (vec_ok && n + 4 <= N) {
      epilogue4(e, m, n, f32x4{acc[i][0], acc[i][1], acc[i][2], acc[i][3]});
    } else {
      for (int j = 0; j < 4 && n + j < N; ++j) epilogue1(e, m, n + j, acc[i][j]);
    }
  }
}

// =============================================================================================
// tuned bf16 kernels
// =============================================================================================
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void tile_ids(int ntm, int ntn, int& tm, int& tn) {
  // XCD-aware, bijective remap: blocks are dealt round-robin over 8 XCDs, so block b and b+8 share an L2.
  // Give each XCD a contiguous run of tile ids; consecutive ids walk tn first (same A row-panel).
  const int nwg = ntm * ntn;
  const int b = blockIdx.x;
  const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
  const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  tm = id / ntn;
  tn = id - tm * ntn;
}

__device__ __forceinline__ uint32_t clamp_records(int64_t bytes) {
  return bytes > 0xFFFFFFFFll ? 0xFFFFFFFFu : (bytes < 0 ? 0u : (uint32_t)bytes);
}

// ---- NT: A [M,K] (lda), B [N,K] (ldb), K % 64 == 0 -------------------------------------------------------------
__global__ void __launch_bounds__(256, 2) gemm_bf16_nt_kernel(int M, int N, int K, const bf16* __restrict__ A, int64_t lda,
                                                              const bf16* __restrict__ B, int64_t ldb, Epilogue e) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[65536];  // [buf][A 16K | B 16K]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ntm = (M + 127) >> 7, ntn = (N + 127) >> 7;
  int tm, tn;
  tile_ids(ntm, ntn, tm, tn);
  const int m0 = tm << 7, n0 = tn << 7;

  const bf16* Ab = A + (int64_t)m0 * lda;
  const bf16* Bb = B + (int64_t)n0 * ldb;
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, clamp_records(((int64_t)(M - m0 - 1) * lda + K) * 2), 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, clamp_records(((int64_t)(N - n0 - 1) * ldb + K) * 2), 0x00020000);

  // staging map: wave w issues chunks c = 4w..4w+3 (1 KiB = 8 rows x 128 B each) of both tiles.
  // lane -> (row = 8c + lane/8, LDS 16-B slot = lane%8); the slot holds logical k-chunk  slot ^ ((row>>1)&7).
  uint32_t voa[4], vob[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = wave * 4 + i;
    const int row = c * 8 + (lane >> 3);
    const int src_chunk = (lane & 7) ^ ((row >> 1) & 7);
    voa[i] = (uint32_t)(row * lda * 2 + src_chunk * 16);
    vob[i] = (uint32_t)(row * ldb * 2 + src_chunk * 16);
  }
  auto stage = [&](int buf, int k0) {
    unsigned char* base = smem + buf * 32768;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = wave * 4 + i;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(base + c * 1024), 16, voa[i] + k0 * 2, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void*)(base + 16384 + c * 1024), 16, vob[i] + k0 * 2, 0, 0, 0);
    }
  };

  const int wr = wave >> 1, wc = wave & 1;
  const int frow = lane & 15, fchk = lane >> 4, swz = (lane >> 1) & 7;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

  const int nk = K >> 6;
  stage(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) stage(buf ^ 1, (kt + 1) << 6);
    const unsigned char* sa = smem + buf * 32768 + (wr * 64 + frow) * 128;
    const unsigned char* sb = smem + buf * 32768 + 16384 + (wc * 64 + frow) * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int off = ((ks * 4 + fchk) ^ swz) << 4;
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sa + i * 2048 + off);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(sb + j * 2048 + off);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  // acc[i][j][e] = C[m0 + wr*64 + i*16 + (lane&15)][n0 + wc*64 + j*16 + (lane>>4)*4 + e]
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    epilogue_tile16x64(e, smem + wave * kStageBytes, lane, m0 + wr * 64 + i * 16, n0 + wc * 64, M, N, acc[i]);
  }
}

// ---- NT, large problems: 256x256 tile, 32-deep stages, 4-stage LDS ring, one barrier per stage -----------------
// 8 waves = 2(M) x 4(N), each 128x64 (8x4 MFMA tiles, 128 accumulator VGPRs).  Three stages (96 KiB) stay in flight
// behind a COUNTED s_waitcnt vmcnt(8) + raw s_barrier, so HBM/L2 latency is covered by ~3 stages of MFMA work instead
// of one (the 128x128 kernel above drains vmcnt(0) every stage).  LDS stage = A[256][32] | B[256][32] bf16, 64-B rows;
// 16-B chunk c of row r sits at chunk  c ^ F[(r>>2)&3],  F = {0,3,2,1}  (conflict-free ds_read_b128 fragments).
__device__ __forceinline__ int swz64(int r) { return (0x1230 >> (((r >> 2) & 3) * 4)) & 3; }  // F[(r>>2)&3]

// XOR-swizzled 16 x 64 fp32 patch (256-B rows, exactly 4 KiB per wave -> 8 waves fit one 32-KiB ring buffer)
__device__ __forceinline__ void epilogue_tile16x64_swz(const Epilogue& e, unsigned char* patch, int lane, int m_base, int n_base,
                                                       int M, int N, const f32x4* acc) {
  const int frow = lane & 15, fchk = lane >> 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(patch + frow * 256 + (((j * 4 + fchk) ^ frow) << 4)) = acc[j];
  const int rr = lane >> 4, cc = lane & 15;
  const int n = n_base + cc * 4;
  f32x4 v[4], r[4], u[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = it * 4 + rr;
    v[it] = *reinterpret_cast<const f32x4*>(patch + row * 256 + ((cc ^ row) << 4));
  }
  const bool nok = n < N;
  f32x4 bv = (e.bias && nok) ? Vec4<float>::load(e.bias + n) : f32x4{0, 0, 0, 0};
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int m = m_base + it * 4 + rr;
    const bool ok = nok && m < M;
    r[it] = (e.residual && ok) ? Vec4<float>::load(e.residual + (int64_t)m * e.ldr + n) : f32x4{0, 0, 0, 0};
    u[it] = (e.act == HCT_ACT_DGELU && ok) ? load4(e.aux, e.aux_dtype, (int64_t)m * e.ldaux + n) : f32x4{0, 0, 0, 0};
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int m = m_base + it * 4 + rr;
    if (!(nok && m < M)) continue;
    f32x4 x = v[it] * e.alpha + bv;
    if (e.act == HCT_ACT_GELU) {
      if (e.aux) {
        f32x4 sv = x;
        if (e.aux_deriv) {
#pragma unroll
          for (int i = 0; i < 4; ++i) sv[i] = dgelu_erf(x[i]);
        }
        store4(e.aux, e.aux_dtype, (int64_t)m * e.ldaux + n, sv);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) x[i] = gelu_erf(x[i]);
    } else if (e.act == HCT_ACT_DGELU) {
#pragma unroll
      for (int i = 0; i < 4; ++i) x[i] *= e.aux_deriv ? u[it][i] : dgelu_erf(u[it][i]);
    }
    x += r[it];
    store4(e.C, e.c_dtype, (int64_t)m * e.ldc + n, x);
    if (e.C2) store4(e.C2, e.c2_dtype, (int64_t)m * e.ldc2 + n, x);
  }
}

// 16 x 128 fp32 patch (512-B rows, 8 KiB per wave; chunk c of row r at c ^ (r&7)): a store instruction then covers
// 2 rows x 512 B (fp32) / 256 B (bf16) -- row pieces of >= 256 B, which HBM takes ~3x faster than 128-B pieces
// (measured: 256 MB of bf16 output, 164 us as 128-B pieces vs 55 us dense).
__device__ __forceinline__ void epilogue_tile16x128(const Epilogue& e, unsigned char* patch, int lane, int m_base, int n_base,
                                                    int M, int N, const f32x4* acc) {
  const int frow = lane & 15, fchk = lane >> 4;
#pragma unroll
  for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4*>(patch + frow * 512 + (((j * 4 + fchk) ^ (frow & 7)) << 4)) = acc[j];
  const int rr = lane >> 5, cc = lane & 31;
  const int n = n_base + cc * 4;
  const bool nok = n < N;
  const f32x4 bv = (e.bias && nok) ? Vec4<float>::load(e.bias + n) : f32x4{0, 0, 0, 0};
#pragma unroll
  for (int h = 0; h < 2; ++h) {  // two batches of 4 instructions (8 rows): loads first, then math + stores
    f32x4 v[4], r[4], u[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = (h * 4 + it) * 2 + rr;
      v[it] = *reinterpret_cast<const f32x4*>(patch + row * 512 + ((cc ^ (row & 7)) << 4));
      const int m = m_base + row;
      const bool ok = nok && m < M;
      r[it] = (e.residual && ok) ? Vec4<float>::load(e.residual + (int64_t)m * e.ldr + n) : f32x4{0, 0, 0, 0};
      u[it] = (e.act == HCT_ACT_DGELU && ok) ? load4(e.aux, e.aux_dtype, (int64_t)m * e.ldaux + n) : f32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int m = m_base + (h * 4 + it) * 2 + rr;
      if (!(nok && m < M)) continue;
      f32x4 x = v[it] * e.alpha + bv;
      if (e.act == HCT_ACT_GELU) {
        if (e.aux) {
        f32x4 sv = x;
        if (e.aux_deriv) {
#pragma unroll
          for (int i = 0; i < 4; ++i) sv[i] = dgelu_erf(x[i]);
        }
        store4(e.aux, e.aux_dtype, (int64_t)m * e.ldaux + n, sv);
      }
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = gelu_erf(x[i]);
      } else if (e.act == HCT_ACT_DGELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] *= e.aux_deriv ? u[it][i] : dgelu_erf(u[it][i]);
      }
      x += r[it];
      store4(e.C, e.c_dtype, (int64_t)m * e.ldc + n, x);
      if (e.C2) store4(e.C2, e.c2_dtype, (int64_t)m * e.ldc2 + n, x);
    }
  }
}

// Mode-specialised tile epilogue (no per-store branches, per-tile scalar bases + 32-bit lane offsets).
//   EPI_PLAIN_BF16 : C(bf16) = alpha*acc (+bias)                       forward qkv / pred / embeds, dgrad
//   EPI_RES_F32    : C(f32)  = alpha*acc (+bias) + residual            proj / linear2 forward (residual stream)
//   EPI_GELU_BF16  : aux(bf16) = alpha*acc + bias ; C(bf16) = gelu(aux)  linear1 forward
//   EPI_DGELU_BF16 : C(bf16) = alpha*acc * gelu'(aux(bf16))             linear2 dgrad
//   EPI_GENERIC    : everything else (runtime flags)
enum { EPI_GENERIC = 0, EPI_PLAIN_BF16 = 1, EPI_RES_F32 = 2, EPI_GELU_BF16 = 3, EPI_DGELU_BF16 = 4, EPI_PLAIN_F32 = 5,
       EPI_DGELU_CS = 6 /* DGELU + fused column sums of the output (own instance: costs registers in the epilogue) */ };

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

struct TileBufs {  // wave-uniform buffer descriptors rooted at the tile origin (m0, n0)
  __amdgpu_buffer_rsrc_t c, res, aux;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const void* base, int64_t ld, int esz, int m0, int n0, int M, int N) {
  const char* p = (const char*)base + ((int64_t)m0 * ld + n0) * esz;
  const int64_t bytes = base ? ((int64_t)(M - m0 - 1) * ld + (N - n0)) * esz : 0;  // rows >= M fall outside -> dropped / zero
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, clamp_records(bytes), 0x00020000);
}

__device__ __forceinline__ void bstore_bf16x4(__amdgpu_buffer_rsrc_t r, uint32_t off, f32x4 v) {
  bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), r, off, 0, 0);
}
__device__ __forceinline__ void bstore_f32x4(__amdgpu_buffer_rsrc_t r, uint32_t off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, 0);
}
__device__ __forceinline__ f32x4 bload_f32x4(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ f32x4 bload_bf16x4(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  const bf16x4 v = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
  return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}

// All global traffic of the specialised epilogue goes through buffer instructions: SGPR descriptor + one 32-bit lane
// offset, so no 64-bit per-lane addresses exist (those were being spilled around every store, and each reload's
// vmcnt(0) serialised the whole store stream: ~20 us per 256x256 tile).
// Epilogue traffic is streamed once: non-temporal policy (aux = 2) keeps it from displacing the A/B panels that the
// LDS-DMA stream re-reads through L2 (measured +9..16 % on the N >= 2304, K = 768 GEMMs).
#ifndef HCT_EPI_CACHE_POLICY
#define HCT_EPI_CACHE_POLICY 2  /* nt */
#endif
constexpr int kNT = HCT_EPI_CACHE_POLICY;
#ifndef HCT_BF16_OUT_POLICY
#define HCT_BF16_OUT_POLICY HCT_EPI_CACHE_POLICY  /* bf16 outputs (qkv, GELU(u), dgrads): nt 40.15 ms per step, sc1 40.93, write-back 41.15 */
#endif
#ifndef HCT_RES_POLICY
#define HCT_RES_POLICY 16  /* the fp32 residual-stream output, read back by the next LayerNorm: sc1 39.76 / write-back 39.77 / nt 39.84 ms per step; with LayerNorm's non-temporal loads: sc1 39.28 / write-back 39.33 / nt 39.45 */
#endif
#ifndef HCT_EPI_LA_WIDE  /* look-ahead (sub-tiles of 16 rows x 64 columns) of the half-width epilogue's loads: bf16 aux (x gelu') / fp32 residual */
#define HCT_EPI_LA_WIDE 1
#endif
#ifndef HCT_EPI_LA_F32
#define HCT_EPI_LA_F32 1
#endif
#ifndef HCT_GELU_VEC8  /* GELU + gelu' of eight elements with the Horner steps of the two 4-vectors interleaved (common.h); 0 = element by element (A/B) */
#define HCT_GELU_VEC8 1
#endif
#ifndef HCT_SLAB_POLICY
#define HCT_SLAB_POLICY 16  /* cache policy of the wgrad's split-K slab stores: 16 = sc1 write-through (39.83 ms per step), 0 = write-back (39.90), 2 = nt (40.17) */
#endif

// Lane -> output mapping of the specialised epilogue:
//   bf16 outputs ("wide" modes): lane = 4 rows x 16 lanes, 8 consecutive columns (16 B) per lane -> dwordx4 stores / loads.
//     The epilogue is store-ISSUE bound (one wave-instruction moves at most 16 B per lane whatever its width: in-kernel
//     stamps showed ~12 B/clk/CU with dwordx2 stores, independent of the other CUs' phase), so bytes per instruction is
//     what counts.
//   f32 outputs: lane = 2 rows x 32 lanes, 4 consecutive columns (16 B) per lane.
template <int MODE> struct EpiTraits {
  static constexpr bool wide = MODE == EPI_PLAIN_BF16 || MODE == EPI_GELU_BF16 || MODE == EPI_DGELU_BF16 || MODE == EPI_DGELU_CS;
  static constexpr bool loads = MODE == EPI_RES_F32 || MODE == EPI_DGELU_BF16 || MODE == EPI_DGELU_CS;
  static constexpr int batches = wide ? 4 : 8;                    // row batches per 16-row patch
  static constexpr int loads_per_rowtile = loads ? batches : 0;   // one 16-B load per lane and batch
  static constexpr int stores_per_rowtile = MODE == EPI_GELU_BF16 ? 2 * batches : batches;
  static constexpr int bias_ops = wide ? 2 : 1;
  static constexpr int ops_per_tile = 4 * (loads_per_rowtile + stores_per_rowtile);  // vector-memory ops per wave and tile
};

struct TileBias { f32x4 lo, hi; };  // bias of this lane's columns (hi unused by the f32 modes)

// compiler-visible bias load (kernels whose operand stream hipcc can count)
template <int MODE>
__device__ __forceinline__ TileBias tile_bias(const Epilogue& e, int n0, int col0, int lane, int N) {
  TileBias b = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
  if (EpiTraits<MODE>::wide) {
    const int n = n0 + col0 + (lane & 15) * 8;
    if (e.bias && n < N) {
      b.lo = Vec4<float>::load(e.bias + n);
      b.hi = Vec4<float>::load(e.bias + n + 4);
    }
  } else {
    const int n = n0 + col0 + (lane & 31) * 4;
    if (e.bias && n < N) b.lo = Vec4<float>::load(e.bias + n);
  }
  return b;
}

typedef __attribute__((ext_vector_type(4))) int i32x4;

// Epilogue of one wave tile (64 rows x 128 columns = 4 row-tiles of 16), through the wave's 8 KiB LDS patch.
// Loads (residual / pre-activation) are software-pipelined one row-tile ahead: L(i+1) is issued BEFORE the stores of
// row-tile i, so waiting for it never drains those stores (vmcnt retires in issue order; with the loads between the
// stores every batch paid a full store round trip: 17 us per tile for the dgelu epilogue).
// `bias` must already be resident in registers (see the callers: a load that hipcc would have to wait for here would
// be waited for with a count that cannot see the asm LDS-DMA prefetch, i.e. with a full drain).
// gfx950 hazard found with in-kernel data checks (scripts/stress_gemm_epilogue.py): a buffer_store_dwordx4 that takes its
// row offset in an SGPR soffset still reads its data VGPRs for a few cycles after issue.  hipcc's hazard recognizer only
// pads the immediate-soffset form, so a VALU (here: the next batch's v_pk_fma_f32 into the same registers) right behind
// the store corrupted single dwords of single lane quads -- intermittently, 0.01-1 % of the outputs of a large GEMM.
// Four wait states behind every 16-byte store remove it (20/20 clean runs per mode and shape, where 15-20/20 failed).
#define HCT_STORE_GUARD()              \
  do {                                 \
    __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_nop 3");           \
    __builtin_amdgcn_sched_barrier(0); \
  } while (0)

template <int MODE, int STORE_POLICY = kNT>
__device__ __forceinline__ void epilogue_wave64x128_m(const Epilogue& e, const TileBufs& tb, unsigned char* patch, int lane, int m0,
                                                      int n0, int row0, int col0, int M, int N, const f32x4 (*acc)[8], const TileBias& bias,
                                                      f32x4& cs0, f32x4& cs1) {
  typedef EpiTraits<MODE> T;
  // the lane id as an opaque per-call value: every LDS patch address and store offset below is then recomputed here (a few
  // VALU per tile) instead of being hoisted out of the persistent tile loop into ~20 VGPRs that live across the main
  // loop, where the register file is full -- hipcc spilled them and reloaded with s_waitcnt vmcnt(0) at the top of every
  // tile, draining the previous tile's stores and the prefetched stages before the first MFMA
  asm volatile("" : "+v"(lane));
  const int frow = lane & 15, fchk = lane >> 4;
  constexpr int RB = T::wide ? 4 : 2;             // rows per batch
  const int rr = T::wide ? (lane >> 4) : (lane >> 5), cc = T::wide ? (lane & 15) : (lane & 31);
  const int col = col0 + cc * (T::wide ? 8 : 4);
  const bool nok = n0 + col < N;
  const uint32_t OOB = 0xFFFFFFF0u;
  // leading dimensions as opaque per-call scalars: keeps the (tile-invariant) offset arithmetic from being hoisted out
  // of the persistent tile loop into long-lived VGPRs
  int ldc = (int)e.ldc, ldr = (int)e.ldr, ldx = (int)e.ldaux;
  asm volatile("" : "+s"(ldc), "+s"(ldr), "+s"(ldx));
  const int rows_left = M - m0 - row0;
  // lane part of the offsets (elements); the batch's first row goes into the scalar offset
  const uint32_t lane_c = (uint32_t)(rr * ldc + col), lane_r = (uint32_t)(rr * ldr + col), lane_x = (uint32_t)(rr * ldx + col);

  u32x4 ld[2][T::loads ? T::batches : 1];
  // f32 mode: 8 loads of 4 registers per row-tile; the look-ahead is issued in two halves (second half once half of the
  // current row-tile's registers are free) so the double buffer peaks at 48 registers, not 64 (which spilled an accumulator)
  constexpr int kHalf = T::wide ? T::batches : T::batches / 2;
  auto issue_loads = [&](int i, int first, int last) {
    if (!T::loads) return;
#pragma unroll
    for (int it = first; it < last; ++it) {
      const int prow = i * 16 + it * RB;  // first row of the batch inside the wave tile
      const bool ok = nok && prow + rr < rows_left;
      if (MODE == EPI_RES_F32) ld[i & 1][it] = __builtin_amdgcn_raw_buffer_load_b128(tb.res, ok ? lane_r * 4u : OOB, (row0 + prow) * ldr * 4, kNT);
      else ld[i & 1][it] = __builtin_amdgcn_raw_buffer_load_b128(tb.aux, ok ? lane_x * 2u : OOB, (row0 + prow) * ldx * 2, kNT);
    }
  };
  issue_loads(0, 0, T::batches);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4*>(patch + frow * 512 + (((j * 4 + fchk) ^ (frow & 7)) << 4)) = acc[i][j];
    if (i < 3) issue_loads(i + 1, 0, kHalf);
#pragma unroll
    for (int it = 0; it < T::batches; ++it) {
      const int pr = it * RB + rr;        // row inside the 16-row patch
      const int prow = i * 16 + it * RB;  // batch's first row inside the wave tile
      const bool ok = nok && prow + rr < rows_left;
      if (T::wide) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(patch + pr * 512 + (((2 * cc) ^ (pr & 7)) << 4));
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(patch + pr * 512 + (((2 * cc + 1) ^ (pr & 7)) << 4));
        f32x4 x0 = v0 * e.alpha + bias.lo, x1 = v1 * e.alpha + bias.hi;
        const uint32_t vc = ok ? lane_c * 2u : OOB, vx = ok ? lane_x * 2u : OOB;
        const int sc = (row0 + prow) * ldc * 2, sx = (row0 + prow) * ldx * 2;
        auto pack = [](f32x4 a, f32x4 b) {
          bf16x8 o = {(bf16)a[0], (bf16)a[1], (bf16)a[2], (bf16)a[3], (bf16)b[0], (bf16)b[1], (bf16)b[2], (bf16)b[3]};
          return __builtin_bit_cast(u32x4, o);
        };
        if (MODE == EPI_GELU_BF16) {
          if (e.aux_deriv) {  // aux receives gelu'(pre-activation): the backward then multiplies by it (no second evaluation)
            f32x4 d0, d1;
#if HCT_GELU_VEC8
            f32x4 g0, g1;
            gelu_both8(x0, x1, g0, d0, g1, d1);
            x0 = g0; x1 = g1;
#else
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              float ga, da, gb, db;
              gelu_both(x0[q], ga, da);
              gelu_both(x1[q], gb, db);
              x0[q] = ga; d0[q] = da; x1[q] = gb; d1[q] = db;
            }
#endif
            __builtin_amdgcn_raw_buffer_store_b128(pack(d0, d1), tb.aux, vx, sx, kNT);
            HCT_STORE_GUARD();
          } else {
            __builtin_amdgcn_raw_buffer_store_b128(pack(x0, x1), tb.aux, vx, sx, kNT);
            HCT_STORE_GUARD();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              x0[q] = gelu_fast(x0[q]);
              x1[q] = gelu_fast(x1[q]);
            }
          }
        } else if (MODE == EPI_DGELU_BF16 || MODE == EPI_DGELU_CS) {
          const bf16x8 t = __builtin_bit_cast(bf16x8, ld[i & 1][it]);
          if (e.aux_deriv) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              x0[q] *= (float)t[q];
              x1[q] *= (float)t[4 + q];
            }
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              x0[q] *= dgelu_fast((float)t[q]);
              x1[q] *= dgelu_fast((float)t[4 + q]);
            }
          }
          if (MODE == EPI_DGELU_CS) {  // fused bias gradient of the Linear feeding the GELU (column sums of this output)
            cs0 += ok ? x0 : f32x4{0, 0, 0, 0};
            cs1 += ok ? x1 : f32x4{0, 0, 0, 0};
          }
        }
        __builtin_amdgcn_raw_buffer_store_b128(pack(x0, x1), tb.c, vc, sc, HCT_BF16_OUT_POLICY);
        HCT_STORE_GUARD();
      } else {
        const f32x4 v = *reinterpret_cast<const f32x4*>(patch + pr * 512 + ((cc ^ (pr & 7)) << 4));
        f32x4 x = v * e.alpha + bias.lo;
        if (MODE == EPI_RES_F32) x += __builtin_bit_cast(f32x4, ld[i & 1][it]);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x), tb.c, ok ? lane_c * 4u : OOB, (row0 + prow) * ldc * 4, STORE_POLICY);
        HCT_STORE_GUARD();
        if (i < 3 && it == kHalf - 1) issue_loads(i + 1, kHalf, T::batches);
      }
    }
  }
}


// Half-width form of the epilogue above for the persistent 256x256 kernel: the wave tile goes through a 4 KiB patch (16 rows
// x 64 columns of fp32) as 8 sub-tiles (row-tile i, column half h).  The eight patches then fit ONE ring buffer, which leaves
// the other four to the next tile's first two stage pairs: those are issued before this epilogue, the next main loop starts
// as soon as they have landed, and the output stores drain under its first K-steps (see the kernel).
//   bf16 outputs: lane = 8 rows x 8 lanes, 8 consecutive columns (16 B) per lane: one 128-B line per row and instruction;
//   f32 outputs : lane = 4 rows x 16 lanes, 4 consecutive columns (16 B) per lane: 256 B per row.
// Same number of vector-memory operations per tile as the full-width form (EpiTraits<MODE>::ops_per_tile).
template <int MODE>
__device__ __forceinline__ void tile_bias_halves(const Epilogue& e, int n0, int col0, int lane, int N, TileBias (&b)[2]) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    b[h] = TileBias{f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    if (EpiTraits<MODE>::wide) {
      const int n = n0 + col0 + h * 64 + (lane & 7) * 8;
      if (e.bias && n < N) {
        b[h].lo = Vec4<float>::load(e.bias + n);
        b[h].hi = Vec4<float>::load(e.bias + n + 4);
      }
    } else {
      const int n = n0 + col0 + h * 64 + (lane & 15) * 4;
      if (e.bias && n < N) b[h].lo = Vec4<float>::load(e.bias + n);
    }
  }
}

template <int MODE, int STORE_POLICY = kNT, int MT = 4>
__device__ __forceinline__ void epilogue_wave64x128_h(const Epilogue& e, const TileBufs& tb, unsigned char* patch, int lane, int m0,
                                                      int n0, int row0, int col0, int M, int N, const f32x4 (*acc)[8],
                                                      const TileBias (&bias)[2], f32x4 (&cs)[2][2]) {
  typedef EpiTraits<MODE> T;
  asm volatile("" : "+v"(lane));  // opaque per call: see epilogue_wave64x128_m
  const int frow = lane & 15, fchk = lane >> 4;
  constexpr int RB = T::wide ? 8 : 4;  // rows per batch (one store instruction)
  constexpr int NB = 16 / RB;          // batches per sub-tile
  const int rr = T::wide ? (lane >> 3) : (lane >> 4), cc = T::wide ? (lane & 7) : (lane & 15);
  const int col = col0 + cc * (T::wide ? 8 : 4);  // in half 0; half 1 is 64 columns further
  const uint32_t OOB = 0xFFFFFFF0u;
  int ldc = (int)e.ldc, ldr = (int)e.ldr, ldx = (int)e.ldaux;
  asm volatile("" : "+s"(ldc), "+s"(ldr), "+s"(ldx));
  const int rows_left = M - m0 - row0;
  const uint32_t lane_c = (uint32_t)(rr * ldc + col), lane_r = (uint32_t)(rr * ldr + col), lane_x = (uint32_t)(rr * ldx + col);

  // Look-ahead of the residual / saved-gelu' loads in sub-tiles: with ONE sub-tile ahead a wave keeps 2 (bf16) or 4 (fp32) KiB in
  // flight, 16 - 32 KiB per CU -- at an HBM round trip of 1 - 2 us under load that is 10 - 30 GB/s per CU, which is what these
  // epilogues ran at (in-kernel stamps: 8.6 us for the x gelu' tile with 44 or with 256 workgroups active alike).  The accumulator
  // registers free up as the sub-tiles leave (16 per sub-tile) and the main loop's 64 fragment registers are dead here.
  constexpr int LA = T::wide ? HCT_EPI_LA_WIDE : HCT_EPI_LA_F32;
  static_assert(LA >= 1 && LA <= 8, "look-ahead in sub-tiles");
  u32x4 ld[T::loads ? 8 : 1][T::loads ? NB : 1];
  auto issue_loads = [&](int s) {  // sub-tile s = 2 i + h
    if (!T::loads) return;
    const int i = s >> 1, h = s & 1;
    const bool nok = n0 + col + h * 64 < N;
#pragma unroll
    for (int it = 0; it < NB; ++it) {
      const int prow = i * 16 + it * RB;
      const bool ok = nok && prow + rr < rows_left;
      if (MODE == EPI_RES_F32) ld[s][it] = __builtin_amdgcn_raw_buffer_load_b128(tb.res, ok ? (lane_r + h * 64) * 4u : OOB, (row0 + prow) * ldr * 4, kNT);
      else ld[s][it] = __builtin_amdgcn_raw_buffer_load_b128(tb.aux, ok ? (lane_x + h * 64) * 2u : OOB, (row0 + prow) * ldx * 2, kNT);
    }
  };
#pragma unroll
  for (int s = 0; s < LA && s < 2 * MT; ++s) issue_loads(s);
#pragma unroll
  for (int s = 0; s < 2 * MT; ++s) {
    const int i = s >> 1, h = s & 1;
    const bool nok = n0 + col + h * 64 < N;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) *reinterpret_cast<f32x4*>(patch + frow * 256 + (((jj * 4 + fchk) ^ frow) << 4)) = acc[i][h * 4 + jj];
    if (s + LA < 2 * MT) issue_loads(s + LA);  // LA sub-tiles ahead and BEFORE this sub-tile's stores: waiting for it never drains them
#pragma unroll
    for (int it = 0; it < NB; ++it) {
      const int pr = it * RB + rr;        // row inside the 16-row patch
      const int prow = i * 16 + it * RB;  // batch's first row inside the wave tile
      const bool ok = nok && prow + rr < rows_left;
      if (T::wide) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(patch + pr * 256 + (((2 * cc) ^ pr) << 4));
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(patch + pr * 256 + (((2 * cc + 1) ^ pr) << 4));
        f32x4 x0 = v0 * e.alpha + bias[h].lo, x1 = v1 * e.alpha + bias[h].hi;
        const uint32_t vc = ok ? (lane_c + h * 64) * 2u : OOB, vx = ok ? (lane_x + h * 64) * 2u : OOB;
        const int sc = (row0 + prow) * ldc * 2, sx = (row0 + prow) * ldx * 2;
        auto pack = [](f32x4 a, f32x4 b) {
          bf16x8 o = {(bf16)a[0], (bf16)a[1], (bf16)a[2], (bf16)a[3], (bf16)b[0], (bf16)b[1], (bf16)b[2], (bf16)b[3]};
          return __builtin_bit_cast(u32x4, o);
        };
        if (MODE == EPI_GELU_BF16) {
          if (e.aux_deriv) {  // aux receives gelu'(pre-activation): the backward then multiplies by it (no second evaluation)
            f32x4 d0, d1;
#if HCT_GELU_VEC8
            f32x4 g0, g1;
            gelu_both8(x0, x1, g0, d0, g1, d1);
            x0 = g0; x1 = g1;
#else
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              float ga, da, gb, db;
              gelu_both(x0[q], ga, da);
              gelu_both(x1[q], gb, db);
              x0[q] = ga; d0[q] = da; x1[q] = gb; d1[q] = db;
            }
#endif
            __builtin_amdgcn_raw_buffer_store_b128(pack(d0, d1), tb.aux, vx, sx, kNT);
            HCT_STORE_GUARD();
          } else {
            __builtin_amdgcn_raw_buffer_store_b128(pack(x0, x1), tb.aux, vx, sx, kNT);
            HCT_STORE_GUARD();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              x0[q] = gelu_fast(x0[q]);
              x1[q] = gelu_fast(x1[q]);
            }
          }
        } else if (MODE == EPI_DGELU_BF16 || MODE == EPI_DGELU_CS) {
          const bf16x8 t = __builtin_bit_cast(bf16x8, ld[s][it]);
          if (e.aux_deriv) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              x0[q] *= (float)t[q];
              x1[q] *= (float)t[4 + q];
            }
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              x0[q] *= dgelu_fast((float)t[q]);
              x1[q] *= dgelu_fast((float)t[4 + q]);
            }
          }
          if (MODE == EPI_DGELU_CS) {  // fused bias gradient of the Linear feeding the GELU (column sums of this output)
            cs[h][0] += ok ? x0 : f32x4{0, 0, 0, 0};
            cs[h][1] += ok ? x1 : f32x4{0, 0, 0, 0};
          }
        }
        __builtin_amdgcn_raw_buffer_store_b128(pack(x0, x1), tb.c, vc, sc, HCT_BF16_OUT_POLICY);
        HCT_STORE_GUARD();
      } else {
        const f32x4 v = *reinterpret_cast<const f32x4*>(patch + pr * 256 + ((cc ^ pr) << 4));
        f32x4 x = v * e.alpha + bias[h].lo;
        if (MODE == EPI_RES_F32) x += __builtin_bit_cast(f32x4, ld[s][it]);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x), tb.c, ok ? (lane_c + h * 64) * 4u : OOB, (row0 + prow) * ldc * 4, STORE_POLICY);
        HCT_STORE_GUARD();
      }
    }
  }
}


// raw buffer descriptor in SGPRs for inline-asm buffer instructions (same 4 words make_buffer_rsrc builds)
__device__ __forceinline__ i32x4 make_srd(const void* base, uint32_t num_records) {
  const uint64_t pa = (uint64_t)base;
  return i32x4{(int)__builtin_amdgcn_readfirstlane((uint32_t)pa), (int)(__builtin_amdgcn_readfirstlane((uint32_t)(pa >> 32)) & 0xFFFF),
               (int)__builtin_amdgcn_readfirstlane(num_records), 0x00020000};
}

// one 1-KiB LDS-DMA piece: lane l's 16 B from base + voffset land at LDS lds_base + 16*l (M0 carries the LDS base)
__device__ __forceinline__ void dma16s(i32x4 rsrc, uint32_t lds_base, uint32_t voff, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_base), "v"(voff), "s"(rsrc), "s"(soff)
               : "memory", "m0");
}
__device__ __forceinline__ void dma16(i32x4 rsrc, uint32_t lds_base, uint32_t voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_base), "v"(voff), "s"(rsrc) : "memory", "m0");
}

// Diagnostic build only (-DHCT_STAMPS, scripts/stamp_gemm.py; never part of libheadct_hip.so): wave 0 of every workgroup
// records s_memrealtime (100 MHz, chip-global) at four points of each tile into one VGPR (lane = slot) and stores it once
// when the workgroup exits.
#ifdef HCT_STAMPS
__device__ uint32_t* g_stamp_ptr = nullptr;
__device__ uint32_t g_stamp_words = 0;  // capacity of the stamp buffer: the only raw-pointer store of the diagnostic build is bounded by it
#define HCT_STAMP(k)                                                                                   \
  do {                                                                                                 \
    unsigned long long t_;                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                     \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    stamps = lane == ((tile_i * 4 + (k)) & 63) ? (uint32_t)t_ : stamps;                                \
  } while (0)
#else
#define HCT_STAMP(k)
#endif

// stream-K region of the NT workspace (its LAST kSkBytes): [flags: one word per workgroup | error word] then one 256-KiB slab of
// raw fp32 accumulators per workgroup
constexpr int kSkMaxWgs = 256;
constexpr size_t kSkHeadBytes = 4096, kSkSlabBytes = 262144;
constexpr size_t kSkBytes = kSkHeadBytes + (size_t)kSkMaxWgs * kSkSlabBytes;
constexpr int kSkErrWord = 512;

// what a follower publishes: the launch's sequence number -- or, with the debug bit 31 of the kernel argument set
// (hct_debug_set_gemm_variant(-8): test of the time-out path), a wrong one, so that every owner times out
__device__ __forceinline__ unsigned sk_pub(unsigned seq_arg) { return ((seq_arg >> 31) ? seq_arg + 1u : seq_arg) & 0x0FFFFFFFu; }

__device__ __forceinline__ uint32_t xcc_id() {  // the XCD (accelerator die) this wave runs on
  uint32_t v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 15;
}

// PERSISTENT: grid = min(tiles, #CUs); each workgroup walks tiles vb = blockIdx.x, +gridDim.x, ... (same XCD every trip,
// consecutive tiles of an XCD share an A row-panel).  At the end of a tile the first pair of stages of the NEXT tile is
// issued before the epilogue, so the output stores (asynchronous) and the next tile's HBM latency drain under each other
// and under the next main loop instead of leaving the CU's matrix pipes idle.
#ifndef HCT_STAGGER_DMA
#define HCT_STAGGER_DMA 0
#endif
// MT = row tiles of 16 per wave: 4 (256-row tiles) or 3 (192-row tiles, wave tiles of 48 x 128, for the plain / +residual shapes
// whose 256-row tiles fill less than one round of CUs while 192-row tiles still fit one: the encoder's M = 14 080, N = 768 GEMMs are 165
// tiles of 256 rows on 256 CUs and 222 of 192).  The A stage keeps its 16-KiB region and has 12 pieces: waves 6 and 7 issue B pieces
// only (4 operations per pair instead of 8; their counted wait at the tile top says so).  (Issued as dummies with an out-of-range
// lane offset instead, the 192-row tiles were only 6 - 10 % faster than the 256-row ones: the main loop pays per DMA INSTRUCTION.)
template <int MODE, bool SK = false, int MT = 4>
__global__ void __launch_bounds__(512, 2) gemm_bf16_nt256_kernel(int M, int N, int K, const bf16* __restrict__ A, int64_t lda,
                                                                 const bf16* __restrict__ B, int64_t ldb, Epilogue e, int ntiles, int stagger,
                                                                 int sk_tiles, int sk_wgs, unsigned char* __restrict__ sk_ws, unsigned sk_seq) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[163840];  // 5 stages x (A 16K | B 16K); stage 4 (generic epilogue: 3 and 4) doubles as the epilogue patches
  // Specialised epilogues: the next tile's first TWO stage pairs (ring buffers 0 .. 3) are issued before the epilogue, whose
  // patches are 4 KiB per wave (buffer 4).  The next main loop then waits only for those pairs -- `vmcnt(kEpiOps)`: vector-memory
  // operations retire in issue order and the epilogue's kEpiOps loads / stores are younger than the pairs -- and the output
  // stores drain under its first two K-steps.  (With one pair ahead and 8-KiB patches the loop start waited for the stores:
  // 4 .. 8 us per tile by the in-kernel stamps.)
#ifndef HCT_NT_TWO_PAIR_MODES  // bit m set: epilogue mode m prefetches two pairs (diagnostic builds override; generic never)
#define HCT_NT_TWO_PAIR_MODES 0  /* measured: 0x7E (all specialised modes) +0.37 ms per step, 0x2A (the modes without epilogue loads) the same */
#endif
  constexpr bool kTwoPairs = MODE != EPI_GENERIC && ((HCT_NT_TWO_PAIR_MODES >> MODE) & 1);
  static_assert(!(SK && kTwoPairs), "stream-K items assume one prefetched pair");
  static_assert(MT == 4 || (MT == 3 && !SK && (MODE == EPI_PLAIN_BF16 || MODE == EPI_RES_F32)), "192-row tiles: whole tiles, plain / +residual epilogues");
  constexpr int kEpiOps = EpiTraits<MODE>::ops_per_tile > 63 ? 63 : EpiTraits<MODE>::ops_per_tile;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ntn = (N + 255) >> 8;
  float* const colsum_out = e.colsum_partial;  // by value: indexing through `e` made hipcc keep a copy of the struct in scratch

  // staging: 1 KiB piece = 16 rows x 64 B; wave w moves pieces 2w, 2w+1 of A and of B each stage
  // HCT_NT_LOADER_WAVES=1 (experiment): waves 0 .. 3 -- one per SIMD -- move ALL the pieces (4 of A and 4 of B each), so that on every
  // SIMD the wave that is held by its DMA issues has a partner that only multiplies
#ifndef HCT_NT_LOADER_WAVES
#define HCT_NT_LOADER_WAVES 0
#endif
  constexpr int NP = HCT_NT_LOADER_WAVES ? 4 : 2;  // pieces of each operand per loading wave and stage
  uint32_t voa[NP], vob[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int row = ((wave & (HCT_NT_LOADER_WAVES ? 3 : 7)) * NP + i) * 16 + (lane >> 2);
    const int src_chunk = (lane & 3) ^ swz64(row);
    voa[i] = (uint32_t)(row * lda * 2 + src_chunk * 16);
    vob[i] = (uint32_t)(row * ldb * 2 + src_chunk * 16);
  }
  const int wm = wave >> 1, wn = wave & 1;  // 4(M) x 2(N) waves, 64 x 128 outputs each
  const int frow = lane & 15, fchk = lane >> 4;
  const int foff = frow * 64 + ((fchk ^ swz64(frow)) << 4);
  const int nk = K >> 5;

  // operand stream as inline-asm LDS-DMA (see make_srd / dma16s): keeps hipcc from draining vmcnt before LDS accesses
  i32x4 ra, rb;
  const uint32_t lds0 = (uint32_t)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);
  int m0 = 0, n0 = 0;
  // item = (tile id, first K-stage): the operand descriptors are rooted at (m0 | n0, s0 * 32), the stage offsets stay relative
  auto set_tile_id = [&](int id, int s0) {
    const int tm = id / ntn, tn = id - tm * ntn;
    m0 = tm * (64 * MT);
    n0 = tn << 8;
    const bf16* Ab = A + (int64_t)m0 * lda + s0 * 32;
    const bf16* Bb = B + (int64_t)n0 * ldb + s0 * 32;
    ra = make_srd(Ab, clamp_records(((int64_t)(M - m0 - 1) * lda + K - s0 * 32) * 2));
    rb = make_srd(Bb, clamp_records(((int64_t)(N - n0 - 1) * ldb + K - s0 * 32) * 2));
  };
  // whole tiles: ids [sk_tiles, ntiles), walked vb = blockIdx.x, +gridDim.x, ... through the XCD-contiguous permutation
  auto dp_id = [&](int vb) {
    const int nwg = ntiles - sk_tiles;
    const int xcd = vb & 7, q = nwg >> 3, r = nwg & 7;
    return sk_tiles + (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
  };
  // Stages are fetched in PAIRS (t even, t+1).  One stage is 32 bf16 = 64 B of every row, i.e. half a 128-B line, and a
  // stream of half-line requests draws only 62 GB/s per CU from L2 where whole lines give 111 (scripts/micro/
  // lds_dma_rate.hip); issuing the two halves of the same lines in consecutive instructions recovers most of it (94 GB/s
  // per CU: the second request meets the line in the vector L1).  In the isolated GEMM benchmark (operands warm in the
  // Infinity Cache) this schedule is 4 % slower than one stage per K-step on a ring of 4; inside the training step, where
  // the operands come from HBM, it is 0.8 ms per step faster (scripts/ab_step.py) -- and it needs one barrier per two
  // K-steps instead of two.
  auto stage_pair = [&](int t) {
    const uint32_t b0 = lds0 + (t % 5) * 32768, b1 = lds0 + ((t + 1) % 5) * 32768;  // wave-uniform: SALU only
    const uint32_t kb = (uint32_t)t * 64;                                            // 32 bf16 = 64 B per stage
    if (HCT_NT_LOADER_WAVES && wave >= 4) return;  // (wave-uniform)
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int c = wave * NP + i;
      // the stage's K offset rides in the scalar soffset operand: the lane offsets stay tile- and stage-invariant (no
      // per-stage VALU, nothing for the compiler to pre-compute and spill); rows past M are still dropped by the
      // descriptor's range check on voffset
#ifdef HCT_TIMING_NO_DMA  // diagnostic build: main loop without its operand stream (outputs are garbage)
      asm volatile("" ::"s"(b0 + c * 1024), "s"(b1), "v"(voa[i]), "v"(vob[i]), "s"(kb), "s"(ra), "s"(rb));
#else
      if (MT == 4 || c < 4 * MT) {  // (wave-uniform; 192-row tiles: the A stage has 12 pieces, waves 6 and 7 issue none)
        dma16s(ra, b0 + c * 1024, voa[i], kb);
        dma16s(ra, b1 + c * 1024, voa[i], kb + 64);
      }
      dma16s(rb, b0 + 16384 + c * 1024, vob[i], kb);
      dma16s(rb, b1 + 16384 + c * 1024, vob[i], kb + 64);
#endif
    }
  };
  // Software pipeline at half-stage granularity (16 live fragments: 4 A + 4 A' + 4 B-low + 4 B-high):
  //   first half : issue the B-high reads of stage t, run the 16 MFMAs of columns 0..63 (B-low)
  //   boundary   : (odd stages only) the next pair has landed (vmcnt + barrier), refill the ring;
  //                issue A' and B-low reads of stage t+1
  //   second half: run the 16 MFMAs of columns 64..127 (B-high) while those reads return
  f32x4 acc[MT][8];
  bf16x8 b_lo[4], b_hi[4], a0[MT], a1[MT];
  auto rd_a = [&](int t, bf16x8* af) {
    const unsigned char* sa = smem + (t % 5) * 32768 + wm * (16 * MT * 64) + foff;
#pragma unroll
    for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sa + i * 1024);
  };
  auto rd_b = [&](int t, int half, bf16x8* bq) {
    const unsigned char* sb = smem + (t % 5) * 32768 + 16384 + wn * (128 * 64) + half * 4096 + foff;
#pragma unroll
    for (int j = 0; j < 4; ++j) bq[j] = *reinterpret_cast<const bf16x8*>(sb + j * 1024);
  };
  // (no s_setprio around the MFMA groups: raising the priority of the issuing wave starves its SIMD partner's LDS reads
  //  and DMA issue -- 0.55 ms of the 43.8 ms step, scripts/ab_step.py)
  auto mma = [&](int half, const bf16x8* af, const bf16x8* bq) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][half * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[j], af[i], acc[i][half * 4 + j], 0, 0, 0);
  };
  // pair landed: this wave's DMA retired (all of it, or all but the 8 operations of the youngest pair) + barrier, after
  // which every wave's pieces are visible
  auto land_all = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  // first odd step of a tile whose first two pairs were waited for at the tile start: nothing has been issued since
  auto land_first = [&](int t) {
    if (kTwoPairs && t == 0) __builtin_amdgcn_s_barrier();
    else land_all();
  };
  auto land_but_youngest_pair = [&]() {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  // Top of an item: pair (0,1) must have landed, pair (2,3) has just been issued.  Between the two this wave issued the previous
  // item's epilogue -- a fixed number of vector-memory operations (loads and stores go out unconditionally, out-of-range lanes
  // by descriptor) -- and operations retire in issue order, so allowing that many PLUS the 8 of pair (2,3) to be outstanding
  // still means (0,1) is in LDS, while the previous tile's output stores keep draining under the first two K-steps instead of
  // in front of them (2.2 us per tile by the in-kernel stamps).  An under-estimate is safe (it only waits longer).
  // MEASURED: no change inside the step (39.83 / 39.91 with, 39.84 / 39.91 ms without, scripts/ab_step.py on one box) -- the drain
  // moves to the full wait of the first odd K-step -- so the plain wait stays the default.
#ifndef HCT_NT_COUNTED_TOP
#define HCT_NT_COUNTED_TOP 0
#endif
  constexpr int kTopOps = 8 + EpiTraits<MODE>::ops_per_tile > 63 ? 63 : 8 + EpiTraits<MODE>::ops_per_tile;
  auto land_top = [&](int younger) {  // younger: 0 = nothing issued since pair (0,1), 1 = a specialised epilogue, 2 = a follower's 32 slab stores
#if HCT_NT_LOADER_WAVES  // operations per pair: 16 on the loading waves (8 where a 192-row tile leaves the wave B pieces only), none on the others
    if (wave >= 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (MT == 3 && wave == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    return;
#endif
    if (MT == 3 && wave >= 6) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // (these waves issue 4 operations per pair: B pieces only)
    else if (!HCT_NT_COUNTED_TOP || MODE == EPI_GENERIC || younger == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (younger == 2) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kTopOps) : "memory");
    __builtin_amdgcn_s_barrier();
  };
  // De-phase the persistent workgroups: all tiles cost the same, so without this every CU reaches its epilogue at the same
  // moment and the chip alternates between an HBM write burst (matrix pipes idle, vmcnt is in-order so the next tile
  // cannot start until the stores drain) and a pure-MFMA phase.  Eight start phases spread the bursts over the main loops
  // of the other CUs.
  if (stagger > 0) {
    const int phase = (blockIdx.x >> 3) & 7;
    for (int i = 0; i < phase * stagger; ++i) __builtin_amdgcn_s_sleep(32);
  }
#ifdef HCT_STAMPS
  uint32_t stamps = 0;
  int tile_i = 0;
#endif
  // ---- work items ---------------------------------------------------------------------------------------------------------
  // Whole tiles cost the same, so ntiles = R * grid + rem leaves (grid - rem) CUs idle for a whole tile time in the last round
  // (651 tiles of the decoder's N = 768 GEMMs on 256 CUs: 2.54 rounds of work in 3).  In the SK instances the first `sk_tiles`
  // (= rem) tile ids are therefore shared out by K range ("stream-K" for the remainder, whole tiles for the rest): in units of
  // stage pairs the rem * P pairs are cut into sk_wgs contiguous ranges, workgroup c takes [bound(c), bound(c+1)), split at the
  // tile boundary into at most one piece that starts inside a tile (FOLLOWER: raw accumulators to slab c, then flag c) and one
  // that starts a tile (OWNER: after its own K range it adds the followers' slabs in workgroup order -- a fixed
  // order, bit-reproducible -- and runs the epilogue).  A follower piece is always the FIRST thing its workgroup does and
  // never waits, so an owner only ever waits for work that started at kernel start on a higher-numbered workgroup: no
  // cycles, and a bounded spin flags an error instead of hanging should the grid not be resident.
  // An item is one packed word -- tile id [0,8) | first pair [8,18) | pairs [18,28) | followers to collect [28,31) -- so that
  // the persistent loop carries two more scalars than the plain instances, not ten.
  const int P = nk >> 1;  // stage pairs per tile
  uint32_t it_first = 0, it_owner = 0;
  // Owner and followers of a tile sit on the SAME XCD (workgroups are dealt round-robin: XCD = blockIdx & 7): XCD x shares out
  // its own slice of the remainder tiles over its own workgroups j = blockIdx >> 3, so a follower's slab is read back from the
  // L2 it was written through, not from HBM.  (Only the speed depends on that placement: stores and flags are agent-scope.)
  if (SK) {
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int t0 = (x * sk_tiles) >> 3, nx = (((x + 1) * sk_tiles) >> 3) - t0;  // this XCD's tiles [t0, t0 + nx)
    const int wx = sk_wgs < nx * 4 ? sk_wgs : nx * 4;                           // its workgroups that take a K range (<= 4 per tile)
    if (j < wx) {
      auto sk_bound = [&](int c) -> int {
        if (c >= wx) return nx * P;
        const int v = (int)(((int64_t)c * nx * P) / wx);
        const int r = v % P;  // no piece shorter than two pairs (the pipeline needs four stages): snap to the tile boundary
        return r == 1 ? v - 1 : (r == P - 1 ? v + 1 : v);
      };
      int b = sk_bound(j);
      const int en = sk_bound(j + 1);
      int t = b / P;
      const int off = b - t * P;
      if (off) {
        const int pe = en < (t + 1) * P ? en : (t + 1) * P;
        it_first = (uint32_t)(t0 + t) | ((uint32_t)off << 8) | ((uint32_t)(pe - b) << 18);
        b = pe;
        ++t;
      }
      if (b < en) {  // b == t * P: this workgroup starts tile t; the host keeps every range shorter than a tile
        const int tend = (t + 1) * P;
        uint32_t nf = 0;
        for (int c2 = j + 1; c2 < wx && sk_bound(c2) < tend; ++c2) ++nf;
        it_owner = (uint32_t)(t0 + t) | ((uint32_t)((en < tend ? en : tend) - b) << 18) | (nf << 28) | 0x80000000u;  // (bit 31: present)
      }
      if (!it_first) { it_first = it_owner; it_owner = 0; }
    }
  }
  int vb = blockIdx.x;
  uint32_t cit = 0;  // the item set_tile_id was last called for
  auto take_item = [&](uint32_t it) {
    set_tile_id((int)(it & 255), (int)((it >> 8) & 1023) * 2);
    cit = it;
  };
  auto next_item = [&]() -> bool {
    if (SK && it_owner) {
      take_item(it_owner);
      it_owner = 0;
      return true;
    }
    if (vb < ntiles - sk_tiles) {
      set_tile_id(dp_id(vb), 0);
      cit = (uint32_t)P << 18;
      vb += gridDim.x;
      return true;
    }
    return false;
  };
#ifdef HCT_PRIO_YOUNG
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
  // Epilogues that LOAD (fp32 residual: 256 KiB per tile; saved gelu': 128 KiB) are bound by the CU's miss path: a CU draws ~30 GB/s
  // from beyond its L2 however many loads it keeps in flight (in-kernel stamps: the x gelu' epilogue takes 8.2 us with one sub-tile
  // of look-ahead and 7.6 + 1 with the whole tile requested up front, HCT_EPI_LA_*; the plain one 3.1), with the matrix pipes idle.
  // HCT_EPI_TOUCH (bit m: epilogue mode m; experiment): one dword of each 128-B line of the tile's residual / aux rows is requested
  // right behind the item's LAST stage pair -- 6 K-steps before the main loop ends; nothing younger is ever waited for inside the
  // loop, the one remaining landing wait is counted -- so that the epilogue's loads find the lines in L2.  (Round 3's first form
  // issued them 12 K-steps early: the in-order vmcnt wait of the NEXT pair then stalled the loop for an HBM round trip.)
#ifndef HCT_EPI_TOUCH
#define HCT_EPI_TOUCH 0
#endif
  constexpr bool kTouch = ((HCT_EPI_TOUCH >> MODE) & 1) && EpiTraits<MODE>::loads;
  constexpr int kTouchOps = MODE == EPI_RES_F32 ? 4 : 2;  // per wave: 2048 / 1024 lines per tile
  uint32_t res_sink = 0;
  auto res_touch = [&]() {
    if (!kTouch) return;
    const bool f32 = MODE == EPI_RES_F32;
    int ldt = f32 ? (int)e.ldr : (int)e.ldaux;
    asm volatile("" : "+s"(ldt));
    const int esz = f32 ? 4 : 2;
    const char* rp = (f32 ? (const char*)e.residual : (const char*)e.aux) + ((int64_t)m0 * ldt + n0) * esz;
    const i32x4 rres = make_srd(rp, clamp_records((((int64_t)(M - m0 - 1) * ldt + (N - n0))) * esz));
    int ln = lane;
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int j = 0; j < kTouchOps; ++j) {
      // fp32: 8 lines per row, 8 rows per load; bf16: 4 lines per row, 16 rows per load
      const int row = f32 ? wave * 32 + j * 8 + (ln >> 3) : wave * 32 + j * 16 + (ln >> 2);
      const uint32_t off = (uint32_t)(row * ldt * esz + (f32 ? (ln & 7) : (ln & 3)) * 128);
      asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "+v"(res_sink) : "v"(off), "s"(rres) : "memory");
    }
  };
  if (SK && it_first) take_item(it_first);
  else if (!next_item()) return;  // (only with stream-K: more workgroups than K ranges and no whole tiles)
  stage_pair(0);
  if (kTwoPairs) {
    stage_pair(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // first tile: no epilogue behind the pairs to leave in flight
  }
  int younger = 0;  // what this wave has issued since the current item's pair (0,1): see land_top
  while (true) {
    HCT_STAMP(0);
    const int cm0 = m0, cn0 = n0;  // item being computed (next_item below moves m0/n0 to the next one)
    const uint32_t item = cit;
    const int cns = SK ? (int)((item >> 18) & 1023) * 2 : nk;  // stages of this item
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    // nk is even and >= 4 (host dispatch: K % 64 == 0, K >= 128).  Ring of 5 stage buffers; pair (0,1) was issued before
    // the previous tile's epilogue.  Buffers 3 and 4 were that epilogue's patches: once every wave is through with them
    // (barrier) pair (2,3) may go.  Pair (0,1) is older than the epilogue's loads/stores and than pair (2,3) (vmcnt retires
    // in issue order), so allowing the 8 youngest operations to be outstanding means (0,1) has landed.
    if (kTwoPairs) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kEpiOps) : "memory");  // pairs (0,1), (2,3): older than the previous epilogue's operations
      __builtin_amdgcn_s_barrier();                                    // ... of every wave; and every wave is done with its patch
    } else {
      __builtin_amdgcn_s_barrier();
      stage_pair(2);
      land_top(younger);
    }
    HCT_STAMP(1);
    rd_a(0, a0);
    rd_b(0, 0, b_lo);
    // Steady state, two K-steps per trip.  Even step t: no synchronisation at all (pair (t, t+1) became visible at the
    // previous odd step).  Odd step t+1: pair (t+2, t+3), issued two steps ago, must have landed; every wave is then past
    // its reads of stages t-1 and t, whose buffers take pair (t+4, t+5).
    int t = 0;
    bool touched = false;
    // The two waves of a SIMD (w and w + 4) run the same program between the same barriers: both issue their 8 DMA pieces
    // right behind the barrier (an LDS-DMA piece holds the issuing wave for 60 - 185 cycles, MI355X_MICROARCH.md) and then both
    // want the matrix pipe.  HCT_STAGGER_DMA: waves 4 - 7 issue theirs one MFMA group later, so that on every SIMD one wave issues
    // DMA while the other multiplies (same order of issue per wave, so the counted waits hold).
    for (; t + 5 < cns; t += 2) {
      rd_b(t, 1, b_hi);
      mma(0, a0, b_lo);
      rd_a(t + 1, a1);
      rd_b(t + 1, 0, b_lo);
      mma(1, a0, b_hi);
      rd_b(t + 1, 1, b_hi);
      mma(0, a1, b_lo);
      land_first(t);
#if HCT_STAGGER_DMA
      if (wave < 4) stage_pair(t + 4);
      __builtin_amdgcn_sched_barrier(0);
#else
      stage_pair(t + 4);
#endif
      if (kTouch && t + 6 == cns && !(SK && ((item >> 8) & 1023))) {  // the item's last pair is out: nothing younger is waited for in the loop
        res_touch();
        touched = true;
      }
      rd_a(t + 2, a0);
      rd_b(t + 2, 0, b_lo);
      mma(1, a1, b_hi);
#if HCT_STAGGER_DMA
      __builtin_amdgcn_sched_barrier(0);
      if (wave >= 4) stage_pair(t + 4);
#endif
    }
    // t == cns - 4: every stage of the item has been issued
    rd_b(t, 1, b_hi);
    mma(0, a0, b_lo);
    rd_a(t + 1, a1);
    rd_b(t + 1, 0, b_lo);
    mma(1, a0, b_hi);
    rd_b(t + 1, 1, b_hi);
    mma(0, a1, b_lo);
    if (kTouch && touched) {  // pair (cns-2, cns-1) is older than the touches: a counted wait
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kTouchOps) : "memory");
      __builtin_amdgcn_s_barrier();
    } else {
      land_first(t);  // pair (cns-2, cns-1)
    }
    rd_a(t + 2, a0);
    rd_b(t + 2, 0, b_lo);
    mma(1, a1, b_hi);
    rd_b(t + 2, 1, b_hi);
    mma(0, a0, b_lo);
    rd_a(t + 3, a1);
    rd_b(t + 3, 0, b_lo);
    mma(1, a0, b_hi);
    // Bias of this lane's columns: requested here, with nothing else outstanding (every stage has landed) and the a0 fragments
    // dead, and pinned as resident after the last MFMA group -- part of its L2 round trip runs under the two groups in between.
    // hipcc cannot count the asm LDS-DMA operations, so a wait it generated for this load after the next tile's prefetch would
    // wait for that prefetch.
    TileBias bv = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    TileBias bh[2] = {bv, bv};
    if (MODE != EPI_GENERIC) tile_bias_halves<MODE>(e, cn0, wn * 128, lane, N, bh);
    rd_b(t + 3, 1, b_hi);
    mma(0, a1, b_lo);
    mma(1, a1, b_hi);

    if (kTouch) {  // the touch loads write res_sink whenever they return: it stays allocated until they have
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("" ::"v"(res_sink));
    }
    __builtin_amdgcn_s_barrier();  // every wave has its last fragments in registers: the whole ring is free
    HCT_STAMP(2);
    if (MODE != EPI_GENERIC) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (EpiTraits<MODE>::wide) asm volatile("" : "+v"(bh[h].lo), "+v"(bh[h].hi));
        else asm volatile("" : "+v"(bh[h].lo));
      }
    }
    const bool more = next_item();
    if (more) {  // prefetch the next item's first pair(s) of stages (ring buffers 0, 1 [, 2, 3]) under this item's epilogue
      stage_pair(0);
      if (kTwoPairs) stage_pair(2);
    }
    if (SK && ((item >> 8) & 1023)) {
      // FOLLOWER piece: the raw accumulators leave in register order (1 KiB per store instruction), write-through (sc1) like
      // the wgrad's split partials -- a write-through store needs no release fence -- then ONE flag per workgroup
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc(sk_ws + kSkHeadBytes + (size_t)blockIdx.x * kSkSlabBytes, 0, (uint32_t)kSkSlabBytes, 0x00020000);
      int ln = lane;
      asm volatile("" : "+v"(ln));  // (not hoisted out of the persistent loop: a register there costs a spill in the main loop)
      const uint32_t off0 = (uint32_t)(wave * 32768 + ln * 16);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rs, off0 + (uint32_t)((i * 8 + j) * 1024), 0, 16);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0)  // flag = launch sequence number | the XCD this workgroup really runs on
        __hip_atomic_store((unsigned*)sk_ws + blockIdx.x, (sk_pub(sk_seq) << 4) | xcc_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      younger = 0;  // (everything was drained for the flag)
    } else {
      younger = 1;
      if (SK && ((item >> 28) & 7)) {
        // OWNER of a shared tile: add the followers' partials, workgroup order c+1, c+2, ...  Their bytes were stored
        // write-through and drained before the flag; the poll is relaxed, ONE agent-scope acquire then drops this CU's stale
        // lines before the plain loads (the protocol of the wgrad's in-launch fold; MI355X_MICROARCH.md, "Valid forms").
        const int c_end = (int)blockIdx.x + 8 * (1 + (int)((item >> 28) & 7));
        int& s_bad = *reinterpret_cast<int*>(smem + 5 * 32768 - 16);  // (end of the last epilogue patch: free until the epilogue below)
        for (int c2 = blockIdx.x + 8; c2 < c_end; c2 += 8) {  // the next workgroups of this XCD
          if (threadIdx.x == 0) {
            unsigned spins = 0, f;
            while (((f = __hip_atomic_load((unsigned*)sk_ws + c2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 4) != (sk_seq & 0x0FFFFFFFu) &&
                   spins < (1u << 20)) {
              __builtin_amdgcn_s_sleep(8);
              ++spins;
            }
            // A partial that never arrives cannot happen while the whole grid is resident (a follower never waits), but a grid that
            // is NOT resident is possible -- e.g. a communication kernel holding more CUs than hct_set_cu_reserve left free.  The
            // error word (hct_gemm_nt_flags_offset) is set AND the tile is poisoned with NaN below: the loss of this or the next
            // step is then not finite, which the engine checks every step (engine_pretrain_mae.py:76-78), instead of training on
            // silently wrong numbers.
            s_bad = spins >= (1u << 20);
            if (spins >= (1u << 20))
              __hip_atomic_store((unsigned*)sk_ws + kSkErrWord, 0xDEADu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // Same XCD (the rule, see the item set-up): the slab was written THROUGH this XCD's L2 and is read below by loads that
            // bypass the vector L1 (sc1), so nothing has to be invalidated.  Another XCD: agent-scope acquire first -- it drops the
            // L2's clean lines, the operand panels of all 32 CUs with them, which is why it is not done unconditionally.
            if ((f & 15) != xcc_id()) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          }
          __syncthreads();
          const __amdgpu_buffer_rsrc_t rs =
              __builtin_amdgcn_make_buffer_rsrc(sk_ws + kSkHeadBytes + (size_t)c2 * kSkSlabBytes, 0, (uint32_t)kSkSlabBytes, 0x00020000);
          int ln = lane;
          asm volatile("" : "+v"(ln));
          const uint32_t off0 = (uint32_t)(wave * 32768 + ln * 16);
#pragma unroll
          for (int i = 0; i + 1 < MT; i += 2) {  // 16 loads (16 KiB per wave) in flight: the fragment registers are dead here
            f32x4 v[2][8];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
              for (int j = 0; j < 8; ++j)
                v[u][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off0 + (uint32_t)(((i + u) * 8 + j) * 1024), 0, 16));
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
              for (int j = 0; j < 8; ++j) acc[i + u][j] += v[u][j];
          }
          // (one poisoned accumulator is enough -- four NaN outputs per lane reach the loss through every later layer -- and costs
          //  no registers: filling all 32 accumulators under a branch made every stream-K instance spill, 112 - 192 B per lane; so did
          //  a branch around one of them, hence the unconditional multiplication by 1 or NaN)
          acc[0][0] = acc[0][0] * (s_bad ? __builtin_nanf("") : 1.0f);
          __syncthreads();  // (s_bad is rewritten by the next follower's poll)
        }
      }
      // generic: ring buffers 3 and 4 (8 KiB per wave), refilled only after the next tile's first barrier; specialised: buffer 4
      unsigned char* patch = kTwoPairs ? smem + 4 * 32768 + wave * 4096 : smem + 3 * 32768 + wave * 8192;
      if (MODE == EPI_GENERIC) {
#pragma unroll
        for (int i = 0; i < MT; ++i) epilogue_tile16x128(e, patch, lane, cm0 + wm * (16 * MT) + i * 16, cn0 + wn * 128, M, N, acc[i]);
      } else {
        TileBufs tb;
        const int csz = (MODE == EPI_RES_F32) ? 4 : 2;
        tb.c = tile_rsrc(e.C, e.ldc, csz, cm0, cn0, M, N);
        tb.res = tile_rsrc(MODE == EPI_RES_F32 ? (const void*)e.residual : nullptr, e.ldr, 4, cm0, cn0, M, N);
        tb.aux = tile_rsrc((MODE == EPI_GELU_BF16 || MODE == EPI_DGELU_BF16 || MODE == EPI_DGELU_CS) ? e.aux : nullptr, e.ldaux, 2, cm0, cn0, M, N);
        f32x4 cs[2][2] = {{f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}}, {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}}};
        epilogue_wave64x128_h<MODE, (MODE == EPI_RES_F32 ? HCT_RES_POLICY : kNT), MT>(e, tb, patch, lane, cm0, cn0, wm * (16 * MT), wn * 128, M, N, acc, bh, cs);
        if (MODE == EPI_DGELU_CS) {  // lanes l, l+8, ..., l+56 hold 8 different rows of the same 8 columns
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                cs[h][u][q] += __shfl_xor(cs[h][u][q], 8, 64);
                cs[h][u][q] += __shfl_xor(cs[h][u][q], 16, 64);
                cs[h][u][q] += __shfl_xor(cs[h][u][q], 32, 64);
              }
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int n = cn0 + wn * 128 + h * 64 + (lane & 7) * 8;
            if (lane < 8 && n < N) {
              float* dst = colsum_out + ((int64_t)((cm0 >> 8) * 4 + wm)) * N + n;
              Vec4<float>::store(dst, cs[h][0]);
              Vec4<float>::store(dst + 4, cs[h][1]);
            }
          }
        }
      }
    }
    HCT_STAMP(3);
#ifdef HCT_STAMPS
    ++tile_i;
#endif
    if (!more) break;
  }
#ifdef HCT_STAMPS
  if (g_stamp_ptr && wave == 0 && blockIdx.x * 64 + lane < g_stamp_words) g_stamp_ptr[blockIdx.x * 64 + lane] = stamps;
#endif
}

// ---- NT, two workgroups per CU: 256x128 tile, 4 waves x (64x128), 3-stage ring (72 KiB) -----------------------------
// Same pipeline and epilogue as the 256x256 kernel, but sized so that TWO workgroups are resident on a CU (2 waves per
// SIMD in total): while one workgroup drains its tile (LDS patch -> buffer stores; vmcnt is in-order, so a wave cannot
// run ahead of its own stores) the other one keeps the matrix pipes busy.  Pays 1.5x the LDS-DMA bytes per FLOP of the
// 256x256 tile, so it is used for the short-K, wide-output GEMMs whose epilogue dominates (measured crossover in
// hct_gemm).  Ring of 3: stage t+3 reuses the buffer of stage t at the mid-stage barrier, after lgkmcnt(0).
template <int MODE>
__global__ void __launch_bounds__(256, 2) gemm_bf16_nt_w4_kernel(int M, int N, int K, const bf16* __restrict__ A, int64_t lda,
                                                                 const bf16* __restrict__ B, int64_t ldb, Epilogue e, int ntiles,
                                                                 int stagger) {
  constexpr int kStage = 24576;  // A 256 x 64 B | B 128 x 64 B
  __shared__ __attribute__((aligned(16))) unsigned char smem[3 * kStage];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ntm = (M + 255) >> 8, ntn = (N + 127) >> 7;

  uint32_t voa[4], vob[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 16 + (lane >> 2);
    voa[i] = (uint32_t)(row * lda * 2 + (((lane & 3) ^ swz64(row)) << 4));
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave * 2 + i) * 16 + (lane >> 2);
    vob[i] = (uint32_t)(row * ldb * 2 + (((lane & 3) ^ swz64(row)) << 4));
  }
  const int wm = wave;
  const int frow = lane & 15, fchk = lane >> 4;
  const int foff = frow * 64 + ((fchk ^ swz64(frow)) << 4);
  const int nk = K >> 5;

  __amdgpu_buffer_rsrc_t ra, rb;
  int m0 = 0, n0 = 0;
  auto set_tile = [&](int vb) {
    const int nwg = ntm * ntn;
    const int xcd = vb & 7, q = nwg >> 3, r = nwg & 7;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
    const int tm = id / ntn, tn = id - tm * ntn;
    m0 = tm << 8;
    n0 = tn << 7;
    ra = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (int64_t)m0 * lda), 0, clamp_records(((int64_t)(M - m0 - 1) * lda + K) * 2), 0x00020000);
    rb = __builtin_amdgcn_make_buffer_rsrc((void*)(B + (int64_t)n0 * ldb), 0, clamp_records(((int64_t)(N - n0 - 1) * ldb + K) * 2), 0x00020000);
  };
  auto bufof = [&](int t) -> unsigned char* { return smem + (t % 3) * kStage; };
  auto stage = [&](int t) {
    unsigned char* base = bufof(t);
    const uint32_t kb = (uint32_t)t * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(base + (wave * 4 + i) * 1024), 16, voa[i], kb, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void*)(base + 16384 + (wave * 2 + i) * 1024), 16, vob[i], kb, 0, 0);
  };
  f32x4 acc[4][8];
  bf16x8 b_lo[4], b_hi[4], a0[4], a1[4];
  auto rd_a = [&](int t, bf16x8* af) {
    const unsigned char* sa = bufof(t) + wm * 4096 + foff;
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sa + i * 1024);
  };
  auto rd_b = [&](int t, int half, bf16x8* bq) {
    const unsigned char* sb = bufof(t) + 16384 + half * 4096 + foff;
#pragma unroll
    for (int j = 0; j < 4; ++j) bq[j] = *reinterpret_cast<const bf16x8*>(sb + j * 1024);
  };
  auto mma = [&](int half, const bf16x8* af, const bf16x8* bq) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][half * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[j], af[i], acc[i][half * 4 + j], 0, 0, 0);
  };
  // own DMA retired (leaving `later` younger stages = 6 loads each in flight), every fragment read issued so far has
  // returned (so the buffer of the current stage may be refilled right after), then barrier
  auto land = [&](int later) {
    if (later >= 2) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
    else if (later == 1) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };

  // the two workgroups of a CU do identical work: start the second half of the grid half a tile late so that one's
  // epilogue falls into the other's main loop instead of both alternating in lockstep
  if (stagger > 0 && (int)blockIdx.x >= (int)(gridDim.x >> 1))
    for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(32);
  for (int vb = blockIdx.x; vb < ntiles; vb += gridDim.x) {
    set_tile(vb);
    stage(0);
    stage(1);
    stage(2);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    land(2);
    rd_a(0, a0);
    rd_b(0, 0, b_lo);
    int t = 0;
    for (; t + 4 < nk; t += 2) {
      rd_b(t, 1, b_hi);
      mma(0, a0, b_lo);
      land(1);
      stage(t + 3);
      rd_a(t + 1, a1);
      rd_b(t + 1, 0, b_lo);
      mma(1, a0, b_hi);
      rd_b(t + 1, 1, b_hi);
      mma(0, a1, b_lo);
      land(1);
      stage(t + 4);
      rd_a(t + 2, a0);
      rd_b(t + 2, 0, b_lo);
      mma(1, a1, b_hi);
    }
    rd_b(t, 1, b_hi);
    mma(0, a0, b_lo);
    land(1);
    stage(t + 3);
    rd_a(t + 1, a1);
    rd_b(t + 1, 0, b_lo);
    mma(1, a0, b_hi);
    rd_b(t + 1, 1, b_hi);
    mma(0, a1, b_lo);
    land(1);
    rd_a(t + 2, a0);
    rd_b(t + 2, 0, b_lo);
    mma(1, a1, b_hi);
    rd_b(t + 2, 1, b_hi);
    mma(0, a0, b_lo);
    land(0);
    rd_a(t + 3, a1);
    rd_b(t + 3, 0, b_lo);
    mma(1, a0, b_hi);
    rd_b(t + 3, 1, b_hi);
    mma(0, a1, b_lo);
    mma(1, a1, b_hi);

    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // ring free: reuse it for the epilogue patches (4 waves x 8 KiB)
    {
      unsigned char* patch = smem + wave * 8192;
      TileBufs tb;
      if (MODE != EPI_GENERIC) {
        const int csz = (MODE == EPI_RES_F32 || MODE == EPI_PLAIN_F32) ? 4 : 2;
        tb.c = tile_rsrc(e.C, e.ldc, csz, m0, n0, M, N);
        tb.res = tile_rsrc(MODE == EPI_RES_F32 ? (const void*)e.residual : nullptr, e.ldr, 4, m0, n0, M, N);
        tb.aux = tile_rsrc((MODE == EPI_GELU_BF16 || MODE == EPI_DGELU_BF16) ? e.aux : nullptr, e.ldaux, 2, m0, n0, M, N);
      }
      if (MODE == EPI_GENERIC) {
#pragma unroll
        for (int i = 0; i < 4; ++i) epilogue_tile16x128(e, patch, lane, m0 + wm * 64 + i * 16, n0, M, N, acc[i]);
      } else {
        TileBias bv = tile_bias<MODE>(e, n0, 0, lane, N);
        f32x4 cs0 = {0, 0, 0, 0}, cs1 = {0, 0, 0, 0};
        epilogue_wave64x128_m<MODE>(e, tb, patch, lane, m0, n0, wm * 64, 0, M, N, acc, bv, cs0, cs1);
      }
    }
    __syncthreads();  // patches dead before the next tile's DMA overwrites the ring
  }
}

// ---- TN, 256x256 tile: C[M,N] (+)= A[R,M]^T . B[R,N]  (wgrad), same ring / pipeline / epilogue as the NT kernel ------
// LDS stage = A[32 r][256 m] | B[32 r][256 n] bf16 (512-B rows = whole lines per DMA piece of 2 rows); the 32-B block nb of
// row r sits at block  nb ^ f(r),  f(r) = (r&3) | ((r>>3)&1)<<2, which makes the ds_read_b64_tr_b16 fragment reads
// conflict-free.  Work item = (split over R, tile); every split reduces r_chunk rows (zero-filled past R) and writes an
// fp32 partial (or the final C when splits == 1); a fold kernel adds the partials in fixed order.
__global__ void __launch_bounds__(512, 2) gemm_bf16_tn256_kernel(int M, int N, int R, int r_chunk, const bf16* __restrict__ A,
                                                                 int64_t lda, const bf16* __restrict__ B, int64_t ldb,
                                                                 float* __restrict__ slab, Epilogue e, int ntiles, int splits,
                                                                 unsigned int* __restrict__ counters) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[163840];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ntm = (M + 255) >> 8, ntn = (N + 255) >> 8, nmn = ntm * ntn;
  const int wm = wave >> 1, wn = wave & 1;
  const int nk = r_chunk >> 5;
  const uint32_t OOB = 0xFFFFFFF0u;

  // staging: 1 KiB piece = 2 reduction rows x 512 B; wave w moves pieces 2w, 2w+1 of A and of B each stage
  uint32_t voa[2], vob[2];
  // The operand stream of this kernel is issued as INLINE-ASM LDS-DMA.  With the builtin, hipcc (ROCm 7.2) inserts
  // s_waitcnt vmcnt(0) between a stage's DMA issue and the ds_read_b64_tr_b16 fragment reads (it treats the transposed
  // read as aliasing every pending LDS-DMA), which drains the whole ring every stage: the kernel then runs at one DMA
  // round trip per 32-row stage (~1.4 us instead of ~0.45).  Ordering is enforced by the counted vmcnt waits + barriers
  // below; compiler-generated vmcnt waits for its own loads/stores only become more conservative.
  i32x4 ra, rb;  // buffer descriptors {base_lo, base_hi, num_records, flags} in SGPRs
  const uint32_t lds0 = (uint32_t)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);
  int m0 = 0, n0 = 0, sp = 0;
  auto set_tile = [&](int vb) {
    const int xcd = vb & 7, q = ntiles >> 3, r = ntiles & 7;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
    sp = id / nmn;
    const int rem = id - sp * nmn;
    const int tm = rem / ntn, tn = rem - tm * ntn;
    m0 = tm << 8;
    n0 = tn << 8;
    const int rbeg = sp * r_chunk;
    const int rows = min(R, rbeg + r_chunk) - rbeg;
    const bf16* Ab = A + (int64_t)rbeg * lda + m0;
    const bf16* Bb = B + (int64_t)rbeg * ldb + n0;
    ra = make_srd(Ab, clamp_records(((int64_t)(rows - 1) * lda + (M - m0)) * 2));
    rb = make_srd(Bb, clamp_records(((int64_t)(rows - 1) * ldb + (N - n0)) * 2));
    // lane offsets recomputed per tile from an opaque lane id (a dozen VALU): as tile-invariant values hipcc kept them in
    // VGPRs across the main loop, spilled them, and reloaded them here one by one with s_waitcnt vmcnt(0)
    int l = lane;
    asm volatile("" : "+v"(l));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int srow = (wave * 2 + i) * 2 + (l >> 5);
      const int slot = l & 31;
      const int f = (srow & 3) | (((srow >> 3) & 1) << 2);
      const int scol = ((((slot >> 1) ^ f) << 1) | (slot & 1)) * 8;
      voa[i] = (m0 + scol < M) ? (uint32_t)((srow * lda + scol) * 2) : OOB;
      vob[i] = (n0 + scol < N) ? (uint32_t)((srow * ldb + scol) * 2) : OOB;
    }
  };
  auto stage = [&](int t) {
    const uint32_t base = lds0 + (t & 3) * 32768;
    uint32_t ka = (uint32_t)(t * 32 * lda * 2), kb = (uint32_t)(t * 32 * ldb * 2);
    // opaque per call: otherwise hipcc pre-adds the stage offsets of the three prefetch stages into 12 long-lived VGPRs,
    // spills them and reloads each with s_waitcnt vmcnt(0) between the prefetch DMAs
    asm volatile("" : "+s"(ka), "+s"(kb));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = wave * 2 + i;
      // NOTE: the reduction-row offset must stay in voffset here (rows past the split's end are zero-filled by the
      // descriptor's range check, which does not see soffset)
      dma16(ra, base + c * 1024, voa[i] == OOB ? OOB : voa[i] + ka);
      dma16(rb, base + 16384 + c * 1024, vob[i] == OOB ? OOB : vob[i] + kb);
    }
  };
  const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
  const int ff = qq | ((g & 1) << 2);
  const int rowb = (8 * g + qq) * 512 + pp * 8;
  auto frag = [&](const unsigned char* tile, int nb) -> bf16x8 {
    const unsigned char* pa = tile + rowb + ((nb ^ ff) << 5);
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pa));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pa + 4 * 512));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  f32x4 acc[4][8];
  bf16x8 b_lo[4], b_hi[4], a0[4], a1[4];
  auto rd_a = [&](int t, bf16x8* af) {
    const unsigned char* sa = smem + (t & 3) * 32768;
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = frag(sa, wm * 4 + i);
  };
  auto rd_b = [&](int t, int half, bf16x8* bq) {
    const unsigned char* sb = smem + (t & 3) * 32768 + 16384;
#pragma unroll
    for (int j = 0; j < 4; ++j) bq[j] = frag(sb, wn * 8 + half * 4 + j);
  };
  auto mma = [&](int half, const bf16x8* af, const bf16x8* bq) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][half * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[j], af[i], acc[i][half * 4 + j], 0, 0, 0);
  };
  // half of a half-stage (rows 32*part .. 32*part+31 of the wave tile): lets the 16 transposed reads of the next stage be
  // issued as 8 + 8 around it, so no wait ever needs more than the 15 outstanding LDS ops lgkmcnt can express
  auto mma_part = [&](int half, int part, const bf16x8* af, const bf16x8* bq) {
#pragma unroll
    for (int i = 2 * part; i < 2 * part + 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][half * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[j], af[i], acc[i][half * 4 + j], 0, 0, 0);
  };
  auto land = [&](int later) {
    if (later >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };

  int vb = blockIdx.x;
  set_tile(vb);
  stage(0);
  stage(1);
  stage(2);
  while (true) {
    const int cm0 = m0, cn0 = n0, csp = sp;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    land(2);
    rd_a(0, a0);
    rd_b(0, 0, b_lo);
    int t = 0;
    for (; t + 4 < nk; t += 2) {
      rd_b(t, 1, b_hi);
      mma(0, a0, b_lo);
      land(1);
      stage(t + 3);
      rd_a(t + 1, a1);
      mma_part(1, 0, a0, b_hi);
      rd_b(t + 1, 0, b_lo);
      mma_part(1, 1, a0, b_hi);
      rd_b(t + 1, 1, b_hi);
      mma(0, a1, b_lo);
      land(1);
      stage(t + 4);
      rd_a(t + 2, a0);
      mma_part(1, 0, a1, b_hi);
      rd_b(t + 2, 0, b_lo);
      mma_part(1, 1, a1, b_hi);
    }
    rd_b(t, 1, b_hi);
    mma(0, a0, b_lo);
    land(1);
    stage(t + 3);
    rd_a(t + 1, a1);
    mma_part(1, 0, a0, b_hi);
    rd_b(t + 1, 0, b_lo);
    mma_part(1, 1, a0, b_hi);
    rd_b(t + 1, 1, b_hi);
    mma(0, a1, b_lo);
    land(1);
    rd_a(t + 2, a0);
    mma_part(1, 0, a1, b_hi);
    rd_b(t + 2, 0, b_lo);
    mma_part(1, 1, a1, b_hi);
    rd_b(t + 2, 1, b_hi);
    mma(0, a0, b_lo);
    land(0);
    rd_a(t + 3, a1);
    mma_part(1, 0, a0, b_hi);
    rd_b(t + 3, 0, b_lo);
    mma_part(1, 1, a0, b_hi);
    rd_b(t + 3, 1, b_hi);
    mma(0, a1, b_lo);
    mma(1, a1, b_hi);

    __builtin_amdgcn_s_barrier();
    TileBias bv = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    if (!slab) bv = tile_bias<EPI_PLAIN_F32>(e, cn0, wn * 128, lane, N);
    vb += gridDim.x;
    const bool more = vb < ntiles;
    if (more) {
      set_tile(vb);
      stage(0);
      stage(1);
      stage(2);
    }
    {
      unsigned char* patch = smem + 3 * 32768 + wave * 8192;
      Epilogue eo = e;
      if (slab) {  // raw fp32 partial of this split; the fold kernel applies alpha and the output dtype
        eo.bias = nullptr; eo.alpha = 1.0f;
        eo.C = slab + (int64_t)csp * M * N; eo.ldc = N;
      }
      TileBufs tb;
      tb.c = tile_rsrc(eo.C, eo.ldc, 4, cm0, cn0, M, N);
      tb.res = tile_rsrc(nullptr, 0, 4, cm0, cn0, M, N);
      tb.aux = tb.res;
      f32x4 cs0 = {0, 0, 0, 0}, cs1 = {0, 0, 0, 0};
      // partial slabs leave write-through (sc1): they are handed to other workgroups below, and a write-through store needs
      // no release fence (publishing 256 KB of plain stores with buffer_wbl2 costs several us per workgroup)
      // (with the separate fold kernel -- the default -- the kernel boundary publishes them: policy HCT_SLAB_POLICY)
      if (slab && counters) epilogue_wave64x128_m<EPI_PLAIN_F32, 16>(eo, tb, patch, lane, cm0, cn0, wm * 64, wn * 128, M, N, acc, bv, cs0, cs1);
      else if (slab) epilogue_wave64x128_m<EPI_PLAIN_F32, HCT_SLAB_POLICY>(eo, tb, patch, lane, cm0, cn0, wm * 64, wn * 128, M, N, acc, bv, cs0, cs1);
      else epilogue_wave64x128_m<EPI_PLAIN_F32>(eo, tb, patch, lane, cm0, cn0, wm * 64, wn * 128, M, N, acc, bv, cs0, cs1);
    }
    if (slab && counters) {  // publish this split's partial: every wave's stores have left, then ONE arrival on the tile's counter
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_fetch_add(counters + (cm0 >> 8) * ntn + (cn0 >> 8), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!more) break;
  }
  if (!slab || !counters) return;
  // ---- split fold inside the launch ------------------------------------------------------------------------------------
  // All splits of a tile are co-resident (grid <= #CUs, one 160-KiB workgroup per CU) or queued behind workgroups that never
  // wait before publishing, so waiting for the tile's arrival count cannot deadlock: a workgroup publishes ALL its items
  // first and only then takes up its fold duties.  Every one of the tile's `splits` workgroups then sums 1/splits of the
  // tile's rows over the slabs in split order 0, 1, 2, ... (fixed order: bit-reproducible whatever the arrival order) and
  // writes the final fp32 C.  Slab bytes were stored write-through (sc1) and drained before the arrival; the consumer
  // polls relaxed, then ONE agent-scope acquire drops its L1 lines before the plain loads: placement-independent
  // (MI355X_MICROARCH.md, "Valid forms").
  unsigned int* const done = counters + 64;
  for (int vb2 = blockIdx.x; vb2 < ntiles; vb2 += gridDim.x) {
    const int xcd = vb2 & 7, q = ntiles >> 3, r = ntiles & 7;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb2 >> 3);
    const int fsp = id / nmn, tile = id - fsp * nmn;
    const int tm = tile / ntn, tn = tile - tm * ntn;
    int& s_timeout = *reinterpret_cast<int*>(smem);  // the ring is idle now (every wave is past the publish barrier)
    if (threadIdx.x == 0) {
      unsigned spins = 0;
      while (__hip_atomic_load(counters + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)splits && spins < (1u << 20)) {
        __builtin_amdgcn_s_sleep(8);
        ++spins;
      }
      s_timeout = spins >= (1u << 20);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // drop this CU's stale L1 lines of the slabs
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const bool fold_bad = s_timeout != 0;  // a split never arrived (cannot happen with a resident grid): flag it, poison this share with NaN
    if (fold_bad && threadIdx.x == 0) __hip_atomic_store(counters + 128, 0xDEADu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // this workgroup's share of the tile: rows [r0, r1)
    const int r0 = (fsp * 256) / splits, r1 = ((fsp + 1) * 256) / splits;
    const int cols4 = 64;  // 256 columns as float4
    const float* tile_base = slab + ((int64_t)tm * 256) * N + tn * 256;
    const int64_t zstride = (int64_t)M * N;
    float* Cf = (float*)e.C;
    // 8 outputs x 4 splits = 32 independent 16-B loads in flight per thread: the slabs come from the Infinity Cache / another
    // XCD's L2 at ~2 us per round trip, so a thread that waits for 4 loads at a time spends 8 round trips on its share
    const int nelem = (r1 - r0) * cols4;
    for (int g0 = 0; g0 < nelem; g0 += 512 * 8) {
      const float* src[8];
      float* dst[8];
      f32x4 sum[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int i = g0 + k * 512 + (int)threadIdx.x;
        const int rr = r0 + i / cols4, c4 = (i % cols4) * 4;
        const int m = tm * 256 + rr, n = tn * 256 + c4;
        const bool ok = i < nelem && m < M && n < N;
        src[k] = ok ? tile_base + (int64_t)rr * N + c4 : nullptr;
        dst[k] = ok ? Cf + (int64_t)m * e.ldc + n : nullptr;
        sum[k] = f32x4{0, 0, 0, 0};
      }
      for (int z = 0; z < splits; z += 4) {
        f32x4 v[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int k = 0; k < 8; ++k)
            v[u][k] = (src[k] && z + u < splits) ? *reinterpret_cast<const f32x4*>(src[k] + (int64_t)(z + u) * zstride) : f32x4{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 4; ++u)  // split order 0, 1, 2, ... per output (adding the zero of an absent split changes nothing)
#pragma unroll
          for (int k = 0; k < 8; ++k) sum[k] += v[u][k];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (dst[k]) *reinterpret_cast<f32x4*>(dst[k]) = fold_bad ? f32x4{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")} : sum[k] * e.alpha;
    }
    // last one out re-arms the tile's counters for the next launch on this workspace
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned prev = __hip_atomic_fetch_add(done + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (prev == (unsigned)splits - 1) {
        __hip_atomic_store(done + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(counters + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// ---- TN, GROUPED: many weight-gradient products in ONE persistent launch, no split over R --------------------------------
// The weight gradients are not on the backward's dependency chain (only the optimizer reads them), so the model driver collects the
// wgrads of several blocks and runs them here together.  With enough output tiles to fill the chip there is no need to split the
// reduction: no fp32 partial slabs (65 MB per product before), no fold launch, one prologue / epilogue per 256x256 tile of the
// FULL reduction instead of one per split, two launches per step instead of 166.
//   tiles    : T tiles in all, listed as SEGMENTS (job, first tile, count) that the host orders so that a window of 32 consecutive
//              tile ids is, where it can be, 32 tiles of ONE product (tn_group_segments); tile t of a job = (t / ntn, t % ntn);
//              G = grid = #CUs.
//   rounds   : the first F = T / G rounds are whole tiles; workgroup c takes tile r * G + q(c) in round r, q(c) = (c & 7) * G/8 +
//              (c >> 3): the 32 workgroups of an XCD (dealt round-robin) sweep the reduction rows of one window together, so the
//              A / B row panels of its product are fetched once per XCD (3 + 12 panels for 32 tiles of a 768 x 3072 gradient;
//              the kernel draws 4.4 TB/s as it is -- windows that mix two products measured 9 % slower).
//   remainder: the last Rm = T - F * G tiles are split over the reduction into `ns_split` pieces each (units of 4 stages, dealt
//              evenly), chosen on the host so that Rm * ns_split fills whole rounds; piece p = split * Rm + tile, workgroup q takes
//              pieces q, q + G, ...: a window of 32 consecutive pieces is the SAME reduction range of 32 consecutive tiles, shared
//              through L2 like a whole-tile window (contiguous per-workgroup stage ranges -- the first form of this kernel -- left
//              every workgroup streaming panels of its own: 25 % slower).  Pieces of splits 0 .. ns_split-2 are FOLLOWERS (raw
//              accumulators to slab p, then flag p); the LAST split's piece OWNS the tile: it adds the slabs in split order (fixed:
//              bit-reproducible) and runs the epilogue.  Every workgroup meets its follower pieces before its owner pieces and a
//              follower never waits: no cycles.  A bounded spin poisons the tile with NaN (the engine's finite-loss check then
//              stops the run) instead of hanging should the grid not be resident.
struct TnJob {                 // 80 bytes, device copy written by tn_group_table_kernel
  const bf16* A; const bf16* B; float* C;
  int M, N, R, lda, ldb, ldc;
  int tile0, ntiles, ntn, nk;  // (tile0 unused by the kernel), tiles, column tiles, stages per tile (R / 32 rounded up to a multiple of 4)
  float alpha; int pad[3];
};
struct TnSeg { int job, tile_first, count, gtile0; };  // tiles [tile_first, tile_first + count) of `job` have the ids gtile0 ..
constexpr int kTnGroupChunk = 32;  // jobs per table-writer launch (kernel arguments stay under 4 KiB)
struct TnJobChunk { TnJob j[kTnGroupChunk]; };
__global__ void tn_group_table_kernel(TnJob* __restrict__ dst, TnJobChunk c, int first, int n) {
  const int i = threadIdx.x;
  if (i < n) dst[first + i] = c.j[i];
}
constexpr int kTnSegChunk = 192;
struct TnSegChunk { TnSeg s[kTnSegChunk]; };
__global__ void tn_group_seg_kernel(TnSeg* __restrict__ dst, TnSegChunk c, int first, int n) {
  const int i = threadIdx.x;
  if (i < n) dst[first + i] = c.s[i];
}
// stream-K region of the grouped kernel: [4 KiB: error word | 12 KiB: one flag per follower piece | slabs]
constexpr int kTnMaxFollowers = 1024;
constexpr size_t kTnSkHeadBytes = 16384;
constexpr size_t kTnSkBytes = kTnSkHeadBytes + (size_t)kTnMaxFollowers * kSkSlabBytes;

__global__ void __launch_bounds__(512, 2) gemm_bf16_tn_group_kernel(const TnJob* __restrict__ jobs, const TnSeg* __restrict__ segs, int nsegs,
                                                                    int T, int F, int ns_split, unsigned char* __restrict__ sk_ws, unsigned sk_seq) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[163840];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const uint32_t OOB = 0xFFFFFFF0u;
  const int G = gridDim.x;
  const int q = __builtin_amdgcn_readfirstlane((G & 7) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3));  // XCD-contiguous index
  const int Rm = T - F * G;  // remainder tiles

  uint32_t voa[2], vob[2];
  i32x4 ra, rb;
  const uint32_t lds0 = (uint32_t)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);
  // ---- item = (job, tile, first stage, stages, kind) ----------------------------------------------------------------------
  int sc = 0;      // segment cursor
  int round = 0;   // whole-tile rounds done
  int rpiece = q;  // next remainder piece of this workgroup
  // the item being SET UP (next to compute): what the main loop needs stays in registers (lda, ldb, descriptors, lane offsets,
  // stage count); what only the epilogue needs is re-read from the job table then -- (job, tile, kind, piece) is all that is
  // carried across the main loop, where the register file is full
  int lda = 0, ldb = 0, ns = 0, nx_j = 0, nx_t = 0, nx_info = 0;
  // (values read from the tables are wave-uniform by construction; readfirstlane tells the compiler so -- they end up in scalar
  //  operands of the DMA descriptors and stage offsets)
  auto rfl = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
  auto find_seg = [&](int g) {  // segment that holds tile id g (ids mostly grow along a workgroup's walk: scan from the cursor)
    if (g < rfl(segs[sc].gtile0)) sc = 0;
    while (g >= rfl(segs[sc].gtile0) + rfl(segs[sc].count)) ++sc;
  };
  auto set_item = [&](int j, int t, int s0, int nst, int info) {  // tile t of job j, stages [s0, s0 + nst)
    TnJob jb = jobs[j];
    jb.lda = rfl(jb.lda); jb.ldb = rfl(jb.ldb); jb.ntn = rfl(jb.ntn); jb.M = rfl(jb.M); jb.N = rfl(jb.N); jb.R = rfl(jb.R);
    lda = jb.lda; ldb = jb.ldb;
    const int tm = rfl(t / jb.ntn), tn = t - tm * jb.ntn;
    const int m0 = tm << 8, n0 = tn << 8;
    ns = nst; nx_j = j; nx_t = t; nx_info = info;
    const int rbeg = s0 * 32;
    const int rows = (jb.R < rbeg + nst * 32 ? jb.R : rbeg + nst * 32) - rbeg;  // (may be <= 0 for the last pieces of a short reduction: all zero-fill)
    const bf16* Ab = jb.A + (int64_t)rbeg * lda + m0;
    const bf16* Bb = jb.B + (int64_t)rbeg * ldb + n0;
    ra = make_srd(Ab, clamp_records(rows > 0 ? ((int64_t)(rows - 1) * lda + (jb.M - m0)) * 2 : 0));
    rb = make_srd(Bb, clamp_records(rows > 0 ? ((int64_t)(rows - 1) * ldb + (jb.N - n0)) * 2 : 0));
    int l = lane;
    asm volatile("" : "+v"(l));  // lane offsets recomputed per item (see gemm_bf16_tn256_kernel)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int srow = (wave * 2 + i) * 2 + (l >> 5);
      const int slot = l & 31;
      const int f = (srow & 3) | (((srow >> 3) & 1) << 2);
      const int scol = ((((slot >> 1) ^ f) << 1) | (slot & 1)) * 8;
      voa[i] = (m0 + scol < jb.M) ? (uint32_t)((srow * lda + scol) * 2) : OOB;
      vob[i] = (n0 + scol < jb.N) ? (uint32_t)((srow * ldb + scol) * 2) : OOB;
    }
  };
  // info word of an item: bits 0-1 kind (0 whole tile, 1 follower piece, 2 owner piece), bits 2.. the piece id
  auto next_item = [&]() -> bool {
    int it_job, it_tile, it_s0, it_ns, it_info;
    if (round < F) {  // whole tile of round `round`
      const int g = round * G + q;
      ++round;
      find_seg(g);
      const TnSeg sg = segs[sc];
      it_job = rfl(sg.job);
      it_tile = rfl(sg.tile_first) + (g - rfl(sg.gtile0));
      it_s0 = 0;
      it_ns = rfl(jobs[it_job].nk);
      it_info = 0;
    } else {
      if (rpiece >= Rm * ns_split) return false;
      const int p = rfl(rpiece);
      rpiece = rfl(rpiece + G);
      const int sp = rfl(p / Rm), tr = p - sp * Rm;  // split sp of remainder tile tr
      find_seg(F * G + tr);
      const TnSeg sg = segs[sc];
      it_job = rfl(sg.job);
      it_tile = rfl(sg.tile_first) + (F * G + tr - rfl(sg.gtile0));
      const int units = rfl(jobs[it_job].nk) >> 2;   // units of 4 stages, dealt evenly over the splits (host: units >= ns_split)
      const int u0 = rfl(sp * units / ns_split), u1 = rfl((sp + 1) * units / ns_split);  // (at most 16 splits, under 2^20 units: 32-bit)
      it_s0 = u0 * 4;
      it_ns = (u1 - u0) * 4;
      it_info = (ns_split == 1 ? 0 : (sp + 1 == ns_split ? 2 : 1)) | (p << 2);
    }
    set_item(rfl(it_job), rfl(it_tile), rfl(it_s0), rfl(it_ns), rfl(it_info));
    return true;
  };
  auto stage = [&](int t) {
    const uint32_t base = lds0 + (t & 3) * 32768;
    uint32_t ka = (uint32_t)rfl(t * 32 * lda * 2), kb = (uint32_t)rfl(t * 32 * ldb * 2);
    asm volatile("" : "+s"(ka), "+s"(kb));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = wave * 2 + i;
      dma16(ra, base + c * 1024, voa[i] == OOB ? OOB : voa[i] + ka);
      dma16(rb, base + 16384 + c * 1024, vob[i] == OOB ? OOB : vob[i] + kb);
    }
  };
  const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
  const int ff = qq | ((g & 1) << 2);
  const int rowb = (8 * g + qq) * 512 + pp * 8;
  auto frag = [&](const unsigned char* tile, int nb) -> bf16x8 {
    const unsigned char* pa = tile + rowb + ((nb ^ ff) << 5);
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pa));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pa + 4 * 512));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  f32x4 acc[4][8];
  bf16x8 b_lo[4], b_hi[4], a0[4], a1[4];
  auto rd_a = [&](int t, bf16x8* af) {
    const unsigned char* sa = smem + (t & 3) * 32768;
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = frag(sa, wm * 4 + i);
  };
  auto rd_b = [&](int t, int half, bf16x8* bq) {
    const unsigned char* sb = smem + (t & 3) * 32768 + 16384;
#pragma unroll
    for (int j = 0; j < 4; ++j) bq[j] = frag(sb, wn * 8 + half * 4 + j);
  };
  auto mma = [&](int half, const bf16x8* af, const bf16x8* bq) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][half * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[j], af[i], acc[i][half * 4 + j], 0, 0, 0);
  };
  auto mma_part = [&](int half, int part, const bf16x8* af, const bf16x8* bq) {
#pragma unroll
    for (int i = 2 * part; i < 2 * part + 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][half * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[j], af[i], acc[i][half * 4 + j], 0, 0, 0);
  };
  auto land = [&](int later) {
    if (later >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };

#ifdef HCT_PRIO_YOUNG  /* experiment: static priority for the second-dispatched half of the waves (MI355X_MICROARCH.md, two waves per SIMD, item 4) */
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
  if (!rfl((int)next_item())) return;
  stage(0);
  stage(1);
  stage(2);
  while (true) {
    // the item being computed (next_item below overwrites the set-up variables)
    const int cur_j = nx_j, cur_t = nx_t, cns = ns, ckind = nx_info & 3, cpiece = nx_info >> 2;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    land(2);
    rd_a(0, a0);
    rd_b(0, 0, b_lo);
    int t = 0;
    for (; t + 4 < cns; t += 2) {
      rd_b(t, 1, b_hi);
      mma(0, a0, b_lo);
      land(1);
#if HCT_STAGGER_DMA  /* waves 4 - 7 issue their DMA one MFMA group behind their SIMD partners (see gemm_bf16_nt256_kernel) */
      if (wave < 4) stage(t + 3);
      __builtin_amdgcn_sched_barrier(0);
#else
      stage(t + 3);
#endif
      rd_a(t + 1, a1);
      mma_part(1, 0, a0, b_hi);
      rd_b(t + 1, 0, b_lo);
      mma_part(1, 1, a0, b_hi);
#if HCT_STAGGER_DMA
      __builtin_amdgcn_sched_barrier(0);
      if (wave >= 4) stage(t + 3);
#endif
      rd_b(t + 1, 1, b_hi);
      mma(0, a1, b_lo);
      land(1);
#if HCT_STAGGER_DMA
      if (wave < 4) stage(t + 4);
      __builtin_amdgcn_sched_barrier(0);
#else
      stage(t + 4);
#endif
      rd_a(t + 2, a0);
      mma_part(1, 0, a1, b_hi);
      rd_b(t + 2, 0, b_lo);
      mma_part(1, 1, a1, b_hi);
#if HCT_STAGGER_DMA
      __builtin_amdgcn_sched_barrier(0);
      if (wave >= 4) stage(t + 4);
#endif
    }
    rd_b(t, 1, b_hi);
    mma(0, a0, b_lo);
    land(1);
    stage(t + 3);
    rd_a(t + 1, a1);
    mma_part(1, 0, a0, b_hi);
    rd_b(t + 1, 0, b_lo);
    mma_part(1, 1, a0, b_hi);
    rd_b(t + 1, 1, b_hi);
    mma(0, a1, b_lo);
    land(1);
    rd_a(t + 2, a0);
    mma_part(1, 0, a1, b_hi);
    rd_b(t + 2, 0, b_lo);
    mma_part(1, 1, a1, b_hi);
    rd_b(t + 2, 1, b_hi);
    mma(0, a0, b_lo);
    land(0);
    rd_a(t + 3, a1);
    mma_part(1, 0, a0, b_hi);
    rd_b(t + 3, 0, b_lo);
    mma_part(1, 1, a0, b_hi);
    rd_b(t + 3, 1, b_hi);
    mma(0, a1, b_lo);
    mma(1, a1, b_hi);

    __builtin_amdgcn_s_barrier();  // every wave has its last fragments: the ring is free
    const bool more = rfl((int)next_item()) != 0;
    if (more) {
      stage(0);
      stage(1);
      stage(2);
    }
    if (__builtin_expect(ckind == 1, 0)) {
      // FOLLOWER piece: raw accumulators in register order (1 KiB per store instruction), write-through, then ONE flag
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc(sk_ws + kTnSkHeadBytes + (size_t)cpiece * kSkSlabBytes, 0, (uint32_t)kSkSlabBytes, 0x00020000);
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const uint32_t off0 = (uint32_t)(wave * 32768 + ln * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rs, off0 + (uint32_t)((i * 8 + j) * 1024), 0, 16);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0)
        __hip_atomic_store((unsigned*)(sk_ws + 4096) + cpiece, (sk_pub(sk_seq) << 4) | xcc_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (__builtin_expect(ckind == 2, 0)) {
        // OWNER (last split of its tile): add the partials of splits 0, 1, ... (protocol of the NT stream-K)
        int& s_bad = *reinterpret_cast<int*>(smem + 3 * 32768 + 65536 - 16);  // (last bytes of the patch area: not written before the epilogue)
        const int tr = cpiece - (ns_split - 1) * Rm;
        for (int c2 = tr; c2 < cpiece; c2 += Rm) {  // pieces of the same tile, splits 0 .. ns_split - 2
          if (threadIdx.x == 0) {
            unsigned spins = 0, f;
            while (((f = __hip_atomic_load((unsigned*)(sk_ws + 4096) + c2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 4) != (sk_seq & 0x0FFFFFFFu) &&
                   spins < (1u << 22)) {
              __builtin_amdgcn_s_sleep(8);
              ++spins;
            }
            s_bad = spins >= (1u << 22);
            if (s_bad) __hip_atomic_store((unsigned*)sk_ws + kSkErrWord, 0xDEADu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((f & 15) != xcc_id()) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          }
          __syncthreads();
          const bool bad = s_bad != 0;
          const __amdgpu_buffer_rsrc_t rs =
              __builtin_amdgcn_make_buffer_rsrc(sk_ws + kTnSkHeadBytes + (size_t)c2 * kSkSlabBytes, 0, (uint32_t)kSkSlabBytes, 0x00020000);
          int ln = lane;
          asm volatile("" : "+v"(ln));
          const uint32_t off0 = (uint32_t)(wave * 32768 + ln * 16);
#pragma unroll
          for (int i = 0; i < 4; ++i) {  // 8 loads (8 KiB per wave) in flight: once per workgroup and launch, registers matter more here
            f32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
              v[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off0 + (uint32_t)((i * 8 + j) * 1024), 0, 16));
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] += v[j];
          }
          acc[0][0] = acc[0][0] * (bad ? __builtin_nanf("") : 1.0f);  // a partial never arrived: poison the tile (the finite-loss check of the engine stops the run)
          __syncthreads();  // s_bad is rewritten by the next follower's poll
        }
      }
      unsigned char* patch = smem + 3 * 32768 + wave * 8192;
      TnJob cj = jobs[cur_j];
      cj.ntn = rfl(cj.ntn); cj.M = rfl(cj.M); cj.N = rfl(cj.N); cj.ldc = rfl(cj.ldc);
      const int ctm = cur_t / cj.ntn, cm0 = ctm << 8, cn0 = (cur_t - ctm * cj.ntn) << 8, cM = cj.M, cN = cj.N;
      Epilogue eo;
      eo.bias = nullptr; eo.residual = nullptr; eo.ldr = 0; eo.act = HCT_ACT_NONE; eo.aux = nullptr; eo.aux_dtype = HCT_F32; eo.ldaux = 0;
      eo.C = cj.C; eo.c_dtype = HCT_F32; eo.ldc = cj.ldc; eo.C2 = nullptr; eo.c2_dtype = HCT_F32; eo.ldc2 = 0; eo.alpha = cj.alpha;
      eo.colsum_partial = nullptr; eo.aux_deriv = 0;
      TileBufs tb;
      tb.c = tile_rsrc(eo.C, eo.ldc, 4, cm0, cn0, cM, cN);
      tb.res = tile_rsrc(nullptr, 0, 4, cm0, cn0, cM, cN);
      tb.aux = tb.res;
      const TileBias bv = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
      f32x4 cs0 = {0, 0, 0, 0}, cs1 = {0, 0, 0, 0};
      epilogue_wave64x128_m<EPI_PLAIN_F32>(eo, tb, patch, lane, cm0, cn0, wm * 64, wn * 128, cM, cN, acc, bv, cs0, cs1);
    }
    if (!more) break;
  }
}

// ---- TN: A stored [R,M] (lda), B stored [R,N] (ldb); C[M,N] = sum_r A[r,m] B[r,n] ------------------------------
// grid.x = tiles, grid.y = splits over R.  splits > 1: fp32 partials to slab[split][M][N].
__global__ void __launch_bounds__(256, 2) gemm_bf16_tn_kernel(int M, int N, int R, int r_chunk, const bf16* __restrict__ A,
                                                              int64_t lda, const bf16* __restrict__ B, int64_t ldb,
                                                              float* __restrict__ slab, Epilogue e) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[65536];  // [buf][A 16K | B 16K], tiles [64 r][128 cols]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ntm = (M + 127) >> 7, ntn = (N + 127) >> 7;
  int tm, tn;
  tile_ids(ntm, ntn, tm, tn);
  const int m0 = tm << 7, n0 = tn << 7;
  const int rbeg = blockIdx.y * r_chunk;
  const int rend = min(R, rbeg + r_chunk);

  const bf16* Ab = A + (int64_t)rbeg * lda;
  const bf16* Bb = B + (int64_t)rbeg * ldb;
  const uint32_t reca = clamp_records(((int64_t)(rend - rbeg - 1) * lda + M) * 2);
  const uint32_t recb = clamp_records(((int64_t)(rend - rbeg - 1) * ldb + N) * 2);
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, reca, 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, recb, 0x00020000);

  // staging: chunk c = 4w+i = 4 reduction rows x 256 B.  lane -> (r = 4c + lane/16, LDS 16-B slot = lane%16).
  // 32-B block nb of the row is stored at block nb ^ f(r), f(r) = (r&3) | ((r>>3)&1)<<2 (conflict-free tr reads).
  uint32_t voa[4], vob[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = wave * 4 + i;
    const int r = c * 4 + (lane >> 4);
    const int slot = lane & 15;
    const int f = (r & 3) | (((r >> 3) & 1) << 2);
    const int col = ((((slot >> 1) ^ f) << 1) | (slot & 1)) * 8;  // element column inside the 128-wide tile
    voa[i] = (m0 + col < M) ? (uint32_t)(r * lda * 2 + (m0 + col) * 2) : 0xFFFFFFF0u;
    vob[i] = (n0 + col < N) ? (uint32_t)(r * ldb * 2 + (n0 + col) * 2) : 0xFFFFFFF0u;
  }
  auto stage = [&](int buf, int r0) {  // r0 relative to rbeg
    unsigned char* base = smem + buf * 32768;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = wave * 4 + i;
      const uint32_t oa = voa[i] == 0xFFFFFFF0u ? voa[i] : voa[i] + (uint32_t)(r0 * lda * 2);
      const uint32_t ob = vob[i] == 0xFFFFFFF0u ? vob[i] : vob[i] + (uint32_t)(r0 * ldb * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(base + c * 1024), 16, oa, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void*)(base + 16384 + c * 1024), 16, ob, 0, 0, 0);
    }
  };

  const int wr = wave >> 1, wc = wave & 1;
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const int f = q | ((g & 1) << 2);
  // byte offset of this lane's tr-read address for (ks, hh) = (0,0), 16-col block nb: rows 8g + q
  const int row_base = (8 * g + q) * 256 + pp * 8;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

  auto frag = [&](const unsigned char* tile, int nb, int ks) -> bf16x8 {
    const unsigned char* p = tile + row_base + ks * (32 * 256) + ((nb ^ f) << 5);
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * 256));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };

  const int nsteps = (rend - rbeg + 63) >> 6;
  if (nsteps > 0) {
    stage(0, 0);
    __syncthreads();
    for (int t = 0; t < nsteps; ++t) {
      const int buf = t & 1;
      if (t + 1 < nsteps) stage(buf ^ 1, (t + 1) << 6);
      const unsigned char* ta = smem + buf * 32768;
      const unsigned char* tb = ta + 16384;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 af[4], bfr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = frag(ta, wr * 4 + i, ks);
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = frag(tb, wc * 4 + j, ks);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
      }
      __syncthreads();
    }
  }
  Epilogue eo = e;
  if (slab) {  // raw fp32 partial of this split; the fold kernel applies the real epilogue
    eo.bias = nullptr; eo.residual = nullptr; eo.act = HCT_ACT_NONE; eo.aux = nullptr; eo.C2 = nullptr; eo.alpha = 1.0f;
    eo.C = slab + (int64_t)blockIdx.y * M * N; eo.c_dtype = HCT_F32; eo.ldc = N;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
    epilogue_tile16x64(eo, smem + wave * kStageBytes, lane, m0 + wr * 64 + i * 16, n0 + wc * 64, M, N, acc[i]);
}

// fold split partials in fixed order and run the epilogue
__global__ void __launch_bounds__(256) gemm_fold_kernel(const float* __restrict__ slab, int splits, int M, int N, Epilogue e) {
  const int64_t total4 = (int64_t)M * N / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t idx = i * 4;
#ifndef HCT_FOLD_NT  // A/B builds: bit 0 = the slab reads (their last use) non-temporal, bit 1 = a plain fp32 result too (it is next read by the clip / AdamW pass)
#define HCT_FOLD_NT 1  /* measured in the step: 1 -> -0.20 ms, 3 -> -0.19 */
#endif
    f32x4 s = (HCT_FOLD_NT & 1) ? Vec4<float>::load_nt(slab + idx) : Vec4<float>::load(slab + idx);
    for (int z = 1; z < splits; ++z)
      s += (HCT_FOLD_NT & 1) ? Vec4<float>::load_nt(slab + (int64_t)z * M * N + idx) : Vec4<float>::load(slab + (int64_t)z * M * N + idx);
    const int m = (int)(idx / N), n = (int)(idx - (int64_t)m * N);
    if ((HCT_FOLD_NT & 2) && e.c_dtype == HCT_F32 && !e.bias && !e.residual && e.act == HCT_ACT_NONE && !e.C2 && !e.aux)
      Vec4<float>::store_nt((float*)e.C + (int64_t)m * e.ldc + n, s * e.alpha);
    else
      epilogue4(e, m, n, s);
  }
}

static int g_cu_reserve = 0;  // CUs left free for concurrently running communication kernels (RCCL) -- hct_set_cu_reserve

static int num_cus_total() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

// grid size of the persistent GEMMs: their workgroups own a CU's whole register file, so a co-running RCCL kernel could
// only start by displacing statically scheduled workgroups (long tail); when data parallelism is active a few CUs are
// left to it instead.
static int num_cus() { return std::max(8, num_cus_total() - g_cu_reserve); }

static bool aligned_to(const void* p, size_t a) { return p == nullptr || ((uintptr_t)p % a) == 0; }

static Epilogue make_epilogue(const hct_gemm_args* a) {
  Epilogue e;
  e.bias = a->bias; e.residual = a->residual; e.ldr = a->ldr;
  e.aux_deriv = a->act == HCT_ACT_GELU_D || a->act == HCT_ACT_MULAUX;
  e.act = a->act == HCT_ACT_GELU_D ? HCT_ACT_GELU : a->act == HCT_ACT_MULAUX ? HCT_ACT_DGELU : a->act;
  e.aux = a->aux; e.aux_dtype = a->aux_dtype; e.ldaux = a->ldaux;
  e.C = a->C; e.c_dtype = a->c_dtype; e.ldc = a->ldc;
  e.C2 = a->C2; e.c2_dtype = a->c2_dtype; e.ldc2 = a->ldc2;
  e.alpha = a->alpha;
  e.colsum_partial = nullptr;
  return e;
}

static bool epilogue_vec_ok(const hct_gemm_args* a) {
  auto ok = [](const void* p, int dt, int64_t ld) { return p == nullptr || (ld % 4 == 0 && aligned_to(p, dt == HCT_BF16 ? 8 : 16)); };
  return a->N % 4 == 0 && ok(a->C, a->c_dtype, a->ldc) && ok(a->C2, a->c2_dtype, a->ldc2) && ok(a->aux, a->aux_dtype, a->ldaux) &&
         ok(a->residual, HCT_F32, a->ldr) && aligned_to(a->bias, 16);
}

enum Path { PATH_GENERIC = 0, PATH_NT = 1, PATH_TN = 2 };

static Path choose_path(const hct_gemm_args* a) {
  if (a->force_generic || a->a_dtype != HCT_BF16 || a->b_dtype != HCT_BF16) return PATH_GENERIC;
  if (!epilogue_vec_ok(a)) return PATH_GENERIC;
  if (!aligned_to(a->A, 16) || !aligned_to(a->B, 16) || a->lda % 8 || a->ldb % 8) return PATH_GENERIC;
  if (a->transA == 0 && a->transB == 1 && a->K % 64 == 0 && a->N % 16 == 0 && a->lda * 2 * 128 < (1ll << 31) && a->ldb * 2 * 128 < (1ll << 31))
    return PATH_NT;
  if (a->transA == 1 && a->transB == 0 && a->M % 16 == 0 && a->N % 16 == 0 && a->act == HCT_ACT_NONE && !a->bias && !a->residual &&
      (int64_t)a->K * a->lda * 2 < (1ll << 40))
    return PATH_TN;
  return PATH_GENERIC;
}

// head of the wgrad workspace: per-tile arrival / completion counters of the in-launch split fold (64 + 64 words + a timeout flag)
constexpr size_t kTnCounterBytes = 1024;
constexpr int kTnMaxFoldTiles = 64;

static bool tn256_ok(const hct_gemm_args* a) {
  return g_nt_variant != 128 && a->c_dtype == HCT_F32 && !a->C2 && a->ldc % 4 == 0 && a->ldc * 256 < (1ll << 28) &&
         a->lda * 2 * 64 < (1ll << 31) && a->ldb * 2 * 64 < (1ll << 31);
}

static void tn256_split(const hct_gemm_args* a, int& splits, int& r_chunk) {
  const int tiles = ((a->M + 255) / 256) * ((a->N + 255) / 256);
  int s = std::max(1, num_cus() / tiles);
  int per = (a->K + s - 1) / s;
  per = std::max(128, (per + 63) / 64 * 64);  // even stage count >= 4
  // keep a split's operand span inside the 32-bit buffer offset range
  while ((int64_t)per * std::max(a->lda, a->ldb) * 2 >= (1ll << 31)) per = std::max(128, per / 2 / 64 * 64);
  r_chunk = per;
  splits = (a->K + per - 1) / per;
}

static int epilogue_mode(const hct_gemm_args* a) {
  if (a->C2) return EPI_GENERIC;
  const bool small = a->ldc * 256 < (1ll << 28) && a->ldr * 256 < (1ll << 28) && a->ldaux * 256 < (1ll << 28);
  if (!small) return EPI_GENERIC;
  // bf16 outputs are stored 8 columns (16 B) per lane
  auto wide_ok = [](const void* p, int64_t ld) { return p == nullptr || (ld % 8 == 0 && aligned_to(p, 16)); };
  if (a->c_dtype == HCT_BF16 && !(a->N % 8 == 0 && wide_ok(a->C, a->ldc) && wide_ok(a->aux, a->ldaux))) return EPI_GENERIC;
  if (a->act == HCT_ACT_NONE && !a->residual && a->c_dtype == HCT_BF16) return EPI_PLAIN_BF16;
  if (a->act == HCT_ACT_NONE && a->residual && a->c_dtype == HCT_F32) return EPI_RES_F32;
  const bool is_gelu = a->act == HCT_ACT_GELU || a->act == HCT_ACT_GELU_D, is_dgelu = a->act == HCT_ACT_DGELU || a->act == HCT_ACT_MULAUX;
  if (is_gelu && !a->residual && a->c_dtype == HCT_BF16 && a->aux && a->aux_dtype == HCT_BF16) return EPI_GELU_BF16;
  if (is_dgelu && !a->residual && a->c_dtype == HCT_BF16 && a->aux_dtype == HCT_BF16) return EPI_DGELU_BF16;
  return EPI_GENERIC;
}

// Stream-K for the remainder round of the persistent 256x256 NT kernel.  What it saves is the idle share of the last round, in
// stage pairs per CU; what it costs is one 256-KiB slab out and one or two in per workgroup, a second pipeline fill, and the
// clock / bandwidth head-room that the idle CUs were leaving to the busy ones.  Measured inside the training step
// (scripts/ab_step.py sk20 / sk16 / skoff): a threshold of 20 pairs -- the decoder's K = 3072 GEMMs with 651 tiles, 22 pairs
// saved -- is 0.23 ms per step faster than whole tiles; 16 (adds the encoder's 165-tile K = 3072 and the decoder's K = 2304
// GEMMs) is 0.10 ms slower, 8 is 0.3 ms slower.
static int g_sk_min_k = 512;       // debug hook: hct_debug_set_gemm_variant(-1000 - k); k > any K switches stream-K off
static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}
static int g_sk_gain_pairs = env_int("HCT_NT_STREAMK_PAIRS", 20);  // debug hook: hct_debug_set_gemm_variant(-100 - n); a huge value = whole tiles only
static bool nt_stream_k(const hct_gemm_args* a, int tiles256, int& sk_tiles, int& sk_wgs) {
  sk_tiles = sk_wgs = 0;
  const int G = num_cus(), P = a->K / 64;
  if ((HCT_NT_TWO_PAIR_MODES) != 0 || G > kSkMaxWgs || a->K < g_sk_min_k || P < 8 || P > 1023) return false;
  const int rem = tiles256 % G;  // (< 256: fits the packed item's tile field)
  if (rem == 0 || (int64_t)(G - rem) * P < (int64_t)g_sk_gain_pairs * G) return false;
  // per XCD (grid / 8 workgroups, ceil(rem / 8) tiles at most): every K range at least four pairs long and shorter than a tile
  const int gx = G / 8, nxmax = (rem + 7) / 8;
  if (G % 8 || (int64_t)nxmax * P > (int64_t)gx * (P - 1)) return false;
  sk_tiles = rem;
  // workgroups per XCD that may take a K range: one per four stage pairs of the XCD's share of the remainder tiles, ceil(rem / 8)
  // of them (rem / 8 gave ONE workgroup per XCD for rem < 8: a stream-K launch that shared nothing)
  sk_wgs = (int)std::min<int64_t>(gx, std::max<int64_t>(1, (int64_t)((rem + 7) / 8) * P / 4));
  return true;
}

static void tn_split(const hct_gemm_args* a, int& splits, int& r_chunk) {
  const int tiles = ((a->M + 127) / 128) * ((a->N + 127) / 128);
  const int steps = (a->K + 63) / 64;
  int s = (1024 + tiles - 1) / tiles;
  if (s > steps / 4) s = steps / 4;
  if (s < 1) s = 1;
  // keep the per-split byte span of an operand inside the 32-bit buffer range
  int per = (steps + s - 1) / s;
  r_chunk = per * 64;
  splits = (a->K + r_chunk - 1) / r_chunk;
}

}  // namespace hct

using namespace hct;

extern "C" {

void hct_set_cu_reserve(int n) { g_cu_reserve = n < 0 ? 0 : n; }
void hct_debug_set_gemm_variant(int v) {
  if (v == -4 || v == -5) { g_w4_auto = v == -4; return; }
  if (v == -8 || v == -9) { g_sk_drop = v == -8; return; }
  if (v == -10 || v == -11) { g_w4_small = v == -10; return; }
  if (v == -12 || v == -13) { g_even_rounds = v == -12; return; }
  if (v == -14 || v == -15) { g_mt3 = v == -14; return; }
  if (v <= -1000) { g_sk_min_k = -v - 1000; return; }       // stream-K of the NT remainder round only for K >= this (huge: off)
  if (v <= -100) { g_sk_gain_pairs = -v - 100; return; }     // ... and only where it saves at least this many stage pairs per CU
  if (v == -6 || v == -7) { g_tn_separate_fold = v == -6; return; }  // -6 / -7: separate fold kernel for the wgrad splits on / off  // -4 / -5: auto-dispatch of the 2-WG/CU variant on / off
  g_nt_variant = v;
}
#ifdef HCT_STAMPS
int hct_debug_set_stamp_buffer(void* p, unsigned int n_words) {  // 64 uint32 per workgroup (diagnostic build only)
  if (int rc = hct::check_hip(hipMemcpyToSymbol(HIP_SYMBOL(hct::g_stamp_words), &n_words, sizeof(n_words)), "stamp buffer size")) return rc;
  return hct::check_hip(hipMemcpyToSymbol(HIP_SYMBOL(hct::g_stamp_ptr), &p, sizeof(p)), "stamp buffer");
}
#endif
void hct_debug_set_gemm_stagger(int v) { g_stagger = v; }

static size_t colsum_ws(const hct_gemm_args* a) {
  if (!a->colsum_out) return 0;
  const size_t fused = (size_t)((a->M + 255) / 256) * 4 * a->N * sizeof(float);
  return std::max(fused, hct_colsum_workspace_bytes(a->M, a->N));
}

static size_t colsum_ws256(const hct_gemm_args* a) { return (colsum_ws(a) + 255) & ~(size_t)255; }

size_t hct_gemm_workspace_bytes(const hct_gemm_args* a) {
  // NT: column-sum partials (if asked for) at the head; the stream-K region of the persistent kernel at the tail (optional: a
  // caller that passes less, or no workspace, gets whole tiles only)
  // (only where the remainder round of THIS shape would be shared out on the present CU count: a caller that allocates per call
  //  -- the DINO head's Linears -- then neither reserves 64 MiB nor resets flags for shapes that never split)
  if (choose_path(a) == PATH_NT) {
    int sk_tiles = 0, sk_wgs = 0;
    const bool sk = a->K % 64 == 0 && a->K >= 128 && nt_stream_k(a, ((a->M + 255) / 256) * ((a->N + 255) / 256), sk_tiles, sk_wgs);
    return sk ? colsum_ws256(a) + kSkBytes : colsum_ws(a);
  }
  if (choose_path(a) != PATH_TN) return colsum_ws(a);
  int splits, r_chunk;
  if (tn256_ok(a)) {
    tn256_split(a, splits, r_chunk);
    return splits > 1 ? kTnCounterBytes + (size_t)splits * a->M * a->N * sizeof(float) : 0;
  }
  tn_split(a, splits, r_chunk);
  return splits > 1 ? (size_t)splits * a->M * a->N * sizeof(float) : 0;
}

size_t hct_gemm_nt_stream_k_bytes(void) { return kSkBytes; }
size_t hct_gemm_nt_flags_offset(size_t workspace_bytes) {
  return workspace_bytes >= kSkBytes ? ((workspace_bytes - kSkBytes) & ~(size_t)255) : (size_t)-1;
}

// ---- grouped wgrad (gemm_bf16_tn_group_kernel) --------------------------------------------------------------------------
static size_t tn_group_table_bytes(int n) { return align_up((size_t)n * sizeof(TnJob), 4096); }
static int tn_group_seg_capacity(int n) { return 16 * n + 64; }
static size_t tn_group_seg_bytes(int n) { return align_up((size_t)tn_group_seg_capacity(n) * sizeof(TnSeg), 4096); }

}  // extern "C"
namespace hct {
bool tn_group_ok(const hct_gemm_args* a) {  // (quiet form for the model driver: falls back to the split-K launch otherwise)
  return a->transA == 1 && a->transB == 0 && a->a_dtype == HCT_BF16 && a->b_dtype == HCT_BF16 && a->c_dtype == HCT_F32 && !a->bias &&
         !a->residual && a->act == HCT_ACT_NONE && !a->aux && !a->C2 && !a->colsum_out && a->M > 0 && a->N > 0 && a->K > 0 && a->M % 16 == 0 &&
         a->N % 16 == 0 && a->lda % 8 == 0 && a->ldb % 8 == 0 && a->ldc % 4 == 0 && aligned_to(a->A, 16) && aligned_to(a->B, 16) &&
         aligned_to(a->C, 16) && a->A && a->B && a->C && a->ldc * 256 < (1ll << 28) && (int64_t)a->K * a->lda * 2 < 0xFFFFFFF0ll &&
         (int64_t)a->K * a->ldb * 2 < 0xFFFFFFF0ll && a->lda < (1 << 24) && a->ldb < (1 << 24);
}
}  // namespace hct
extern "C" {

static int tn_group_check(const hct_gemm_args* a, int i) {
  if (!hct::tn_group_ok(a)) {
    set_error("hct_gemm_tn_group: job %d is not a plain bf16 weight-gradient product (transA = 1, transB = 0, fp32 C, no epilogue extras, "
              "M / N multiples of 16, 16-byte aligned operands, reduction span under 4 GiB)", i);
    return HCT_E_BADARG;
  }
  return 0;
}

static TnJob tn_group_job(const hct_gemm_args* a, int tile0) {
  TnJob j;
  memset(&j, 0, sizeof(j));
  j.A = (const bf16*)a->A; j.B = (const bf16*)a->B; j.C = (float*)a->C;
  j.M = a->M; j.N = a->N; j.R = a->K; j.lda = (int)a->lda; j.ldb = (int)a->ldb; j.ldc = (int)a->ldc;
  j.ntn = (a->N + 255) / 256;
  j.ntiles = ((a->M + 255) / 256) * j.ntn;
  j.tile0 = tile0;
  j.nk = std::max(4, ((a->K + 31) / 32 + 3) / 4 * 4);
  j.alpha = a->alpha;
  return j;
}

// Tile order of a grouped launch.  Jobs by falling reduction length (stable), so that the whole-tile rounds are homogeneous and the
// shortest products end up in the remainder; inside a class of equal length the tiles are dealt in windows of 32 ids (= what the
// 32 workgroups of an XCD work on at a time): whole chunks of 32 tiles of ONE product while there are any, the left-overs packed
// largest-first into the windows that remain (a left-over is cut only where nothing fits).
static std::vector<TnSeg> tn_group_segments(const std::vector<TnJob>& jobs) {
  std::vector<int> order(jobs.size());
  for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return jobs[x].nk > jobs[y].nk; });
  std::vector<TnSeg> segs;
  int gid = 0;
  auto emit = [&](int job, int first, int count) {
    if (!segs.empty() && segs.back().job == job && segs.back().tile_first + segs.back().count == first) segs.back().count += count;
    else segs.push_back(TnSeg{job, first, count, gid});
    gid += count;
  };
  size_t i = 0;
  while (i < order.size()) {
    size_t e = i;
    while (e < order.size() && jobs[order[e]].nk == jobs[order[i]].nk) ++e;
    struct Item { int job, first, count; };
    std::vector<Item> full, rest;  // chunks of 32, left-overs (< 32)
    for (size_t k = i; k < e; ++k) {
      const int j = order[k], nt = jobs[j].ntiles;
      for (int c = 0; c + 32 <= nt; c += 32) full.push_back(Item{j, c, 32});
      if (nt % 32) rest.push_back(Item{j, nt / 32 * 32, nt % 32});
    }
    std::stable_sort(rest.begin(), rest.end(), [](const Item& x, const Item& y) { return x.count > y.count; });
    size_t fi = 0;
    while (fi < full.size() || !rest.empty()) {
      const int room = 32 - gid % 32;
      if (room == 32 && fi < full.size()) { emit(full[fi].job, full[fi].first, 32); ++fi; continue; }
      // the largest left-over that fits the window; none: a piece of the largest one (or of a whole chunk) closes the window
      size_t pick = rest.size();
      for (size_t k = 0; k < rest.size(); ++k)
        if (rest[k].count <= room) { pick = k; break; }
      if (pick < rest.size()) {
        emit(rest[pick].job, rest[pick].first, rest[pick].count);
        rest.erase(rest.begin() + pick);
      } else if (!rest.empty()) {
        emit(rest[0].job, rest[0].first, room);
        rest[0].first += room; rest[0].count -= room;
        std::stable_sort(rest.begin(), rest.end(), [](const Item& x, const Item& y) { return x.count > y.count; });
      } else {  // only whole chunks left and the window is open: cut one
        Item it = full[fi++];
        emit(it.job, it.first, room);
        rest.push_back(Item{it.job, it.first + room, 32 - room});
      }
    }
    i = e;
  }
  return segs;
}

// splits of the remainder tiles: least (rounds of pieces) / splits, a small price per split for the fix-up; every piece at least
// 4 stages, at most kTnMaxFollowers follower pieces
static int tn_group_splits(int Rm, int G, int min_nk) {
  if (Rm <= 0) return 1;
  int best = 1;
  double best_cost = 1e30;
  for (int sp = 1; sp <= 16; ++sp) {
    if (sp > 1 && ((int64_t)(sp - 1) * Rm > kTnMaxFollowers || min_nk / 4 < sp * 4)) break;  // (pieces of at least 16 stages)
    const double cost = (double)(((int64_t)Rm * sp + G - 1) / G) / sp + 0.004 * sp;
    if (cost < best_cost - 1e-12) { best_cost = cost; best = sp; }
  }
  return best;
}

size_t hct_gemm_tn_group_workspace_bytes(int n_jobs) { return n_jobs > 0 ? tn_group_table_bytes(n_jobs) + tn_group_seg_bytes(n_jobs) + kTnSkBytes : 0; }

static int tn_group_build(const hct_gemm_args* jobs, int n, bool check, std::vector<TnJob>& tj, std::vector<TnSeg>& segs) {
  tj.resize(n);
  int tile0 = 0;
  for (int i = 0; i < n; ++i) {
    if (check)
      if (int rc = tn_group_check(jobs + i, i)) return rc;
    tj[i] = tn_group_job(jobs + i, tile0);
    tile0 += tj[i].ntiles;
  }
  segs = tn_group_segments(tj);
  if ((int)segs.size() > tn_group_seg_capacity(n)) {
    set_error("hct_gemm_tn_group: %zu tile segments exceed the workspace's table (%d)", segs.size(), tn_group_seg_capacity(n));
    return HCT_E_WORKSPACE;
  }
  return 0;
}

int hct_gemm_tn_group_prepare(const hct_gemm_args* jobs, int n, void* workspace, size_t workspace_bytes, void* stream) {
  HCT_REQUIRE(jobs && n > 0 && n <= 4096 && workspace, "hct_gemm_tn_group_prepare: bad arguments");
  if (workspace_bytes < hct_gemm_tn_group_workspace_bytes(n) || ((uintptr_t)workspace % 256) != 0) {
    set_error("hct_gemm_tn_group: workspace too small or not 256-byte aligned (%zu < %zu)", workspace_bytes, hct_gemm_tn_group_workspace_bytes(n));
    return HCT_E_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  std::vector<TnJob> tj;
  std::vector<TnSeg> segs;
  if (int rc = tn_group_build(jobs, n, true, tj, segs)) return rc;
  for (int first = 0; first < n; first += kTnGroupChunk) {
    TnJobChunk c;
    memset(&c, 0, sizeof(c));
    const int cnt = std::min(kTnGroupChunk, n - first);
    for (int i = 0; i < cnt; ++i) c.j[i] = tj[first + i];
    hipLaunchKernelGGL(tn_group_table_kernel, dim3(1), dim3(64), 0, s, (TnJob*)workspace, c, first, cnt);
  }
  TnSeg* dsegs = (TnSeg*)((unsigned char*)workspace + tn_group_table_bytes(n));
  for (int first = 0; first < (int)segs.size(); first += kTnSegChunk) {
    TnSegChunk c;
    memset(&c, 0, sizeof(c));
    const int cnt = std::min(kTnSegChunk, (int)segs.size() - first);
    for (int i = 0; i < cnt; ++i) c.s[i] = segs[first + i];
    hipLaunchKernelGGL(tn_group_seg_kernel, dim3(1), dim3(256), 0, s, dsegs, c, first, cnt);
  }
  if (int rc = check_hip(hipMemsetAsync((unsigned char*)workspace + tn_group_table_bytes(n) + tn_group_seg_bytes(n), 0, kTnSkHeadBytes, s),
                         "hct_gemm_tn_group: flag reset"))
    return rc;
  HCT_CHECK_LAUNCH("hct_gemm_tn_group_prepare");
  return 0;
}

int hct_gemm_tn_group_run(const hct_gemm_args* jobs, int n, void* workspace, size_t workspace_bytes, void* stream) {
  HCT_REQUIRE(jobs && n > 0 && workspace && workspace_bytes >= hct_gemm_tn_group_workspace_bytes(n), "hct_gemm_tn_group_run: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const int G = std::min(num_cus(), kSkMaxWgs);
  std::vector<TnJob> tj;
  std::vector<TnSeg> segs;
  if (int rc = tn_group_build(jobs, n, false, tj, segs)) return rc;  // (the same order `prepare` wrote: host arithmetic only)
  int T = 0;
  double flops = 0, bytes = 0;
  for (int i = 0; i < n; ++i) {
    T += tj[i].ntiles;
    flops += 2.0 * jobs[i].M * jobs[i].N * jobs[i].K;
    bytes += 2.0 * jobs[i].K * (jobs[i].M + jobs[i].N) + 4.0 * jobs[i].M * jobs[i].N;
  }
  const int F = T / G, Rm = T - F * G;
  int min_nk = 1 << 30;  // shortest reduction among the remainder tiles (ids >= F * G)
  for (const TnSeg& sg : segs)
    if (sg.gtile0 + sg.count > F * G) min_nk = std::min(min_nk, tj[sg.job].nk);
  const int ns_split = tn_group_splits(Rm, G, min_nk);
  ProfScope ps(PROF_GEMM_TN, flops, s, bytes);
  ps.tag(n, T, ns_split, -1, T, Rm);
  static unsigned seq = 0;
  unsigned sk_seq = (++seq) & 0x0FFFFFFFu;
  if (sk_seq == 0) sk_seq = (++seq) & 0x0FFFFFFFu;
  if (g_sk_drop) sk_seq |= 0x80000000u;
  unsigned char* base = (unsigned char*)workspace;
  hipLaunchKernelGGL(gemm_bf16_tn_group_kernel, dim3(G), dim3(512), 0, s, (const TnJob*)base, (const TnSeg*)(base + tn_group_table_bytes(n)),
                     (int)segs.size(), T, F, ns_split, base + tn_group_table_bytes(n) + tn_group_seg_bytes(n), sk_seq);
  HCT_CHECK_LAUNCH("hct_gemm_tn_group_run");
  return 0;
}

int hct_gemm(const hct_gemm_args* a, void* workspace, size_t workspace_bytes, void* stream) {
  HCT_REQUIRE(a && a->M >= 0 && a->N >= 0 && a->K >= 0, "hct_gemm: bad shape");
  HCT_REQUIRE((a->act != HCT_ACT_DGELU && a->act != HCT_ACT_MULAUX && a->act != HCT_ACT_GELU_D) || a->aux, "hct_gemm: DGELU / MULAUX / GELU_D need aux");
  HCT_REQUIRE(a->act >= HCT_ACT_NONE && a->act <= HCT_ACT_MULAUX && a->act != HCT_ACT_TANH, "hct_gemm: unknown activation code %d", a->act);
  if (a->M == 0 || a->N == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  Epilogue e = make_epilogue(a);
  const Path path = choose_path(a);
  if (a->colsum_out) {
    if (colsum_ws(a) > workspace_bytes || !workspace) {
      set_error("hct_gemm: colsum_out needs %zu workspace bytes", colsum_ws(a));
      return HCT_E_WORKSPACE;
    }
    HCT_REQUIRE(path != PATH_TN, "hct_gemm: colsum_out is not supported on the wgrad path");
  }
  // column sums of C: fused into the persistent NT kernel's DGELU epilogue, otherwise a separate pass over C
  auto finish_colsum = [&](bool fused) -> int {
    if (!a->colsum_out) return 0;
    if (fused) return fold_rows((const float*)workspace, ((a->M + 255) / 256) * 4, a->N, a->colsum_out, s);
    // (the tail of a workspace that is large enough for it belongs to the stream-K flags and slabs)
    const size_t head = workspace_bytes >= colsum_ws256(a) + kSkBytes ? ((workspace_bytes - kSkBytes) & ~(size_t)255) : workspace_bytes;
    return hct_colsum(a->C, a->c_dtype, a->M, a->N, a->ldc, a->colsum_out, workspace, head, stream);
  };
  const double flops = 2.0 * a->M * a->N * a->K;
  // algorithmic bytes: each operand read once, each output written once (SURVEY 8d secondary report)
  const double mn = (double)a->M * a->N;
  const double bytes = (double)a->M * a->K * dtype_size(a->a_dtype) + (double)a->N * a->K * dtype_size(a->b_dtype) +
                       mn * dtype_size(a->c_dtype) + (a->residual ? mn * 4 : 0.0) + (a->aux ? mn * dtype_size(a->aux_dtype) : 0.0) +
                       (a->C2 ? mn * dtype_size(a->c2_dtype) : 0.0);
  if (path == PATH_NT) {
    ProfScope ps(PROF_GEMM_NT, flops, s, bytes);
    const int tiles256 = ((a->M + 255) / 256) * ((a->N + 255) / 256);
    const bool ok256 = a->K % 64 == 0 && a->K >= 128;
    const bool big = ok256 && (g_nt_variant == 256 || g_nt_variant == 4 || g_nt_variant == 0);
    if (big) {
      const int mode = epilogue_mode(a);
      // two-workgroups-per-CU variant for epilogue-dominated shapes (short K, wide output)
      // measured crossover (scripts/bench_gemm_model.py, B=256): with two workgroups per CU one drains its tile while the
      // other computes, which pays for the 1.5x operand traffic only where the epilogue is heavy (GELU: two outputs;
      // +residual: fp32 read + write), K is short and there are several tiles per CU
      const bool w4_shape = a->K <= 1024 && ((mode == EPI_GELU_BF16 && tiles256 >= 4 * num_cus()) ||
                                             (mode == EPI_RES_F32 && tiles256 >= 2 * num_cus()));
      const bool w4_small = g_w4_small && tiles256 < num_cus() && 2 * tiles256 > num_cus() && mode != EPI_GENERIC && !a->colsum_out;
      const bool w4 = (g_nt_variant == 4) || (g_nt_variant == 0 && ((g_w4_auto && w4_shape) || w4_small));
      if (w4) {
        const int tiles = ((a->M + 255) / 256) * ((a->N + 127) / 128);
        const dim3 g4(std::min(tiles, 2 * num_cus()));
#define HCT_NTW4(MODE_)                                                                                            \
  hipLaunchKernelGGL(gemm_bf16_nt_w4_kernel<MODE_>, g4, dim3(256), 0, s, a->M, a->N, a->K, (const bf16*)a->A, a->lda, \
                     (const bf16*)a->B, a->ldb, e, tiles, st4)
        const int st4 = g_stagger >= 0 ? g_stagger : (a->K / 32) * 2 / 5;
        switch (mode) {
          case EPI_PLAIN_BF16: HCT_NTW4(EPI_PLAIN_BF16); break;
          case EPI_RES_F32: HCT_NTW4(EPI_RES_F32); break;
          case EPI_GELU_BF16: HCT_NTW4(EPI_GELU_BF16); break;
          case EPI_DGELU_BF16: HCT_NTW4(EPI_DGELU_BF16); break;
          default: HCT_NTW4(EPI_GENERIC); break;
        }
#undef HCT_NTW4
        HCT_CHECK_LAUNCH("hct_gemm(nt_w4)");
        return finish_colsum(false);
      }
      // (any M: rows past M are masked out of the sums, and every (row tile, wave) partial row is written -- zeros where a
      //  wave's 64 rows lie wholly past M -- so the fixed-order fold over ceil(M / 256) * 4 rows sees no stale data)
      const bool fuse_cs = a->colsum_out && mode == EPI_DGELU_BF16 && !w4;
      if (fuse_cs) e.colsum_partial = (float*)workspace;
      // stream-K for the remainder round (see the kernel): needs its region at the END of the workspace
      int sk_tiles = 0, sk_wgs = 0;
      unsigned char* sk_ws = nullptr;
      unsigned sk_seq = 0;
      if (workspace && workspace_bytes >= colsum_ws256(a) + kSkBytes && nt_stream_k(a, tiles256, sk_tiles, sk_wgs)) {
        sk_ws = (unsigned char*)workspace + ((workspace_bytes - kSkBytes) & ~(size_t)255);
        static unsigned seq = 0;
        sk_seq = (++seq) & 0x0FFFFFFFu;  // (28 bits: the flag word also carries the writer's XCD)
        if (sk_seq == 0) sk_seq = (++seq) & 0x0FFFFFFFu;  // zero is what an armed region starts from
        if (g_sk_drop) sk_seq |= 0x80000000u;
        if (!a->workspace_armed)
          if (int rc = check_hip(hipMemsetAsync(sk_ws, 0, kSkHeadBytes, s), "hct_gemm(nt256): stream-K flag reset")) return rc;
      }
      // Whole-tile launches: R = ceil(tiles / CUs) rounds take the same time on ceil(tiles / R) workgroups as on all CUs -- the last
      // round is then full and the CUs left out idle for the whole launch instead of for its last round only, which leaves their
      // share of the power budget to the others (hct_debug_set_gemm_variant(-12 / -13): A/B hook)
      int gsz = std::min(tiles256, num_cus());
      if (g_even_rounds && !sk_tiles && tiles256 > num_cus()) {
        const int rounds = (tiles256 + num_cus() - 1) / num_cus();
        gsz = (tiles256 + rounds - 1) / rounds;
        gsz = std::min(num_cus(), (gsz + 7) / 8 * 8);  // (a multiple of 8: the tile walk deals ids per XCD)
      }
      // 192-row tiles for the plain / +residual shapes whose 256-row tiles fill less than one round of CUs while 192-row tiles still
      // fit one (the encoder's M = 14 080, N = 768 products: 165 -> 222 tiles, each 3/4 of the work); hct_debug_set_gemm_variant(-14 / -15)
      const int tiles192 = ((a->M + 191) / 192) * ((a->N + 255) / 256);
      const bool mt3 = g_mt3 && !sk_tiles && (mode == EPI_PLAIN_BF16 || mode == EPI_RES_F32) && tiles256 < num_cus() && tiles192 <= num_cus() &&
                       tiles192 > tiles256 && a->M >= 192;
      const dim3 grid(sk_tiles ? num_cus() : (mt3 ? tiles192 : gsz));
      ps.tag(a->M, a->N, a->K, fuse_cs ? EPI_DGELU_CS : mode, mt3 ? tiles192 : tiles256, sk_tiles);
      if (mt3) {
        if (mode == EPI_PLAIN_BF16)
          hipLaunchKernelGGL((gemm_bf16_nt256_kernel<EPI_PLAIN_BF16, false, 3>), grid, dim3(512), 0, s, a->M, a->N, a->K, (const bf16*)a->A, a->lda,
                             (const bf16*)a->B, a->ldb, e, tiles192, 0, 0, 0, (unsigned char*)nullptr, 0u);
        else
          hipLaunchKernelGGL((gemm_bf16_nt256_kernel<EPI_RES_F32, false, 3>), grid, dim3(512), 0, s, a->M, a->N, a->K, (const bf16*)a->A, a->lda,
                             (const bf16*)a->B, a->ldb, e, tiles192, 0, 0, 0, (unsigned char*)nullptr, 0u);
        HCT_CHECK_LAUNCH("hct_gemm(nt256, 192-row tiles)");
        return finish_colsum(false);
      }
      // one start phase = 1/8 of a tile's main loop (nk stages x ~1000 cycles; s_sleep(32) = 2048 cycles); only when each
      // CU runs several tiles (otherwise the delay is pure loss)
      // (a start-phase stagger of the workgroups helped the earlier one-stage-per-step schedule by ~0.1 ms per step; with
      //  the paired schedule it is neutral to slightly negative: off unless forced through the debug hook)
      const int stagger = g_stagger >= 0 ? g_stagger : 0;
#define HCT_NT256(MODE_)                                                                                                       \
  do {                                                                                                                         \
    if (sk_tiles)                                                                                                              \
      hipLaunchKernelGGL((gemm_bf16_nt256_kernel<MODE_, (HCT_NT_TWO_PAIR_MODES) == 0>), grid, dim3(512), 0, s, a->M, a->N, a->K, \
                         (const bf16*)a->A, a->lda, (const bf16*)a->B, a->ldb, e, tiles256, stagger, sk_tiles, sk_wgs, sk_ws, sk_seq); \
    else                                                                                                                       \
      hipLaunchKernelGGL((gemm_bf16_nt256_kernel<MODE_, false>), grid, dim3(512), 0, s, a->M, a->N, a->K, (const bf16*)a->A,     \
                         a->lda, (const bf16*)a->B, a->ldb, e, tiles256, stagger, 0, 0, (unsigned char*)nullptr, 0u);             \
  } while (0)
      switch (mode) {
        case EPI_PLAIN_BF16: HCT_NT256(EPI_PLAIN_BF16); break;
        case EPI_RES_F32: HCT_NT256(EPI_RES_F32); break;
        case EPI_GELU_BF16: HCT_NT256(EPI_GELU_BF16); break;
        case EPI_DGELU_BF16: if (fuse_cs) HCT_NT256(EPI_DGELU_CS); else HCT_NT256(EPI_DGELU_BF16); break;
        default: HCT_NT256(EPI_GENERIC); break;
      }
#undef HCT_NT256
      HCT_CHECK_LAUNCH("hct_gemm(nt256)");
      return finish_colsum(fuse_cs);
    }
    const int tiles = ((a->M + 127) / 128) * ((a->N + 127) / 128);
    hipLaunchKernelGGL(gemm_bf16_nt_kernel, dim3(tiles), dim3(256), 0, s, a->M, a->N, a->K, (const bf16*)a->A, a->lda,
                       (const bf16*)a->B, a->ldb, e);
    HCT_CHECK_LAUNCH("hct_gemm(nt)");
    return finish_colsum(false);
  }
  if (path == PATH_TN && tn256_ok(a)) {
    ProfScope ps(PROF_GEMM_TN, flops, s, bytes);
    int splits, r_chunk;
    tn256_split(a, splits, r_chunk);
    ps.tag(a->M, a->N, a->K, splits, ((a->M + 255) / 256) * ((a->N + 255) / 256) * splits, 0);
    const int tiles_mn = ((a->M + 255) / 256) * ((a->N + 255) / 256);
    const int tiles = tiles_mn * splits;
    float* slab = nullptr;
    unsigned int* counters = nullptr;
    if (splits > 1) {
      const size_t need = kTnCounterBytes + (size_t)splits * a->M * a->N * sizeof(float);
      if (workspace_bytes < need || !workspace) {
        set_error("hct_gemm(tn256): workspace too small (%zu < %zu)", workspace_bytes, need);
        return HCT_E_WORKSPACE;
      }
      counters = (unsigned int*)workspace;
      slab = (float*)((char*)workspace + kTnCounterBytes);
    }
    // the in-launch fold needs one counter pair per output tile and writes fp32 C directly; anything else keeps the fold kernel
    const bool fold_in_launch = slab && tiles_mn <= kTnMaxFoldTiles && !g_tn_separate_fold;
    if (fold_in_launch) {
      // the counters re-arm themselves at the end of every launch; a caller that keeps the workspace head to itself says so
      // (the plan does: its workspace is zeroed at allocation), anyone else gets a reset in front of the launch
      if (!a->workspace_armed)
        if (int rc = check_hip(hipMemsetAsync(workspace, 0, kTnCounterBytes, s), "hct_gemm(tn256): counter reset")) return rc;
    }
    hipLaunchKernelGGL(gemm_bf16_tn256_kernel, dim3(std::min(tiles, num_cus())), dim3(512), 0, s, a->M, a->N, a->K, r_chunk,
                       (const bf16*)a->A, a->lda, (const bf16*)a->B, a->ldb, slab, e, tiles, fold_in_launch ? splits : 0,
                       fold_in_launch ? counters : nullptr);
    if (slab && !fold_in_launch) {
      const int64_t total4 = (int64_t)a->M * a->N / 4;
      const int blocks = (int)std::min<int64_t>(2048, (total4 + 255) / 256);
      hipLaunchKernelGGL(gemm_fold_kernel, dim3(blocks), dim3(256), 0, s, slab, splits, a->M, a->N, e);
    }
    HCT_CHECK_LAUNCH("hct_gemm(tn256)");
    return 0;
  }
  if (path == PATH_TN) {
    ProfScope ps(PROF_GEMM_TN, flops, s, bytes);
    int splits, r_chunk;
    tn_split(a, splits, r_chunk);
    const int tiles = ((a->M + 127) / 128) * ((a->N + 127) / 128);
    float* slab = nullptr;
    if (splits > 1) {
      if (workspace_bytes < (size_t)splits * a->M * a->N * sizeof(float) || !workspace) {
        set_error("hct_gemm(tn): workspace too small (%zu < %zu)", workspace_bytes, (size_t)splits * a->M * a->N * sizeof(float));
        return HCT_E_WORKSPACE;
      }
      slab = (float*)workspace;
    }
    hipLaunchKernelGGL(gemm_bf16_tn_kernel, dim3(tiles, splits), dim3(256), 0, s, a->M, a->N, a->K, r_chunk,
                       (const bf16*)a->A, a->lda, (const bf16*)a->B, a->ldb, slab, e);
    if (slab) {
      const int64_t total4 = (int64_t)a->M * a->N / 4;
      const int blocks = (int)std::min<int64_t>(2048, (total4 + 255) / 256);
      hipLaunchKernelGGL(gemm_fold_kernel, dim3(blocks), dim3(256), 0, s, slab, splits, a->M, a->N, e);
    }
    HCT_CHECK_LAUNCH("hct_gemm(tn)");
    return 0;
  }
  // generic
  ProfScope ps(PROF_GEMM_GENERIC, flops, s, bytes);
  const int64_t sam = a->transA ? 1 : a->lda, sak = a->transA ? a->lda : 1;
  const int64_t sbk = a->transB ? 1 : a->ldb, sbn = a->transB ? a->ldb : 1;
  const int vec_ok = epilogue_vec_ok(a) ? 1 : 0;
  dim3 grid((a->N + 63) / 64, (a->M + 63) / 64);
#define HCT_GEN(TA, TB)                                                                                              \
  hipLaunchKernelGGL((gemm_generic_kernel<TA, TB>), grid, dim3(256), 0, s, a->M, a->N, a->K, (const TA*)a->A, sam, sak, \
                     (const TB*)a->B, sbk, sbn, e, vec_ok)
  if (a->a_dtype == HCT_BF16 && a->b_dtype == HCT_BF16) HCT_GEN(bf16, bf16);
  else if (a->a_dtype == HCT_BF16) HCT_GEN(bf16, float);
  else if (a->b_dtype == HCT_BF16) HCT_GEN(float, bf16);
  else HCT_GEN(float, float);
#undef HCT_GEN
  HCT_CHECK_LAUNCH("hct_gemm(generic)");
  return finish_colsum(false);
}

}  // extern "C"
