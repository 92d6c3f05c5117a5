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

namespace hct {

struct Epilogue {
  const float* bias;
  const float* residual;
  int64_t ldr;
  int act;
  void* aux; int aux_dtype; int64_t ldaux;
  void* C; int c_dtype; int64_t ldc;
  void* C2; int c2_dtype; int64_t ldc2;
  float alpha;
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
    if (e.aux) store4(e.aux, e.aux_dtype, (int64_t)m * e.ldaux + n, v);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = gelu_erf(v[i]);
  } else if (e.act == HCT_ACT_DGELU) {
    const f32x4 u = load4(e.aux, e.aux_dtype, (int64_t)m * e.ldaux + n);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] *= dgelu_erf(u[i]);
  }
  if (e.residual) v += Vec4<float>::load(e.residual + (int64_t)m * e.ldr + n);
  store4(e.C, e.c_dtype, (int64_t)m * e.ldc + n, v);
  if (e.C2) store4(e.C2, e.c2_dtype, (int64_t)m * e.ldc2 + n, v);
}
__device__ __forceinline__ void epilogue1(const Epilogue& e, int m, int n, float acc) {
  float v = acc * e.alpha;
  if (e.bias) v += e.bias[n];
  if (e.act == HCT_ACT_GELU) {
    if (e.aux) store1(e.aux, e.aux_dtype, (int64_t)m * e.ldaux + n, v);
    v = gelu_erf(v);
  } else if (e.act == HCT_ACT_DGELU) {
    v *= dgelu_erf(load1(e.aux, e.aux_dtype, (int64_t)m * e.ldaux + n));
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
    const int m = m0 + wr * 64 + i * 16 + frow;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wc * 64 + j * 16 + fchk * 4;
      if (n < N) epilogue4(e, m, n, acc[i][j]);
    }
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
  const int frow = lane & 15, fchk = lane >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wr * 64 + i * 16 + frow;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wc * 64 + j * 16 + fchk * 4;
      if (n >= N) continue;
      if (slab) Vec4<float>::store(slab + ((int64_t)blockIdx.y * M + m) * N + n, acc[i][j]);
      else epilogue4(e, m, n, acc[i][j]);
    }
  }
}

// fold split partials in fixed order and run the epilogue
__global__ void __launch_bounds__(256) gemm_fold_kernel(const float* __restrict__ slab, int splits, int M, int N, Epilogue e) {
  const int64_t total4 = (int64_t)M * N / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t idx = i * 4;
    f32x4 s = Vec4<float>::load(slab + idx);
    for (int z = 1; z < splits; ++z) s += Vec4<float>::load(slab + (int64_t)z * M * N + idx);
    const int m = (int)(idx / N), n = (int)(idx - (int64_t)m * N);
    epilogue4(e, m, n, s);
  }
}

static bool aligned_to(const void* p, size_t a) { return p == nullptr || ((uintptr_t)p % a) == 0; }

static Epilogue make_epilogue(const hct_gemm_args* a) {
  Epilogue e;
  e.bias = a->bias; e.residual = a->residual; e.ldr = a->ldr; e.act = a->act;
  e.aux = a->aux; e.aux_dtype = a->aux_dtype; e.ldaux = a->ldaux;
  e.C = a->C; e.c_dtype = a->c_dtype; e.ldc = a->ldc;
  e.C2 = a->C2; e.c2_dtype = a->c2_dtype; e.ldc2 = a->ldc2;
  e.alpha = a->alpha;
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

size_t hct_gemm_workspace_bytes(const hct_gemm_args* a) {
  if (choose_path(a) != PATH_TN) return 0;
  int splits, r_chunk;
  tn_split(a, splits, r_chunk);
  return splits > 1 ? (size_t)splits * a->M * a->N * sizeof(float) : 0;
}

int hct_gemm(const hct_gemm_args* a, void* workspace, size_t workspace_bytes, void* stream) {
  HCT_REQUIRE(a && a->M >= 0 && a->N >= 0 && a->K >= 0, "hct_gemm: bad shape");
  HCT_REQUIRE(a->act != HCT_ACT_DGELU || a->aux, "hct_gemm: DGELU needs aux");
  if (a->M == 0 || a->N == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  Epilogue e = make_epilogue(a);
  const Path path = choose_path(a);
  const double flops = 2.0 * a->M * a->N * a->K;
  if (path == PATH_NT) {
    ProfScope ps(PROF_GEMM_NT, flops, s);
    const int tiles = ((a->M + 127) / 128) * ((a->N + 127) / 128);
    hipLaunchKernelGGL(gemm_bf16_nt_kernel, dim3(tiles), dim3(256), 0, s, a->M, a->N, a->K, (const bf16*)a->A, a->lda,
                       (const bf16*)a->B, a->ldb, e);
    HCT_CHECK_LAUNCH("hct_gemm(nt)");
    return 0;
  }
  if (path == PATH_TN) {
    ProfScope ps(PROF_GEMM_TN, flops, s);
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
  ProfScope ps(PROF_GEMM_GENERIC, flops, s);
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
  return 0;
}

}  // extern "C"
