// Whole-model driver: parameter layout, activation workspace layout, forward and staged backward of
// MaskedAutoencoderViT (src/models/mae.py:220-317; AttentionBlock attentionblock.py:96-99) on one GPU.
// Host-only logic: it owns no device memory and launches the kernels of this library on the caller's stream.
//
// Flat parameter layout (element offsets, every tensor starts on a 1024-element unit): tensors are laid out in
// FORWARD-USE order  [patch-embed w,b | pos | cls | enc block 0 .. | norm | decoder_embed | mask_token | dec_cls |
// dec_pos | dec block 0 .. | decoder_norm | decoder_pred]  so that each backward stage completes one contiguous
// range and the ranges finish from the end of the buffer towards its start (gradient-bucket order for the
// data-parallel all-reduce).  Names/shapes are the reference's; the Python side builds
// nn.Parameter views by name, so state_dict order/keys are the reference's regardless of this layout.
#include <cstdlib>
#include "common.h"

#include <algorithm>
#include <map>
#include <string>
#include <vector>

using namespace hct;

namespace {

constexpr int64_t kUnitElems = 1024;

struct BlockP {  // parameter indices
  int ln1_w, ln1_b, qkv_w, qkv_b, proj_w, proj_b, ln2_w, ln2_b, fc1_w, fc1_b, fc2_w, fc2_b;
};
struct BlockA {  // byte offsets into the workspace
  size_t x1, mean1, rstd1, qkv, o, lse, h_mid, x2, mean2, rstd2, u, g;
};
struct Act { size_t off; int64_t rows, cols; int dtype; };
// gradient operands of one block's four weight gradients (compute dtype): gradient wrt the block output [M,d], wrt the MLP's
// pre-activation [M,m], wrt h_mid [M,d], wrt qkv [M,3d].  They stay untouched until the block's weight gradients have run (the
// grouped launch, see flush_wgrads): a ring of sets per side.
struct BlockG { size_t out, big, mid, qkv; };
constexpr int kWgSlots = 8;   // grouped weight-gradient launches per backward, at most (each keeps a prepared job table)

}  // namespace

struct hct_mae_plan {
  hct_mae_config cfg;
  int B, dt;           // batch, compute dtype
  int g, L, K, pd, Ne, Nd, Me, Md;
  int D, Dd, Mlp, Mlpd, H, Hd;
  std::vector<hct_param_info> params;
  std::map<std::string, int> pindex;
  int64_t param_elems = 0, bf16t_elems = 0;
  // parameter indices
  bool vit = false;  // encoder-only plan (plain ViT backbone)
  int R = 0;         // register tokens
  float norm_eps = 1e-5f;
  int p_reg = -1;
  int p_pe_w, p_pe_b, p_pos, p_cls, p_norm_w, p_norm_b, p_de_w, p_de_b, p_mask, p_dcls, p_dpos, p_dnorm_w, p_dnorm_b,
      p_pred_w, p_pred_b;
  std::vector<BlockP> enc, dec;
  // stage -> parameter range
  std::vector<std::pair<int64_t, int64_t>> stage_range;
  // workspace
  size_t ws_bytes = 0;
  size_t a_ids_restore, a_ids_shuffle, a_mask, a_row_loss, a_patches, a_tok, a_latent, a_lat_mean, a_lat_rstd, a_e,
      a_ynorm, a_yn_mean, a_yn_rstd, a_pred, a_dpred;
  // Compact tail of the decoder (hct_mae_plan_set_tail): the loss takes only the masked patches' rows (mae.py:298-299), so
  // everything behind the LAST decoder block's attention -- its proj / MLP, decoder_norm, decoder_pred, the loss and their
  // backward -- runs on those Mc = B * (L - K) rows alone (75 % of the rows at mask_ratio 0.75), gathered into compact
  // matrices; the block's LN1 / qkv / attention keep every row (keys and values of all tokens are needed).
  bool tail = false;      // mode of the next forward
  bool tail_fwd = false;  // mode the last forward ran in (its backward follows it)
  bool tail_ok = false;   // the geometry allows it
  int Mc = 0;
  size_t a_tail_rows = 0, a_tail_inv = 0, a_oc = 0, s_dhc = 0;
  // First decoder block on "cat" rows (hct_mae_plan_set_dec0): its masked tokens enter as mask_token + pos[l] in EVERY volume
  // (mae.py:259-265), so LayerNorm1 and the qkv Linear of that block -- forward, weight gradient, input gradient, LayerNorm backward --
  // need one row per patch position for them, not one per (volume, position): Rcat = B (K+1) kept / class rows + L table rows
  // instead of B (L+1).
  bool dec0 = false, dec0_fwd = false, dec0_ok = false;
  int Rcat = 0;
  size_t a_kept_rows = 0, a_cat_idx = 0, a_hk = 0, a_x1c = 0, a_mean_c = 0, a_rstd_c = 0, a_qkv_cat = 0, a_dqkv_cat = 0, a_dcat = 0;
  std::vector<size_t> h_enc, h_dec;  // fp32 residual-stream chain
  std::vector<BlockA> aenc, adec;
  size_t s_dh, s_dh_shadow, s_dbig, s_dx, s_do, s_dqkv, s_small, s_small_bytes, s_gemm, s_gemm_bytes, s_nt_bytes = 0;
  size_t s_fold_a = 0, s_fold_b = 0, s_fold_bytes = 0, s_small2 = 0, s_small2_bytes = 0;  // partial buffers of the deferred folds (block_backward)
  std::map<std::string, Act> acts;
  // Deferred weight gradients.  dW = dY^T . X feeds nothing in the backward (only the optimizer reads it), so the products are
  // queued (linear_wgrad) and run together in one persistent grouped launch (hct_gemm_tn_group_*: whole tiles over the full
  // reduction, no split partials, no fold launches) when the decoder's / the encoder's backward is through, or every
  // `wg_blocks` block stages (data parallelism: the gradient buckets then become final earlier).
  bool wg_defer = false;
  int wg_blocks = 0;                           // flush after this many block stages (0: at the decoder / encoder boundaries only)
  int wg_blocks_pending = 0, wg_slot = 0;
  std::vector<BlockG> genc, gdec;              // rings of gradient-operand sets; block i uses set i % size
  size_t a_de = 0, a_dtok = 0;                 // gradient operands of decoder_embed / the patch embedding
  std::vector<hct_gemm_args> wg_pending;
  std::vector<hct_gemm_args> wg_prepared[kWgSlots];
  size_t s_wg[kWgSlots] = {0}, s_wg_bytes = 0;
  int64_t final_off = 0;                       // every gradient element at or behind this offset is final (hct_mae_backward_final_offset)
  // bound buffers
  float* params_f32 = nullptr;
  float* grads = nullptr;
  bf16* params_bf16 = nullptr;
  bf16* params_bf16_t = nullptr;
  unsigned char* ws = nullptr;
  bool fwd_done = false;
  bool gemm_ws_armed = false;
  bool nt_ws_armed = false;  // the stream-K tail of [s_small | s_nt] has been zeroed since the last bind
  bool dpred_done = false;  // the last forward also wrote d(loss)/d(pred) (training forward)
  const float* dloss = nullptr;

  size_t esz() const { return dtype_size(dt); }
  template <typename T> T* W(size_t off) const { return reinterpret_cast<T*>(ws + off); }
  const float* pf(int i) const { return i < 0 ? nullptr : params_f32 + params[i].offset; }
  float* gf(int i) const { return i < 0 ? nullptr : grads + params[i].offset; }
  // weight operand in the compute dtype
  const void* wop(int i) const { return dt == HCT_BF16 ? (const void*)(params_bf16 + params[i].offset) : (const void*)(params_f32 + params[i].offset); }
  const void* wop_t(int i) const { return (const void*)(params_bf16_t + params[i].bf16_t_offset); }
};

namespace {

int add_param(hct_mae_plan* p, const std::string& name, std::vector<int64_t> shape, bool rg, bool matrix, bool needs_t) {
  hct_param_info pi;
  memset(&pi, 0, sizeof(pi));
  snprintf(pi.name, sizeof(pi.name), "%s", name.c_str());
  pi.ndim = (int)shape.size();
  int64_t n = 1;
  for (size_t i = 0; i < shape.size(); ++i) { pi.shape[i] = shape[i]; n *= shape[i]; }
  pi.numel = n;
  pi.offset = p->param_elems;
  pi.requires_grad = rg ? 1 : 0;
  pi.is_matrix = matrix ? 1 : 0;
  pi.bf16_t_offset = -1;
  if (needs_t) {
    pi.bf16_t_offset = p->bf16t_elems;
    p->bf16t_elems += (n + 63) / 64 * 64;
  }
  p->param_elems += (n + kUnitElems - 1) / kUnitElems * kUnitElems;
  p->params.push_back(pi);
  p->pindex[name] = (int)p->params.size() - 1;
  return (int)p->params.size() - 1;
}

BlockP add_block(hct_mae_plan* p, const std::string& pre, int d, int m, bool use_bias) {
  BlockP b;
  b.ln1_w = add_param(p, pre + ".att_norm.weight", {d}, true, false, false);
  b.ln1_b = add_param(p, pre + ".att_norm.bias", {d}, true, false, false);
  b.qkv_w = add_param(p, pre + ".attn.qkv.weight", {3 * d, d}, true, true, true);
  b.qkv_b = use_bias ? add_param(p, pre + ".attn.qkv.bias", {3 * d}, true, false, false) : -1;
  b.proj_w = add_param(p, pre + ".attn.proj.weight", {d, d}, true, true, true);
  b.proj_b = add_param(p, pre + ".attn.proj.bias", {d}, true, false, false);
  b.ln2_w = add_param(p, pre + ".ffn_norm.weight", {d}, true, false, false);
  b.ln2_b = add_param(p, pre + ".ffn_norm.bias", {d}, true, false, false);
  b.fc1_w = add_param(p, pre + ".mlp.linear1.weight", {m, d}, true, true, true);
  b.fc1_b = add_param(p, pre + ".mlp.linear1.bias", {m}, true, false, false);
  b.fc2_w = add_param(p, pre + ".mlp.linear2.weight", {d, m}, true, true, true);
  b.fc2_b = add_param(p, pre + ".mlp.linear2.bias", {d}, true, false, false);
  return b;
}

// All transposed bf16 weight copies in ONE launch (82 matrices for ViT-B): table passed by value in the kernel arguments.
constexpr int kMaxT = 112;
struct TransposeTable {
  int n;
  int tile_start[kMaxT + 1];  // prefix sum of 64x64 tiles
  int rows[kMaxT], cols[kMaxT];
  long long src[kMaxT], dst[kMaxT];  // element offsets
};

__global__ void __launch_bounds__(256) transpose_cast_batched_kernel(const float* __restrict__ src, bf16* __restrict__ dst,
                                                                     TransposeTable t) {
  __shared__ float tile[64][65];
  int lo = 0, hi = t.n;  // tile_start[lo] <= blockIdx.x < tile_start[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (t.tile_start[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
  }
  const int rows = t.rows[lo], cols = t.cols[lo];
  const int local = blockIdx.x - t.tile_start[lo];
  const int tcols = (cols + 63) >> 6;
  const int r0 = (local / tcols) * 64, c0 = (local % tcols) * 64;
  const float* s = src + t.src[lo];
  bf16* d = dst + t.dst[lo];
  // reads: 16 lanes x 16 B per source row (4 rows per wave-instruction); writes: 16 lanes x 8 B per destination row.  All
  // weight matrices of the path have rows and cols that are multiples of 4 (plan creation checks the dims).
  const int q = threadIdx.x & 15, rr = threadIdx.x >> 4;  // 16 quads x 16 row slots
  const bool vec = (cols & 3) == 0 && (rows & 3) == 0;
  if (vec) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = r0 + rr + 16 * i, c = c0 + q * 4;
      f32x4 v = {0, 0, 0, 0};
      if (r < rows && c < cols) v = Vec4<float>::load(s + (size_t)r * cols + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) tile[rr + 16 * i][q * 4 + e] = v[e];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = c0 + rr + 16 * i, r = r0 + q * 4;
      if (c < cols && r < rows) {
        const f32x4 v = {tile[q * 4 + 0][rr + 16 * i], tile[q * 4 + 1][rr + 16 * i], tile[q * 4 + 2][rr + 16 * i], tile[q * 4 + 3][rr + 16 * i]};
        Vec4<bf16>::store(d + (size_t)c * rows + r, v);
      }
    }
    return;
  }
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? s[(size_t)r * cols + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows) d[(size_t)c * rows + r] = (bf16)tile[tx][i];
  }
}

struct WsAlloc {
  size_t cur = 0;
  size_t take(size_t bytes) {
    size_t o = cur;
    cur += align_up(bytes ? bytes : 16, 256);
    return o;
  }
};

BlockA alloc_block(WsAlloc& w, size_t M, size_t d, size_t m, size_t heads_tokens, size_t es) {
  BlockA a;
  a.x1 = w.take(M * d * es);
  a.mean1 = w.take(M * 4);
  a.rstd1 = w.take(M * 4);
  a.qkv = w.take(M * 3 * d * es);
  a.o = w.take(M * d * es);
  a.lse = w.take(heads_tokens * 4);
  a.h_mid = w.take(M * d * 4);
  a.x2 = w.take(M * d * es);
  a.mean2 = w.take(M * 4);
  a.rstd2 = w.take(M * 4);
  a.u = w.take(M * m * es);
  a.g = w.take(M * m * es);
  return a;
}

// ---- GEMM helpers -----------------------------------------------------------------------------------------------
hct_gemm_args base_args() {
  hct_gemm_args a;
  memset(&a, 0, sizeof(a));
  a.alpha = 1.0f;
  return a;
}

// Workspace of the forward / dgrad GEMMs: [s_small (column-sum partials) | stream-K region of the persistent NT kernel].  The
// two regions are adjacent in the plan's layout and the kernel takes its region from the END of what it is given, so its flags
// always sit at the same address, which nothing else writes; zeroed by the first GEMM after a bind.
int nt_gemm(hct_mae_plan* p, hct_gemm_args& a, hipStream_t s) {
  if (p->s_nt_bytes == 0) return hct_gemm(&a, a.colsum_out ? p->ws + p->s_small : nullptr, a.colsum_out ? p->s_small_bytes : 0, s);
  if (!p->nt_ws_armed) {
    if (hipMemsetAsync(p->ws + p->s_small + p->s_small_bytes, 0, p->s_nt_bytes, s) != hipSuccess) {
      set_error("plan: clearing the stream-K region of the GEMM workspace failed");
      return HCT_E_WORKSPACE;
    }
    p->nt_ws_armed = true;
  }
  a.workspace_armed = 1;
  return hct_gemm(&a, p->ws + p->s_small, p->s_small_bytes + p->s_nt_bytes, s);
}

// Y[M,N] = act(X[M,K] . W[N,K]^T + b) (+ residual)
int linear_fwd(hct_mae_plan* p, const void* X, int M, int K, int w, int b, int N, void* Y, int y_dtype, int act, void* aux,
               const float* residual, hipStream_t s) {
  hct_gemm_args a = base_args();
  a.M = M; a.N = N; a.K = K;
  a.A = X; a.a_dtype = p->dt; a.lda = K; a.transA = 0;
  a.B = p->wop(w); a.b_dtype = p->dt; a.ldb = K; a.transB = 1;
  a.C = Y; a.c_dtype = y_dtype; a.ldc = N;
  a.bias = p->pf(b);
  a.residual = residual; a.ldr = N;
  a.act = act; a.aux = aux; a.aux_dtype = p->dt; a.ldaux = N;
  return nt_gemm(p, a, s);
}

// dX[M,K] = dY[M,N] . W[N,K]   (optionally * gelu'(aux))
int linear_dgrad(hct_mae_plan* p, const void* dY, int M, int N, int w, int K, void* dX, int act, void* aux, hipStream_t s,
                 float* colsum_out = nullptr) {
  hct_gemm_args a = base_args();
  a.colsum_out = colsum_out;
  a.M = M; a.N = K; a.K = N;
  a.A = dY; a.a_dtype = p->dt; a.lda = N; a.transA = 0;
  if (p->dt == HCT_BF16) { a.B = p->wop_t(w); a.b_dtype = HCT_BF16; a.ldb = N; a.transB = 1; }  // W^T stored [K,N]
  else { a.B = p->wop(w); a.b_dtype = HCT_F32; a.ldb = K; a.transB = 0; }
  a.C = dX; a.c_dtype = p->dt; a.ldc = K;
  a.act = act; a.aux = aux; a.aux_dtype = p->dt; a.ldaux = K;
  return nt_gemm(p, a, s);
}

// dW[N,K] = dY[M,N]^T . X[M,K]  (fp32 gradient); optional bias gradient = colsum(dY)
int linear_wgrad(hct_mae_plan* p, const void* dY, const void* X, int M, int N, int K, int w, int b, hipStream_t s) {
  hct_gemm_args a = base_args();
  a.M = N; a.N = K; a.K = M;
  a.A = dY; a.a_dtype = p->dt; a.lda = N; a.transA = 1;
  a.B = X; a.b_dtype = p->dt; a.ldb = K; a.transB = 0;
  a.C = p->gf(w); a.c_dtype = HCT_F32; a.ldc = K;
  int rc = 0;
  // (a reduction of more than 4096 stages stays on the split-K launch: the workgroups of a window sweep a whole-tile reduction
  //  without meeting again, and over 10 000 stages -- DINO at 640 crops x 517 tokens -- they drift out of the L2's reach of each
  //  other: 281.9 against 278.4 ms per DINO iteration, where the MAE configurations gain 1.1 - 1.5 ms)
  if (p->wg_defer && M <= 131072 && tn_group_ok(&a)) {
    p->wg_pending.push_back(a);  // runs with the next grouped launch (flush_wgrads); dY and X stay untouched until then
  } else {
    a.workspace_armed = p->gemm_ws_armed ? 1 : 0;  // s_gemm is this plan's alone: its fold counters are reset by the first wgrad after a bind
    rc = hct_gemm(&a, p->ws + p->s_gemm, p->s_gemm_bytes, s);
    p->gemm_ws_armed = rc == 0;
    if (rc) return rc;
  }
  if (b >= 0) rc = hct_colsum(dY, p->dt, M, N, N, p->gf(b), p->ws + p->s_small2, p->s_small2_bytes, s);  // (s_small's head may hold a deferred fold's partials)
  return rc;
}

size_t wgrad_ws(int dt, int M, int N, int K) {
  hct_gemm_args a = base_args();
  a.M = N; a.N = K; a.K = M;
  a.a_dtype = dt; a.b_dtype = dt; a.lda = N; a.ldb = K; a.transA = 1; a.transB = 0; a.ldc = K; a.c_dtype = HCT_F32;
  a.A = (const void*)256; a.B = (const void*)256; a.C = (void*)256;  // alignment probes only
  return hct_gemm_workspace_bytes(&a);
}

#define RC(x)            \
  do {                   \
    int _rc = (x);       \
    if (_rc) return _rc; \
  } while (0)

// Run the queued weight gradients as one grouped launch.  Longest reductions first (stable): the whole-tile rounds are then
// homogeneous and the shorter products (the compact decoder tail, decoder_pred) end up in the last round and the stream-K
// remainder, which is balanced by reduction stages.  The job table of a slot (= the n-th flush of a backward) is rewritten only
// when the jobs differ from the ones it was last prepared for (pointers and shapes are the same every step).
int flush_wgrads(hct_mae_plan* p, hipStream_t s) {
  p->wg_blocks_pending = 0;
  if (p->wg_pending.empty()) return 0;
  std::vector<hct_gemm_args>& jobs = p->wg_pending;
  std::stable_sort(jobs.begin(), jobs.end(), [](const hct_gemm_args& x, const hct_gemm_args& y) { return x.K > y.K; });
  const int slot = p->wg_slot < kWgSlots ? p->wg_slot : kWgSlots - 1;
  ++p->wg_slot;
  const int n = (int)jobs.size();
  if (hct_gemm_tn_group_workspace_bytes(n) > p->s_wg_bytes) {
    set_error("plan: %d queued weight gradients exceed the grouped launch's workspace", n);
    return HCT_E_WORKSPACE;
  }
  std::vector<hct_gemm_args>& prep = p->wg_prepared[slot];
  if (prep.size() != jobs.size() || memcmp(prep.data(), jobs.data(), jobs.size() * sizeof(hct_gemm_args)) != 0) {
    RC(hct_gemm_tn_group_prepare(jobs.data(), n, p->ws + p->s_wg[slot], p->s_wg_bytes, s));
    prep = jobs;
  }
  RC(hct_gemm_tn_group_run(jobs.data(), n, p->ws + p->s_wg[slot], p->s_wg_bytes, s));
  jobs.clear();
  return 0;
}

// end of backward stage `stage`: flush where the policy says so, then move the "final" watermark
int end_stage(hct_mae_plan* p, int stage, bool block_stage, bool boundary, int ring, int depth, hipStream_t s) {
  if (block_stage) ++p->wg_blocks_pending;
  const bool last = stage + 1 == (int)p->stage_range.size();
  // a ring of r < depth gradient-operand sets allows at most r - 1 block stages between flushes (the r-th block would overwrite
  // the output gradient that the oldest queued product still reads); with one set per block nothing is ever overwritten
  int limit = p->wg_blocks > 0 ? p->wg_blocks : (1 << 30);
  if (ring > 0 && ring < depth && ring - 1 < limit) limit = ring - 1;
  if (last || boundary || p->wg_blocks_pending >= limit) RC(flush_wgrads(p, s));
  if (p->wg_pending.empty()) p->final_off = p->stage_range[stage].first;
  return 0;
}

// The MLP's saved activation is gelu'(pre-activation), written by the fc1 epilogue from the unrounded value; the fc2 dgrad then
// multiplies by it (HCT_ACT_GELU_D / HCT_ACT_MULAUX) instead of evaluating gelu' a second time from a bf16-rounded input.
#ifdef HCT_PLAN_RECOMPUTE_DGELU  // diagnostic build (A/B): the pre-activation is saved and gelu' recomputed in the backward
constexpr int kActFc1 = HCT_ACT_GELU, kActFc2Dgrad = HCT_ACT_DGELU;
#else
constexpr int kActFc1 = HCT_ACT_GELU_D, kActFc2Dgrad = HCT_ACT_MULAUX;
#endif

int block_forward(hct_mae_plan* p, const BlockP& bp, const BlockA& ba, const float* h_in, float* h_out, int B, int N, int d,
                  int m, int heads, hipStream_t s) {
  const int M = B * N;
  unsigned char* ws = p->ws;
  RC(hct_layernorm_fwd(h_in, p->pf(bp.ln1_w), p->pf(bp.ln1_b), M, d, 1e-5f, ws + ba.x1, p->dt, (float*)(ws + ba.mean1), (float*)(ws + ba.rstd1), s));
  RC(linear_fwd(p, ws + ba.x1, M, d, bp.qkv_w, bp.qkv_b, 3 * d, ws + ba.qkv, p->dt, HCT_ACT_NONE, nullptr, nullptr, s));
  RC(hct_attention_fwd(ws + ba.qkv, B, N, heads, d / heads, p->dt, ws + ba.o, (float*)(ws + ba.lse), s));
  RC(linear_fwd(p, ws + ba.o, M, d, bp.proj_w, bp.proj_b, d, ws + ba.h_mid, HCT_F32, HCT_ACT_NONE, nullptr, h_in, s));
  RC(hct_layernorm_fwd((const float*)(ws + ba.h_mid), p->pf(bp.ln2_w), p->pf(bp.ln2_b), M, d, 1e-5f, ws + ba.x2, p->dt, (float*)(ws + ba.mean2), (float*)(ws + ba.rstd2), s));
  RC(linear_fwd(p, ws + ba.x2, M, d, bp.fc1_w, bp.fc1_b, m, ws + ba.g, p->dt, kActFc1, ws + ba.u, nullptr, s));  // ba.u holds gelu'(pre-activation)
  RC(linear_fwd(p, ws + ba.g, M, m, bp.fc2_w, bp.fc2_b, d, h_out, HCT_F32, HCT_ACT_NONE, nullptr, (const float*)(ws + ba.h_mid), s));
  return 0;
}

// in: dh (fp32 [M,d]) + shadow (compute dtype) = gradient wrt the block output.  out: same buffers hold the gradient
// wrt the block input.  prev_fc2_b: bias index that receives colsum(d h_in) (the previous block's linear2 bias), or -1.
bool defer_folds() {  // HCT_DEFER_FOLDS=0: every fold as a launch of its own (A/B runs)
  static const bool on = [] { const char* v = getenv("HCT_DEFER_FOLDS"); return !(v && v[0] == '0'); }();
  return on;
}

int block_backward(hct_mae_plan* p, const BlockP& bp, const BlockA& ba, const BlockG& bg, size_t dhs_in_off, const float* h_in, int B, int N,
                   int d, int m, int heads, int prev_fc2_b, hipStream_t s) {
  const int M = B * N;
  unsigned char* ws = p->ws;
  float* dh = (float*)(ws + p->s_dh);
  void* dhs = ws + bg.out;        // gradient wrt the block output (written by the stage before)
  void* dhs_mid = ws + bg.mid;    // ... wrt h_mid
  void* dhs_in = ws + dhs_in_off; // ... wrt the block input = the next stage's `out`
  void* dbig = ws + bg.big;
  void* dx = ws + p->s_dx;
  void* d_o = ws + p->s_do;
  void* dqkv = ws + bg.qkv;
  // The three fixed-order folds of this block (fc1 bias column sums; ln2 and ln1 gamma / beta / column sums) are recorded and run as
  // ONE launch at the end (common.h, FoldSink): each keeps a partial buffer of its own until then -- s_small's head, s_fold_a,
  // s_fold_b -- and nothing else in between writes those.
  FoldSink sink;
  struct SinkScope {
    FoldSink* prev;
    explicit SinkScope(FoldSink* s_) : prev(g_fold_sink) { g_fold_sink = s_; }
    ~SinkScope() { g_fold_sink = prev; }
  } scope(defer_folds() ? &sink : g_fold_sink);
  // MLP branch
  RC(linear_wgrad(p, dhs, ws + ba.g, M, d, m, bp.fc2_w, -1, s));
  // d(pre-GELU) = (dh . W2) * gelu'(u); the linear1 bias gradient = column sums of this output rides in the same
  // epilogue (colsum_out: per-row-tile partials + a fixed-order fold; hct_gemm falls back to a separate pass over the
  // output where the fused instance does not apply)
  RC(linear_dgrad(p, dhs, M, d, bp.fc2_w, m, dbig, kActFc2Dgrad, ws + ba.u, s, p->gf(bp.fc1_b)));
  RC(linear_wgrad(p, dbig, ws + ba.x2, M, m, d, bp.fc1_w, -1, s));
  RC(linear_dgrad(p, dbig, M, m, bp.fc1_w, d, dx, HCT_ACT_NONE, nullptr, s));
  RC(hct_layernorm_bwd(dx, p->dt, (const float*)(ws + ba.h_mid), (const float*)(ws + ba.mean2), (const float*)(ws + ba.rstd2),
                       p->pf(bp.ln2_w), dh, M, d, dh, dhs_mid, p->dt, p->gf(bp.ln2_w), p->gf(bp.ln2_b), p->gf(bp.proj_b), ws + p->s_fold_a,
                       p->s_fold_bytes, s));
  // attention branch
  RC(linear_wgrad(p, dhs_mid, ws + ba.o, M, d, d, bp.proj_w, -1, s));
  RC(linear_dgrad(p, dhs_mid, M, d, bp.proj_w, d, d_o, HCT_ACT_NONE, nullptr, s));
  RC(hct_attention_bwd(ws + ba.qkv, ws + ba.o, d_o, (const float*)(ws + ba.lse), B, N, heads, d / heads, p->dt, dqkv, s));
  RC(linear_wgrad(p, dqkv, ws + ba.x1, M, 3 * d, d, bp.qkv_w, bp.qkv_b, s));
  RC(linear_dgrad(p, dqkv, M, 3 * d, bp.qkv_w, d, dx, HCT_ACT_NONE, nullptr, s));
  RC(hct_layernorm_bwd(dx, p->dt, h_in, (const float*)(ws + ba.mean1), (const float*)(ws + ba.rstd1), p->pf(bp.ln1_w), dh, M, d,
                       dh, dhs_in, p->dt, p->gf(bp.ln1_w), p->gf(bp.ln1_b), prev_fc2_b >= 0 ? p->gf(prev_fc2_b) : nullptr, ws + p->s_fold_b,
                       p->s_fold_bytes, s));
  return fold_flush(sink, s);
}

// First decoder block on cat rows (see hct_mae_plan::dec0).  Forward: LayerNorm1 + qkv on the Rcat rows, the full qkv matrix that
// the attention reads is assembled from them (a row gather); everything behind the attention as block_forward.
int block_forward_dec0(hct_mae_plan* p, const BlockP& bp, const BlockA& ba, const float* h_in, float* h_out, hipStream_t s) {
  const int B = p->B, N = p->Nd, d = p->Dd, m = p->Mlpd, heads = p->Hd, M = B * N, Rc = p->Rcat, Nc = p->Me;
  unsigned char* ws = p->ws;
  const int32_t* kept = (const int32_t*)(ws + p->a_kept_rows);
  float* hk = (float*)(ws + p->a_hk);
  RC(hct_gather_rows(h_in, kept, Nc, d * 4, hk, s));
  RC(dec0_table(p->pf(p->p_mask), p->pf(p->p_dpos), p->L, d, hk + (size_t)Nc * d, s));
  RC(hct_layernorm_fwd(hk, p->pf(bp.ln1_w), p->pf(bp.ln1_b), Rc, d, 1e-5f, ws + p->a_x1c, p->dt, (float*)(ws + p->a_mean_c), (float*)(ws + p->a_rstd_c), s));
  RC(linear_fwd(p, ws + p->a_x1c, Rc, d, bp.qkv_w, bp.qkv_b, 3 * d, ws + p->a_qkv_cat, p->dt, HCT_ACT_NONE, nullptr, nullptr, s));
  RC(hct_gather_rows(ws + p->a_qkv_cat, (const int32_t*)(ws + p->a_cat_idx), M, (int)(3 * d * p->esz()), ws + ba.qkv, s));
  RC(hct_attention_fwd(ws + ba.qkv, B, N, heads, d / heads, p->dt, ws + ba.o, (float*)(ws + ba.lse), s));
  RC(linear_fwd(p, ws + ba.o, M, d, bp.proj_w, bp.proj_b, d, ws + ba.h_mid, HCT_F32, HCT_ACT_NONE, nullptr, h_in, s));
  RC(hct_layernorm_fwd((const float*)(ws + ba.h_mid), p->pf(bp.ln2_w), p->pf(bp.ln2_b), M, d, 1e-5f, ws + ba.x2, p->dt, (float*)(ws + ba.mean2), (float*)(ws + ba.rstd2), s));
  RC(linear_fwd(p, ws + ba.x2, M, d, bp.fc1_w, bp.fc1_b, m, ws + ba.g, p->dt, kActFc1, ws + ba.u, nullptr, s));
  RC(linear_fwd(p, ws + ba.g, M, m, bp.fc2_w, bp.fc2_b, d, h_out, HCT_F32, HCT_ACT_NONE, nullptr, (const float*)(ws + ba.h_mid), s));
  return 0;
}

// Its backward: as block_backward down to the attention; the qkv gradient is then reduced to cat rows (kept rows copied, the masked
// rows summed per patch position: weight gradient, input gradient and LayerNorm backward are linear in it for a fixed row input), and
// the LayerNorm backward leaves the gradient wrt the kept / class rows of the decoder input in `a_dcat` (fp32) and `a_de` (compute
// dtype = the operand of decoder_embed's backward), the table rows' part behind them.  The residual gradient `s_dh` keeps the part of
// the masked rows that does not pass through LayerNorm1 (dec0_token_grads sums it).
int block_backward_dec0(hct_mae_plan* p, const BlockP& bp, const BlockA& ba, const BlockG& bg, hipStream_t s) {
  const int B = p->B, N = p->Nd, d = p->Dd, m = p->Mlpd, heads = p->Hd, M = B * N, Rc = p->Rcat;
  unsigned char* ws = p->ws;
  float* dh = (float*)(ws + p->s_dh);
  void* dhs = ws + bg.out;
  void* dhs_mid = ws + bg.mid;
  void* dbig = ws + bg.big;
  void* dx = ws + p->s_dx;
  void* d_o = ws + p->s_do;
  void* dqkv = ws + bg.qkv;
  const int32_t* kept = (const int32_t*)(ws + p->a_kept_rows);
  FoldSink sink;
  struct SinkScope {
    FoldSink* prev;
    explicit SinkScope(FoldSink* s_) : prev(g_fold_sink) { g_fold_sink = s_; }
    ~SinkScope() { g_fold_sink = prev; }
  } scope(defer_folds() ? &sink : g_fold_sink);
  RC(linear_wgrad(p, dhs, ws + ba.g, M, d, m, bp.fc2_w, -1, s));
  RC(linear_dgrad(p, dhs, M, d, bp.fc2_w, m, dbig, kActFc2Dgrad, ws + ba.u, s, p->gf(bp.fc1_b)));
  RC(linear_wgrad(p, dbig, ws + ba.x2, M, m, d, bp.fc1_w, -1, s));
  RC(linear_dgrad(p, dbig, M, m, bp.fc1_w, d, dx, HCT_ACT_NONE, nullptr, s));
  RC(hct_layernorm_bwd(dx, p->dt, (const float*)(ws + ba.h_mid), (const float*)(ws + ba.mean2), (const float*)(ws + ba.rstd2),
                       p->pf(bp.ln2_w), dh, M, d, dh, dhs_mid, p->dt, p->gf(bp.ln2_w), p->gf(bp.ln2_b), p->gf(bp.proj_b), ws + p->s_fold_a,
                       p->s_fold_bytes, s));
  RC(linear_wgrad(p, dhs_mid, ws + ba.o, M, d, d, bp.proj_w, -1, s));
  RC(linear_dgrad(p, dhs_mid, M, d, bp.proj_w, d, d_o, HCT_ACT_NONE, nullptr, s));
  RC(hct_attention_bwd(ws + ba.qkv, ws + ba.o, d_o, (const float*)(ws + ba.lse), B, N, heads, d / heads, p->dt, dqkv, s));
  if (bp.qkv_b >= 0) RC(hct_colsum(dqkv, p->dt, M, 3 * d, 3 * d, p->gf(bp.qkv_b), ws + p->s_small2, p->s_small2_bytes, s));  // bias: every row
  RC(dec0_aggregate(dqkv, p->dt, kept, (const int32_t*)(ws + p->a_ids_restore), B, p->L, p->K, 3 * d, ws + p->a_dqkv_cat, s));
  RC(linear_wgrad(p, ws + p->a_dqkv_cat, ws + p->a_x1c, Rc, 3 * d, d, bp.qkv_w, -1, s));
  RC(linear_dgrad(p, ws + p->a_dqkv_cat, Rc, 3 * d, bp.qkv_w, d, dx, HCT_ACT_NONE, nullptr, s));
  RC(hct_layernorm_bwd_mapped(dx, p->dt, (const float*)(ws + p->a_hk), (const float*)(ws + p->a_mean_c), (const float*)(ws + p->a_rstd_c),
                              p->pf(bp.ln1_w), dh, kept, Rc, d, (float*)(ws + p->a_dcat), ws + p->a_de, p->dt, p->gf(bp.ln1_w), p->gf(bp.ln1_b),
                              nullptr, ws + p->s_fold_b, p->s_fold_bytes, s));
  return fold_flush(sink, s);
}

// Last decoder block in compact-tail mode: LN1 / qkv / attention on every row, then the masked patches' rows of the attention
// output and of the residual stream are gathered (tail_rows) and proj / LN2 / MLP run on Mc rows.  h_out_c [Mc, d].
int block_forward_tail(hct_mae_plan* p, const BlockP& bp, const BlockA& ba, const float* h_in, float* h_out_c, hipStream_t s) {
  const int B = p->B, N = p->Nd, d = p->Dd, m = p->Mlpd, heads = p->Hd, M = B * N, Mc = p->Mc;
  unsigned char* ws = p->ws;
  const int32_t* rows = (const int32_t*)(ws + p->a_tail_rows);
  float* hin_c = (float*)(ws + p->s_dh);  // (a scratch buffer of the backward: free during the forward)
  RC(hct_layernorm_fwd(h_in, p->pf(bp.ln1_w), p->pf(bp.ln1_b), M, d, 1e-5f, ws + ba.x1, p->dt, (float*)(ws + ba.mean1), (float*)(ws + ba.rstd1), s));
  RC(linear_fwd(p, ws + ba.x1, M, d, bp.qkv_w, bp.qkv_b, 3 * d, ws + ba.qkv, p->dt, HCT_ACT_NONE, nullptr, nullptr, s));
  RC(hct_attention_fwd(ws + ba.qkv, B, N, heads, d / heads, p->dt, ws + ba.o, (float*)(ws + ba.lse), s));
  RC(hct_gather_rows(ws + ba.o, rows, Mc, (int)(d * p->esz()), ws + p->a_oc, s));
  RC(hct_gather_rows(h_in, rows, Mc, d * 4, hin_c, s));
  RC(linear_fwd(p, ws + p->a_oc, Mc, d, bp.proj_w, bp.proj_b, d, ws + ba.h_mid, HCT_F32, HCT_ACT_NONE, nullptr, hin_c, s));
  RC(hct_layernorm_fwd((const float*)(ws + ba.h_mid), p->pf(bp.ln2_w), p->pf(bp.ln2_b), Mc, d, 1e-5f, ws + ba.x2, p->dt, (float*)(ws + ba.mean2), (float*)(ws + ba.rstd2), s));
  RC(linear_fwd(p, ws + ba.x2, Mc, d, bp.fc1_w, bp.fc1_b, m, ws + ba.g, p->dt, kActFc1, ws + ba.u, nullptr, s));
  RC(linear_fwd(p, ws + ba.g, Mc, m, bp.fc2_w, bp.fc2_b, d, h_out_c, HCT_F32, HCT_ACT_NONE, nullptr, (const float*)(ws + ba.h_mid), s));
  return 0;
}

// Its backward.  in: s_dhc (fp32 [Mc,d]) + s_dh_shadow (compute dtype, compact rows) = gradient wrt the block's compact output.
// out: s_dh / s_dh_shadow hold the gradient wrt the block input on ALL rows.
int block_backward_tail(hct_mae_plan* p, const BlockP& bp, const BlockA& ba, const BlockG& bg, size_t dhs_in_off, const float* h_in,
                        int prev_fc2_b, hipStream_t s) {
  const int B = p->B, N = p->Nd, d = p->Dd, m = p->Mlpd, heads = p->Hd, M = B * N, Mc = p->Mc;
  unsigned char* ws = p->ws;
  float* dh = (float*)(ws + p->s_dh);
  float* dhc = (float*)(ws + p->s_dhc);
  void* dhs = ws + bg.out;         // compact rows
  void* dhs_mid = ws + bg.mid;     // compact rows
  void* dhs_in = ws + dhs_in_off;  // all rows
  void* dbig = ws + bg.big;
  void* dx = ws + p->s_dx;
  void* d_o = ws + p->s_do;
  void* dqkv = ws + bg.qkv;
  const int32_t* inv = (const int32_t*)(ws + p->a_tail_inv);
  FoldSink sink;
  struct SinkScope {
    FoldSink* prev;
    explicit SinkScope(FoldSink* s_) : prev(g_fold_sink) { g_fold_sink = s_; }
    ~SinkScope() { g_fold_sink = prev; }
  } scope(defer_folds() ? &sink : g_fold_sink);
  // MLP branch, compact rows
  RC(linear_wgrad(p, dhs, ws + ba.g, Mc, d, m, bp.fc2_w, -1, s));
  RC(linear_dgrad(p, dhs, Mc, d, bp.fc2_w, m, dbig, kActFc2Dgrad, ws + ba.u, s, p->gf(bp.fc1_b)));
  RC(linear_wgrad(p, dbig, ws + ba.x2, Mc, m, d, bp.fc1_w, -1, s));
  RC(linear_dgrad(p, dbig, Mc, m, bp.fc1_w, d, dx, HCT_ACT_NONE, nullptr, s));
  RC(hct_layernorm_bwd(dx, p->dt, (const float*)(ws + ba.h_mid), (const float*)(ws + ba.mean2), (const float*)(ws + ba.rstd2),
                       p->pf(bp.ln2_w), dhc, Mc, d, dhc, dhs_mid, p->dt, p->gf(bp.ln2_w), p->gf(bp.ln2_b), p->gf(bp.proj_b), ws + p->s_fold_a,
                       p->s_fold_bytes, s));
  // attention branch: proj on the compact rows, its input gradient scattered back (zeros on the rows the loss never sees)
  RC(linear_wgrad(p, dhs_mid, ws + p->a_oc, Mc, d, d, bp.proj_w, -1, s));
  RC(linear_dgrad(p, dhs_mid, Mc, d, bp.proj_w, d, dx, HCT_ACT_NONE, nullptr, s));
  RC(hct_gather_rows(dx, inv, M, (int)(d * p->esz()), d_o, s));
  RC(hct_attention_bwd(ws + ba.qkv, ws + ba.o, d_o, (const float*)(ws + ba.lse), B, N, heads, d / heads, p->dt, dqkv, s));
  RC(linear_wgrad(p, dqkv, ws + ba.x1, M, 3 * d, d, bp.qkv_w, bp.qkv_b, s));
  RC(linear_dgrad(p, dqkv, M, 3 * d, bp.qkv_w, d, dx, HCT_ACT_NONE, nullptr, s));
  // residual gradient of row r = the compact gradient's row tail_inv[r], nothing for the class token and the kept patches
  RC(hct_layernorm_bwd_mapped(dx, p->dt, h_in, (const float*)(ws + ba.mean1), (const float*)(ws + ba.rstd1), p->pf(bp.ln1_w), dhc, inv, M, d,
                              dh, dhs_in, p->dt, p->gf(bp.ln1_w), p->gf(bp.ln1_b), prev_fc2_b >= 0 ? p->gf(prev_fc2_b) : nullptr, ws + p->s_fold_b,
                              p->s_fold_bytes, s));
  return fold_flush(sink, s);
}

}  // namespace

extern "C" {

hct_mae_plan* hct_mae_plan_create(const hct_mae_config* c, int batch, int compute_dtype) {
  if (!c || batch <= 0 || (compute_dtype != HCT_F32 && compute_dtype != HCT_BF16)) { set_error("hct_mae_plan_create: bad arguments"); return nullptr; }
  if (c->patch_size <= 0 || c->input_size % c->patch_size || c->patch_size % 4) { set_error("hct_mae_plan_create: input_size %% patch_size != 0 or patch_size %% 4 != 0"); return nullptr; }
  const bool enc_only = c->encoder_only != 0;
  if (c->encoder_embed_dim % c->encoder_num_heads || (!enc_only && c->decoder_embed_dim % c->decoder_num_heads)) { set_error("hidden size should be divisible by num_heads"); return nullptr; }
  if (c->encoder_embed_dim % 4 || c->encoder_mlp_dim % 4 || (!enc_only && (c->decoder_embed_dim % 4 || c->decoder_mlp_dim % 4))) { set_error("embed/mlp dims must be multiples of 4"); return nullptr; }
  hct_mae_plan* p = new hct_mae_plan();
  p->cfg = *c;
  p->B = batch; p->dt = compute_dtype;
  p->g = c->input_size / c->patch_size;
  p->L = p->g * p->g * p->g;
  p->vit = c->encoder_only != 0;
  p->R = p->vit ? c->num_register_tokens : 0;
  p->norm_eps = (p->vit && c->final_norm_eps > 0.f) ? c->final_norm_eps : 1e-5f;
  if (p->R < 0) { set_error("num_register_tokens < 0"); delete p; return nullptr; }
  p->K = (int)((double)p->L * (1.0 - c->mask_ratio));  // int(L * (1 - mask_ratio)) in double, as Python evaluates mae.py:205
  if (p->vit) p->K = p->L;  // the plain ViT embeds every patch
  p->pd = c->in_chans * c->patch_size * c->patch_size * c->patch_size;
  p->Ne = p->K + 1 + p->R; p->Nd = p->vit ? 0 : p->L + 1;
  p->Me = batch * p->Ne; p->Md = batch * p->Nd;
  if (p->vit) p->cfg.decoder_depth = 0;  // no decoder: its loops below run zero times, its buffers have zero rows
  p->D = c->encoder_embed_dim; p->Dd = p->vit ? c->encoder_embed_dim : c->decoder_embed_dim;
  p->Mlp = c->encoder_mlp_dim; p->Mlpd = p->vit ? c->encoder_mlp_dim : c->decoder_mlp_dim;
  p->H = c->encoder_num_heads; p->Hd = p->vit ? c->encoder_num_heads : c->decoder_num_heads;
  if (p->K < 1) { set_error("mask_ratio leaves no visible patches"); delete p; return nullptr; }
  const bool ub = c->use_bias != 0;
  const int P = c->patch_size, C = c->in_chans;
  // ---- parameters, forward-use order ----
  std::vector<std::pair<int64_t, int64_t>> seg;  // per group [begin,end)
  int64_t g0 = p->param_elems;
  p->p_pe_w = add_param(p, "patch_embedding.patch_embeddings.weight", {p->D, C, P, P, P}, true, true, false);
  p->p_pe_b = add_param(p, "patch_embedding.patch_embeddings.bias", {p->D}, true, false, false);
  p->p_pos = c->pos_embed ? add_param(p, "patch_embedding.position_embeddings", {1, p->L, p->D}, true, false, false) : -1;
  p->p_cls = add_param(p, "cls_token", {1, 1, p->D}, true, false, false);
  if (p->R > 0) p->p_reg = add_param(p, "register_tokens", {1, p->R, p->D}, true, false, false);
  seg.push_back({g0, p->param_elems});
  for (int i = 0; i < c->encoder_depth; ++i) {
    g0 = p->param_elems;
    p->enc.push_back(add_block(p, "blocks." + std::to_string(i), p->D, p->Mlp, ub));
    seg.push_back({g0, p->param_elems});
  }
  g0 = p->param_elems;
  p->p_norm_w = add_param(p, "norm.weight", {p->D}, true, false, false);
  p->p_norm_b = add_param(p, "norm.bias", {p->D}, true, false, false);
  if (p->vit) {
    seg.push_back({g0, p->param_elems});
    p->p_de_w = p->p_de_b = p->p_mask = p->p_dcls = p->p_dpos = p->p_dnorm_w = p->p_dnorm_b = p->p_pred_w = p->p_pred_b = -1;
  } else {
  p->p_de_w = add_param(p, "decoder_embed.weight", {p->Dd, p->D}, true, true, true);
  p->p_de_b = ub ? add_param(p, "decoder_embed.bias", {p->Dd}, true, false, false) : -1;
  p->p_mask = add_param(p, "mask_token", {1, 1, p->Dd}, true, false, false);
  p->p_dcls = add_param(p, "decoder_cls_token", {1, 1, p->Dd}, true, false, false);
  p->p_dpos = add_param(p, "decoder_pos_embed", {1, p->L, p->Dd}, false, false, false);  // mae.py:92 frozen
  seg.push_back({g0, p->param_elems});
  for (int i = 0; i < p->cfg.decoder_depth; ++i) {
    g0 = p->param_elems;
    p->dec.push_back(add_block(p, "decoder_blocks." + std::to_string(i), p->Dd, p->Mlpd, ub));
    seg.push_back({g0, p->param_elems});
  }
  g0 = p->param_elems;
  p->p_dnorm_w = add_param(p, "decoder_norm.weight", {p->Dd}, true, false, false);
  p->p_dnorm_b = add_param(p, "decoder_norm.bias", {p->Dd}, true, false, false);
  p->p_pred_w = add_param(p, "decoder_pred.weight", {p->pd, p->Dd}, true, true, true);
  p->p_pred_b = ub ? add_param(p, "decoder_pred.bias", {p->pd}, true, false, false) : -1;
  seg.push_back({g0, p->param_elems});
  }
  // backward stages complete the groups in reverse
  for (int i = (int)seg.size() - 1; i >= 0; --i) p->stage_range.push_back(seg[i]);

  // ---- workspace ----
  WsAlloc w;
  const size_t es = p->esz();
  const size_t B = batch, L = p->L, K = p->K, Me = p->Me, Md = p->Md, D = p->D, Dd = p->Dd;
  p->a_ids_restore = w.take(B * L * 4);
  p->a_ids_shuffle = w.take(B * L * 4);
  p->a_mask = w.take(B * L * 4);
  p->a_row_loss = w.take(B * L * 4);
  p->a_patches = w.take(B * K * p->pd * es);
  p->a_tok = w.take(B * K * D * es);
  for (int i = 0; i <= c->encoder_depth; ++i) p->h_enc.push_back(w.take(Me * D * 4));
  for (int i = 0; i < c->encoder_depth; ++i) p->aenc.push_back(alloc_block(w, Me, D, p->Mlp, (size_t)batch * p->H * p->Ne, es));
  p->a_latent = w.take(Me * D * es);
  p->a_lat_mean = w.take(Me * 4);
  p->a_lat_rstd = w.take(Me * 4);
  p->a_e = w.take(Me * Dd * es);
  for (int i = 0; i <= p->cfg.decoder_depth; ++i) p->h_dec.push_back(w.take(Md * Dd * 4));
  for (int i = 0; i < p->cfg.decoder_depth; ++i) p->adec.push_back(alloc_block(w, Md, Dd, p->Mlpd, (size_t)batch * p->Hd * p->Nd, es));
  p->a_ynorm = w.take(Md * Dd * es);
  p->a_yn_mean = w.take(Md * 4);
  p->a_yn_rstd = w.take(Md * 4);
  p->a_pred = w.take(Md * p->pd * es);
  p->a_dpred = w.take(Md * p->pd * es);
  p->Mc = p->vit ? 0 : batch * (p->L - p->K);
  p->tail_ok = !p->vit && p->cfg.decoder_depth >= 1 && p->Mc > 0 && (Dd * es) % 16 == 0;
  p->a_tail_rows = w.take((size_t)p->Mc * 4);
  p->a_tail_inv = w.take(Md * 4);
  p->a_oc = w.take((size_t)p->Mc * Dd * es);
  p->s_dhc = w.take((size_t)p->Mc * Dd * 4);
  const size_t Mx = Md > Me ? Md : Me, Dx = Dd > D ? Dd : D;
  const size_t mlpx = (size_t)(p->Mlp > p->Mlpd ? p->Mlp : p->Mlpd);
  p->s_dh = w.take(Mx * Dx * 4);
  p->s_dh_shadow = w.take(Mx * Dx * es);  // (gradient wrt a side's first block input in the compute dtype: written, read by nobody)
  p->s_dbig = p->s_dqkv = 0;              // (per block now: BlockG)
  p->s_dx = w.take(Mx * Dx * es);
  p->s_do = w.take(Mx * Dx * es);
  // Gradient operands per block (see BlockG): one set per block while that stays under 32 GB per side, otherwise a ring of as
  // many sets as fit (at least 2: the grouped launches are then flushed every ring - 1 block stages, end_stage).
  auto alloc_ring = [&](std::vector<BlockG>& ring, int depth, size_t M, size_t d, size_t m) {
    const size_t set_bytes = M * (5 * d + m) * es, budget = (size_t)32 << 30;
    int nsets = depth;
    if (depth > 0 && set_bytes * (size_t)depth > budget) nsets = std::max(2, (int)(budget / set_bytes));
    nsets = std::min(nsets, depth);
    for (int i = 0; i < nsets; ++i) {
      BlockG g;
      g.out = w.take(M * d * es);
      g.big = w.take(M * m * es);
      g.mid = w.take(M * d * es);
      g.qkv = w.take(M * 3 * d * es);
      ring.push_back(g);
    }
  };
  alloc_ring(p->genc, c->encoder_depth, Me, D, (size_t)p->Mlp);
  alloc_ring(p->gdec, p->cfg.decoder_depth, Md, Dd, (size_t)p->Mlpd);
  p->Rcat = p->vit ? 0 : (int)(Me + L);
  p->dec0_ok = !p->vit && p->cfg.decoder_depth >= 2 && (Dd * es) % 16 == 0 && (3 * Dd * es) % 16 == 0;
  {
    const char* ev = getenv("HCT_DEC0_TABLE");
    p->dec0 = p->dec0_ok && !(ev && ev[0] == '0');
    const size_t Rc = (size_t)p->Rcat;
    p->a_kept_rows = w.take(Rc * 4);
    p->a_cat_idx = w.take(Md * 4);
    p->a_hk = w.take(Rc * Dd * 4);
    p->a_x1c = w.take(Rc * Dd * es);
    p->a_mean_c = w.take(Rc * 4);
    p->a_rstd_c = w.take(Rc * 4);
    p->a_qkv_cat = w.take(Rc * 3 * Dd * es);
    p->a_dqkv_cat = w.take(Rc * 3 * Dd * es);
    p->a_dcat = w.take(Rc * Dd * 4);
  }
  p->a_de = w.take((Me + L) * Dd * es);  // (+ L rows: the table rows' part of the cat gradient's shadow)
  p->a_dtok = w.take(B * K * D * es);
  size_t small = hct_layernorm_bwd_workspace_bytes((int)Mx, (int)Dx);
  small = std::max(small, hct_assemble_bwd_workspace_bytes((int)Dx));
  small = std::max(small, hct_colsum_workspace_bytes((int)Mx, (int)(3 * Dx)));
  small = std::max(small, hct_colsum_workspace_bytes((int)Mx, (int)mlpx));
  small = std::max(small, hct_colsum_workspace_bytes((int)Md, p->pd));
  small = std::max(small, (size_t)((Mx + 255) / 256) * 4 * mlpx * sizeof(float));  // fused colsum partials of the dGELU dgrad
  small = align_up(small, 256);
  p->s_fold_bytes = hct_layernorm_bwd_workspace_bytes((int)Mx, (int)Dx);
  p->s_fold_a = w.take(p->s_fold_bytes);
  p->s_fold_b = w.take(p->s_fold_bytes);
  p->s_small2_bytes = small;  // (scratch of the wgrads' bias column sums: any of the sizes s_small was sized for)
  p->s_small2 = w.take(p->s_small2_bytes);
  p->s_small_bytes = small;
  p->s_small = w.take(small);
  if (p->dt == HCT_BF16) {  // stream-K region of the persistent NT GEMM, directly behind s_small (see nt_gemm)
    p->s_nt_bytes = hct_gemm_nt_stream_k_bytes();
    w.take(p->s_nt_bytes);
  }
  size_t gw = 0;
  for (int side = 0; side < 2; ++side) {
    const int M = side ? p->Md : p->Me, d = side ? p->Dd : p->D, m = side ? p->Mlpd : p->Mlp;
    gw = std::max(gw, wgrad_ws(p->dt, M, d, m));
    gw = std::max(gw, wgrad_ws(p->dt, M, m, d));
    gw = std::max(gw, wgrad_ws(p->dt, M, d, d));
    gw = std::max(gw, wgrad_ws(p->dt, M, 3 * d, d));
  }
  gw = std::max(gw, wgrad_ws(p->dt, p->Md, p->pd, p->Dd));
  gw = std::max(gw, wgrad_ws(p->dt, p->Me, p->Dd, p->D));
  gw = std::max(gw, wgrad_ws(p->dt, batch * p->K, p->D, p->pd));
  p->s_gemm_bytes = gw;
  p->s_gemm = w.take(gw);
  {  // grouped weight-gradient launches: one workspace (job table + stream-K slabs) per flush slot
    const char* ev = getenv("HCT_WGRAD_DEFER");
    p->wg_defer = p->dt == HCT_BF16 && !(ev && ev[0] == '0');
    const char* eb = getenv("HCT_WGRAD_GROUP_BLOCKS");
    p->wg_blocks = eb && *eb ? std::max(0, atoi(eb)) : 0;
    p->s_wg_bytes = hct_gemm_tn_group_workspace_bytes(4 * (c->encoder_depth + p->cfg.decoder_depth) + 4);
    for (int i = 0; i < kWgSlots; ++i) p->s_wg[i] = p->wg_defer ? w.take(p->s_wg_bytes) : 0;
  }
  p->final_off = p->param_elems;
  p->ws_bytes = w.cur;

  // ---- named activations (parity tests) ----
  auto reg = [&](const std::string& n, size_t off, int64_t r, int64_t cc, int dtp) { p->acts[n] = Act{off, r, cc, dtp}; };
  reg("ids_restore", p->a_ids_restore, batch, p->L, 2);
  reg("ids_shuffle", p->a_ids_shuffle, batch, p->L, 2);
  reg("mask", p->a_mask, batch, p->L, HCT_F32);
  reg("patches", p->a_patches, (int64_t)batch * p->K, p->pd, p->dt);
  reg("tok", p->a_tok, (int64_t)batch * p->K, p->D, p->dt);
  reg("enc_in", p->h_enc[0], p->Me, p->D, HCT_F32);
  for (int i = 0; i < c->encoder_depth; ++i) reg("enc" + std::to_string(i) + ".out", p->h_enc[i + 1], p->Me, p->D, HCT_F32);
  reg("latent", p->a_latent, p->Me, p->D, p->dt);
  reg("dec_in", p->h_dec[0], p->Md, p->Dd, HCT_F32);
  for (int i = 0; i < p->cfg.decoder_depth; ++i) reg("dec" + std::to_string(i) + ".out", p->h_dec[i + 1], p->Md, p->Dd, HCT_F32);
  reg("tail_rows", p->a_tail_rows, 1, p->Mc, 2);
  reg("pred_full", p->a_pred, p->Md, p->pd, p->dt);
  reg("dpred_full", p->a_dpred, p->Md, p->pd, p->dt);
  if (c->encoder_depth > 0) {
    reg("enc0.qkv", p->aenc[0].qkv, p->Me, 3 * p->D, p->dt);
    reg("enc0.attn_o", p->aenc[0].o, p->Me, p->D, p->dt);
  }
  return p;
}

void hct_mae_plan_destroy(hct_mae_plan* p) { delete p; }
int hct_mae_plan_num_params(const hct_mae_plan* p) { return (int)p->params.size(); }
int hct_mae_plan_param_info(const hct_mae_plan* p, int index, hct_param_info* out) {
  HCT_REQUIRE(p && out && index >= 0 && index < (int)p->params.size(), "hct_mae_plan_param_info: bad index");
  *out = p->params[index];
  return 0;
}
int64_t hct_mae_plan_param_elems(const hct_mae_plan* p) { return p->param_elems; }
int64_t hct_mae_plan_bf16_t_elems(const hct_mae_plan* p) { return p->bf16t_elems; }
size_t hct_mae_plan_workspace_bytes(const hct_mae_plan* p) { return p->ws_bytes; }
int hct_mae_plan_len_keep(const hct_mae_plan* p) { return p ? p->K : -1; }
int hct_mae_plan_set_dec0(hct_mae_plan* p, int on) {
  if (!p) return -1;
  p->dec0 = on != 0 && p->dec0_ok;
  return p->dec0 ? 1 : 0;
}
int hct_mae_plan_set_tail(hct_mae_plan* p, int compact) {
  if (!p) return -1;
  p->tail = compact != 0 && p->tail_ok;
  return p->tail ? 1 : 0;
}

int hct_mae_plan_bind(hct_mae_plan* p, float* params, float* grads, void* params_bf16, void* params_bf16_t, void* workspace,
                      size_t workspace_bytes) {
  HCT_REQUIRE(p && params && workspace, "hct_mae_plan_bind: null buffer");
  HCT_REQUIRE(p->dt == HCT_F32 || (params_bf16 && params_bf16_t), "hct_mae_plan_bind: bf16 mode needs the bf16 weight buffers");
  if (workspace_bytes < p->ws_bytes) { set_error("hct_mae_plan_bind: workspace too small"); return HCT_E_WORKSPACE; }
  p->params_f32 = params; p->grads = grads;
  p->params_bf16 = (bf16*)params_bf16; p->params_bf16_t = (bf16*)params_bf16_t;
  p->ws = (unsigned char*)workspace;
  p->gemm_ws_armed = false;
  p->nt_ws_armed = false;
  p->fwd_done = false;
  for (auto& v : p->wg_prepared) v.clear();  // the job tables held the old buffers' addresses
  p->wg_pending.clear();
  return 0;
}

int hct_mae_refresh_weights(hct_mae_plan* p, int with_plain, void* stream) {
  HCT_REQUIRE(p && p->params_f32, "hct_mae_refresh_weights: plan not bound");
  if (p->dt != HCT_BF16) return 0;
  if (with_plain) RC(hct_cast(p->params_f32, HCT_F32, p->params_bf16, HCT_BF16, p->param_elems, stream));
  TransposeTable t;
  t.n = 0;
  t.tile_start[0] = 0;
  auto flush = [&]() -> int {
    if (t.n == 0) return 0;
    hipLaunchKernelGGL(transpose_cast_batched_kernel, dim3(t.tile_start[t.n]), dim3(256), 0, (hipStream_t)stream, p->params_f32,
                       p->params_bf16_t, t);
    t.n = 0;
    return check_hip(hipGetLastError(), "hct_mae_refresh_weights");
  };
  for (const auto& pi : p->params) {
    if (pi.bf16_t_offset < 0) continue;
    const int rows = (int)pi.shape[0], cols = (int)(pi.numel / pi.shape[0]);
    t.rows[t.n] = rows; t.cols[t.n] = cols; t.src[t.n] = pi.offset; t.dst[t.n] = pi.bf16_t_offset;
    t.tile_start[t.n + 1] = t.tile_start[t.n] + ((rows + 63) / 64) * ((cols + 63) / 64);
    if (++t.n == kMaxT) RC(flush());
  }
  RC(flush());
  return 0;
}

int hct_mae_forward(hct_mae_plan* p, const void* x, int x_dtype, const float* noise, float* loss, float grad_scale, void* stream) {
  HCT_REQUIRE(p && p->ws && x && noise && loss, "hct_mae_forward: plan not bound or null argument");
  HCT_REQUIRE(!p->vit, "hct_mae_forward: encoder-only plans are driven by hct_vit_forward");
  HCT_REQUIRE(x_dtype == HCT_F32 || x_dtype == HCT_F16, "hct_mae_forward: volumes are fp32 or fp16");
  hipStream_t s = (hipStream_t)stream;
  unsigned char* ws = p->ws;
  const hct_mae_config& c = p->cfg;
  const int B = p->B;
  int32_t* ids_restore = (int32_t*)(ws + p->a_ids_restore);
  int32_t* ids_shuffle = (int32_t*)(ws + p->a_ids_shuffle);
  float* mask = (float*)(ws + p->a_mask);
  RC(hct_mask_rank(noise, B, p->L, p->K, ids_restore, ids_shuffle, mask, s));
  RC(hct_patch_gather(x, x_dtype, ids_shuffle, B, c.in_chans, c.input_size, c.patch_size, p->L, p->K, ws + p->a_patches, p->dt, s));
  RC(linear_fwd(p, ws + p->a_patches, B * p->K, p->pd, p->p_pe_w, p->p_pe_b, p->D, ws + p->a_tok, p->dt, HCT_ACT_NONE, nullptr, nullptr, s));
  RC(hct_encoder_assemble_fwd(ws + p->a_tok, p->dt, p->pf(p->p_cls), p->pf(p->p_pos), ids_shuffle, B, p->L, p->K, p->D, (float*)(ws + p->h_enc[0]), s));
  for (int i = 0; i < c.encoder_depth; ++i)
    RC(block_forward(p, p->enc[i], p->aenc[i], (const float*)(ws + p->h_enc[i]), (float*)(ws + p->h_enc[i + 1]), B, p->Ne, p->D, p->Mlp, p->H, s));
  RC(hct_layernorm_fwd((const float*)(ws + p->h_enc[c.encoder_depth]), p->pf(p->p_norm_w), p->pf(p->p_norm_b), p->Me, p->D, 1e-5f,
                       ws + p->a_latent, p->dt, (float*)(ws + p->a_lat_mean), (float*)(ws + p->a_lat_rstd), s));
  RC(linear_fwd(p, ws + p->a_latent, p->Me, p->D, p->p_de_w, p->p_de_b, p->Dd, ws + p->a_e, p->dt, HCT_ACT_NONE, nullptr, nullptr, s));
  RC(hct_decoder_assemble_fwd(ws + p->a_e, p->dt, p->pf(p->p_mask), p->pf(p->p_dcls), p->pf(p->p_dpos), ids_restore, B, p->L, p->K, p->Dd,
                              (float*)(ws + p->h_dec[0]), s));
  const bool tail = p->tail && p->tail_ok;
  p->tail_fwd = tail;
  if (tail) RC(hct_tail_rows(ids_restore, B, p->L, p->K, (int32_t*)(ws + p->a_tail_rows), (int32_t*)(ws + p->a_tail_inv), s));
  const bool dec0 = p->dec0 && p->dec0_ok;
  p->dec0_fwd = dec0;
  if (dec0) RC(dec0_index(ids_restore, B, p->L, p->K, (int32_t*)(ws + p->a_kept_rows), (int32_t*)(ws + p->a_cat_idx), s));
  for (int i = 0; i < c.decoder_depth; ++i) {
    if (dec0 && i == 0)
      RC(block_forward_dec0(p, p->dec[i], p->adec[i], (const float*)(ws + p->h_dec[i]), (float*)(ws + p->h_dec[i + 1]), s));
    else if (tail && i == c.decoder_depth - 1)
      RC(block_forward_tail(p, p->dec[i], p->adec[i], (const float*)(ws + p->h_dec[i]), (float*)(ws + p->h_dec[i + 1]), s));
    else
      RC(block_forward(p, p->dec[i], p->adec[i], (const float*)(ws + p->h_dec[i]), (float*)(ws + p->h_dec[i + 1]), B, p->Nd, p->Dd, p->Mlpd, p->Hd, s));
  }
  const int Mt = tail ? p->Mc : p->Md;  // rows of the decoder's tail: the masked patches' (compact), or all
  RC(hct_layernorm_fwd((const float*)(ws + p->h_dec[c.decoder_depth]), p->pf(p->p_dnorm_w), p->pf(p->p_dnorm_b), Mt, p->Dd, 1e-5f,
                       ws + p->a_ynorm, p->dt, (float*)(ws + p->a_yn_mean), (float*)(ws + p->a_yn_rstd), s));
  RC(linear_fwd(p, ws + p->a_ynorm, Mt, p->Dd, p->p_pred_w, p->p_pred_b, p->pd, ws + p->a_pred, p->dt, HCT_ACT_NONE, nullptr, nullptr, s));
  // One pass over the volume and the prediction gives the loss and, in a training forward (grad_scale != 0), the seed of
  // the backward d(loss)/d(pred) * grad_scale as well (grad_scale = the data-parallel 1 / world_size, known on the host);
  // the incoming dLoss is a device scalar that is only known in the backward and is applied there if it is not 1.
  const bool train = grad_scale != 0.0f;
  RC(masked_mse_launch(ws + p->a_pred, p->dt, x, x_dtype, mask, B, c.in_chans, c.input_size, c.patch_size, c.norm_pix_loss,
                       (float)((int64_t)B * (p->L - p->K)), (float*)(ws + p->a_row_loss), loss, train ? ws + p->a_dpred : nullptr, nullptr,
                       train ? grad_scale : 1.0f, s, tail ? ids_shuffle : nullptr, p->K));
  p->fwd_done = true;
  p->dpred_done = train;
  return 0;
}

int hct_vit_forward(hct_mae_plan* p, const void* x, int x_dtype, void* stream) {
  HCT_REQUIRE(x, "hct_vit_forward: null argument");
  return hct_vit_forward_parts(p, &x, 1, x_dtype, stream);
}

int hct_vit_forward_parts(hct_mae_plan* p, const void* const* xs, int n_parts, int x_dtype, void* stream) {
  HCT_REQUIRE(p && p->ws && xs && n_parts > 0, "hct_vit_forward: plan not bound or null argument");
  HCT_REQUIRE(p->vit, "hct_vit_forward: the plan was not created with encoder_only = 1");
  HCT_REQUIRE(x_dtype == HCT_F32 || x_dtype == HCT_F16, "hct_vit_forward: volumes are fp32 or fp16");
  HCT_REQUIRE(p->B % n_parts == 0, "hct_vit_forward_parts: the plan's batch (%d) is not a multiple of the number of parts (%d)", p->B, n_parts);
  hipStream_t s = (hipStream_t)stream;
  unsigned char* ws = p->ws;
  const hct_mae_config& c = p->cfg;
  const int B = p->B, Bp = B / n_parts;
  // PatchEmbeddingBlock on every patch (patch_embedding.py:149-156), class token and register tokens (vit.py:147-160).  The batch
  // may arrive as several equally sized tensors (the crops MultiCropWrapper concatenates, misc.py:467-480): the patch rows are
  // gathered straight from each of them, which is the concatenation without its copy.
  for (int i = 0; i < n_parts; ++i) {
    HCT_REQUIRE(xs[i], "hct_vit_forward_parts: null part %d", i);
    RC(hct_patch_gather(xs[i], x_dtype, nullptr, Bp, c.in_chans, c.input_size, c.patch_size, p->L, p->L,
                        ws + p->a_patches + (size_t)i * Bp * p->L * p->pd * p->esz(), p->dt, s));
  }
  RC(linear_fwd(p, ws + p->a_patches, B * p->L, p->pd, p->p_pe_w, p->p_pe_b, p->D, ws + p->a_tok, p->dt, HCT_ACT_NONE, nullptr, nullptr, s));
  RC(hct_vit_assemble_fwd(ws + p->a_tok, p->dt, p->pf(p->p_cls), p->p_reg >= 0 ? p->pf(p->p_reg) : nullptr, p->p_pos >= 0 ? p->pf(p->p_pos) : nullptr, B,
                          p->L, p->R, p->D, (float*)(ws + p->h_enc[0]), s));
  for (int i = 0; i < c.encoder_depth; ++i)
    RC(block_forward(p, p->enc[i], p->aenc[i], (const float*)(ws + p->h_enc[i]), (float*)(ws + p->h_enc[i + 1]), B, p->Ne, p->D, p->Mlp, p->H, s));
  RC(hct_layernorm_fwd((const float*)(ws + p->h_enc[c.encoder_depth]), p->pf(p->p_norm_w), p->pf(p->p_norm_b), p->Me, p->D, p->norm_eps,
                       ws + p->a_latent, p->dt, (float*)(ws + p->a_lat_mean), (float*)(ws + p->a_lat_rstd), s));
  p->fwd_done = true;
  return 0;
}

int hct_vit_backward_stage(hct_mae_plan* p, int stage, const void* dlatent, void* stream) {
  HCT_REQUIRE(p && p->ws && p->grads && p->vit, "hct_vit_backward_stage: plan not bound, no gradient buffer, or not an encoder-only plan");
  if (!p->fwd_done) { set_error("hct_vit_backward_stage: forward has not run"); return HCT_E_STATE; }
  hipStream_t s = (hipStream_t)stream;
  unsigned char* ws = p->ws;
  const int B = p->B, ne = p->cfg.encoder_depth;
  float* dh = (float*)(ws + p->s_dh);
  void* small = ws + p->s_small;
  const int Re = (int)p->genc.size();
  auto enc_out = [&](int i) { return i >= 0 ? p->genc[i % Re].out : p->s_dh_shadow; };  // gradient wrt block i's output (compute dtype)
  if (stage == 0) {  // final norm
    HCT_REQUIRE(dlatent, "hct_vit_backward_stage: stage 0 needs dlatent");
    p->wg_pending.clear(); p->wg_slot = 0; p->wg_blocks_pending = 0; p->final_off = p->param_elems;
    RC(hct_layernorm_bwd(dlatent, p->dt, (const float*)(ws + p->h_enc[ne]), (const float*)(ws + p->a_lat_mean), (const float*)(ws + p->a_lat_rstd),
                         p->pf(p->p_norm_w), nullptr, p->Me, p->D, dh, ws + enc_out(ne - 1), p->dt, p->gf(p->p_norm_w), p->gf(p->p_norm_b),
                         ne > 0 ? p->gf(p->enc[ne - 1].fc2_b) : nullptr, small, p->s_small_bytes, s));
    return end_stage(p, stage, false, false, Re, ne, s);
  }
  if (stage <= ne) {
    const int i = ne - stage;
    RC(block_backward(p, p->enc[i], p->aenc[i], p->genc[i % Re], enc_out(i - 1), (const float*)(ws + p->h_enc[i]), B, p->Ne, p->D, p->Mlp, p->H,
                      i > 0 ? p->enc[i - 1].fc2_b : -1, s));
    return end_stage(p, stage, true, false, Re, ne, s);
  }
  if (stage == ne + 1) {  // input assembly -> patch embedding
    void* dtok = ws + p->a_dtok;
    RC(hct_vit_assemble_bwd(dh, B, p->L, p->R, p->D, dtok, p->dt, p->gf(p->p_cls), p->p_reg >= 0 ? p->gf(p->p_reg) : nullptr,
                            p->p_pos >= 0 ? p->gf(p->p_pos) : nullptr, s));
    RC(linear_wgrad(p, dtok, ws + p->a_patches, B * p->L, p->D, p->pd, p->p_pe_w, p->p_pe_b, s));
    return end_stage(p, stage, false, false, Re, ne, s);
  }
  set_error("hct_vit_backward_stage: stage %d out of range", stage);
  return HCT_E_BADARG;
}

int hct_mae_set_loss_grad(hct_mae_plan* p, const float* dloss) {
  HCT_REQUIRE(p, "hct_mae_set_loss_grad: null plan");
  p->dloss = dloss;
  return 0;
}

int hct_mae_num_backward_stages(const hct_mae_plan* p) { return (int)p->stage_range.size(); }
int64_t hct_mae_backward_final_offset(const hct_mae_plan* p) { return p ? p->final_off : -1; }
int hct_mae_plan_set_wgrad_defer(hct_mae_plan* p, int defer, int group_blocks) {
  if (!p) return -1;
  p->wg_defer = defer != 0 && p->dt == HCT_BF16 && p->s_wg_bytes > 0 && p->s_wg[kWgSlots - 1] != 0;
  p->wg_blocks = group_blocks > 0 ? group_blocks : 0;
  return p->wg_defer ? 1 : 0;
}
int hct_mae_backward_stage_range(const hct_mae_plan* p, int stage, int64_t* begin, int64_t* end) {
  HCT_REQUIRE(p && stage >= 0 && stage < (int)p->stage_range.size(), "hct_mae_backward_stage_range: bad stage");
  *begin = p->stage_range[stage].first;
  *end = p->stage_range[stage].second;
  return 0;
}

int hct_mae_backward_stage(hct_mae_plan* p, int stage, void* stream) {
  HCT_REQUIRE(p && p->ws && p->grads, "hct_mae_backward_stage: plan not bound (or no gradient buffer)");
  if (!p->fwd_done) { set_error("hct_mae_backward_stage: forward has not run"); return HCT_E_STATE; }
  hipStream_t s = (hipStream_t)stream;
  unsigned char* ws = p->ws;
  const hct_mae_config& c = p->cfg;
  const int B = p->B;
  const int nd = c.decoder_depth, ne = c.encoder_depth;
  float* dh = (float*)(ws + p->s_dh);
  void* dx = ws + p->s_dx;
  void* small = ws + p->s_small;
  const int Re = (int)p->genc.size(), Rd = (int)p->gdec.size();
  auto enc_out = [&](int i) { return i >= 0 ? p->genc[i % Re].out : p->s_dh_shadow; };  // gradient wrt encoder block i's output (compute dtype)
  auto dec_out = [&](int i) { return i >= 0 ? p->gdec[i % Rd].out : p->s_dh_shadow; };
  if (stage == 0) {  // loss seed -> decoder_pred -> decoder_norm
    if (!p->dpred_done) { set_error("hct_mae_backward_stage: the last forward ran without grad_scale (inference forward)"); return HCT_E_STATE; }
    p->wg_pending.clear(); p->wg_slot = 0; p->wg_blocks_pending = 0; p->final_off = p->param_elems;
    void* dhs = ws + dec_out(nd - 1);
    const int Mt = p->tail_fwd ? p->Mc : p->Md;                       // compact tail: the masked patches' rows only
    float* dht = p->tail_fwd ? (float*)(ws + p->s_dhc) : dh;          // ... whose fp32 gradient has a buffer of its own
    if (p->dloss) RC(scale_unless_one(ws + p->a_dpred, p->dt, (int64_t)Mt * p->pd, p->dloss, s));
    RC(linear_wgrad(p, ws + p->a_dpred, ws + p->a_ynorm, Mt, p->pd, p->Dd, p->p_pred_w, p->p_pred_b, s));
    RC(linear_dgrad(p, ws + p->a_dpred, Mt, p->pd, p->p_pred_w, p->Dd, dx, HCT_ACT_NONE, nullptr, s));
    RC(hct_layernorm_bwd(dx, p->dt, (const float*)(ws + p->h_dec[nd]), (const float*)(ws + p->a_yn_mean), (const float*)(ws + p->a_yn_rstd),
                         p->pf(p->p_dnorm_w), nullptr, Mt, p->Dd, dht, dhs, p->dt, p->gf(p->p_dnorm_w), p->gf(p->p_dnorm_b),
                         nd > 0 ? p->gf(p->dec[nd - 1].fc2_b) : nullptr, small, p->s_small_bytes, s));
    return end_stage(p, stage, false, nd == 0, Rd, nd, s);
  }
  if (stage <= nd) {
    const int i = nd - stage;
    if (p->dec0_fwd && i == 0)
      RC(block_backward_dec0(p, p->dec[i], p->adec[i], p->gdec[i % Rd], s));
    else if (p->tail_fwd && i == nd - 1)
      RC(block_backward_tail(p, p->dec[i], p->adec[i], p->gdec[i % Rd], dec_out(i - 1), (const float*)(ws + p->h_dec[i]), i > 0 ? p->dec[i - 1].fc2_b : -1, s));
    else
      RC(block_backward(p, p->dec[i], p->adec[i], p->gdec[i % Rd], dec_out(i - 1), (const float*)(ws + p->h_dec[i]), B, p->Nd, p->Dd, p->Mlpd, p->Hd,
                        i > 0 ? p->dec[i - 1].fc2_b : -1, s));
    return end_stage(p, stage, true, stage == nd, Rd, nd, s);  // (the decoder's products run together once its backward is through)
  }
  if (stage == nd + 1) {  // decoder input assembly -> decoder_embed -> encoder norm
    void* de = ws + p->a_de;
    void* dhs = ws + enc_out(ne - 1);
    if (p->dec0_fwd)  // the kept rows' gradient is in a_de already (block_backward_dec0); the two token gradients from its pieces
      RC(dec0_token_grads(dh, (const int32_t*)(ws + p->a_ids_shuffle), (const float*)(ws + p->a_dcat), B, p->L, p->K, p->Dd, p->gf(p->p_mask),
                          p->gf(p->p_dcls), small, p->s_small_bytes, s));
    else
      RC(hct_decoder_assemble_bwd(dh, (const int32_t*)(ws + p->a_ids_restore), (const int32_t*)(ws + p->a_ids_shuffle), B, p->L, p->K, p->Dd, de,
                                  p->dt, p->gf(p->p_mask), p->gf(p->p_dcls), small, p->s_small_bytes, s));
    RC(linear_wgrad(p, de, ws + p->a_latent, p->Me, p->Dd, p->D, p->p_de_w, p->p_de_b, s));
    RC(linear_dgrad(p, de, p->Me, p->Dd, p->p_de_w, p->D, dx, HCT_ACT_NONE, nullptr, s));
    RC(hct_layernorm_bwd(dx, p->dt, (const float*)(ws + p->h_enc[ne]), (const float*)(ws + p->a_lat_mean), (const float*)(ws + p->a_lat_rstd),
                         p->pf(p->p_norm_w), nullptr, p->Me, p->D, dh, dhs, p->dt, p->gf(p->p_norm_w), p->gf(p->p_norm_b),
                         ne > 0 ? p->gf(p->enc[ne - 1].fc2_b) : nullptr, small, p->s_small_bytes, s));
    return end_stage(p, stage, false, false, Re, ne, s);
  }
  if (stage <= nd + 1 + ne) {
    const int i = ne - (stage - nd - 1);
    RC(block_backward(p, p->enc[i], p->aenc[i], p->genc[i % Re], enc_out(i - 1), (const float*)(ws + p->h_enc[i]), B, p->Ne, p->D, p->Mlp, p->H,
                      i > 0 ? p->enc[i - 1].fc2_b : -1, s));
    return end_stage(p, stage, true, false, Re, ne, s);
  }
  if (stage == nd + ne + 2) {  // encoder input assembly -> patch embedding
    void* dtok = ws + p->a_dtok;
    RC(hct_encoder_assemble_bwd(dh, (const int32_t*)(ws + p->a_ids_restore), B, p->L, p->K, p->D, dtok, p->dt, p->gf(p->p_cls),
                                p->p_pos >= 0 ? p->gf(p->p_pos) : nullptr, small, p->s_small_bytes, s));
    RC(linear_wgrad(p, dtok, ws + p->a_patches, B * p->K, p->D, p->pd, p->p_pe_w, p->p_pe_b, s));
    return end_stage(p, stage, false, false, Re, ne, s);
  }
  set_error("hct_mae_backward_stage: stage %d out of range", stage);
  return HCT_E_BADARG;
}

const void* hct_mae_plan_activation(const hct_mae_plan* p, const char* name, int64_t* rows, int64_t* cols, int* dtype) {
  if (!p || !p->ws || !name) return nullptr;
  auto it = p->acts.find(name);
  if (it == p->acts.end()) return nullptr;
  if (rows) *rows = it->second.rows;
  if (cols) *cols = it->second.cols;
  if (dtype) *dtype = it->second.dtype;
  return p->ws + it->second.off;
}

}  // extern "C"
