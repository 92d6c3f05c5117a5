/*
 * headct_hip.h -- C ABI of libheadct_hip.so: the MI355X (gfx950) MAE pre-training hot path.
 *
 * The reference (nirvanesque/headCT_foundation) has no FFI seam: the hot path sits behind the
 * Python nn.Module contract of MaskedAutoencoderViT and the engine_pretrain_mae functions
 * (SURVEY.md 8b).  This header is the boundary a maintainer would bind instead of the PyTorch
 * operators that path dispatches today; each entry point cites the reference lines it replaces
 * (paths relative to the reference root).  INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; the caller owns all memory,
 *     the library never allocates or frees device memory;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); calls are asynchronous;
 *   - return value: 0 on success, a positive hipError_t, or a negative HCT_E_* code;
 *     hct_last_error_string() gives text for the calling thread's last failure;
 *   - dtype codes: HCT_F32 = 0 (fp32 storage), HCT_BF16 = 1 (bfloat16 storage, fp32 accumulation);
 *   - matrices are row-major; "ld" = row stride in elements.
 */
#ifndef HEADCT_HIP_H
#define HEADCT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HCT_F32 0
#define HCT_BF16 1
#define HCT_F16 2 /* IEEE half: input volumes of the persistent cache only (hct_augment_volume) */

#define HCT_E_BADARG (-1)
#define HCT_E_UNSUPPORTED (-2)
#define HCT_E_WORKSPACE (-3)
#define HCT_E_STATE (-4)

#define HCT_ACT_NONE 0
#define HCT_ACT_GELU 1  /* out = gelu_erf(acc + bias); aux (if given) receives the pre-activation */
#define HCT_ACT_DGELU 2 /* out = acc * gelu_erf'(aux)   (aux = saved pre-activation)             */
#define HCT_ACT_TANH 3  /* hct_head_linear only: out = tanh(acc + bias)                           */
#define HCT_ACT_GELU_D 4 /* hct_gemm: out = gelu_erf(acc + bias); aux receives gelu_erf'(acc + bias) -- what the backward     */
#define HCT_ACT_MULAUX 5 /* hct_gemm: out = acc * aux -- needs (the derivative is evaluated once, on the unrounded value)   */

const char* hct_last_error_string(void);
int hct_version(void);
/* 1 when the tuned gfx950 MFMA kernels are compiled in (always, in this build). */
int hct_has_mfma_kernels(void);

/* ------------------------------------------------------------------------------------------
 * GEMM with fused epilogue:  C = act(alpha * op(A) . op(B) + bias) (+ residual)
 * Replaces nn.Linear / Conv3d-as-GEMM forward and the autograd dgrad / wgrad products
 * (attentionblock.py:41-42,54,64; MONAI MLPBlock linear1/linear2; mae.py:118-119;
 *  patch_embedding.py:102-105,149).
 *   transA = 0: A stored [M,K];  transA = 1: A stored [K,M]  (op(A) = A^T)
 *   transB = 1: B stored [N,K] (the nn.Linear weight layout, "NT"); transB = 0: B stored [K,N]
 * Dispatch: bf16 x bf16, transA=0, transB=1, K%64==0, N%16==0 -> tuned MFMA "NT" kernel;
 *           bf16 x bf16, transA=1, transB=0, M%16==0, N%16==0   -> tuned MFMA "TN" kernel (split-K
 *           over the reduction, partial slabs in `workspace`, deterministic reduction);
 *           anything else (all fp32 work) -> generic strided kernel (fp32 FMA accumulation).
 * ------------------------------------------------------------------------------------------ */
typedef struct hct_gemm_args {
  int M, N, K;
  const void* A; int a_dtype; int64_t lda; int transA;
  const void* B; int b_dtype; int64_t ldb; int transB;
  void* C; int c_dtype; int64_t ldc;
  const float* bias;     /* [N] fp32 or NULL */
  const float* residual; /* [M,N] fp32 (row stride ldr) or NULL; added after the activation */
  int64_t ldr;
  int act;               /* HCT_ACT_* */
  void* aux; int aux_dtype; int64_t ldaux;
  void* C2; int c2_dtype; int64_t ldc2; /* optional second copy of the output (e.g. bf16 shadow) */
  float alpha;
  int force_generic;     /* testing: always take the generic kernel */
  float* colsum_out;     /* optional [N] fp32: column sums of the OUTPUT C (the bias gradient of the Linear that produced the
                            operand of this dgrad); fused into the epilogue where the kernel supports it */
  int workspace_armed;   /* wgrad split fold: 1 = the first 1 KiB of `workspace` was zero when first used and has only been
                            touched by hct_gemm since (its counters re-arm themselves): skips the per-call reset.  0 = reset it.
                            NT path: the same promise for the head of the stream-K region (hct_gemm_nt_flags_offset). */
} hct_gemm_args;

size_t hct_gemm_workspace_bytes(const hct_gemm_args* a);
/* NT path (forward Linears / dgrads), persistent 256x256 kernel: when the tile count leaves a partly filled last round, the
 * remainder tiles are shared out by K range over all CUs ("stream-K"; whole tiles for the rest), partial accumulators passing
 * through the LAST 64 MiB + 4 KiB of `workspace` in a fixed summation order (bit-reproducible).  Optional: with a smaller (or
 * no) workspace the launch uses whole tiles only.  hct_gemm_workspace_bytes includes the region only for shapes whose
 * remainder round would be shared out on the present CU count.  An owner that waits in vain for a partial (a grid that is
 * not wholly resident) sets the error word AND fills its tile with NaN.  hct_gemm_nt_flags_offset = byte offset of that region's head inside a
 * workspace of the given size ((size_t)-1: too small): 256 32-bit arrival flags, and at byte 2048 an error word that a launch
 * sets to 0xDEAD if a partial never arrived (cannot happen while the grid is resident; it is flagged rather than waited for).
 * The head is reset before every such launch unless `workspace_armed` says that it started zeroed and only hct_gemm has
 * written it since.  Environment: HCT_NT_STREAMK_PAIRS = least number of K-stage pairs per CU the sharing must save for it
 * to be used (default 20, tuned on the MAE step; a huge value switches it off). */
size_t hct_gemm_nt_flags_offset(size_t workspace_bytes);
size_t hct_gemm_nt_stream_k_bytes(void); /* size of that region: what a caller that keeps ONE workspace for many shapes appends to it */
/* Leave `n` CUs out of the persistent GEMM grids (default 0) so that communication kernels (RCCL all-reduce overlapped
 * with the backward) have somewhere to run; set by the data-parallel wrapper when world_size > 1. */
void hct_set_cu_reserve(int n);
int hct_gemm(const hct_gemm_args* a, void* workspace, size_t workspace_bytes, void* stream);
/* Grouped weight gradients: n "TN" products dW_i[M_i,N_i] = alpha_i * A_i[K_i,M_i]^T . B_i[K_i,N_i] (bf16 operands, fp32 C, no
 * epilogue extras) in ONE persistent launch, each 256x256 output tile reducing over ALL K_i rows (no split partials, no fold
 * launch); the partly filled last round of tiles is shared out by reduction range with a fixed summation order
 * (bit-reproducible).  The weight gradients of a training step do not feed its backward chain, so the model driver collects
 * those of several blocks and runs them here (the reference computes each inside autograd's backward of its Linear,
 * attentionblock.py:41-42, MLPBlock).  `prepare` writes the job table into the workspace and zeroes its flags (once per set of
 * pointers / shapes, stream-ordered); `run` launches on the prepared table (same jobs array).  workspace: 256-byte aligned,
 * hct_gemm_tn_group_workspace_bytes(n), used by nothing else between prepare and the last run.                              */
size_t hct_gemm_tn_group_workspace_bytes(int n_jobs);
int hct_gemm_tn_group_prepare(const hct_gemm_args* jobs, int n_jobs, void* workspace, size_t workspace_bytes, void* stream);
int hct_gemm_tn_group_run(const hct_gemm_args* jobs, int n_jobs, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Random masking from supplied noise (mae.py:204-216).  Stable ranking: ties -> lower index first.
 *   noise [B,L] fp32  ->  ids_restore [B,L] i32, ids_shuffle [B,L] i32 (first K = ids_keep),
 *   mask [B,L] fp32 (0 keep, 1 masked).
 * ------------------------------------------------------------------------------------------ */
int hct_mask_rank(const float* noise, int B, int L, int K, int32_t* ids_restore, int32_t* ids_shuffle,
                  float* mask, void* stream);

/* ------------------------------------------------------------------------------------------
 * Patch gather (im2col of the stride==kernel Conv3d, kept tokens only; patch_embedding.py:149-152
 * + mae.py:212).   x [B,C,S,S,S] fp32 -> rows [B*K, C*P^3] in Conv3d weight order (c,ph,pw,pd).
 * ------------------------------------------------------------------------------------------ */
int hct_patch_gather(const void* x, int x_dtype /* HCT_F32 | HCT_F16 */, const int32_t* ids_shuffle, int B, int C, int S, int P, int L, int K,
                     void* rows, int rows_dtype, void* stream);

/* Encoder input assembly: h0[b,0,:] = cls;  h0[b,1+j,:] = tok[b*K+j,:] + pos[ids_keep[b,j],:]
 * (patch_embedding.py:155-156, mae.py:212,233-234).  pos may be NULL (pos_embed == "none").   */
int hct_encoder_assemble_fwd(const void* tok, int tok_dtype, const float* cls, const float* pos,
                             const int32_t* ids_shuffle, int B, int L, int K, int D, float* h0, void* stream);
/* backward: dtok[b*K+j] = dh0[b,1+j]; dcls = sum_b dh0[b,0]; dpos[l] = sum_{b: l kept} dh0[b,1+rank]. */
int hct_encoder_assemble_bwd(const float* dh0, const int32_t* ids_restore, int B, int L, int K, int D,
                             void* dtok, int dtok_dtype, float* dcls, float* dpos, void* workspace,
                             size_t workspace_bytes, void* stream);
/* workspace for the *_assemble_bwd reductions */
size_t hct_assemble_bwd_workspace_bytes(int D);

/* ------------------------------------------------------------------------------------------
 * LayerNorm over the last dim, eps = 1e-5 (nn.LayerNorm default; attentionblock.py:92-93,
 * mae.py:116-117).  x fp32 [rows,D] (the residual stream); y in y_dtype; mean/rstd saved fp32.
 * ------------------------------------------------------------------------------------------ */
int hct_layernorm_fwd(const float* x, const float* gamma, const float* beta, int rows, int D, float eps,
                      void* y, int y_dtype, float* mean, float* rstd, void* stream);
/* dx_total = dres + LN'(dy)  written fp32 to dx (may alias dres) and, if dx_shadow != NULL, also
 * in shadow_dtype.  dgamma/dbeta [D] fp32 (overwritten); if dcolsum != NULL it receives
 * sum_rows(dx_total) [D] (the bias gradient of the Linear that produced this residual branch).
 * workspace: hct_layernorm_bwd_workspace_bytes(rows, D).                                       */
size_t hct_layernorm_bwd_workspace_bytes(int rows, int D);
int hct_layernorm_bwd(const void* dy, int dy_dtype, const float* x, const float* mean, const float* rstd,
                      const float* gamma, const float* dres, int rows, int D, float* dx, void* dx_shadow,
                      int shadow_dtype, float* dgamma, float* dbeta, float* dcolsum, void* workspace,
                      size_t workspace_bytes, void* stream);
/* The same with the residual gradient taken from a COMPACT matrix: row r adds dres[dres_rows[r]], nothing where
 * dres_rows[r] < 0 (dres_rows == NULL: as hct_layernorm_bwd).  dres must not alias dx then.  Used by the plan where the
 * tail of the MAE decoder runs on the masked patches' rows only (mae.py:298-299: the loss takes no other row).      */
int hct_layernorm_bwd_mapped(const void* dy, int dy_dtype, const float* x, const float* mean, const float* rstd,
                             const float* gamma, const float* dres, const int32_t* dres_rows, int rows, int D, float* dx,
                             void* dx_shadow, int shadow_dtype, float* dgamma, float* dbeta, float* dcolsum, void* workspace,
                             size_t workspace_bytes, void* stream);
/* Rows of the [B, L+1] decoder layout that hold masked patches (mae.py:207-214: ids_restore[b, l] >= K), per volume in
 * shuffle order: tail_rows [B*(L-K)] and its inverse tail_inv [B*(L+1)] (-1 for the class token and the kept patches). */
int hct_tail_rows(const int32_t* ids_restore, int B, int L, int K, int32_t* tail_rows, int32_t* tail_inv, void* stream);
/* dst row r = src row idx[r], zeros where idx[r] < 0; rows of row_bytes (multiple of 16) bytes, 16-byte aligned bases.  */
int hct_gather_rows(const void* src, const int32_t* idx, int n_rows, int row_bytes, void* dst, void* stream);

/* ------------------------------------------------------------------------------------------
 * Multi-head self-attention, no mask, scale dh^-1/2 (attentionblock.py:54-62,
 * F.scaled_dot_product_attention).  qkv [B,N,3,H,dh] (the fused-QKV Linear output, as the
 * reference views it :54); o [B,N,H*dh] (head-merged, as :62); lse [B,H,N] fp32 saved for backward.
 * dtype = storage of qkv / o / do / dqkv.
 * ------------------------------------------------------------------------------------------ */
int hct_attention_fwd(const void* qkv, int B, int N, int H, int dh, int dtype, void* o, float* lse, void* stream);
int hct_attention_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, int B, int N, int H,
                      int dh, int dtype, void* dqkv, void* stream);

/* ------------------------------------------------------------------------------------------
 * Decoder input assembly (mae.py:257-265):
 *   y[b,0]   = e[b,0] + dec_cls
 *   y[b,1+l] = (ids_restore[b,l] < K ? e[b,1+ids_restore[b,l]] : mask_token) + dec_pos[l]
 * e [B,K+1,D] in e_dtype; y fp32 [B,L+1,D].
 * backward: de (e_dtype) gather of dy; dmask_token = sum over masked rows; ddec_cls = sum_b dy[b,0].
 * ------------------------------------------------------------------------------------------ */
int hct_decoder_assemble_fwd(const void* e, int e_dtype, const float* mask_token, const float* dec_cls,
                             const float* dec_pos, const int32_t* ids_restore, int B, int L, int K, int D,
                             float* y, void* stream);
int hct_decoder_assemble_bwd(const float* dy, const int32_t* ids_restore, const int32_t* ids_shuffle, int B,
                             int L, int K, int D, void* de, int de_dtype, float* dmask_token, float* ddec_cls,
                             void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Masked-voxel MSE (mae.py:277-301) fused with patchify (mae.py:160-170) and its backward seed.
 *   pred [B, L+1, pd] (row 0 of each volume = cls row, ignored; mae.py:273), pd = P^3*C (C fastest)
 *   loss (1 fp32, overwritten) = sum_l mask*mean_k (pred-tgt)^2 / sum(mask)
 *   dpred (may be NULL) same shape/dtype as pred: s*2*mask*(pred-tgt)/(pd*sum(mask)); 0 on cls/kept rows;
 *   s = *dpred_scale (device fp32, the incoming dLoss, e.g. a GradScaler factor) or 1 if NULL.
 *   loss may be NULL when only dpred is wanted.
 *   norm_pix: per-patch (t-mean)/sqrt(var_unbiased+1e-6) target (mae.py:290-293).
 *   mask_sum = sum(mask) (= B*(L-K) for masks from hct_mask_rank); row_loss: workspace of B*L floats.
 * ------------------------------------------------------------------------------------------ */
int hct_masked_mse(const void* pred, int pred_dtype, const void* x, int x_dtype /* HCT_F32 | HCT_F16 */, const float* mask, int B, int C, int S,
                   int P, int norm_pix, float mask_sum, float* row_loss, float* loss, void* dpred,
                   const float* dpred_scale, void* stream);
/* reconstructed voxels: unpatchify (mae.py:172-192). pred rows [B, L(+1 if has_cls_row), pd] -> [B,C,S,S,S] fp32 */
int hct_unpatchify(const void* pred, int pred_dtype, int has_cls_row, int B, int C, int S, int P, float* vol,
                   void* stream);

/* Feature-extraction input of the plain ViT encoder (src/models/vit.py:144-162): every patch is embedded, the class token
 * is prepended and the optional register tokens are inserted behind it:
 *   h[b,0,:] = cls;  h[b,1+r,:] = reg[r,:] (r < R);  h[b,1+R+l,:] = tok[b*L+l,:] + pos[l,:]        (pos / reg may be NULL) */
int hct_vit_assemble_fwd(const void* tok, int tok_dtype, const float* cls, const float* reg, const float* pos, int B,
                         int L, int R, int D, float* h, void* stream);

/* Classification heads over the ViT features, forward / eval-mode arithmetic (src/models/classifier.py:7-99 and the
 * `classification_head` of src/models/vit.py:133-137, :170-171).  All fp32 unless a dtype is given.
 *
 * hct_channel_norm: nn.BatchNorm1d(C, affine=False) in eval mode on [rows, C] (classifier.py:89 `bn1`):
 *   out[r, c] = (x[r, c] - mean[c]) / sqrt(var[c] + eps)                                   C % 4 == 0
 * hct_query_attention: the learnt queries of AttentionClassifier against every token (classifier.py:84-93):
 *   out[b, h, q, :] = softmax_n(logit_scale * <q[q, h*dh:(h+1)*dh], K[b, n, h, :]>) @ V[b, :, h, :]
 *   q: [Q, H*dh] fp32 (the raw `cls_token`; the reference scales it by `scale` and SDPA by dh^-1/2 again, so the caller
 *   passes logit_scale = scale * dh^-1/2); kv: [B, N, 2, H, dh] (the `wkv` output as it lies, classifier.py:90);
 *   out: [B, H, Q, dh] fp32 - the layout classifier.py:95 reshapes WITHOUT a permute.  Q*N + 4*Q*dh + Q <= 16384.
 * hct_head_linear: an optional eval-mode BatchNorm1d, the mean over nq consecutive D-vectors, a Linear and an optional Tanh:
 *   out[r, c] = act( sum_k (1/nq * sum_q (x[r*ldx + q*D + k] - mean[k]) / sqrt(var[k] + eps)) * W[c, k] + bias[c] )
 *   mean = var = NULL skips the normalisation; act = HCT_ACT_NONE | HCT_ACT_TANH.  (LinearClassifier: nq = 1, ldx = D;
 *   AttentionClassifier's bn2 + mean + linear: nq = Q, ldx = Q*D; ViT classification_head on x[:, 0]: ldx = T*D.) */
int hct_channel_norm(const float* x, const float* mean, const float* var, float eps, void* out, int out_dtype, int64_t rows,
                     int C, void* stream);
int hct_query_attention(const float* q, int Q, const void* kv, int kv_dtype, int B, int N, int H, int dh, float logit_scale,
                        float* out, void* stream);
int hct_head_linear(const float* x, int64_t ldx, int nq, const float* mean, const float* var, float eps, const float* W,
                    const float* bias, int act, float* out, int rows, int D, int n_out, void* stream);

/* Linear probing: LinearClassifier in TRAINING mode on frozen (detached) features with nn.CrossEntropyLoss
 * (main_downstream.py:142-146, :214; engine_downstream.py:70-117).  All fp32, fixed summation orders.
 *
 * hct_batchnorm_stats: nn.BatchNorm1d training statistics of x [B, D] (B > 1): mean[k], var[k] = biased variance (what the
 *   forward normalises with); if running_* are given: running = (1-momentum)*running + momentum*{mean, unbiased variance}.
 *   The forward itself is hct_head_linear with these mean / var.
 * hct_softmax_xent: loss = mean_b(logsumexp(logits[b]) - logits[b, target[b]]) (loss may be NULL);
 *   dlogits[b, c] = (softmax(logits[b])[c] - [c == target[b]]) * (dloss ? *dloss : 1) / B (dlogits may be NULL).
 * hct_head_linear_wgrad: dW[c, k] = sum_b dlogits[b, c] * (x[b, k] - mean[k]) / sqrt(var[k] + eps), db[c] = sum_b dlogits[b, c]
 *   (mean = var = NULL: no normalisation; db may be NULL).  The statistics are treated as constants: exact for the
 *   parameter gradients; the gradient with respect to x (fine-tuning an unfrozen backbone) is not built. */
int hct_batchnorm_stats(const float* x, int B, int D, float momentum, float* mean, float* var, float* running_mean,
                        float* running_var, void* stream);
int hct_softmax_xent(const float* logits, const int64_t* target, int B, int n_classes, const float* dloss, float* loss,
                     float* dlogits, void* stream);
int hct_head_linear_wgrad(const float* x, const float* mean, const float* var, float eps, const float* dlogits, int B, int D,
                          int n_out, float* dW, float* db, void* stream);

/* Device side of the reference's per-sample MAE input transforms, mae3d_transforms(mode='train'), src/data/transforms.py:
 * 193-228: CastToTyped(float32) of the cached volume (fp16 on disk, transforms.py:170-175) -> RandFlipd on spatial axes
 * 0, 1, 2 -> RandShiftIntensityd.  The random draws stay on the host (one byte of flip flags and one offset per sample);
 *   out[b, c, i, j, k] = (float)in[b, c, f0(i), f1(j), f2(k)] + shift[b],   f_a(t) = S-1-t if flip[b] bit a else t.
 * in: [B, C, S, S, S] of in_dtype (HCT_F16 / HCT_BF16 / HCT_F32), out fp32 (may not alias in).  flip / shift may be NULL.
 * RandGaussianSmoothd (transforms.py:230-238) is not part of this call. */
/* ------------------------------------------------------------------------------------------
 * DINO self-distillation (BASELINE config #5; reference engine_pretrain_dino.py:14-130).
 *   hct_dino_loss: DINOLoss.forward, src/losses/losses.py:63-91, plus the gradient w.r.t. the student logits and the column
 *     sums of the teacher logits that update_center (:93-102) all-reduces.  student [V*B, K] (crop-major: row v*B + b),
 *     teacher [2*B, K], both `dtype`; center [K] fp32; loss: 1 device fp32.  dstudent (same shape / dtype as student) and
 *     batch_center_sum [K] may be NULL; dloss: device scalar multiplying the gradient, or NULL for 1.
 *   hct_dino_center_update: center = center * momentum + (batch_center_sum / count) * (1 - momentum), count = 2*B*world.
 *   hct_ema_update: momentum encoder, src/utils/misc.py:386-397: k = k * m + (1 - m) * q over a flat fp32 buffer.
 * ------------------------------------------------------------------------------------------ */
size_t hct_dino_loss_workspace_bytes(int n_crops, int B, int K);
int hct_dino_loss(const void* student, const void* teacher, int dtype, int n_crops, int B, int K, const float* center, float student_temp,
                  float teacher_temp, float* loss, void* dstudent, const float* dloss, float* batch_center_sum, void* workspace,
                  size_t workspace_bytes, void* stream);
int hct_dino_center_update(float* center, const float* batch_center_sum, int K, double momentum, double count, void* stream);
int hct_ema_update(float* momentum_params, const float* params, int64_t n, double m, void* stream);
/* DINOHead tail (src/models/dino_head.py:37-41): rows L2-normalised (F.normalize, eps 1e-12), prototype weights weight-normalised
 * (torch.nn.utils.weight_norm, dim 0: W[k,:] = g[k] v[k,:] / ||v[k,:]||); forward keeps 1 / norm per row for the backward. */
int hct_l2norm_rows_fwd(const float* z, int M, int n, void* zn, int zn_dtype, float* inv_norm, void* stream);
/* Projection head with use_bn (dino_head.py:15-21, the default of config.py:86): Linear -> BatchNorm1d -> GELU.  u [M, D] fp32 is the
 * Linear's output, mean / var [D] the statistics to normalise with -- hct_batchnorm_stats of u in training (the caller all-reduces
 * them under data parallelism, as the SyncBatchNorm of main_pretrain_dino.py:183-185 does), the running ones in eval:
 *   fwd   : xhat = (u - mean) / sqrt(var + eps); h = gelu(gamma * xhat + beta) in h_dtype; xhat and dact = gelu'(.) fp32 kept for the
 *           backward (either may be NULL)
 *   sums  : sums[0..D) = sum_rows dh * dact (= dbeta), sums[D..2D) = sum_rows dh * dact * xhat (= dgamma) of this rank's rows
 *   apply : du = gamma / sqrt(var + eps) * (dh * dact - sums[0] / count - xhat * sums[1] / count), count = rows that shared the
 *           statistics (all ranks; `sums` all-reduced then), du in dh's dtype or fp32                                          */
int hct_bn_gelu_fwd(const float* u, const float* mean, const float* var, const float* gamma, const float* beta, float eps, int M, int D, void* h,
                    int h_dtype, float* xhat, float* dact, void* stream);
int hct_bn_gelu_bwd_sums(const void* dh, int dh_dtype, const float* dact, const float* xhat, int M, int D, float* sums, void* stream);
int hct_bn_gelu_bwd_apply(const void* dh, int dh_dtype, const float* dact, const float* xhat, const float* gamma, const float* var, float eps,
                          const float* sums, double count, int M, int D, void* du, int du_dtype, void* stream);
int hct_l2norm_rows_bwd(const float* dzn, const void* zn, int zn_dtype, const float* inv_norm, int M, int n, float* dz, void* stream);
int hct_weight_norm_fwd(const float* v, const float* g, int K, int n, void* w, int w_dtype, float* inv_norm, void* stream);
int hct_weight_norm_bwd(const float* dw, const float* v, const float* g, const float* inv_norm, int K, int n, float* dv, float* dg /* or NULL */,
                        void* stream);

/* HU windowing of loading_transforms (src/data/transforms.py:108-133): ScaleIntensityRanged(a_min, a_max, 0, 1, clip) for one
 * channel (window 40 +- 150), MultipleWindowScaleStack (transforms.py:8-36) for three ((40,80), (80,200), (600,2800) as
 * centre, width -> a_min = l - w/2, a_max = l + w/2), stacked on the channel axis:
 *   out[b, w, v] = clip((hu[b, v] - a_min[w]) / (a_max[w] - a_min[w]), 0, 1),   hu [B, voxels] -> out [B, n_windows, voxels].
 * in / out dtype HCT_F32 or HCT_F16 (fp16 = the persistent cache's type, transforms.py:171-178); a_min / a_max device fp32. */
int hct_hu_window(const void* hu, int in_dtype, void* out, int out_dtype, int B, int64_t voxels, int n_windows, const float* a_min,
                  const float* a_max, void* stream);
int hct_augment_volume(const void* in, int in_dtype, float* out, int B, int C, int S, const unsigned char* flip,
                       const float* shift, void* stream);
/* RandGaussianSmoothd of mae3d_transforms(reshape=False) (src/data/transforms.py:230-238; MONAI GaussianSmooth ->
 * GaussianFilter -> separable_filtering with zero padding): in [B,C,S,S,S] fp32 -> out, one 1-D pass per spatial axis.
 *   taps  [B][3][9] device fp32: sample b's centred kernel of spatial axis a (0 = slowest), zero beyond its tail; the host
 *         computes them as MONAI's gaussian_1d(sigma, truncated=4, approx="erf") does (sigma <= 1.06 keeps 9 taps);
 *   apply [B] device bytes: 0 = the transform did not fire for this sample (out = in).
 * tmp: scratch of the same size as in/out; the three buffers must differ. */
int hct_gaussian_smooth3d(const float* in, float* out, float* tmp, int B, int C, int S, const float* taps, const unsigned char* apply,
                          void* stream);

/* Resume at another resolution: trilinear resize (align_corners = false) of the learnable position table
 * src [extra + g_src^3, D] -> dst [extra + g_dst^3, D], the `extra` leading (class) rows copied unchanged.
 * Replaces interpolate_pos_embed's 3-D branch, src/utils/pos_embed.py:102-153 (called at main_pretrain_mae.py:132). */
int hct_pos_embed_interp3d(const float* src, int g_src, float* dst, int g_dst, int D, int extra, void* stream);

/* Column sum of a [rows, cols] matrix -> fp32 [cols] (bias gradients). workspace >= hct_colsum_workspace_bytes. */
size_t hct_colsum_workspace_bytes(int rows, int cols);
int hct_colsum(const void* x, int dtype, int rows, int cols, int64_t ld, float* out, void* workspace,
               size_t workspace_bytes, void* stream);

/* dtype conversion / transposed conversion (bf16 working copies of the fp32 master weights). */
int hct_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream);
int hct_transpose_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int rows, int cols, void* stream);

/* ------------------------------------------------------------------------------------------
 * Per-parameter gradient clip (src/utils/misc.py:374-383) + AdamW (src/utils/optimizers.py:354-360
 * -> torch.optim.AdamW defaults) over a FLAT fp32 parameter/gradient/state buffer described by a
 * device segment table seg_off[nseg+1] (element offsets; segment i = [seg_off[i], seg_off[i+1])).
 *   hct_grad_norms : norms[i] = ||g_i||_2 ; coef[i] = clip/(norm+1e-6) if < 1 else 1 (clip <= 0: all 1).
 *                    scale_in_place != 0 also multiplies the gradients (the reference's in-place form).
 *   hct_adamw_step : g <- g*coef (written back), decoupled weight decay on every element, bias-corrected
 *                    update; `step` is 1-based; optionally refreshes a bf16 shadow of the parameters.
 *                    skip[i] != 0 freezes segment i (requires_grad = False).
 * ------------------------------------------------------------------------------------------ */
size_t hct_grad_norms_workspace_bytes(int64_t total);
int hct_grad_norms(float* grads, const int64_t* seg_off, int nseg, int64_t total, float clip, int scale_in_place,
                   float* norms, float* coef, void* workspace, size_t workspace_bytes, void* stream);
int hct_adamw_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, const int64_t* seg_off,
                   const float* coef, const uint8_t* skip, int nseg, int64_t total, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int step, void* params_bf16, void* stream);

/* ------------------------------------------------------------------------------------------
 * Whole-model driver (MaskedAutoencoderViT.forward mae.py:303-317 and its autograd backward,
 * engine_pretrain_mae.py:58-62).  The plan is a HOST object describing the parameter layout
 * (names / shapes / offsets into one flat buffer, in the reference's registration order) and the
 * activation workspace layout for a fixed batch size; it owns no device memory.
 * ------------------------------------------------------------------------------------------ */
typedef struct hct_mae_config {
  int input_size, patch_size, in_chans;
  double mask_ratio; /* double: len_keep = (int)(L * (1 - mask_ratio)) must round as the reference's Python float does (mae.py:205) */
  int pos_embed; /* 0 none, 1 learnable, 2 sincos (same storage; init differs on the host) */
  int encoder_depth, encoder_embed_dim, encoder_mlp_dim, encoder_num_heads;
  int decoder_depth, decoder_embed_dim, decoder_mlp_dim, decoder_num_heads;
  int norm_pix_loss, use_bias;
  /* encoder_only = 1 turns the plan into the plain ViT backbone of DINO pre-training (src/models/vit.py:144-173): every patch is
   * embedded (no masking, mask_ratio ignored), class token, num_register_tokens register tokens behind it, encoder blocks, final
   * LayerNorm with final_norm_eps (vit.py:124: 1e-6; 0 = the MAE default 1e-5); no decoder.  Driven by hct_vit_forward /
   * hct_vit_backward_stage instead of hct_mae_forward / hct_mae_backward_stage. */
  int encoder_only, num_register_tokens;
  float final_norm_eps;
} hct_mae_config;

typedef struct hct_mae_plan hct_mae_plan;

typedef struct hct_param_info {
  char name[96];
  int ndim;
  int64_t shape[5];
  int64_t offset; /* element offset into the flat fp32 parameter / gradient buffers */
  int64_t numel;
  int requires_grad;
  int is_matrix;      /* 1: a GEMM weight that gets bf16 (+ transposed bf16) working copies */
  int64_t bf16_t_offset; /* element offset of the transposed bf16 copy in the bf16-T buffer, or -1 */
} hct_param_info;

/* compute_dtype: HCT_F32 (parity mode: every kernel fp32) or HCT_BF16 (bf16 storage + MFMA). */
hct_mae_plan* hct_mae_plan_create(const hct_mae_config* cfg, int batch, int compute_dtype);
void hct_mae_plan_destroy(hct_mae_plan*);
int hct_mae_plan_num_params(const hct_mae_plan*);
int hct_mae_plan_param_info(const hct_mae_plan*, int index, hct_param_info* out);
int64_t hct_mae_plan_param_elems(const hct_mae_plan*);     /* flat fp32 params / grads length (padded) */
int64_t hct_mae_plan_bf16_t_elems(const hct_mae_plan*);    /* transposed-bf16 weight buffer length      */
size_t hct_mae_plan_workspace_bytes(const hct_mae_plan*);  /* activations + scratch                      */
int hct_mae_plan_len_keep(const hct_mae_plan*);            /* visible patches per volume = int(L * (1 - mask_ratio)) */
/* Compact tail (MAE plans): the loss takes only the masked patches' rows (mae.py:298-299), so with compact != 0 the next
 * forwards run everything behind the last decoder block's attention (its proj / MLP, decoder_norm, decoder_pred, the loss)
 * and the matching backward on those B*(L-K) rows alone.  Loss and every parameter gradient are those of the full
 * computation; the activations "dec<last>.out", "pred_full", "dpred_full" then hold the compact rows (row j of volume b =
 * decoder row tail_rows[b*(L-K)+j], activation "tail_rows") and the kept patches have NO prediction: leave it off (the
 * default) for a forward whose reconstruction of every patch is wanted.  Returns the mode in effect (0 where the geometry
 * does not allow it), < 0 on a null plan.                                                                                */
int hct_mae_plan_set_tail(hct_mae_plan*, int compact);
/* First decoder block on "cat" rows (default on for MAE plans with two or more decoder blocks; HCT_DEC0_TABLE=0 at plan creation or
 * on = 0 here turns it off): the masked tokens enter the decoder as mask_token + pos[l] in every volume (mae.py:259-265), so
 * LayerNorm1 and the qkv Linear of that block run on B*(K+1) kept / class rows + L table rows instead of B*(L+1), forward and
 * backward; same loss, same gradients (the sums over the masked rows are taken per patch position first).  Returns the mode in
 * effect. */
int hct_mae_plan_set_dec0(hct_mae_plan*, int on);
/* bind caller-owned device buffers. params_bf16 / params_bf16_t may be NULL in HCT_F32 mode. */
int hct_mae_plan_bind(hct_mae_plan*, float* params, float* grads, void* params_bf16, void* params_bf16_t,
                      void* workspace, size_t workspace_bytes);
/* refresh bf16 + transposed-bf16 working copies from the fp32 master weights (after load / optimizer step).
 * with_plain = 0 skips the plain bf16 copy (hct_adamw_step already wrote it). */
int hct_mae_refresh_weights(hct_mae_plan*, int with_plain, void* stream);
/* forward: x [B,C,S,S,S] of x_dtype (HCT_F32, or HCT_F16 = the persistent cache's storage type, transforms.py:171-178),
 * noise [B,L] fp32 -> *loss (device fp32).  Saves activations in the workspace.
 * grad_scale != 0: training forward -- the loss pass also writes the backward's seed d(loss)/d(pred) * grad_scale (the
 * volume and the prediction are read once per step); grad_scale = 1 / world_size under data parallelism, else 1.
 * grad_scale == 0: inference forward (no backward may follow). */
int hct_mae_forward(hct_mae_plan*, const void* x, int x_dtype, const float* noise, float* loss, float grad_scale, void* stream);
/* device pointer to the scalar dLoss that multiplies the backward's seed (NULL = 1.0; a value of 1.0 costs nothing). */
int hct_mae_set_loss_grad(hct_mae_plan*, const float* dloss);
/* backward in stages so the host can launch the per-bucket gradient all-reduce between them:
 * stage 0 .. hct_mae_num_backward_stages()-1, in order; stage s computes the gradients of the parameter range reported by
 * hct_mae_backward_stage_range (element offsets into the flat buffer; the ranges tile the buffer from its end to its start).
 * In bf16 plans the WEIGHT gradients (dW = dY^T . X, which feed nothing in the backward) of several stages are queued and run
 * together in one grouped launch (hct_gemm_tn_group_*): a range is therefore FINAL only once
 * hct_mae_backward_final_offset() -- every element at or behind it is final -- has moved down to its begin; after the last
 * stage it is 0.  hct_mae_plan_set_wgrad_defer(plan, defer, group_blocks): defer = 0 runs every weight gradient inside its stage
 * (split-K launches, ranges final stage by stage); group_blocks > 0 flushes the queue at least every that many block stages
 * (default 0: once after the decoder's and once after the encoder's backward); returns the mode in effect.  Environment at plan
 * creation: HCT_WGRAD_DEFER=0, HCT_WGRAD_GROUP_BLOCKS=n. */
int hct_mae_num_backward_stages(const hct_mae_plan*);
int hct_mae_backward_stage_range(const hct_mae_plan*, int stage, int64_t* begin, int64_t* end);
int hct_mae_backward_stage(hct_mae_plan*, int stage, void* stream);
int64_t hct_mae_backward_final_offset(const hct_mae_plan*);
int hct_mae_plan_set_wgrad_defer(hct_mae_plan*, int defer, int group_blocks);
/* Plain ViT backbone (plans created with encoder_only = 1).  forward: x [B,C,S,S,S] -> "latent" [B*(1+R+L), D] in the compute
 * dtype = norm(blocks(...)) of every token (hct_mae_plan_activation(plan, "latent")); row b*(1+R+L) is volume b's class token.
 * backward: stages 0 .. hct_mae_num_backward_stages()-1 like the MAE plan (final norm, blocks in reverse, input assembly + patch
 * embedding); stage 0 takes dlatent [B*(1+R+L), D] in the compute dtype (the gradient w.r.t. "latent"; rows that do not
 * feed the loss are zero). */
int hct_vit_forward(hct_mae_plan*, const void* x, int x_dtype, void* stream);
/* The same forward with the batch given as `n_parts` tensors of batch / n_parts volumes each, in batch order: what
 * MultiCropWrapper's torch.cat of equally sized crops (misc.py:467-480) would have produced, without the copy. */
int hct_vit_forward_parts(hct_mae_plan*, const void* const* xs, int n_parts, int x_dtype, void* stream);
int hct_vit_backward_stage(hct_mae_plan*, int stage, const void* dlatent, void* stream);
int hct_vit_assemble_bwd(const float* dh0, int B, int L, int R, int D, void* dtok, int dtok_dtype, float* dcls, float* dreg, float* dpos,
                         void* stream);
/* named activation lookup for parity tests: returns device pointer + shape/dtype, or NULL. */
const void* hct_mae_plan_activation(const hct_mae_plan*, const char* name, int64_t* rows, int64_t* cols, int* dtype);

/* ------------------------------------------------------------------------------------------
 * Measurement hooks (bench.py roofline leg): when enabled, every launch of a kernel class is bracketed
 * by HIP events on its own stream.  id: 0 GEMM-NT (MFMA), 1 GEMM-TN (MFMA), 2 GEMM-generic,
 * 3 attention fwd, 4 attention bwd.  hct_prof_read blocks until the recorded launches finished and returns
 * their summed duration, launch count and summed algorithmic work (FLOPs).
 * ------------------------------------------------------------------------------------------ */
void hct_prof_enable(int mask); /* bit i enables kernel class i; 0 = off */
void hct_prof_reset(void);
int hct_prof_read(int id, double* total_ms, int64_t* launches, double* work);
int hct_prof_read_bytes(int id, double* bytes); /* algorithmic bytes (operands read once + outputs written once) of those launches */
/* per-shape view of the same records (GEMM classes): one entry per distinct (M, N, K, epilogue mode, tiles, stream-K tiles) */
typedef struct hct_prof_shape {
  int M, N, K, mode, tiles, sk_tiles;
  int64_t launches;
  double total_ms, work, bytes;
} hct_prof_shape;
int hct_prof_shapes(int id, hct_prof_shape* out, int cap); /* returns the number of distinct keys, -1 on a HIP error */
/* testing hook, attention kernel choice (tests and scripts/ab_step.py only):
 *   0 / 1        default / every call through the fp32-math kernels;   2 / 3  online-softmax / full-row MFMA forward
 *   10 + bits    backward experiment bits (4 single-phase, 8 four-wave two-phase, 32 two-phase everywhere, 64 one wave per
 *                SIMD, 0x80 phase stamps, 0x100 .. 0x800 timing ablations -- outputs are then garbage)
 *   100000 + m   which shapes take the key-owner backward kernels: bit0 bwd3 for head dim 48, bit1 bwd3 for head dim 64,
 *                bit2 persistent bwd4 (head dim 48, 193 .. 224 tokens), bit3 persistent forward fwd4 (same shapes),
 *                bit4 bwd4 as 16 waves x one key tile, bit5 encoder bwd3 as four waves x one key tile; default 54 */
void hct_debug_force_simple_attention(int on);
/* testing hook: force the NT GEMM tile variant (0 auto, 128, 256) */
void hct_debug_set_gemm_variant(int v);
/* testing hook: start-phase stagger of the persistent GEMM workgroups (-1 auto, 0 off, n = units) */
void hct_debug_set_gemm_stagger(int v);

#ifdef __cplusplus
}
#endif
#endif /* HEADCT_HIP_H */
