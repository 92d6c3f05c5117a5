// C ABI for multi-head self-attention (attentionblock.py:54-62).  Dispatch: bf16 storage with a head size the
// MFMA kernels cover -> attention_mfma.hip; everything else (all fp32 parity work) -> attention_simple.hip.
#include "common.h"
#include "prof.h"

namespace hct {
int attention_fwd_simple(const void* qkv, int B, int N, int H, int dh, int dtype, void* o, float* lse, hipStream_t s);
int attention_bwd_simple(const void* qkv, const void* o, const void* d_o, const float* lse, int B, int N, int H, int dh,
                         int dtype, void* dqkv, hipStream_t s);
bool attention_mfma_supported(int N, int H, int dh);
int attention_fwd_mfma(const void* qkv, int B, int N, int H, int dh, void* o, float* lse, hipStream_t s);
int attention_bwd_mfma(const void* qkv, const void* o, const void* d_o, const float* lse, int B, int N, int H, int dh,
                       void* dqkv, hipStream_t s);
int g_force_simple_attention = 0;
extern int g_attn_row;
extern int g_attn_dbg;
extern int g_attn_bwd3;
}  // namespace hct

using namespace hct;

extern "C" {

// testing hook: route bf16 attention through the simple kernels (A/B comparisons)
void hct_debug_force_simple_attention(int on) {
  if (on >= 100000) { g_attn_bwd3 = on - 100000; return; }  // which shapes take the key-owner backward (bit0 dh 48, bit1 dh 64)
  if (on >= 10) { g_attn_dbg = on - 10; return; }
  if (on >= 2) { g_force_simple_attention = 0; g_attn_row = on == 2 ? 0 : 1; return; }  // 2: online-softmax MFMA kernel, 3: full-row
  g_force_simple_attention = on;
  g_attn_row = 1;
}

int hct_attention_fwd(const void* qkv, int B, int N, int H, int dh, int dtype, void* o, float* lse, void* stream) {
  HCT_REQUIRE(B > 0 && N > 0 && H > 0 && dh > 0, "hct_attention_fwd: bad shape");
  ProfScope ps(PROF_ATTN_FWD, 4.0 * B * H * (double)N * N * dh, (hipStream_t)stream);
  if (dtype == HCT_BF16 && !g_force_simple_attention && attention_mfma_supported(N, H, dh))
    return attention_fwd_mfma(qkv, B, N, H, dh, o, lse, (hipStream_t)stream);
  return attention_fwd_simple(qkv, B, N, H, dh, dtype, o, lse, (hipStream_t)stream);
}

int hct_attention_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, int B, int N, int H, int dh,
                      int dtype, void* dqkv, void* stream) {
  HCT_REQUIRE(B > 0 && N > 0 && H > 0 && dh > 0, "hct_attention_bwd: bad shape");
  ProfScope ps(PROF_ATTN_BWD, 10.0 * B * H * (double)N * N * dh, (hipStream_t)stream);
  if (dtype == HCT_BF16 && !g_force_simple_attention && attention_mfma_supported(N, H, dh))
    return attention_bwd_mfma(qkv, o, d_o, lse, B, N, H, dh, dqkv, (hipStream_t)stream);
  return attention_bwd_simple(qkv, o, d_o, lse, B, N, H, dh, dtype, dqkv, (hipStream_t)stream);
}

}  // extern "C"
