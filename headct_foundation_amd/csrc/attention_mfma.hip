// bf16 MFMA attention (placeholder until the LDS-resident MFMA kernels land; dispatcher falls back to simple).
#include "common.h"
namespace hct {
bool attention_mfma_supported(int N, int H, int dh) { (void)N; (void)H; (void)dh; return false; }
int attention_fwd_mfma(const void*, int, int, int, int, void*, float*, hipStream_t) { set_error("mfma attention not built"); return HCT_E_UNSUPPORTED; }
int attention_bwd_mfma(const void*, const void*, const void*, const float*, int, int, int, int, void*, hipStream_t) { set_error("mfma attention not built"); return HCT_E_UNSUPPORTED; }
}  // namespace hct
