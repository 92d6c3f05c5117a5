#pragma once
#include <hip/hip_runtime.h>
namespace hct {
// kernel classes for hct_prof_read
enum { PROF_GEMM_NT = 0, PROF_GEMM_TN = 1, PROF_GEMM_GENERIC = 2, PROF_ATTN_FWD = 3, PROF_ATTN_BWD = 4, PROF_LN = 5,
       PROF_OPTIM = 6 };
bool prof_enabled();
struct ProfScope {
  ProfScope(int id, double work, hipStream_t s, double bytes = 0.0);
  ~ProfScope();
  // shape key of the launch (GEMMs: M, N, K, epilogue mode, tiles, stream-K tiles) for hct_prof_shapes
  void tag(int m, int n, int k, int mode, int tiles, int sk_tiles) { tag_[0] = m; tag_[1] = n; tag_[2] = k; tag_[3] = mode; tag_[4] = tiles; tag_[5] = sk_tiles; }
  int id_; double work_; double bytes_ = 0.0; hipStream_t s_; bool on_; void* a_ = nullptr; void* b_ = nullptr;
  int tag_[6] = {0, 0, 0, 0, 0, 0};
};
}  // namespace hct
