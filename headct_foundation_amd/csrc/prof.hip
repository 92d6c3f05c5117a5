// In-library kernel timing with HIP events on the launch stream (bench.py's roofline leg).
// Disabled by default: PROF_SCOPE costs one predictable branch per launch.
#include "common.h"
#include "prof.h"

#include <mutex>
#include <vector>

namespace hct {

struct ProfRec { int id; hipEvent_t a, b; double work, bytes; int tag[6]; };
static unsigned g_prof_mask = 0;  // bit i = kernel class i is timed
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;
static std::mutex g_mu;

bool prof_enabled() { return g_prof_mask != 0; }

static hipEvent_t take_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}

ProfScope::ProfScope(int id, double work, hipStream_t s, double bytes) : id_(id), work_(work), bytes_(bytes), s_(s), on_((g_prof_mask >> id) & 1u) {
  if (!on_) return;
  std::lock_guard<std::mutex> lk(g_mu);
  a_ = take_event();
  b_ = take_event();
  (void)hipEventRecord((hipEvent_t)a_, s_);
}
ProfScope::~ProfScope() {
  if (!on_) return;
  (void)hipEventRecord((hipEvent_t)b_, s_);
  std::lock_guard<std::mutex> lk(g_mu);
  g_recs.push_back(ProfRec{id_, (hipEvent_t)a_, (hipEvent_t)b_, work_, bytes_, {tag_[0], tag_[1], tag_[2], tag_[3], tag_[4], tag_[5]}});
}

}  // namespace hct

using namespace hct;

extern "C" {

void hct_prof_enable(int mask) { g_prof_mask = (unsigned)mask; }

void hct_prof_reset(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& r : g_recs) { g_pool.push_back(r.a); g_pool.push_back(r.b); }
  g_recs.clear();
}

// Sums over all recorded launches of kernel class `id` (see prof.h); blocks until they have finished.
int hct_prof_read(int id, double* total_ms, int64_t* launches, double* work) {
  std::lock_guard<std::mutex> lk(g_mu);
  double ms = 0, w = 0;
  int64_t n = 0;
  for (auto& r : g_recs) {
    if (r.id != id) continue;
    hipError_t e = hipEventSynchronize(r.b);
    if (e != hipSuccess) return check_hip(e, "hct_prof_read");
    float t = 0.f;
    e = hipEventElapsedTime(&t, r.a, r.b);
    if (e != hipSuccess) return check_hip(e, "hct_prof_read");
    ms += t; w += r.work; ++n;
  }
  if (total_ms) *total_ms = ms;
  if (launches) *launches = n;
  if (work) *work = w;
  return 0;
}

// Sum of the algorithmic bytes (operands read once + outputs written once) the recorded launches of class `id` declared.
int hct_prof_read_bytes(int id, double* bytes) {
  std::lock_guard<std::mutex> lk(g_mu);
  double b = 0;
  for (auto& r : g_recs)
    if (r.id == id) b += r.bytes;
  if (bytes) *bytes = b;
  return 0;
}

// Per-shape view of the recorded launches of class `id`: one entry per distinct (M, N, K, mode, tiles, stream-K tiles) key, in
// order of first appearance.  Returns the number of distinct keys (entries beyond `cap` are counted, not written).
int hct_prof_shapes(int id, hct_prof_shape* out, int cap) {
  std::lock_guard<std::mutex> lk(g_mu);
  std::vector<hct_prof_shape> tab;
  for (auto& r : g_recs) {
    if (r.id != id) continue;
    if (hipEventSynchronize(r.b) != hipSuccess) return -1;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) return -1;
    hct_prof_shape* e = nullptr;
    for (auto& x : tab)
      if (x.M == r.tag[0] && x.N == r.tag[1] && x.K == r.tag[2] && x.mode == r.tag[3] && x.tiles == r.tag[4] && x.sk_tiles == r.tag[5]) { e = &x; break; }
    if (!e) {
      tab.push_back(hct_prof_shape{r.tag[0], r.tag[1], r.tag[2], r.tag[3], r.tag[4], r.tag[5], 0, 0.0, 0.0, 0.0});
      e = &tab.back();
    }
    e->launches += 1; e->total_ms += t; e->work += r.work; e->bytes += r.bytes;
  }
  for (size_t i = 0; i < tab.size() && (int)i < cap; ++i) out[i] = tab[i];
  return (int)tab.size();
}

}  // extern "C"
