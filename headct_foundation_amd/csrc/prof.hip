// In-library kernel timing with HIP events on the launch stream (bench.py's roofline leg).
// Disabled by default: PROF_SCOPE costs one predictable branch per launch.
#include "common.h"
#include "prof.h"

#include <mutex>
#include <vector>

namespace hct {

struct ProfRec { int id; hipEvent_t a, b; double work, bytes; };
static unsigned g_prof_mask = 0;  // bit i = kernel class i is timed
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;
static std::mutex g_mu;

bool prof_enabled() { return g_prof_mask != 0; }

static hipEvent_t take_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e;
  hipEventCreate(&e);
  return e;
}

ProfScope::ProfScope(int id, double work, hipStream_t s, double bytes) : id_(id), work_(work), bytes_(bytes), s_(s), on_((g_prof_mask >> id) & 1u) {
  if (!on_) return;
  std::lock_guard<std::mutex> lk(g_mu);
  a_ = take_event();
  b_ = take_event();
  hipEventRecord((hipEvent_t)a_, s_);
}
ProfScope::~ProfScope() {
  if (!on_) return;
  hipEventRecord((hipEvent_t)b_, s_);
  std::lock_guard<std::mutex> lk(g_mu);
  g_recs.push_back(ProfRec{id_, (hipEvent_t)a_, (hipEvent_t)b_, work_, bytes_});
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

}  // extern "C"
