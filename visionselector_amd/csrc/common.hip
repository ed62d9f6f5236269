// Error plumbing and version for libvsel.
#include "common.h"

#include <stdarg.h>
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include <string.h>
#include <utility>
#include <vector>

namespace vsel {
static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }

int fail(vsel_status st, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return (int)st;
}
const std::string& last_error() { return g_last_error; }
}  // namespace vsel

namespace vsel {
// ---- diagnostic knobs (include/vsel_debug.h) -----------------------------------------------------------------------
namespace {
struct KnobSpec { const char* env; int def, lo, hi; };
constexpr KnobSpec kKnobs[VSEL_KNOB_COUNT] = {
    {"VSEL_PIPELINE", 0, 0, 1},        // VSEL_KNOB_LIS_PIPELINE
    {"VSEL_SMALL_PATH", 4, 0, 8},      // VSEL_KNOB_LIS_SMALL_PATH (8 = lis_small.h's kSmallMaxSeg)
    {"VSEL_FUSED_SELECT", 48, 0, 1 << 30},
    {nullptr, 1, 0, 1},                // VSEL_KNOB_ATTN_USE_TR
    {nullptr, 0, 0, 8},                // VSEL_KNOB_ATTN_WAVES
    {nullptr, 2, 0, 2},                // VSEL_KNOB_ATTN_PACK
    {nullptr, 2, 0, 2},                // VSEL_KNOB_ATTN_SPLIT
    {nullptr, 1, 0, 1},                // VSEL_KNOB_ATTN_SPLIT_Q64
    {nullptr, -1, -1, 4},              // VSEL_KNOB_ATTN_BWD_SPLIT
    {"VSEL_SPLICE_FUSED", 1, 0, 1},    // VSEL_KNOB_LIS_SPLICE_FUSED
    {"VSEL_ATTN_BWD_WAVES", 8, 4, 8},  // VSEL_KNOB_ATTN_BWD_WAVES
    {nullptr, -1, -1, 1},              // VSEL_KNOB_ATTN_TAIL_FIRST
    {"VSEL_SEG_SUMS", 896, 0, 1 << 30},  // VSEL_KNOB_LIS_SEG_SUMS
    {"VSEL_ATTN_XCD_QUEUE", -1, -1, 1},  // VSEL_KNOB_ATTN_XCD_QUEUE
    {"VSEL_ATTN_ROWS64", -1, -1, 1},     // VSEL_KNOB_ATTN_ROWS64
    {"VSEL_ATTN_BWD_DQ64", -1, -1, 1},   // VSEL_KNOB_ATTN_BWD_DQ64
    {"VSEL_ATTN_BWD_DKDV64", -1, -1, 1}, // VSEL_KNOB_ATTN_BWD_DKDV64
    {"VSEL_ATTN_STATIC", -1, -1, 1},     // VSEL_KNOB_ATTN_STATIC
    {"VSEL_ATTN_SKIP_EMPTY", 1, 0, 1},   // VSEL_KNOB_ATTN_SKIP_EMPTY
    {"VSEL_ATTN_GQA", -1, -1, 1},        // VSEL_KNOB_ATTN_GQA
    {"VSEL_ATTN_GQA_FORM", -1, -1, 1},   // VSEL_KNOB_ATTN_GQA_FORM
    {"VSEL_ATTN_BWD_UPDOWN", 1, 0, 1},   // VSEL_KNOB_ATTN_BWD_UPDOWN
    {"VSEL_ATTN_KEY_PARTS", -1, -1, 64}, // VSEL_KNOB_ATTN_KEY_PARTS
    {"VSEL_GATHER", 82, 0, 84},          // VSEL_KNOB_LIS_GATHER
    {"VSEL_TRAIN_FUSED", 1, 0, 1},       // VSEL_KNOB_TRAIN_FUSED
};
int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
int knob_default(int id) {
  const KnobSpec& k = kKnobs[id];
  const char* e = k.env ? getenv(k.env) : nullptr;
  return clampi(e ? atoi(e) : k.def, k.lo, k.hi);
}
struct KnobTable {
  std::atomic<int> v[VSEL_KNOB_COUNT];
  KnobTable() { for (int i = 0; i < VSEL_KNOB_COUNT; ++i) v[i].store(knob_default(i), std::memory_order_relaxed); }
};
KnobTable& knobs() {
  static KnobTable t;          // constructed on first use (thread-safe), environment read once
  return t;
}
}  // namespace
int knob(int id) { return knobs().v[id].load(std::memory_order_relaxed); }
int attn_use_xcd_queues(int64_t max_seqlen, int64_t n_pairs, int64_t min_len, int64_t min_pairs) {
  const int k = knob(VSEL_KNOB_ATTN_XCD_QUEUE);
  return k >= 0 ? k : ((max_seqlen >= min_len && n_pairs >= min_pairs) ? 1 : 0);
}
bool attn_static_deal(int64_t n_items, int64_t slots, bool auto_ok, int rounds20) {
  if (n_items <= slots) return false;
  const int k = knob(VSEL_KNOB_ATTN_STATIC);
  return k >= 0 ? k != 0 : (auto_ok && 20 * n_items <= rounds20 * slots);
}
}  // namespace vsel

extern "C" int vsel_debug_set(int id, int value, int* previous) {
  using namespace vsel;
  if (id < 0 || id >= VSEL_KNOB_COUNT) return fail(VSEL_ERR_INVALID, "vsel_debug_set: unknown knob %d", id);
  const int old = knobs().v[id].exchange(clampi(value, kKnobs[id].lo, kKnobs[id].hi), std::memory_order_relaxed);
  if (previous) *previous = old;
  return VSEL_OK;
}
extern "C" int vsel_debug_get(int id, int* value) {
  using namespace vsel;
  if (id < 0 || id >= VSEL_KNOB_COUNT || !value) return fail(VSEL_ERR_INVALID, "vsel_debug_get: unknown knob %d", id);
  *value = knob(id);
  return VSEL_OK;
}
extern "C" void vsel_debug_reset(void) {
  using namespace vsel;
  for (int i = 0; i < VSEL_KNOB_COUNT; ++i) knobs().v[i].store(knob_default(i), std::memory_order_relaxed);
}

namespace vsel {
namespace {
// slot -> owning stream.  Launches of one stream always take that stream's slot (stream order protects the counter: the memset that
// re-arms it runs behind the previous launch), so the common cases -- one stream, or a handful -- cost a table lookup and put nothing
// into the stream (a HIP event per launch, the first form of this, added ~6 us of marker packets to every 25 us launch).
struct SlotTable {
  hipStream_t owner[64];
  bool used[64];
};
std::mutex g_slot_mu;
SlotTable g_slot_tables[16][kSlotFamilies];
}  // namespace
int queue_slot_acquire(int family, hipStream_t st, int* slot) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return fail(VSEL_ERR_HIP, "hipGetDevice");
  // hipStreamPerThread is ONE handle value for a different stream in every host thread: a slot keyed by it would be shared by launches
  // that stream order does not serialise
  if (st == hipStreamPerThread)
    return fail(VSEL_ERR_UNSUPPORTED, "queued attention launches on hipStreamPerThread: pass an explicit stream (work-queue counters are per stream handle)");
  std::lock_guard<std::mutex> lock(g_slot_mu);
  SlotTable& t = g_slot_tables[dev][family];
  int free_slot = -1;
  for (int i = 0; i < 64; ++i) {
    if (t.used[i] && t.owner[i] == st) { *slot = i; return VSEL_OK; }
    if (!t.used[i] && free_slot < 0) free_slot = i;
  }
  if (free_slot < 0) {
    // 64 other streams hold the slots: take over one whose stream has nothing in flight (or no longer exists)
    for (int i = 0; i < 64 && free_slot < 0; ++i) {
      const hipError_t q = hipStreamQuery(t.owner[i]);
      if (q != hipErrorNotReady) { (void)hipGetLastError(); free_slot = i; }
    }
    if (free_slot < 0)
      return fail(VSEL_ERR_BUSY, "64 other streams have launches of this attention kernel in flight: every work-queue counter is taken");
  }
  t.used[free_slot] = true;
  t.owner[free_slot] = st;
  *slot = free_slot;
  return VSEL_OK;
}
void queue_slot_launched(int, int, hipStream_t) {}      // (nothing to record: ownership is per stream)
}  // namespace vsel

namespace vsel {
// Per-kernel timing with HIP events on the launch stream: VSEL_LAUNCH records a "<begin>" mark in front of every kernel and
// VSEL_AFTER_LAUNCH the named mark behind it; a kernel's time is the interval between the two (the host gap since the previous
// launch is in front of the begin mark and is not booked to anybody).  Off by default; bench.py turns it on for an instrumented
// repeat of the timed region.  Single-threaded use (one process per GPU).
struct Profiler {
  bool on = false;
  std::vector<hipEvent_t> pool;
  std::vector<std::pair<const char*, hipEvent_t>> marks;
};
static Profiler g_prof;
bool prof_enabled() { return g_prof.on; }
void prof_mark(hipStream_t st, const char* name) {
  const size_t i = g_prof.marks.size();
  if (i >= g_prof.pool.size()) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    g_prof.pool.push_back(e);
  }
  (void)hipEventRecord(g_prof.pool[i], st);
  g_prof.marks.emplace_back(name, g_prof.pool[i]);
}
}  // namespace vsel

extern "C" int vsel_profile_start(void) {
  // the event pool is created HERE (and kept for the life of the process), never inside an interval it times: a hipEventCreate on
  // the first instrumented pass used to land between two marks and was booked to whatever kernel came next
  while (vsel::g_prof.pool.size() < 4096) {
    hipEvent_t e;
    VSEL_HIP_CHECK(hipEventCreate(&e));
    vsel::g_prof.pool.push_back(e);
  }
  vsel::g_prof.marks.clear();
  vsel::g_prof.on = true;
  return VSEL_OK;
}

extern "C" int vsel_profile_stop(char* names, size_t names_len, float* total_ms, int64_t* calls, int max_entries,
                                 int* n_entries) {
  using namespace vsel;
  g_prof.on = false;
  if (!names || !total_ms || !calls || !n_entries || max_entries < 1) return fail(VSEL_ERR_INVALID, "NULL pointer");
  std::vector<std::string> keys;
  std::vector<double> ms;
  std::vector<int64_t> cnt;
  if (!g_prof.marks.empty()) VSEL_HIP_CHECK(hipEventSynchronize(g_prof.marks.back().second));
  for (size_t i = 1; i < g_prof.marks.size(); ++i) {
    const char* nm = g_prof.marks[i].first;
    if (!strcmp(nm, "<begin>")) continue;
    float dt = 0.f;
    VSEL_HIP_CHECK(hipEventElapsedTime(&dt, g_prof.marks[i - 1].second, g_prof.marks[i].second));
    size_t j = 0;
    for (; j < keys.size(); ++j) if (keys[j] == nm) break;
    if (j == keys.size()) { keys.emplace_back(nm); ms.push_back(0); cnt.push_back(0); }
    ms[j] += dt;
    cnt[j] += 1;
  }
  g_prof.marks.clear();
  std::string joined;
  int n = 0;
  for (size_t j = 0; j < keys.size() && n < max_entries; ++j, ++n) {
    if (j) joined += ",";
    joined += keys[j];
    total_ms[n] = (float)ms[j];
    calls[n] = cnt[j];
  }
  if (joined.size() + 1 > names_len) return fail(VSEL_ERR_INVALID, "names buffer too small");
  memcpy(names, joined.c_str(), joined.size() + 1);
  *n_entries = n;
  return VSEL_OK;
}

extern "C" const char* vsel_version(void) { return "vsel 0.1.0 (gfx950)"; }
extern "C" const char* vsel_last_error(void) {
  static thread_local std::string copy;
  copy = vsel::last_error();
  return copy.c_str();
}
