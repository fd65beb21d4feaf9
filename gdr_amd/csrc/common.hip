#include <atomic>

#include "common.h"

#include <mutex>
#include <vector>

namespace gdr {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---------------------------------------------------------------------------------------- per-device kernel attributes
namespace {
struct AttrKey {
  const void* fn;
  int dev;
};
std::mutex g_attr_mu;
std::vector<AttrKey> g_attr_done;
}  // namespace

int ensure_dyn_lds(const void* kernel, int bytes, const char* what) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    set_error("%s: hipGetDevice failed", what);
    return GDR_EHIP;
  }
  std::lock_guard<std::mutex> lock(g_attr_mu);
  for (const AttrKey& k : g_attr_done)
    if (k.fn == kernel && k.dev == dev) return GDR_OK;
  hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    set_error("%s: hipFuncSetAttribute: %s", what, hipGetErrorString(e));
    return GDR_EHIP;
  }
  g_attr_done.push_back(AttrKey{kernel, dev});
  return GDR_OK;
}

// ---------------------------------------------------------------------------------------- profiler
struct ProfState {
  bool on = false;
  bool gate = true;  // gdr_prof_gate: while false the enabled profiler records nothing (sampling by step)
  int cap = 0, used = 0;
  hipEvent_t* start = nullptr;
  hipEvent_t* stop = nullptr;
  int* cls = nullptr;
  double* work = nullptr;
};
static ProfState g_prof;

ProfScope::ProfScope(int c, double w, hipStream_t s) : slot(-1), stream(s) {
  if (!g_prof.on || !g_prof.gate || g_prof.used >= g_prof.cap) return;
  slot = g_prof.used++;
  g_prof.cls[slot] = c;
  g_prof.work[slot] = w;
  (void)hipEventRecord(g_prof.start[slot], stream);
}
ProfScope::~ProfScope() {
  if (slot >= 0) (void)hipEventRecord(g_prof.stop[slot], stream);
}
}  // namespace gdr

extern "C" void gdr_prof_gate(int on) { gdr::g_prof.gate = on != 0; }

extern "C" int gdr_prof_enable(int max_events) {
  using namespace gdr;
  if (g_prof.on || max_events <= 0) {
    set_error("prof_enable: already enabled or bad size");
    return GDR_EINVAL;
  }
  g_prof.start = new hipEvent_t[max_events];
  g_prof.stop = new hipEvent_t[max_events];
  g_prof.cls = new int[max_events];
  g_prof.work = new double[max_events];
  for (int i = 0; i < max_events; ++i) {
    if (hipEventCreate(&g_prof.start[i]) != hipSuccess || hipEventCreate(&g_prof.stop[i]) != hipSuccess) {
      set_error("prof_enable: hipEventCreate failed");
      return GDR_EHIP;
    }
  }
  g_prof.cap = max_events, g_prof.used = 0, g_prof.on = true;
  return GDR_OK;
}

// Synchronises the recorded events, accumulates per class {launches, total ms, total work} into the three
// host arrays of length 8, then disables and frees the profiler.  Returns the number of events lost to a
// full buffer (0 = none) or a negative error code.
extern "C" int gdr_prof_collect(int64_t* launches, double* total_ms, double* total_work) {
  using namespace gdr;
  if (!g_prof.on) {
    set_error("prof_collect: profiler not enabled");
    return GDR_EINVAL;
  }
  for (int c = 0; c < PROF_NCLASS; ++c) launches[c] = 0, total_ms[c] = 0.0, total_work[c] = 0.0;
  int rc = GDR_OK;
  for (int i = 0; i < g_prof.used; ++i) {
    float ms = 0.f;
    if (hipEventSynchronize(g_prof.stop[i]) != hipSuccess ||
        hipEventElapsedTime(&ms, g_prof.start[i], g_prof.stop[i]) != hipSuccess) {
      set_error("prof_collect: event query failed");
      rc = GDR_EHIP;
      break;
    }
    const int c = g_prof.cls[i];
    launches[c] += 1, total_ms[c] += ms, total_work[c] += g_prof.work[i];
  }
  for (int i = 0; i < g_prof.cap; ++i) {
    (void)hipEventDestroy(g_prof.start[i]);
    (void)hipEventDestroy(g_prof.stop[i]);
  }
  delete[] g_prof.start, delete[] g_prof.stop, delete[] g_prof.cls, delete[] g_prof.work;
  g_prof = ProfState();
  return rc;
}

namespace gdr {
static std::atomic<int64_t> g_launches{0};
void count_launch() { g_launches.fetch_add(1, std::memory_order_relaxed); }
}  // namespace gdr
extern "C" int64_t gdr_launch_count(void) { return gdr::g_launches.load(std::memory_order_relaxed); }
extern "C" const char* gdr_last_error(void) { return gdr::g_err; }
// 2: gdr_t5_generate takes a GdrPrefixTable; GdrTrie carries V
// 3: gdr_rerank_topk takes a shard range, flags and a workspace; gdr_rerank_topk_bf16, gdr_cluster_candidates added
// 4: gdr_t5_generate_early_exits added (gdr_t5_generate leaves its step loop when every query is done);
//    GdrPrefixTable.complete_levels
// 5: gdr_rerank_wire_pack / _unpack / gdr_rerank_positions_to_ids (the sharded two-stage path's exchange row)
// 6: gdr_device_fault_pending / _clear / _inject_for_tests (a stream-K hand-off that times out no longer traps: it raises a sticky
//    process-wide word and every stream-K launch enqueued while it is raised fails with GDR_EHIP); gdr_t5_layer_norm (a2 as an
//    operator of its own); gdr_t5_generate_last_done_step
// 7: gdr_sim_topk_prefilter (+ _workspace_bytes), gdr_row_norm2_max: the fp32 top-k through a bf16 pre-filter
extern "C" int gdr_abi_version(void) { return 8; }
