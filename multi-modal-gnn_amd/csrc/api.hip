// Error plumbing + version of libmmgnn's C ABI (include/mmgnn.h).
#include "common.h"
#include <atomic>
#include <mutex>
#include <vector>

static thread_local char g_err[512] = "";

void mmg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int mmg_version(void) { return 100; }   // 0.1.0
extern "C" const char* mmg_last_error(void) { return g_err; }

// ---- measurement hook: (event pair, kernel family, shape) of the launches made while armed.  This is the ONE piece of
// process-wide mutable state in the library (the backward of a step runs on the autograd engine's thread, so a
// per-thread hook would miss half of the step); it is mutex-guarded, never armed by the product path, and costs the
// unarmed launch path one relaxed atomic load.
namespace {
struct ProbeEntry { hipEvent_t e0, e1; int tag; int64_t M; int N, K, flags; };
std::mutex g_probe_mu;
std::vector<ProbeEntry> g_probe;
std::atomic<int> g_probe_left{0};
void probe_clear() {
  for (auto& e : g_probe) { (void)hipEventDestroy(e.e0); (void)hipEventDestroy(e.e1); }
  g_probe.clear();
}
}  // namespace

bool mmg_probe_take(int tag, int64_t M, int N, int K, int flags, hipEvent_t* e0, hipEvent_t* e1) {
  if (g_probe_left.load(std::memory_order_relaxed) <= 0) return false;
  std::lock_guard<std::mutex> lk(g_probe_mu);
  if (g_probe_left.load(std::memory_order_relaxed) <= 0) return false;
  ProbeEntry e{nullptr, nullptr, tag, M, N, K, flags};
  if (hipEventCreate(&e.e0) != hipSuccess) return false;
  if (hipEventCreate(&e.e1) != hipSuccess) { (void)hipEventDestroy(e.e0); return false; }
  g_probe.push_back(e);
  g_probe_left.fetch_sub(1, std::memory_order_relaxed);
  *e0 = e.e0; *e1 = e.e1;
  return true;
}

extern "C" int mmg_probe_arm(int n_launches) {
  std::lock_guard<std::mutex> lk(g_probe_mu);
  probe_clear();
  g_probe_left.store(n_launches > 0 ? n_launches : 0, std::memory_order_relaxed);
  return MMG_OK;
}

extern "C" int mmg_probe_read(float* ms, int* tag, int64_t* M, int* N, int* K, int* flags, int cap) {
  std::lock_guard<std::mutex> lk(g_probe_mu);
  g_probe_left.store(0, std::memory_order_relaxed);
  int n = 0;
  for (auto& e : g_probe) {
    float t = 0.f;
    if (n < cap && hipEventSynchronize(e.e1) == hipSuccess && hipEventElapsedTime(&t, e.e0, e.e1) == hipSuccess) {
      ms[n] = t; tag[n] = e.tag; M[n] = e.M; N[n] = e.N; K[n] = e.K; flags[n] = e.flags;
      ++n;
    }
  }
  probe_clear();
  return n;
}
