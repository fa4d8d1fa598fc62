// Error plumbing + version of libmmgnn's C ABI (include/mmgnn.h).
#include "common.h"
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

// ---- measurement hook: per-thread list of (event pair, kernel family, shape) for the launches made while armed
namespace {
struct ProbeEntry { hipEvent_t e0, e1; int tag; int64_t M; int N, K, flags; };
thread_local std::vector<ProbeEntry> t_probe;
thread_local int t_probe_left = 0;
void probe_clear() {
  for (auto& e : t_probe) { (void)hipEventDestroy(e.e0); (void)hipEventDestroy(e.e1); }
  t_probe.clear();
}
}  // namespace

bool mmg_probe_take(int tag, int64_t M, int N, int K, int flags, hipEvent_t* e0, hipEvent_t* e1) {
  if (t_probe_left <= 0) return false;
  ProbeEntry e{nullptr, nullptr, tag, M, N, K, flags};
  if (hipEventCreate(&e.e0) != hipSuccess) return false;
  if (hipEventCreate(&e.e1) != hipSuccess) { (void)hipEventDestroy(e.e0); return false; }
  t_probe.push_back(e);
  --t_probe_left;
  *e0 = e.e0; *e1 = e.e1;
  return true;
}

extern "C" int mmg_probe_arm(int n_launches) {
  probe_clear();
  t_probe_left = n_launches > 0 ? n_launches : 0;
  return MMG_OK;
}

extern "C" int mmg_probe_read(float* ms, int* tag, int64_t* M, int* N, int* K, int* flags, int cap) {
  t_probe_left = 0;
  int n = 0;
  for (auto& e : t_probe) {
    float t = 0.f;
    if (n < cap && hipEventSynchronize(e.e1) == hipSuccess && hipEventElapsedTime(&t, e.e0, e.e1) == hipSuccess) {
      ms[n] = t; tag[n] = e.tag; M[n] = e.M; N[n] = e.N; K[n] = e.K; flags[n] = e.flags;
      ++n;
    }
  }
  probe_clear();
  return n;
}
