// Error plumbing + version of libmmgnn's C ABI (include/mmgnn.h).
#include "common.h"
#include <atomic>
#include <mutex>
#include <vector>
#include <string>
#include <string.h>
#include <ctype.h>
#include <utility>

static thread_local char g_err[512] = "";

void mmg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int mmg_version(void) { return 100; }   // 0.1.0
extern "C" const char* mmg_last_error(void) { return g_err; }

extern "C" int mmg_stream_create(void** stream_out) {
  MMG_CHECK_ARG(stream_out, "stream_create: null output");
  hipStream_t st = nullptr;
  const hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  if (e != hipSuccess) { mmg_set_error("stream_create: %s", hipGetErrorString(e)); return MMG_E_LAUNCH; }
  *stream_out = (void*)st;
  return MMG_OK;
}
extern "C" int mmg_stream_destroy(void* stream) {
  if (!stream) return MMG_OK;
  const hipError_t e = hipStreamDestroy((hipStream_t)stream);
  if (e != hipSuccess) { mmg_set_error("stream_destroy: %s", hipGetErrorString(e)); return MMG_E_LAUNCH; }
  return MMG_OK;
}

// ---- measurement hook: (event pair, kernel family, shape) of the launches made while armed.  This is the ONE piece of
// process-wide mutable state in the library (the backward of a step runs on the autograd engine's thread, so a
// per-thread hook would miss half of the step); it is mutex-guarded, never armed by the product path, and costs the
// unarmed launch path one relaxed atomic load.
namespace {
struct ProbeEntry { hipEvent_t e0, e1; int tag; int64_t M; int N, K, flags; const char* text; const char* launcher; };
std::mutex g_probe_mu;
std::vector<ProbeEntry> g_probe;
std::atomic<int> g_probe_left{0};
void probe_clear() {
  for (auto& e : g_probe) { (void)hipEventDestroy(e.e0); (void)hipEventDestroy(e.e1); }
  g_probe.clear();
}
}  // namespace

bool mmg_probe_take(int tag, int64_t M, int N, int K, int flags, const char* kernel_text, const char* launcher,
                    hipEvent_t* e0, hipEvent_t* e1) {
  if (g_probe_left.load(std::memory_order_relaxed) <= 0) return false;
  std::lock_guard<std::mutex> lk(g_probe_mu);
  if (g_probe_left.load(std::memory_order_relaxed) <= 0) return false;
  ProbeEntry e{nullptr, nullptr, tag, M, N, K, flags, kernel_text, launcher};
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

// Instantiated kernel symbol of a probed launch, as rocprofv3 prints it: the kernel expression of the launch site,
// e.g. "(k_linear_fwd_x6<K, WN, PRO, ACC, L2>)", with the template parameters bound by the launcher's
// __PRETTY_FUNCTION__ suffix "[K = 128, WN = 4, PRO = false, ACC = false, L2 = false]" -> "k_linear_fwd_x6<128, 4, false, false, false>".
static void probe_symbol(const char* text, const char* launcher, char* out, int cap) {
  out[0] = 0;
  if (!text || cap <= 1) return;
  std::vector<std::pair<std::string, std::string>> bind;
  if (launcher) {
    const char* lb = strrchr(launcher, '[');
    const char* rb = lb ? strrchr(lb, ']') : nullptr;
    if (lb && rb) {
      std::string body(lb + 1, rb);
      size_t pos = 0;
      while (pos < body.size()) {
        size_t eq = body.find(" = ", pos);
        if (eq == std::string::npos) break;
        size_t end = eq + 3;
        int depth = 0;
        for (; end < body.size(); ++end) {                 // a value may itself contain commas inside <> or ()
          const char c = body[end];
          if (c == '<' || c == '(') ++depth;
          else if (c == '>' || c == ')') --depth;
          else if (c == ',' && depth == 0) break;
        }
        std::string name = body.substr(pos, eq - pos), val = body.substr(eq + 3, end - eq - 3);
        while (!name.empty() && name.front() == ' ') name.erase(0, 1);
        bind.emplace_back(name, val);
        pos = end + 1;
      }
    }
  }
  std::string t(text), r;
  for (size_t i = 0; i < t.size();) {
    const char c = t[i];
    if (isalpha((unsigned char)c) || c == '_') {
      size_t j = i;
      while (j < t.size() && (isalnum((unsigned char)t[j]) || t[j] == '_')) ++j;
      std::string id = t.substr(i, j - i);
      for (auto& b : bind) if (b.first == id) { id = b.second; break; }
      r += id;
      i = j;
    } else {
      r += c;
      ++i;
    }
  }
  while (!r.empty() && (r.front() == '(' || r.front() == ' ' || r.front() == '&')) r.erase(0, 1);
  while (!r.empty() && (r.back() == ')' || r.back() == ' ')) r.pop_back();
  snprintf(out, (size_t)cap, "%s", r.c_str());
}

extern "C" int mmg_probe_read(float* ms, int* tag, int64_t* M, int* N, int* K, int* flags, char* names, int cap) {
  std::lock_guard<std::mutex> lk(g_probe_mu);
  g_probe_left.store(0, std::memory_order_relaxed);
  int n = 0;
  for (auto& e : g_probe) {
    float t = 0.f;
    if (n < cap && hipEventSynchronize(e.e1) == hipSuccess && hipEventElapsedTime(&t, e.e0, e.e1) == hipSuccess) {
      ms[n] = t; tag[n] = e.tag; M[n] = e.M; N[n] = e.N; K[n] = e.K; flags[n] = e.flags;
      if (names) probe_symbol(e.text, e.launcher, names + (size_t)n * MMG_PROBE_NAME_LEN, MMG_PROBE_NAME_LEN);
      ++n;
    }
  }
  probe_clear();
  return n;
}
