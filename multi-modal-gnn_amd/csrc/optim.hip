// Optimizer step and small vector sums on the device (caller side of the hot path: the reference's
// torch.optim.Adam(model.parameters()).step() of src/train.py:219,390).
//
// mmg_adam_step: ONE launch updates every parameter.  Parameters, first and second moments live in three flat fp32
// buckets (483,970 elements for the reference model without its frozen embedding tables); the gradients stay where the
// backward kernels wrote them and are found through a pointer table passed by value (<= MMG_ADAM_MAX_TENSORS entries
// per launch; longer lists take several launches that share the step counter).  Arithmetic = torch.optim.Adam
// (amsgrad = False, maximize = False): g += wd * p;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;
// p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps).  The step counter t is a device-resident float (hipGraph
// replays advance it): every workgroup reads it on entry, a one-thread launch behind the update writes t + 1 (`ticket` is
// unused since round 4 and stays in the signature for the ABI).
#include "common.h"

namespace {

struct AdamTable {
  const float* grad[MMG_ADAM_MAX_TENSORS];   // NULL: the parameter received no gradient this step (left untouched)
  int32_t off[MMG_ADAM_MAX_TENSORS + 1];     // element offsets of the tensors inside the flat buckets (ascending)
  int n;
};

__global__ __launch_bounds__(256) void k_adam(AdamTable tb, float* __restrict__ p, float* __restrict__ m,
                                              float* __restrict__ v, float lr, float b1, float b2, float eps, float wd,
                                              const float* __restrict__ hyper, const float* __restrict__ step) {
  // device-resident hyper-parameters (a captured step keeps its launch arguments; a scheduler changes lr between replays)
  if (hyper) { lr = hyper[0]; b1 = hyper[1]; b2 = hyper[2]; eps = hyper[3]; wd = hyper[4]; }
  // the bias corrections are two powf per step: one wave computes them, LDS hands them to the workgroup
  __shared__ float s_bc[3];
  // The table travels as a kernel argument, and the kernarg segment is HOST memory: a lane-indexed read of it (the binary
  // search below, then every advance of `lo`) is a round trip over the host link -- nine dependent ones per thread made this
  // kernel 27 us for 484 k parameters.  ONE round brings the table into LDS (lane i reads entry i), the searches run there.
  __shared__ int s_off[MMG_ADAM_MAX_TENSORS + 1];
  __shared__ const float* s_grad[MMG_ADAM_MAX_TENSORS];
  for (int i = threadIdx.x; i <= tb.n; i += 256) s_off[i] = tb.off[i];
  for (int i = threadIdx.x; i < tb.n; i += 256) s_grad[i] = tb.grad[i];
  if (threadIdx.x == 0) {
    const float t0 = *step + 1.f;
    s_bc[0] = t0; s_bc[1] = 1.f - powf(b1, t0); s_bc[2] = sqrtf(1.f - powf(b2, t0));
  }
  __syncthreads();
  const float bc1 = s_bc[1], bc2s = s_bc[2];
  const float step_size = lr / bc1;
  const int n_t = tb.n, off0 = s_off[0];
  const int total = s_off[n_t] - off0;
  // every workgroup owns one contiguous chunk of the bucket, so a thread's elements (256 apart) cross a tensor boundary
  // rarely: one binary search for its first element, then a linear advance
  const int chunk = ((total + gridDim.x - 1) / gridDim.x + 255) & ~255;
  const int beg = blockIdx.x * chunk, end = min(total, beg + chunk);
  int lo = 0;
  {
    const int e0 = off0 + beg + threadIdx.x;
    int hi = n_t;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (s_off[mid] <= e0) lo = mid; else hi = mid;
    }
  }
  for (int i = beg + threadIdx.x; i < end; i += 256) {
    const int e = off0 + i;
    while (lo + 1 < n_t && s_off[lo + 1] <= e) ++lo;
    const float* gp = s_grad[lo];
    if (!gp) continue;
    float g = gp[e - s_off[lo]];
    const float pv = p[e];
    if (wd != 0.f) g = fmaf(wd, pv, g);
    const float mv = fmaf(b1, m[e], (1.f - b1) * g);
    const float vv = fmaf(b2, v[e], (1.f - b2) * g * g);
    m[e] = mv; v[e] = vv;
    p[e] = pv - step_size * (mv / (sqrtf(vv) / bc2s + eps));
  }
}

// The step counter advances in a launch of its own behind the update (one thread).  It used to be the last workgroup to
// finish -- __threadfence() + a ticket per workgroup: on this chip a device-scope fence writes the XCD's L2 back, and a
// thousand of them made the kernel 33 us instead of 4 (measured with the tail compiled out).
__global__ void k_adam_bump(float* __restrict__ step) { *step += 1.f; }

struct SumJobs {
  float* dst[MMG_SUM_MAX_JOBS];
  const float* src[MMG_SUM_MAX_JOBS][4];
  int n_src[MMG_SUM_MAX_JOBS], len[MMG_SUM_MAX_JOBS];
  int cols[MMG_SUM_MAX_JOBS], ld_dst[MMG_SUM_MAX_JOBS], ld_src[MMG_SUM_MAX_JOBS][4];
  int n;
};

__global__ __launch_bounds__(256) void k_vec_sums(SumJobs jb) {
  const int j = blockIdx.y;
  const int n = jb.len[j], cols = jb.cols[j];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int r = i / cols, c = i - r * cols;                         // (a flat job is one row)
    float s = jb.src[j][0][(size_t)r * jb.ld_src[j][0] + c];
    for (int q = 1; q < jb.n_src[j]; ++q) s += jb.src[j][q][(size_t)r * jb.ld_src[j][q] + c];      // fixed order
    jb.dst[j][(size_t)r * jb.ld_dst[j] + c] = s;
  }
}

struct CounterTable { int64_t* p[MMG_COUNTERS_MAX]; int64_t inc[MMG_COUNTERS_MAX]; int n; };
__global__ void k_counters_add(CounterTable tb) {
  const int i = threadIdx.x;
  if (i < tb.n) *tb.p[i] += tb.inc[i];
}

__global__ void k_seed_advance(uint64_t* state) {
  uint64_t z = (state[1] += 0x9E3779B97F4A7C15ull);                   // SplitMix64
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  state[0] = (z ^ (z >> 31)) >> 2;
}

}  // namespace

static int adam_launch(float* p, float* m, float* v, const float* const* grads, const int32_t* offsets, int n_tensors,
                       float lr, float beta1, float beta2, float eps, float weight_decay, const float* hyper, float* step,
                       uint32_t* ticket, void* stream) {
  MMG_CHECK_ARG(p && m && v && grads && offsets && step && ticket, "adam_step: null buffer");
  MMG_CHECK_ARG(n_tensors >= 1, "adam_step: no tensors");
  hipStream_t st = (hipStream_t)stream;
  for (int t0 = 0; t0 < n_tensors; t0 += MMG_ADAM_MAX_TENSORS) {
    AdamTable tb;
    const int n = n_tensors - t0 < MMG_ADAM_MAX_TENSORS ? n_tensors - t0 : MMG_ADAM_MAX_TENSORS;
    tb.n = n;
    for (int i = 0; i < n; ++i) { tb.grad[i] = grads[t0 + i]; tb.off[i] = offsets[t0 + i]; }
    tb.off[n] = offsets[t0 + n];
    MMG_CHECK_ARG(tb.off[n] >= tb.off[0], "adam_step: offsets must ascend");
    const int total = tb.off[n] - tb.off[0];
    // two elements per thread up to 2048 workgroups: the launch is all latency (a few loads per element), not bandwidth
    int nb = (total + 256 * 2 - 1) / (256 * 2);
    if (nb < 1) nb = 1;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_adam, dim3(nb), dim3(256), 0, st, tb, p, m, v, lr, beta1, beta2, eps, weight_decay, hyper, step);
  }
  hipLaunchKernelGGL(k_adam_bump, dim3(1), dim3(1), 0, st, step);       // every launch above read the same t
  MMG_CHECK_LAUNCH("adam_step");
  return MMG_OK;
}

extern "C" int mmg_adam_step(float* p, float* m, float* v, const float* const* grads, const int32_t* offsets, int n_tensors,
                             float lr, float beta1, float beta2, float eps, float weight_decay, float* step,
                             uint32_t* ticket, void* stream) {
  return adam_launch(p, m, v, grads, offsets, n_tensors, lr, beta1, beta2, eps, weight_decay, nullptr, step, ticket, stream);
}

extern "C" int mmg_adam_step_dev(float* p, float* m, float* v, const float* const* grads, const int32_t* offsets,
                                 int n_tensors, const float* hyper, float* step, uint32_t* ticket, void* stream) {
  MMG_CHECK_ARG(hyper, "adam_step_dev: null hyper-parameter buffer");
  return adam_launch(p, m, v, grads, offsets, n_tensors, 0.f, 0.f, 0.f, 0.f, 0.f, hyper, step, ticket, stream);
}

extern "C" int mmg_vec_sums(const mmg_sum_job_t* jobs, int n_jobs, void* stream) {
  MMG_CHECK_ARG(jobs && n_jobs >= 1 && n_jobs <= MMG_SUM_MAX_JOBS, "vec_sums: 1..%d jobs", MMG_SUM_MAX_JOBS);
  SumJobs jb;
  jb.n = n_jobs;
  int maxlen = 0;
  for (int j = 0; j < n_jobs; ++j) {
    MMG_CHECK_ARG(jobs[j].dst && jobs[j].n_src >= 1 && jobs[j].n_src <= 4 && jobs[j].len >= 0, "vec_sums: bad job %d", j);
    jb.dst[j] = jobs[j].dst; jb.n_src[j] = jobs[j].n_src; jb.len[j] = jobs[j].len;
    const int cols = jobs[j].cols;
    MMG_CHECK_ARG(cols >= 0 && (cols == 0 || jobs[j].len % cols == 0), "vec_sums: job %d: len is not a multiple of cols", j);
    jb.cols[j] = cols > 0 ? cols : (jobs[j].len > 0 ? jobs[j].len : 1);
    jb.ld_dst[j] = cols > 0 ? jobs[j].ld_dst : 0;
    MMG_CHECK_ARG(cols == 0 || jobs[j].ld_dst >= cols, "vec_sums: job %d: ld_dst < cols", j);
    for (int q = 0; q < 4; ++q) {
      jb.src[j][q] = q < jobs[j].n_src ? jobs[j].src[q] : nullptr;
      jb.ld_src[j][q] = cols > 0 ? jobs[j].ld_src[q] : 0;
      MMG_CHECK_ARG(cols == 0 || q >= jobs[j].n_src || jobs[j].ld_src[q] >= cols, "vec_sums: job %d: ld_src < cols", j);
    }
    for (int q = 0; q < jobs[j].n_src; ++q) MMG_CHECK_ARG(jobs[j].src[q], "vec_sums: job %d has a null source", j);
    if (jobs[j].len > maxlen) maxlen = jobs[j].len;
    // the jobs of a launch run concurrently: two of them writing one destination would lose a contribution
    for (int i = 0; i < j; ++i) MMG_CHECK_ARG(jobs[i].dst != jobs[j].dst, "vec_sums: jobs %d and %d share a destination", i, j);
  }
  if (maxlen == 0) return MMG_OK;
  int nb = (maxlen + 256 * 4 - 1) / (256 * 4);
  if (nb > 64) nb = 64;
  hipLaunchKernelGGL(k_vec_sums, dim3(nb, n_jobs), dim3(256), 0, (hipStream_t)stream, jb);
  MMG_CHECK_LAUNCH("vec_sums");
  return MMG_OK;
}

extern "C" int mmg_counters_add(int64_t* const* counters, const int64_t* incs, int n, void* stream) {
  MMG_CHECK_ARG(n >= 0 && n <= MMG_COUNTERS_MAX, "counters_add: 0..%d counters", MMG_COUNTERS_MAX);
  if (n == 0) return MMG_OK;
  MMG_CHECK_ARG(counters && incs, "counters_add: null table");
  CounterTable tb;
  tb.n = n;
  for (int i = 0; i < n; ++i) {
    MMG_CHECK_ARG(counters[i], "counters_add: counter %d is null", i);
    tb.p[i] = counters[i]; tb.inc[i] = incs[i];
  }
  hipLaunchKernelGGL(k_counters_add, dim3(1), dim3(64), 0, (hipStream_t)stream, tb);
  MMG_CHECK_LAUNCH("counters_add");
  return MMG_OK;
}

extern "C" int mmg_seed_advance(uint64_t* state, void* stream) {
  MMG_CHECK_ARG(state, "seed_advance: null state");
  hipLaunchKernelGGL(k_seed_advance, dim3(1), dim3(1), 0, (hipStream_t)stream, state);
  MMG_CHECK_LAUNCH("seed_advance");
  return MMG_OK;
}

extern "C" int mmg_fill_zero(void* ptr, size_t bytes, void* stream) {
  if (bytes == 0) return MMG_OK;
  MMG_CHECK_ARG(ptr, "fill_zero: null buffer");
  MMG_CHECK_HIP(mmg_zero_async(ptr, bytes, (hipStream_t)stream), "fill_zero");
  return MMG_OK;
}
