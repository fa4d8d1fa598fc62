// Evaluation reducers on the device (reference src/evaluate.py:36-82 metrics, :417-440 per-lab +-3 sigma winsorisation,
// :89-141 per-lab rows, :237-342 stratifications): everything evaluate_model derives from the predictions is a set of
// SEGMENT sums over the prediction pairs -- segment = lab index (50 segments), patient-degree bucket, lab-frequency
// bucket -- so the predictions never leave the device; only [n_seg, 8] doubles do.
//   pass 1  mmg_seg_moments : per segment  n, sum r, sum r^2            (r = pred - target)
//   pass 2  mmg_seg_metrics : per segment, after clipping r to mean +- n_sigma * std of ITS segment (segments with > 1
//                             sample; n_sigma <= 0: no clipping):  n, sum |e|, sum e^2, sum t, sum t^2,
//                             sum |e / t| over t != 0, count(t != 0), count(clipped);  e = t - (t + clip(r)) in fp32 as
//                             the reference forms it; optionally writes the adjusted predictions.
// Accumulation: fp64, per-workgroup LDS accumulators (ds_add_f64), one partial row per workgroup summed in fixed order.
// Within a workgroup the order of the fp64 adds is not fixed: results agree to ~1e-15 relative between runs.
#include "common.h"

namespace {

constexpr int EV_MAXSEG = 2048;      // 2048 x 8 doubles = 128 KB of LDS
constexpr int EV_BLOCKS = 256;

__global__ __launch_bounds__(256) void k_seg_moments(const float* __restrict__ pred, const float* __restrict__ target,
                                                     const int64_t* __restrict__ seg, int64_t n, int n_seg,
                                                     double* __restrict__ partial) {
  extern __shared__ double ev_acc[];
  for (int i = threadIdx.x; i < n_seg * 3; i += 256) ev_acc[i] = 0.0;
  __syncthreads();
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) {
    const int64_t s = seg[k];
    if (s < 0 || s >= n_seg) continue;
    const float r = pred[k] - target[k];
    atomicAdd(&ev_acc[s * 3 + 0], 1.0);
    atomicAdd(&ev_acc[s * 3 + 1], (double)r);
    atomicAdd(&ev_acc[s * 3 + 2], (double)r * (double)r);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n_seg * 3; i += 256) partial[(size_t)blockIdx.x * n_seg * 3 + i] = ev_acc[i];
}

__global__ __launch_bounds__(256) void k_seg_metrics(const float* __restrict__ pred, const float* __restrict__ target,
                                                     const int64_t* __restrict__ seg, int64_t n, int n_seg,
                                                     const double* __restrict__ moments, float n_sigma,
                                                     float* __restrict__ pred_out, double* __restrict__ partial) {
  extern __shared__ double ev_acc[];
  for (int i = threadIdx.x; i < n_seg * 8; i += 256) ev_acc[i] = 0.0;
  __syncthreads();
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) {
    const int64_t s = seg[k];
    const float p = pred[k], t = target[k];
    float pa = p;
    bool capped = false;
    const bool in = s >= 0 && s < n_seg;
    if (in && n_sigma > 0.f && moments) {
      const double cnt = moments[s * 3 + 0];
      if (cnt > 1.0) {                                  // evaluate.py:426: labs with a single sample are left alone
        const double mu = moments[s * 3 + 1] / cnt;
        double var = moments[s * 3 + 2] / cnt - mu * mu;  // np.std: population standard deviation
        var = var > 0.0 ? var : 0.0;
        const float muf = (float)mu, sdf = (float)sqrt(var);
        const float lo = muf - n_sigma * sdf, hi = muf + n_sigma * sdf;
        const float r = p - t;
        const float rc = fminf(fmaxf(r, lo), hi);
        capped = rc != r;
        pa = t + rc;
      }
    }
    if (pred_out) pred_out[k] = pa;
    if (!in) continue;
    const float e = t - pa;
    double* a = ev_acc + s * 8;
    atomicAdd(a + 0, 1.0);
    atomicAdd(a + 1, (double)fabsf(e));
    atomicAdd(a + 2, (double)e * (double)e);
    atomicAdd(a + 3, (double)t);
    atomicAdd(a + 4, (double)t * (double)t);
    if (t != 0.f) {
      atomicAdd(a + 5, (double)fabsf(e / t));
      atomicAdd(a + 6, 1.0);
    }
    if (capped) atomicAdd(a + 7, 1.0);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n_seg * 8; i += 256) partial[(size_t)blockIdx.x * n_seg * 8 + i] = ev_acc[i];
}

inline int ev_blocks(int64_t n) {
  int64_t b = (n + 256 * 16 - 1) / (256 * 16);
  if (b > EV_BLOCKS) b = EV_BLOCKS;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" int mmg_partial_sum(const double* partial, double* out, int n, int n_rows, void* stream);

extern "C" size_t mmg_seg_reduce_ws_bytes(int64_t n, int n_seg) {
  if (n < 0 || n_seg <= 0 || n_seg > EV_MAXSEG) return 0;
  return (size_t)EV_BLOCKS * n_seg * 8 * sizeof(double) + 256;
}

extern "C" int mmg_seg_moments(const float* pred, const float* target, const int64_t* seg, int64_t n, int n_seg,
                               double* moments, void* ws, size_t ws_bytes, void* stream) {
  MMG_CHECK_ARG(n >= 0 && n_seg > 0 && n_seg <= EV_MAXSEG, "seg_moments: n_seg must be 1..%d", EV_MAXSEG);
  MMG_CHECK_ARG(moments && ws && ws_bytes >= mmg_seg_reduce_ws_bytes(n, n_seg), "seg_moments: bad buffer / workspace");
  MMG_CHECK_ARG(n == 0 || (pred && target && seg), "seg_moments: null buffer");
  hipStream_t st = (hipStream_t)stream;
  double* partial = (double*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  const int nb = ev_blocks(n);
  constexpr int lds_max = EV_MAXSEG * 8 * 8;
  MMG_CHECK_HIP((MmgMaxLds<&k_seg_moments, lds_max>::set()), "seg_moments(attr)");
  hipLaunchKernelGGL(k_seg_moments, dim3(nb), dim3(256), (size_t)n_seg * 3 * 8, st, pred, target, seg, n, n_seg, partial);
  MMG_CHECK_LAUNCH("seg_moments");
  return mmg_partial_sum(partial, moments, n_seg * 3, nb, stream);
}

extern "C" int mmg_seg_metrics(const float* pred, const float* target, const int64_t* seg, int64_t n, int n_seg,
                               const double* moments, float n_sigma, float* pred_out, double* sums, void* ws,
                               size_t ws_bytes, void* stream) {
  MMG_CHECK_ARG(n >= 0 && n_seg > 0 && n_seg <= EV_MAXSEG, "seg_metrics: n_seg must be 1..%d", EV_MAXSEG);
  MMG_CHECK_ARG(sums && ws && ws_bytes >= mmg_seg_reduce_ws_bytes(n, n_seg), "seg_metrics: bad buffer / workspace");
  MMG_CHECK_ARG(n == 0 || (pred && target && seg), "seg_metrics: null buffer");
  MMG_CHECK_ARG(!(n_sigma > 0.f) || moments, "seg_metrics: clipping needs the moments of pass 1");
  hipStream_t st = (hipStream_t)stream;
  double* partial = (double*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  const int nb = ev_blocks(n);
  constexpr int lds_max = EV_MAXSEG * 8 * 8;
  MMG_CHECK_HIP((MmgMaxLds<&k_seg_metrics, lds_max>::set()), "seg_metrics(attr)");
  hipLaunchKernelGGL(k_seg_metrics, dim3(nb), dim3(256), (size_t)n_seg * 8 * 8, st, pred, target, seg, n, n_seg, moments,
                     n_sigma, pred_out, partial);
  MMG_CHECK_LAUNCH("seg_metrics");
  return mmg_partial_sum(partial, sums, n_seg * 8, nb, stream);
}
