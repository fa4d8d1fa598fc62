// HBM-bound row/column passes around the dense layers: BatchNorm statistics and backward,
// ReLU/Dropout, row L2 normalisation (src/model.py:93-105, 229-232, 258-269 of the reference).
// All loads/stores are 16 B per lane; column reductions accumulate in fp64 and are summed in a
// fixed order (deterministic).
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
// Big inputs: one 1024-thread workgroup per CU (16 waves keep 128 KB of loads in flight per CU) so that only 256
// partial rows are left for the second pass; small inputs: 256-thread workgroups of 64 rows.
constexpr int CR_BIG_ROWS = 32768, CR_BIG_BLOCKS = 256, CR_MAX_BLOCKS = 2048;

struct ColGeom { int cg, rl; int nblk; int64_t rows_per_blk; int nthr; };
inline bool col_geom(int64_t M, int N, ColGeom* g) {
  if (N % 4 || N / 4 > 256 || 256 % (N / 4)) return false;
  g->nthr = M >= CR_BIG_ROWS ? 1024 : 256;
  g->cg = N / 4; g->rl = g->nthr / g->cg;
  int64_t nb = (M + 63) / 64;
  const int64_t cap = M >= CR_BIG_ROWS ? CR_BIG_BLOCKS : CR_MAX_BLOCKS;
  if (nb > cap) nb = cap;
  if (nb < 1) nb = 1;
  g->nblk = (int)nb;
  g->rows_per_blk = (M + nb - 1) / nb;
  return true;
}

// MODE 0: a = A, b = B (or A if B null)       -> (sum a, sum a*b)
// MODE 1: BN backward stats: a = g_out(G,Y), b = xhat(Y)
template <int MODE, int NT>
__global__ __launch_bounds__(NT) void k_col_reduce(const float* __restrict__ A, const float* __restrict__ B,
                                                   ProDev pr, const float* __restrict__ mean,
                                                   const float* __restrict__ rstd, double* __restrict__ partial,
                                                   int64_t M, int N, int64_t rows_per_blk,
                                                   const float* __restrict__ A2, ProDev pr2) {
  // A2 / pr2 (MODE 1, nullable): a second upstream gradient through the SAME BatchNorm + ReLU with its own dropout
  // mask (the two encode_nodes passes share their first layer): a = g_out(A, Y; pr) + g_out(A2, Y; pr2)
  pr.resolve();
  if (MODE == 1 && A2) pr2.resolve();
  __shared__ double red[8 * NT];     // [rl][2][N] with rl*N == 4*NT
  const int cg = N / 4, rl = NT / cg;
  const int c4 = threadIdx.x % cg, rr = threadIdx.x / cg;
  double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f}, mu = {0.f, 0.f, 0.f, 0.f}, rs = {1.f, 1.f, 1.f, 1.f};
  if (MODE == 1) {
    if (pr.scale) {
      sc = *reinterpret_cast<const f32x4*>(pr.scale + c4 * 4);
      sh = *reinterpret_cast<const f32x4*>(pr.shift + c4 * 4);
    }
    mu = *reinterpret_cast<const f32x4*>(mean + c4 * 4);
    rs = *reinterpret_cast<const f32x4*>(rstd + c4 * 4);
  }
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk, r1 = min(M, r0 + rows_per_blk);
  auto body = [&](int64_t r, const f32x4& a_in, const f32x4& b) {
    f32x4 a = a_in;
    if (MODE == 1) {                 // a = upstream grad G, b = pre-BN activation Y: g through relu, then dropout
      f32x4 a2 = {0.f, 0.f, 0.f, 0.f};
      if (A2) a2 = *reinterpret_cast<const f32x4*>(A2 + (size_t)r * N + c4 * 4);
      if (pr.relu == MMG_ACT_RELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float o = pr.scale ? fmaf(b[j], sc[j], sh[j]) : b[j];
          if (!(o > 0.f)) { a[j] = 0.f; a2[j] = 0.f; }
        }
      } else if (pr.relu) {            // elu / leaky_relu: the slope at the pre-activation value
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float d = mmg_act_grad(pr.relu, pr.scale ? fmaf(b[j], sc[j], sh[j]) : b[j]);
          a[j] *= d; a2[j] *= d;
        }
      }
      if (pr.p > 0.f)
        mmg_drop4(a, pr.key, (uint64_t)(pr.row_offset + r) * (uint64_t)N + (uint64_t)(c4 * 4), pr.thr, pr.inv_keep);
      if (A2) {
        if (pr2.p > 0.f)
          mmg_drop4(a2, pr2.key, (uint64_t)(pr2.row_offset + r) * (uint64_t)N + (uint64_t)(c4 * 4), pr2.thr, pr2.inv_keep);
        a += a2;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float av = a[j], bv = b[j];
      if (MODE == 1) {
        bv = (bv - mu[j]) * rs[j];
      }
      s0[j] += (double)av;
      s1[j] += (double)av * (double)bv;
    }
  };
  // Chunks of 4*rl rows are dealt round-robin over the workgroups (the grid then reads one contiguous window that
  // spreads over every HBM channel; one contiguous range per workgroup makes the streams advance a fixed stride apart).
  (void)r0; (void)r1;
  const int64_t chunk = 4 * (int64_t)rl;
  for (int64_t cb = (int64_t)blockIdx.x * chunk; cb < M; cb += (int64_t)gridDim.x * chunk) {
    const int64_t r = cb + rr;
    if (cb + chunk <= M) {                       // 4 independent 16-B loads per operand in flight
      f32x4 a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] = *reinterpret_cast<const f32x4*>(A + (size_t)(r + u * rl) * N + c4 * 4);
#pragma unroll
      for (int u = 0; u < 4; ++u) b[u] = B ? *reinterpret_cast<const f32x4*>(B + (size_t)(r + u * rl) * N + c4 * 4) : a[u];
#pragma unroll
      for (int u = 0; u < 4; ++u) body(r + u * rl, a[u], b[u]);
    } else {
      for (int64_t q = r; q < M; q += rl) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(A + (size_t)q * N + c4 * 4);
        const f32x4 b = B ? *reinterpret_cast<const f32x4*>(B + (size_t)q * N + c4 * 4) : a;
        body(q, a, b);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    red[(rr * 2 + 0) * N + c4 * 4 + j] = s0[j];
    red[(rr * 2 + 1) * N + c4 * 4 + j] = s1[j];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * N; i += NT) {
    double s = 0;
    for (int q = 0; q < rl; ++q) s += red[q * 2 * N + i];
    partial[(size_t)blockIdx.x * 2 * N + i] = s;
  }
}

// one wave per output element: lanes stride over the block partials, then a fixed-order wave reduction
__global__ __launch_bounds__(256) void k_partial_sum(const double* __restrict__ partial, double* __restrict__ out, int n,
                                                     int nblk, const double* add = nullptr) {
  const int lane = threadIdx.x & 63;
  const int i = (blockIdx.x * 256 + threadIdx.x) >> 6;
  if (i >= n) return;
  double s = 0;
  for (int b = lane; b < nblk; b += 64) s += partial[(size_t)b * n + i];
  s = wave_sum_d(s);
  if (lane == 0) out[i] = add ? add[i] + s : s;     // (add may alias out: one reader-writer per element)
}

// one column of the BatchNorm fold (shared by k_bn_finalize and k_partial_sum_bn: the same arithmetic, bit for bit)
__device__ inline void bn_fold_col(double sum1, double sum2, int64_t count, int i, const float* gamma, const float* beta,
                                   float* running_mean, float* running_var, int training, int n_updates, float momentum,
                                   float eps, float* scale, float* shift, float* mean_out, float* rstd_out) {
  float mean, var;
  if (training) {
    const double m = sum1 / (double)count;
    double v = sum2 / (double)count - m * m;
    if (v < 0) v = 0;
    mean = (float)m; var = (float)v;
    if (running_mean) {
      const float unb = (float)(v * ((double)count / (double)(count > 1 ? count - 1 : 1)));
      float rm = running_mean[i], rv = running_var[i];
      for (int u = 0; u < n_updates; ++u) {
        rm = (1.f - momentum) * rm + momentum * mean;
        rv = (1.f - momentum) * rv + momentum * unb;
      }
      running_mean[i] = rm; running_var[i] = rv;
    }
  } else {
    mean = running_mean[i]; var = running_var[i];
  }
  const float rstd = 1.0f / sqrtf(var + eps);
  const float g = gamma ? gamma[i] : 1.f, b = beta ? beta[i] : 0.f;
  const float sc = g * rstd;
  scale[i] = sc;
  shift[i] = b - mean * sc;
  if (mean_out) mean_out[i] = mean;
  if (rstd_out) rstd_out[i] = rstd;
}

__global__ __launch_bounds__(256) void k_bn_finalize(const double* __restrict__ sums, int64_t count,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* running_mean, float* running_var, int training,
                                                     int n_updates, float momentum, float eps, float* scale,
                                                     float* shift, float* mean_out, float* rstd_out, int N) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  bn_fold_col(training ? sums[i] : 0.0, training ? sums[N + i] : 0.0, count, i, gamma, beta, running_mean, running_var,
              training, n_updates, momentum, eps, scale, shift, mean_out, rstd_out);
}

// the partial statistics rows of a producer kernel ([nblk][2][N] fp64) summed in fixed order -- one wave per column, as
// k_partial_sum -- and folded at once (training mode)
__global__ __launch_bounds__(256) void k_partial_sum_bn(const double* __restrict__ partial, double* __restrict__ out, int N,
                                                        int nblk, mmg_bn_fin_t f) {
  const int lane = threadIdx.x & 63;
  const int i = (blockIdx.x * 256 + threadIdx.x) >> 6;
  if (i >= N) return;
  double s1 = 0, s2 = 0;
  for (int b = lane; b < nblk; b += 64) {
    s1 += partial[(size_t)b * 2 * N + i];
    s2 += partial[(size_t)b * 2 * N + N + i];
  }
  s1 = wave_sum_d(s1);
  s2 = wave_sum_d(s2);
  if (lane == 0) {
    out[i] = s1; out[N + i] = s2;
    bn_fold_col(s1, s2, f.count, i, f.gamma, f.beta, f.running_mean, f.running_var, 1, f.n_updates, f.momentum, f.eps,
                f.scale, f.shift, f.mean, f.rstd);
  }
}

// Both passes are pure HBM streams: the per-column constants are hoisted into registers (the grid stride is a
// multiple of N/4, so a thread always works on the same four columns) -- per-element L1 lookups had capped
// the pass at ~2 TB/s.
__global__ __launch_bounds__(256) void k_affine_act_drop(const float* __restrict__ Y, ProDev pr, float* __restrict__ out,
                                                         int64_t M, int N) {
  pr.resolve();
  const int64_t n4 = M * (int64_t)(N / 4);
  const int64_t stride = (int64_t)gridDim.x * 256;          // host guarantees stride % (N/4) == 0
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int c = (int)(i % (N / 4)) * 4;
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if (pr.scale) { sc = *reinterpret_cast<const f32x4*>(pr.scale + c); sh = *reinterpret_cast<const f32x4*>(pr.shift + c); }
  for (; i < n4; i += stride) {
    const int64_t r = i / (N / 4);
    f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Y + (size_t)i * 4));   // (next read: the backward)
    mmg_pro_apply4(pr, v, sc, sh, r, c, N);
    *reinterpret_cast<f32x4*>(out + (size_t)i * 4) = v;
  }
}

__global__ __launch_bounds__(256) void k_affine_act_drop_rows(const float* __restrict__ Y, ProDev pr,
                                                              const int64_t* __restrict__ rows, int64_t n_sel,
                                                              float* __restrict__ out, int N) {
  pr.resolve();
  const int64_t n4 = n_sel * (int64_t)(N / 4);
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const int64_t s = i / (N / 4);
  const int c = (int)(i % (N / 4)) * 4;
  const int64_t r = rows[s];
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if (pr.scale) { sc = *reinterpret_cast<const f32x4*>(pr.scale + c); sh = *reinterpret_cast<const f32x4*>(pr.shift + c); }
  f32x4 v = *reinterpret_cast<const f32x4*>(Y + (size_t)r * N + c);
  mmg_pro_apply4(pr, v, sc, sh, r, c, N);
  *reinterpret_cast<f32x4*>(out + (size_t)i * 4) = v;
}

// BatchNorm backward with an upstream gradient that is zero outside a short row list (the tabular head only reaches the
// low-degree patients): the column statistics need those rows only, the dense apply pass runs without G, and the listed
// rows are patched afterwards.  g' = G_rows * [y*scale+shift > 0] * keepmask/(1-p) with the masks of the ORIGINAL rows.
__device__ __forceinline__ f32x4 bn_rows_gprime(const ProDev& pr, f32x4 g, const f32x4& y, const f32x4& sc, const f32x4& sh,
                                                int64_t row, int c, int N) {
  if (pr.relu == MMG_ACT_RELU) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float act = pr.scale ? fmaf(y[j], sc[j], sh[j]) : y[j];
      if (!(act > 0.f)) g[j] = 0.f;
    }
  } else if (pr.relu) {
#pragma unroll
    for (int j = 0; j < 4; ++j) g[j] *= mmg_act_grad(pr.relu, pr.scale ? fmaf(y[j], sc[j], sh[j]) : y[j]);
  }
  if (pr.p > 0.f) mmg_drop4(g, pr.key, (uint64_t)(pr.row_offset + row) * (uint64_t)N + (uint64_t)c, pr.thr, pr.inv_keep);
  return g;
}

// thread = (row lane, column quad); partial[block][0,:] = sum g', partial[block][1,:] = sum g' * xhat over the block's share
// of the listed rows (dealt round-robin); k_partial_sum adds the blocks in fixed order
constexpr int BSR_MAX_BLOCKS = 128;
__global__ __launch_bounds__(256) void k_bn_bwd_stats_rows(const float* __restrict__ Grows, const float* __restrict__ Y,
                                                           const int64_t* __restrict__ rows, int64_t n_sel, ProDev pr,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           double* __restrict__ partial, int N) {
  pr.resolve();
  __shared__ double red[8 * 2 * 256];             // [row lanes <= 8][2][N <= 256]
  const int cg = N / 4, rl = 256 / cg;
  const int c4 = threadIdx.x % cg, rr = threadIdx.x / cg, c = c4 * 4;
  const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 sc = one, sh = zero;
  if (pr.scale) { sc = *reinterpret_cast<const f32x4*>(pr.scale + c); sh = *reinterpret_cast<const f32x4*>(pr.shift + c); }
  const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), rs = *reinterpret_cast<const f32x4*>(rstd + c);
  double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
  for (int64_t s = (int64_t)blockIdx.x * rl + rr; s < n_sel; s += (int64_t)gridDim.x * rl) {
    const int64_t r = rows[s];
    const f32x4 y = *reinterpret_cast<const f32x4*>(Y + (size_t)r * N + c);
    const f32x4 g = bn_rows_gprime(pr, *reinterpret_cast<const f32x4*>(Grows + (size_t)s * N + c), y, sc, sh, r, c, N);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s0[j] += (double)g[j];
      s1[j] += (double)g[j] * (double)((y[j] - mu[j]) * rs[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    red[(rr * 2 + 0) * N + c + j] = s0[j];
    red[(rr * 2 + 1) * N + c + j] = s1[j];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * N; i += 256) {
    double t = 0;
    for (int q = 0; q < rl; ++q) t += red[q * 2 * N + i];
    partial[(size_t)blockIdx.x * 2 * N + i] = t;
  }
}

// dY[rows[s], :] += scale * g'   (the g' term the G-less apply pass left out)
__global__ __launch_bounds__(256) void k_bn_bwd_apply_rows(const float* __restrict__ Grows, const float* __restrict__ Y,
                                                           const int64_t* __restrict__ rows, int64_t n_sel, ProDev pr,
                                                           float* __restrict__ dY, int N) {
  pr.resolve();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_sel * (int64_t)(N / 4)) return;
  const int64_t s = i / (N / 4);
  const int c = (int)(i % (N / 4)) * 4;
  const int64_t r = rows[s];
  const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 sc = one, sh = zero;
  if (pr.scale) { sc = *reinterpret_cast<const f32x4*>(pr.scale + c); sh = *reinterpret_cast<const f32x4*>(pr.shift + c); }
  const f32x4 y = *reinterpret_cast<const f32x4*>(Y + (size_t)r * N + c);
  const f32x4 g = bn_rows_gprime(pr, *reinterpret_cast<const f32x4*>(Grows + (size_t)s * N + c), y, sc, sh, r, c, N);
  f32x4* d = reinterpret_cast<f32x4*>(dY + (size_t)r * N + c);
  f32x4 o = *d;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = fmaf(sc[j], g[j], o[j]);
  *d = o;
}

__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float* __restrict__ G, const float* __restrict__ Y, ProDev pr,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      const double* __restrict__ sums, double inv_count,
                                                      float* __restrict__ dbeta, float* __restrict__ dgamma,
                                                      float* __restrict__ dY, int64_t M, int N, int accumulate,
                                                      const float* __restrict__ G2, ProDev pr2) {
  pr.resolve();
  if (G2) pr2.resolve();
  const int64_t n4 = M * (int64_t)(N / 4);
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int c = (int)(i % (N / 4)) * 4;
  const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 sc = one, sh = zero, mu = zero, rs = one, a0 = zero, a1 = zero;
  if (pr.scale) {
    sc = *reinterpret_cast<const f32x4*>(pr.scale + c); sh = *reinterpret_cast<const f32x4*>(pr.shift + c);
    mu = *reinterpret_cast<const f32x4*>(mean + c); rs = *reinterpret_cast<const f32x4*>(rstd + c);
    if (sums) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a0[j] = (float)(sums[c + j] * inv_count);
        a1[j] = (float)(sums[N + c + j] * inv_count);
      }
    }
  }
  if (sums && blockIdx.x == 0 && threadIdx.x < N / 4) {       // d beta / d gamma ride along (one thread per 4 columns)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (dbeta) dbeta[c + j] = (float)sums[c + j];
      if (dgamma) dgamma[c + j] = (float)sums[N + c + j];
    }
  }
  for (; i < n4; i += stride) {
    const int64_t r = i / (N / 4);
    // (G and Y are read for the last time here: non-temporal loads leave the cache to dY, which the next kernels read)
    const f32x4 g4 = G ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(G + (size_t)i * 4)) : zero;   // no G: zero upstream
    const f32x4 y4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Y + (size_t)i * 4));
    f32x4 prev = zero;
    if (accumulate) prev = *reinterpret_cast<const f32x4*>(dY + (size_t)i * 4);
    f32x4 o, gm = g4, gm2 = zero;
    if (G2) gm2 = *reinterpret_cast<const f32x4*>(G2 + (size_t)i * 4);      // second upstream gradient (see k_col_reduce)
    if (pr.relu == MMG_ACT_RELU) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float act = pr.scale ? fmaf(y4[j], sc[j], sh[j]) : y4[j];
        if (!(act > 0.f)) { gm[j] = 0.f; gm2[j] = 0.f; }
      }
    } else if (pr.relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = mmg_act_grad(pr.relu, pr.scale ? fmaf(y4[j], sc[j], sh[j]) : y4[j]);
        gm[j] *= d; gm2[j] *= d;
      }
    }
    if (pr.p > 0.f) mmg_drop4(gm, pr.key, (uint64_t)(pr.row_offset + r) * (uint64_t)N + (uint64_t)c, pr.thr, pr.inv_keep);
    if (G2) {
      if (pr2.p > 0.f)
        mmg_drop4(gm2, pr2.key, (uint64_t)(pr2.row_offset + r) * (uint64_t)N + (uint64_t)c, pr2.thr, pr2.inv_keep);
      gm += gm2;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float g = gm[j];
      if (pr.scale) {
        const float xh = (y4[j] - mu[j]) * rs[j];
        g = sc[j] * (g - a0[j] - xh * a1[j]);
      }
      o[j] = g + prev[j];
    }
    *reinterpret_cast<f32x4*>(dY + (size_t)i * 4) = o;
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void k_l2norm_fwd(const float* __restrict__ Z, float* __restrict__ out,
                                                    float* __restrict__ rnorm, int64_t M, float eps) {
  constexpr int N = VEC * 64;
  const int lane = threadIdx.x & 63;
  const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  if (row >= M) return;
  float v[VEC], ss = 0.f;
#pragma unroll
  for (int j = 0; j < VEC; ++j) { v[j] = Z[(size_t)row * N + lane * VEC + j]; ss = fmaf(v[j], v[j], ss); }
  ss = wave_sum(ss);
  const float nrm = sqrtf(ss);
  const float r = 1.0f / fmaxf(nrm, eps);
#pragma unroll
  for (int j = 0; j < VEC; ++j) out[(size_t)row * N + lane * VEC + j] = v[j] * r;
  if (lane == 0 && rnorm) rnorm[row] = r;
}

template <int VEC>
__global__ __launch_bounds__(256) void k_l2norm_bwd(const float* __restrict__ G, const float* __restrict__ out,
                                                    const float* __restrict__ rnorm, float* __restrict__ dZ, int64_t M,
                                                    float eps) {
  constexpr int N = VEC * 64;
  const int lane = threadIdx.x & 63;
  const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  if (row >= M) return;
  float g[VEC], o[VEC], dot = 0.f;
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    g[j] = G[(size_t)row * N + lane * VEC + j];
    o[j] = out[(size_t)row * N + lane * VEC + j];
    dot = fmaf(g[j], o[j], dot);
  }
  dot = wave_sum(dot);
  const float r = rnorm[row];
  const bool clamped = !(r * eps < 1.0f);   // ||z|| <= eps: the denominator was the constant eps
  if (clamped) dot = 0.f;
#pragma unroll
  for (int j = 0; j < VEC; ++j) dZ[(size_t)row * N + lane * VEC + j] = r * (g[j] - o[j] * dot);
}

__global__ __launch_bounds__(256) void k_dropout_mask(uint64_t seed, const uint64_t* seed_ptr, uint32_t site, int64_t first,
                                                      int64_t n, float p, uint8_t* mask) {
  if (seed_ptr) seed = *seed_ptr;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) mask[i] = mmg_keep(seed, site, (uint64_t)(first + i), p) ? 1 : 0;
}

constexpr int PL_BLOCKS = 1024;
__global__ __launch_bounds__(256) void k_pair_loss(const float* __restrict__ pred, const float* __restrict__ y,
                                                   const float* __restrict__ w, const float* __restrict__ sup, int64_t n,
                                                   double inv_den, const double* __restrict__ inv_den_ptr, int loss_type,
                                                   float* __restrict__ dpred, double* __restrict__ partial) {
  __shared__ double red[4];
  double s = 0;
  if (inv_den_ptr) inv_den = *inv_den_ptr;          // device-resident normaliser (hipGraph replays with a new mask)
  const float invf = (float)inv_den;
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) {
    const float d = pred[k] - y[k];
    const float ws = (w ? w[k] : 1.f) * (sup ? sup[k] : 1.f);
    float per, g;
    if (loss_type == 0) { per = fabsf(d); g = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }
    else if (loss_type == 1) { per = d * d; g = 2.f * d; }
    else {                                           // huber, delta = 1 (F.huber_loss default)
      const float a = fabsf(d);
      if (a <= 1.f) { per = 0.5f * d * d; g = d; } else { per = a - 0.5f; g = d > 0.f ? 1.f : -1.f; }
    }
    s += (double)(per * ws);
    if (dpred) dpred[k] = g * ws * invf;
  }
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1] + red[2] + red[3]) * inv_den;
}

// Supervision subset of one epoch (src/train.py:150-176: torch.rand(n) < mask_fraction, redrawn every epoch): drawn on
// the device from the counter RNG, keyed on (the seed the captured step reads, a site of its own, the pair's id).
constexpr uint32_t SUP_SITE = 0x53555031u;
__global__ __launch_bounds__(256) void k_sup_mask_draw(const uint64_t* __restrict__ seed_ptr, uint64_t seed,
                                                       const int64_t* __restrict__ ids, int64_t n, float keep_p,
                                                       float* __restrict__ sup, double* __restrict__ partial) {
  __shared__ double red[4];
  if (seed_ptr) seed = *seed_ptr;
  double s = 0;
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) {
    const uint64_t e = ids ? (uint64_t)ids[k] : (uint64_t)k;
    const float m = mmg_keep(seed, SUP_SITE, e, keep_p) ? 1.f : 0.f;
    if (sup) sup[k] = m;                           // (sup == NULL: count only)
    s += (double)m;
  }
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(64) void k_sup_mask_count(const double* __restrict__ partial, int nblk, double* __restrict__ count,
                                                       double* __restrict__ inv_den) {
  double s = 0;
  for (int b = threadIdx.x; b < nblk; b += 64) s += partial[b];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) {
    if (count) *count = s;
    if (inv_den) *inv_den = 1.0 / (s > 1.0 ? s : 1.0);
  }
}

inline unsigned ew_grid(int64_t n4) {
  int64_t b = (n4 + 255) / 256;
  if (b > 256 * 128) b = 256 * 128;   // everything in flight: these passes are pure HBM streams
  if (b < 1) b = 1;
  return (unsigned)b;
}

template <int MODE>
int run_col_reduce(const float* A, const float* B, const ProDev& pr, const float* mean, const float* rstd,
                   double* out, int64_t M, int N, void* ws, size_t ws_bytes, hipStream_t st, const char* what,
                   const float* A2 = nullptr, const ProDev* pr2p = nullptr) {
  const ProDev pr2 = pr2p ? *pr2p : pr;
  ColGeom g;
  MMG_CHECK_ARG(col_geom(M, N, &g), "%s: N=%d unsupported", what, N);
  const size_t need = (size_t)g.nblk * 2 * N * 8 + 256;
  if (ws_bytes < need) { mmg_set_error("%s: workspace %zu < %zu", what, ws_bytes, need); return MMG_E_WS; }
  double* partial = (double*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  if (g.nblk == 1) {      // small tables (vocab side): the single workgroup's result IS the answer
    hipLaunchKernelGGL((k_col_reduce<MODE, 256>), dim3(1), dim3(256), 0, st, A, B, pr, mean, rstd, out, M, N, g.rows_per_blk, A2,
                       pr2);
    return MMG_OK;
  }
  if (g.nthr == 1024)
    hipLaunchKernelGGL((k_col_reduce<MODE, 1024>), dim3(g.nblk), dim3(1024), 0, st, A, B, pr, mean, rstd, partial, M, N,
                       g.rows_per_blk, A2, pr2);
  else
    hipLaunchKernelGGL((k_col_reduce<MODE, 256>), dim3(g.nblk), dim3(256), 0, st, A, B, pr, mean, rstd, partial, M, N,
                       g.rows_per_blk, A2, pr2);
  hipLaunchKernelGGL(k_partial_sum, dim3((2 * N + 3) / 4), dim3(256), 0, st, partial, out, 2 * N, g.nblk);
  return MMG_OK;
}

}  // namespace

// internal (gemm.hip): out[i] = sum over n_rows partial rows, one wave per output, fixed order
extern "C" int mmg_partial_sum(const double* partial, double* out, int n, int n_rows, void* stream) {
  hipLaunchKernelGGL(k_partial_sum, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, partial, out, n, n_rows);
  MMG_CHECK_LAUNCH("partial_sum");
  return MMG_OK;
}

// internal (gemm.hip, aggregate.hip): out[i] = add[i] + the sum (add nullable, may alias out)
extern "C" int mmg_partial_sum_add(const double* partial, double* out, int n, int n_rows, const double* add, void* stream) {
  hipLaunchKernelGGL(k_partial_sum, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, partial, out, n, n_rows, add);
  MMG_CHECK_LAUNCH("partial_sum");
  return MMG_OK;
}

// internal (gemm.hip, aggregate.hip): the same with the BatchNorm fold of the summed statistics (fin != NULL)
extern "C" int mmg_partial_sum_bn(const double* partial, double* col_sums, int N, int n_rows, const mmg_bn_fin_t* fin,
                                  void* stream) {
  if (!fin) return mmg_partial_sum(partial, col_sums, 2 * N, n_rows, stream);
  MMG_CHECK_ARG(fin->count > 0 && fin->scale && fin->shift, "partial_sum_bn: bad BatchNorm fold descriptor");
  hipLaunchKernelGGL(k_partial_sum_bn, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, partial, col_sums, N, n_rows, *fin);
  MMG_CHECK_LAUNCH("partial_sum_bn");
  return MMG_OK;
}

extern "C" size_t mmg_col_reduce2_ws_bytes(int64_t M, int N) {
  ColGeom g;
  if (M < 0 || !col_geom(M, N, &g)) return 0;
  return (size_t)g.nblk * 2 * N * 8 + 256;
}

extern "C" int mmg_col_reduce2(const float* A, const float* B, double* out, int64_t M, int N, void* ws, size_t ws_bytes,
                               void* stream) {
  MMG_CHECK_ARG(M >= 0 && out, "col_reduce2: bad args");
  hipStream_t st = (hipStream_t)stream;
  if (M == 0) { MMG_CHECK_HIP(mmg_zero_async(out, (size_t)2 * N * 8, st), "col_reduce2(memset)"); return MMG_OK; }
  MMG_CHECK_ARG(A && ws, "col_reduce2: null buffer");
  int rc = run_col_reduce<0>(A, B, mmg_pro_dev(nullptr), nullptr, nullptr, out, M, N, ws, ws_bytes, st, "col_reduce2");
  if (rc) return rc;
  MMG_CHECK_LAUNCH("col_reduce2");
  return MMG_OK;
}

extern "C" int mmg_bn_finalize(const double* sums, int64_t count, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, int training, int n_updates, float momentum,
                               float eps, float* scale, float* shift, float* mean, float* rstd, int N, void* stream) {
  MMG_CHECK_ARG(N > 0 && scale && shift, "bn_finalize: bad args");
  MMG_CHECK_ARG(training ? (sums != nullptr && count > 0) : (running_mean && running_var),
                "bn_finalize: missing statistics for the requested mode");
  hipLaunchKernelGGL(k_bn_finalize, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, count, gamma, beta,
                     running_mean, running_var, training, n_updates, momentum, eps, scale, shift, mean, rstd, N);
  MMG_CHECK_LAUNCH("bn_finalize");
  return MMG_OK;
}

extern "C" int mmg_affine_act_drop(const float* Y, const mmg_prologue_t* pro, float* out, int64_t M, int N, void* stream) {
  MMG_CHECK_ARG(M >= 0 && N > 0 && N % 4 == 0 && 256 % (N / 4) == 0, "affine_act_drop: N=%d must be 4*2^k <= 1024", N);
  if (M == 0) return MMG_OK;
  MMG_CHECK_ARG(Y && out, "affine_act_drop: null buffer");
  hipLaunchKernelGGL(k_affine_act_drop, dim3(ew_grid(M * (N / 4))), dim3(256), 0, (hipStream_t)stream, Y, mmg_pro_dev(pro),
                     out, M, N);
  MMG_CHECK_LAUNCH("affine_act_drop");
  return MMG_OK;
}

extern "C" int mmg_affine_act_drop_rows(const float* Y, const mmg_prologue_t* pro, const int64_t* rows, int64_t n_sel,
                                        float* out, int N, void* stream) {
  MMG_CHECK_ARG(n_sel >= 0 && N > 0 && N % 4 == 0, "affine_act_drop_rows: N=%d must be a multiple of 4", N);
  MMG_CHECK_ARG(!pro || !pro->scale || pro->shift, "affine_act_drop_rows: prologue scale without shift");
  if (n_sel == 0) return MMG_OK;
  MMG_CHECK_ARG(Y && rows && out, "affine_act_drop_rows: null buffer");
  const int64_t n4 = n_sel * (int64_t)(N / 4);
  hipLaunchKernelGGL(k_affine_act_drop_rows, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, Y,
                     mmg_pro_dev(pro), rows, n_sel, out, N);
  MMG_CHECK_LAUNCH("affine_act_drop_rows");
  return MMG_OK;
}

extern "C" int mmg_bn_bwd_stats(const float* G, const float* Y, const mmg_prologue_t* pro, const float* mean,
                                const float* rstd, double* sums, int64_t M, int N, void* ws, size_t ws_bytes,
                                void* stream) {
  MMG_CHECK_ARG(M >= 0 && sums && mean && rstd, "bn_bwd_stats: bad args");
  hipStream_t st = (hipStream_t)stream;
  if (M == 0) { MMG_CHECK_HIP(mmg_zero_async(sums, (size_t)2 * N * 8, st), "bn_bwd_stats(memset)"); return MMG_OK; }
  MMG_CHECK_ARG(G && Y && ws, "bn_bwd_stats: null buffer");
  int rc = run_col_reduce<1>(G, Y, mmg_pro_dev(pro), mean, rstd, sums, M, N, ws, ws_bytes, st, "bn_bwd_stats");
  if (rc) return rc;
  MMG_CHECK_LAUNCH("bn_bwd_stats");
  return MMG_OK;
}

extern "C" int mmg_bn_bwd_stats2(const float* G, const float* G2, const float* Y, const mmg_prologue_t* pro,
                                 const mmg_prologue_t* pro2, const float* mean, const float* rstd, double* sums, int64_t M,
                                 int N, void* ws, size_t ws_bytes, void* stream) {
  MMG_CHECK_ARG(M >= 0 && sums && mean && rstd, "bn_bwd_stats2: bad args");
  hipStream_t st = (hipStream_t)stream;
  if (M == 0) { MMG_CHECK_HIP(mmg_zero_async(sums, (size_t)2 * N * 8, st), "bn_bwd_stats(memset)"); return MMG_OK; }
  MMG_CHECK_ARG(G && G2 && Y && ws && pro && pro2, "bn_bwd_stats2: null buffer");
  const ProDev p2 = mmg_pro_dev(pro2);
  int rc = run_col_reduce<1>(G, Y, mmg_pro_dev(pro), mean, rstd, sums, M, N, ws, ws_bytes, st, "bn_bwd_stats2", G2, &p2);
  if (rc) return rc;
  MMG_CHECK_LAUNCH("bn_bwd_stats2");
  return MMG_OK;
}

extern "C" size_t mmg_bn_bwd_stats_rows_ws_bytes(int N) { return (size_t)BSR_MAX_BLOCKS * 2 * (size_t)(N > 0 ? N : 0) * 8 + 256; }

extern "C" int mmg_bn_bwd_stats_rows(const float* G_rows, const float* Y, const int64_t* rows, int64_t n_sel,
                                     const mmg_prologue_t* pro, const float* mean, const float* rstd, double* sums, int N,
                                     void* ws, size_t ws_bytes, void* stream) {
  MMG_CHECK_ARG(n_sel >= 0 && sums && mean && rstd, "bn_bwd_stats_rows: bad args");
  MMG_CHECK_ARG(N > 0 && N % 4 == 0 && N <= 256 && 256 % (N / 4) == 0, "bn_bwd_stats_rows: N=%d unsupported", N);
  hipStream_t st = (hipStream_t)stream;
  if (n_sel == 0) { MMG_CHECK_HIP(mmg_zero_async(sums, (size_t)2 * N * 8, st), "bn_bwd_stats_rows(memset)"); return MMG_OK; }
  MMG_CHECK_ARG(G_rows && Y && rows && ws, "bn_bwd_stats_rows: null buffer");
  MMG_CHECK_ARG(ws_bytes >= mmg_bn_bwd_stats_rows_ws_bytes(N), "bn_bwd_stats_rows: workspace too small");
  double* partial = (double*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  const int rl = 256 / (N / 4);
  int64_t nblk = (n_sel + rl - 1) / rl;
  if (nblk > BSR_MAX_BLOCKS) nblk = BSR_MAX_BLOCKS;
  hipLaunchKernelGGL(k_bn_bwd_stats_rows, dim3((unsigned)nblk), dim3(256), 0, st, G_rows, Y, rows, n_sel, mmg_pro_dev(pro),
                     mean, rstd, partial, N);
  hipLaunchKernelGGL(k_partial_sum, dim3((2 * N + 3) / 4), dim3(256), 0, st, partial, sums, 2 * N, (int)nblk);
  MMG_CHECK_LAUNCH("bn_bwd_stats_rows");
  return MMG_OK;
}

extern "C" int mmg_bn_bwd_apply_rows(const float* G_rows, const float* Y, const int64_t* rows, int64_t n_sel,
                                     const mmg_prologue_t* pro, float* dY, int N, void* stream) {
  MMG_CHECK_ARG(n_sel >= 0 && N > 0 && N % 4 == 0, "bn_bwd_apply_rows: bad args");
  if (n_sel == 0) return MMG_OK;
  MMG_CHECK_ARG(G_rows && Y && rows && dY, "bn_bwd_apply_rows: null buffer");
  const int64_t n4 = n_sel * (int64_t)(N / 4);
  hipLaunchKernelGGL(k_bn_bwd_apply_rows, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, G_rows, Y,
                     rows, n_sel, mmg_pro_dev(pro), dY, N);
  MMG_CHECK_LAUNCH("bn_bwd_apply_rows");
  return MMG_OK;
}

extern "C" int mmg_bn_bwd_apply(const float* G, const float* Y, const mmg_prologue_t* pro, const float* mean,
                                const float* rstd, const double* sums, double inv_count, float* dbeta, float* dgamma,
                                float* dY, int64_t M, int N, int accumulate, void* stream) {
  MMG_CHECK_ARG(M >= 0 && N > 0 && N % 4 == 0 && 256 % (N / 4) == 0, "bn_bwd_apply: N=%d must be 4*2^k <= 1024", N);
  if (M == 0) return MMG_OK;
  MMG_CHECK_ARG(Y && dY, "bn_bwd_apply: null buffer");          // (G may be NULL: an all-zero upstream gradient)
  MMG_CHECK_ARG(!pro || !pro->scale || (mean && rstd), "bn_bwd_apply: affine prologue needs mean/rstd");
  hipLaunchKernelGGL(k_bn_bwd_apply, dim3(ew_grid(M * (N / 4))), dim3(256), 0, (hipStream_t)stream, G, Y, mmg_pro_dev(pro),
                     mean, rstd, sums, inv_count, dbeta, dgamma, dY, M, N, accumulate, (const float*)nullptr,
                     mmg_pro_dev(pro));
  MMG_CHECK_LAUNCH("bn_bwd_apply");
  return MMG_OK;
}

extern "C" int mmg_bn_bwd_apply2(const float* G, const float* G2, const float* Y, const mmg_prologue_t* pro,
                                 const mmg_prologue_t* pro2, const float* mean, const float* rstd, const double* sums,
                                 double inv_count, float* dbeta, float* dgamma, float* dY, int64_t M, int N, void* stream) {
  MMG_CHECK_ARG(M >= 0 && N > 0 && N % 4 == 0 && 256 % (N / 4) == 0, "bn_bwd_apply2: N=%d must be 4*2^k <= 1024", N);
  if (M == 0) return MMG_OK;
  MMG_CHECK_ARG(G && G2 && Y && dY && pro && pro2 && mean && rstd, "bn_bwd_apply2: null buffer");
  hipLaunchKernelGGL(k_bn_bwd_apply, dim3(ew_grid(M * (N / 4))), dim3(256), 0, (hipStream_t)stream, G, Y, mmg_pro_dev(pro),
                     mean, rstd, sums, inv_count, dbeta, dgamma, dY, M, N, 0, G2, mmg_pro_dev(pro2));
  MMG_CHECK_LAUNCH("bn_bwd_apply2");
  return MMG_OK;
}

extern "C" int mmg_l2norm_fwd(const float* Z, float* out, float* rnorm, int64_t M, int N, float eps, void* stream) {
  MMG_CHECK_ARG(mmg_valid_D(N), "l2norm_fwd: N=%d unsupported (64|128|256)", N);
  MMG_CHECK_ARG(M >= 0, "l2norm_fwd: M < 0");
  if (M == 0) return MMG_OK;
  MMG_CHECK_ARG(Z && out, "l2norm_fwd: null buffer");
  const unsigned nb = (unsigned)((M + 3) / 4);
  hipStream_t st = (hipStream_t)stream;
  if (N == 64) hipLaunchKernelGGL(k_l2norm_fwd<1>, dim3(nb), dim3(256), 0, st, Z, out, rnorm, M, eps);
  else if (N == 128) hipLaunchKernelGGL(k_l2norm_fwd<2>, dim3(nb), dim3(256), 0, st, Z, out, rnorm, M, eps);
  else hipLaunchKernelGGL(k_l2norm_fwd<4>, dim3(nb), dim3(256), 0, st, Z, out, rnorm, M, eps);
  MMG_CHECK_LAUNCH("l2norm_fwd");
  return MMG_OK;
}

extern "C" int mmg_l2norm_bwd(const float* G, const float* out, const float* rnorm, float* dZ, int64_t M, int N, float eps,
                              void* stream) {
  MMG_CHECK_ARG(mmg_valid_D(N), "l2norm_bwd: N=%d unsupported (64|128|256)", N);
  MMG_CHECK_ARG(M >= 0, "l2norm_bwd: M < 0");
  if (M == 0) return MMG_OK;
  MMG_CHECK_ARG(G && out && rnorm && dZ, "l2norm_bwd: null buffer");
  const unsigned nb = (unsigned)((M + 3) / 4);
  hipStream_t st = (hipStream_t)stream;
  if (N == 64) hipLaunchKernelGGL(k_l2norm_bwd<1>, dim3(nb), dim3(256), 0, st, G, out, rnorm, dZ, M, eps);
  else if (N == 128) hipLaunchKernelGGL(k_l2norm_bwd<2>, dim3(nb), dim3(256), 0, st, G, out, rnorm, dZ, M, eps);
  else hipLaunchKernelGGL(k_l2norm_bwd<4>, dim3(nb), dim3(256), 0, st, G, out, rnorm, dZ, M, eps);
  MMG_CHECK_LAUNCH("l2norm_bwd");
  return MMG_OK;
}

extern "C" int mmg_dropout_mask(uint64_t seed, const uint64_t* seed_ptr, uint32_t site, int64_t first_elem, int64_t n_elems,
                                float p, uint8_t* mask, void* stream) {
  MMG_CHECK_ARG(n_elems >= 0 && (mask || n_elems == 0), "dropout_mask: bad args");
  if (n_elems == 0) return MMG_OK;
  hipLaunchKernelGGL(k_dropout_mask, dim3((unsigned)((n_elems + 255) / 256)), dim3(256), 0, (hipStream_t)stream, seed, seed_ptr,
                     site, first_elem, n_elems, p, mask);
  MMG_CHECK_LAUNCH("dropout_mask");
  return MMG_OK;
}

extern "C" size_t mmg_sup_mask_ws_bytes(int64_t n) { return n < 0 ? 0 : (size_t)PL_BLOCKS * 8 + 256; }

extern "C" int mmg_sup_mask_draw(const uint64_t* seed_ptr, uint64_t seed, const int64_t* ids, int64_t n, float fraction,
                                 float* sup, double* count, double* inv_den, void* ws, size_t ws_bytes, void* stream) {
  MMG_CHECK_ARG(n >= 0 && ws && (sup || count || inv_den), "sup_mask_draw: bad args");
  MMG_CHECK_ARG(fraction >= 0.f && fraction <= 1.f, "sup_mask_draw: fraction outside [0, 1]");
  if (ws_bytes < mmg_sup_mask_ws_bytes(n)) { mmg_set_error("sup_mask_draw: workspace too small"); return MMG_E_WS; }
  hipStream_t st = (hipStream_t)stream;
  double* partial = (double*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  int64_t nb = (n + 255) / 256;
  if (nb > PL_BLOCKS) nb = PL_BLOCKS;
  if (nb < 1) nb = 1;
  // mmg_keep keeps with probability 1 - p: an element is supervised with probability `fraction`
  hipLaunchKernelGGL(k_sup_mask_draw, dim3((unsigned)nb), dim3(256), 0, st, seed_ptr, seed, ids, n, 1.0f - fraction, sup, partial);
  hipLaunchKernelGGL(k_sup_mask_count, dim3(1), dim3(64), 0, st, partial, (int)nb, count, inv_den);
  MMG_CHECK_LAUNCH("sup_mask_draw");
  return MMG_OK;
}

extern "C" size_t mmg_pair_loss_ws_bytes(int64_t n) { return n < 0 ? 0 : (size_t)PL_BLOCKS * 8 + 256; }

extern "C" int mmg_pair_loss(const float* pred, const float* y, const float* w, const float* sup, int64_t n, double inv_den,
                             const double* inv_den_ptr, int loss_type, float* dpred, double* loss, void* ws,
                             size_t ws_bytes, void* stream) {
  MMG_CHECK_ARG(n >= 0 && loss && ws, "pair_loss: bad args");
  MMG_CHECK_ARG(loss_type >= 0 && loss_type <= 2, "pair_loss: loss_type must be 0 (mae), 1 (mse) or 2 (huber)");
  MMG_CHECK_ARG(n == 0 || (pred && y), "pair_loss: null buffer");
  if (ws_bytes < mmg_pair_loss_ws_bytes(n)) { mmg_set_error("pair_loss: workspace too small"); return MMG_E_WS; }
  hipStream_t st = (hipStream_t)stream;
  double* partial = (double*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  int64_t nb = (n + 255) / 256;
  if (nb > PL_BLOCKS) nb = PL_BLOCKS;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(k_pair_loss, dim3((unsigned)nb), dim3(256), 0, st, pred, y, w, sup, n, inv_den, inv_den_ptr, loss_type, dpred, partial);
  hipLaunchKernelGGL(k_partial_sum, dim3(1), dim3(256), 0, st, partial, loss, 1, (int)nb);
  MMG_CHECK_LAUNCH("pair_loss");
  return MMG_OK;
}
