// Grouped launches for the vocab side of the path.  The lab / diagnosis / medication tables have 50 .. 200 rows: every
// dense op on them (the lin_l / lin_r of the SAGEConv into a vocab type, the transformed tables the patient-side gather
// reads, their weight and data gradients; call site src/model.py:125-131,256 of the reference) is a few microseconds
// of work behind a launch, and a HeteroConv layer issues dozens of them, independent of each other per relation.  One
// launch here runs a GROUP of such problems (blockIdx.z = problem): at the x100 scale they are 22 % of the step's
// kernel time, at the x1 scale (BASELINE config 2) the step is nothing but launches.
//   mmg_small_fwd_group  : Y[M,N] (+)= X . W^T (+ X2 . W2^T) + bias      fp32 matrix cores (exact fp32 products)
//   mmg_small_wgrad_group: dW[N,K] (+)= dY^T . X ;  dbias (+)= column sums of dY
//   mmg_small_bn_act_group / mmg_small_bn_bwd_group: the per-type BatchNorm -> activation -> dropout of a HeteroConv
//                          layer (src/model.py:258-269) and its backward for ALL small node types in one launch each
//                          (statistics, folded scale / shift, running-statistics update and the apply pass fused:
//                          nine, respectively six, launches per layer before)
#include "common.h"

namespace {

typedef float sf32x16 __attribute__((ext_vector_type(16)));
typedef float sf32x4 __attribute__((ext_vector_type(4)));

struct FwdGroup { mmg_small_fwd_t p[MMG_SMALL_MAX]; int n; };
struct WgradGroup { mmg_small_wgrad_t p[MMG_SMALL_MAX]; int n; };

// One workgroup = one [32 rows x 32 columns] output tile of one problem; the K axis is split over the four waves (and
// the lane halves inside a wave), every operand load issued up front, the four partial tiles summed through LDS in
// fixed order.  Same arithmetic as k_linear_small (gemm.hip).
template <int K>
__global__ __launch_bounds__(256) void k_small_fwd_group(FwdGroup g, int N) {
  const mmg_small_fwd_t& P = g.p[blockIdx.z];
  const int64_t M = P.M;
  const int64_t row0 = (int64_t)blockIdx.y * 32;
  if (row0 >= M) return;
  __shared__ float part[4][16][64];
  const bool acc_out = (P.flags & MMG_LIN_ACCUMULATE) != 0, wkn = (P.flags & MMG_LIN_W_KN) != 0;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, h = lane >> 5, l31 = lane & 31;
  constexpr int KW = K / 8;
  const int kb = wid * (K / 4) + h * KW;
  const int n0 = blockIdx.x * 32;
  const int64_t ar = row0 + l31;
  const int n_terms = P.X2 ? 2 : 1;
  sf32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  sf32x4 a[2][KW / 4], w[2][KW / 4];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (t < n_terms) {
      const float* X = t ? P.X2 : P.X;
      const float* W = t ? P.W2 : P.W;
      const float* xp = X + (size_t)(ar < M ? ar : 0) * K + kb;
      const float* wp = wkn ? W + (size_t)kb * N + n0 + l31 : W + (size_t)(n0 + l31) * K + kb;
#pragma unroll
      for (int q = 0; q < KW / 4; ++q) {
        a[t][q] = *reinterpret_cast<const sf32x4*>(xp + q * 4);
        if (wkn) {
#pragma unroll
          for (int j = 0; j < 4; ++j) w[t][q][j] = wp[(size_t)(q * 4 + j) * N];
        } else {
          w[t][q] = *reinterpret_cast<const sf32x4*>(wp + q * 4);
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (t < n_terms) {
#pragma unroll
      for (int q = 0; q < KW / 4; ++q) {
        if (ar >= M) a[t][q] = sf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][q][j], w[t][q][j], acc, 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) part[wid][i][lane] = acc[i];
  __syncthreads();
  const int col = n0 + l31;
  const float bv = P.bias ? P.bias[col] : 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int i = 4 * wid + e;
    const float v0 = part[0][i][lane] + part[1][i][lane] + part[2][i][lane] + part[3][i][lane];
    const int64_t gr = row0 + (i & 3) + 8 * (i >> 2) + 4 * h;
    if (gr < M) {
      float* dst = P.Y + (size_t)gr * N + col;
      float v = v0 + bv;
      if (acc_out) v += *dst;
      *dst = v;
    }
  }
}

// One workgroup = one [32 x 32] tile of dW of one problem: A = dY^T (lane = output row n, lane half = row parity of the
// contraction), B = X (lane = k column); the M rows are dealt over the four waves, partial tiles summed through LDS in
// fixed order.  The workgroups of k-tile 0 also take the column sums of dY (the bias gradient).
__global__ __launch_bounds__(256) void k_small_wgrad_group(WgradGroup g, int N, int K) {
  const mmg_small_wgrad_t& P = g.p[blockIdx.z];
  const int M = (int)P.M;
  __shared__ float part[4][16][64];
  __shared__ float bpart[4][64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, h = lane >> 5, l31 = lane & 31;
  const int k0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  sf32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float bsum = 0.f;
  // wave w takes rows m = 8 (4 j + w) + 2 s + h, s = 0..3: eight rows per step, all loads of a step issued together
  for (int mb = wid * 8; mb < M; mb += 32) {
    float av[4], bv[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int m = mb + 2 * s + h;
      const bool ok = m < M;
      av[s] = ok ? P.dY[(size_t)m * N + n0 + l31] : 0.f;
      bv[s] = ok ? P.X[(size_t)m * K + k0 + l31] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s], acc, 0, 0, 0);
      bsum += av[s];
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) part[wid][i][lane] = acc[i];
  bpart[wid][lane] = bsum;
  __syncthreads();
  const bool acc_out = P.accumulate != 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int i = 4 * wid + e;
    const float v0 = part[0][i][lane] + part[1][i][lane] + part[2][i][lane] + part[3][i][lane];
    const int n = n0 + (i & 3) + 8 * (i >> 2) + 4 * h;
    float* dst = P.dW + (size_t)n * K + k0 + l31;
    *dst = acc_out ? *dst + v0 : v0;
  }
  if (P.dbias && blockIdx.x == 0 && tid < 32) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) t += bpart[w][tid] + bpart[w][32 + tid];      // fixed order: wave, then row parity
    float* dst = P.dbias + n0 + tid;
    *dst = acc_out ? *dst + t : t;
  }
}

struct BnGroup { mmg_small_bn_t p[MMG_SMALL_MAX]; int n; };

__device__ inline ProDev small_pro(const mmg_small_bn_t& P, const float* sc, const float* sh) {
  ProDev pr;
  pr.scale = sc; pr.shift = sh; pr.relu = P.act; pr.p = P.drop_p;
  pr.inv_keep = P.drop_p > 0.f ? 1.0f / (1.0f - P.drop_p) : 1.0f;
  pr.seed = P.seed; pr.site = P.site; pr.row_offset = P.row_offset; pr.seed_ptr = P.seed_ptr;
  pr.key = 0; pr.thr = 0;
  pr.resolve();
  return pr;
}

// workgroup = 32 columns of one problem; thread = (4 columns, one of 32 row lanes).  Training: fp64 column sums over the
// row lanes (fixed order), the BatchNorm fold exactly as k_bn_finalize forms it (elementwise.hip), then the apply pass
// re-reads Y (L2-resident: <= 4096 rows).
__global__ __launch_bounds__(256) void k_small_bn_act_group(BnGroup g, int N, float momentum, float eps) {
  const mmg_small_bn_t& P = g.p[blockIdx.y];
  const int M = (int)P.M;
  if (M == 0) return;
  __shared__ double red[32][8][8];
  __shared__ double tot[2][32];
  __shared__ __attribute__((aligned(16))) float scs[32], shs[32];
  const int tid = threadIdx.x, c4 = tid & 7, rl = tid >> 3;
  const int c0 = blockIdx.x * 32, c = c0 + c4 * 4;
  const bool has_bn = P.gamma != nullptr;
  if (has_bn && P.training) {
    double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
    for (int r = rl; r < M; r += 32) {
      const sf32x4 v = *reinterpret_cast<const sf32x4*>(P.Y + (size_t)r * N + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) { s0[j] += (double)v[j]; s1[j] += (double)v[j] * (double)v[j]; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[rl][c4][j] = s0[j]; red[rl][c4][4 + j] = s1[j]; }
    __syncthreads();
    if (tid < 64) {
      const int col = tid & 31, which = tid >> 5;
      double t = 0;
      for (int q = 0; q < 32; ++q) t += red[q][col >> 2][which * 4 + (col & 3)];
      tot[which][col] = t;
    }
    __syncthreads();
  }
  if (tid < 32) {
    const int i = c0 + tid;
    float sc = 1.f, sh = 0.f;
    if (has_bn) {
      float mean, var;
      if (P.training) {
        const double m = tot[0][tid] / (double)M;
        double v = tot[1][tid] / (double)M - m * m;
        if (v < 0) v = 0;
        mean = (float)m; var = (float)v;
        if (P.running_mean) {
          const float unb = (float)(v * ((double)M / (double)(M > 1 ? M - 1 : 1)));
          P.running_mean[i] = (1.f - momentum) * P.running_mean[i] + momentum * mean;
          P.running_var[i] = (1.f - momentum) * P.running_var[i] + momentum * unb;
        }
      } else {
        mean = P.running_mean[i]; var = P.running_var[i];
      }
      const float rstd = 1.0f / sqrtf(var + eps);
      sc = P.gamma[i] * rstd;
      sh = (P.beta ? P.beta[i] : 0.f) - mean * sc;
      P.stats_out[i] = sc; P.stats_out[N + i] = sh; P.stats_out[2 * N + i] = mean; P.stats_out[3 * N + i] = rstd;
    }
    scs[tid] = sc; shs[tid] = sh;
  }
  __syncthreads();
  const ProDev pr = small_pro(P, has_bn ? scs : nullptr, shs);
  const sf32x4 sc4 = *reinterpret_cast<const sf32x4*>(&scs[c4 * 4]), sh4 = *reinterpret_cast<const sf32x4*>(&shs[c4 * 4]);
  for (int r = rl; r < M; r += 32) {
    sf32x4 v = *reinterpret_cast<const sf32x4*>(P.Y + (size_t)r * N + c);
    mmg_pro_apply4(pr, v, sc4, sh4, r, c, N);
    *reinterpret_cast<sf32x4*>(P.out + (size_t)r * N + c) = v;
  }
}

struct BnBwdGroup { mmg_small_bn_bwd_t p[MMG_SMALL_MAX]; int n; };

// g' = G * act'(y * scale + shift) * keep / (1 - p);  training: dY = scale * (g' - mean(g') - xhat * mean(g' xhat)),
// d beta = sum g', d gamma = sum g' xhat;  eval: dY = scale * g';  no BatchNorm: dY = g'.
__global__ __launch_bounds__(256) void k_small_bn_bwd_group(BnBwdGroup g, int N) {
  const mmg_small_bn_bwd_t& P = g.p[blockIdx.y];
  const int M = (int)P.M;
  if (M == 0) return;
  __shared__ double red[32][8][8];
  __shared__ float a0s[32], a1s[32];
  const int tid = threadIdx.x, c4 = tid & 7, rl = tid >> 3;
  const int c0 = blockIdx.x * 32, c = c0 + c4 * 4;
  const bool has_bn = P.scale != nullptr;
  const sf32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
  sf32x4 sc = one, sh = zero, mu = zero, rs = one;
  if (has_bn) {
    sc = *reinterpret_cast<const sf32x4*>(P.scale + c); sh = *reinterpret_cast<const sf32x4*>(P.shift + c);
    mu = *reinterpret_cast<const sf32x4*>(P.mean + c); rs = *reinterpret_cast<const sf32x4*>(P.rstd + c);
  }
  mmg_small_bn_t fwd{};                       // (reuses the forward's prologue builder for the dropout fields)
  fwd.act = P.act; fwd.drop_p = P.drop_p; fwd.seed = P.seed; fwd.site = P.site; fwd.row_offset = P.row_offset;
  fwd.seed_ptr = P.seed_ptr;
  const ProDev pr = small_pro(fwd, nullptr, nullptr);
  auto gprime = [&](int r) {
    sf32x4 gv = *reinterpret_cast<const sf32x4*>(P.G + (size_t)r * N + c);
    const sf32x4 y = *reinterpret_cast<const sf32x4*>(P.Y + (size_t)r * N + c);
    if (P.act) {
#pragma unroll
      for (int j = 0; j < 4; ++j) gv[j] *= mmg_act_grad(P.act, has_bn ? fmaf(y[j], sc[j], sh[j]) : y[j]);
    }
    if (pr.p > 0.f) mmg_drop4(gv, pr.key, (uint64_t)(pr.row_offset + r) * (uint64_t)N + (uint64_t)c, pr.thr, pr.inv_keep);
    return gv;
  };
  const bool stats = has_bn && P.training;
  if (stats) {
    double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
    for (int r = rl; r < M; r += 32) {
      const sf32x4 gv = gprime(r);
      const sf32x4 y = *reinterpret_cast<const sf32x4*>(P.Y + (size_t)r * N + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s0[j] += (double)gv[j];
        s1[j] += (double)gv[j] * (double)((y[j] - mu[j]) * rs[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[rl][c4][j] = s0[j]; red[rl][c4][4 + j] = s1[j]; }
    __syncthreads();
    if (tid < 64) {
      const int col = tid & 31, which = tid >> 5;
      double t = 0;
      for (int q = 0; q < 32; ++q) t += red[q][col >> 2][which * 4 + (col & 3)];
      if (which == 0) { a0s[col] = (float)(t / (double)M); if (P.dbeta) P.dbeta[c0 + col] = (float)t; }
      else { a1s[col] = (float)(t / (double)M); if (P.dgamma) P.dgamma[c0 + col] = (float)t; }
    }
    __syncthreads();
  } else if (has_bn && (P.dbeta || P.dgamma)) {
    // eval mode: d beta / d gamma are still the column sums (no mean subtraction in dY)
    double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
    for (int r = rl; r < M; r += 32) {
      const sf32x4 gv = gprime(r);
      const sf32x4 y = *reinterpret_cast<const sf32x4*>(P.Y + (size_t)r * N + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) { s0[j] += (double)gv[j]; s1[j] += (double)gv[j] * (double)((y[j] - mu[j]) * rs[j]); }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[rl][c4][j] = s0[j]; red[rl][c4][4 + j] = s1[j]; }
    __syncthreads();
    if (tid < 64) {
      const int col = tid & 31, which = tid >> 5;
      double t = 0;
      for (int q = 0; q < 32; ++q) t += red[q][col >> 2][which * 4 + (col & 3)];
      if (which == 0) { if (P.dbeta) P.dbeta[c0 + col] = (float)t; }
      else if (P.dgamma) P.dgamma[c0 + col] = (float)t;
    }
    __syncthreads();
  }
  sf32x4 a0 = zero, a1 = zero;
  if (stats) { a0 = *reinterpret_cast<const sf32x4*>(&a0s[c4 * 4]); a1 = *reinterpret_cast<const sf32x4*>(&a1s[c4 * 4]); }
  for (int r = rl; r < M; r += 32) {
    const sf32x4 gv = gprime(r);
    sf32x4 o = gv;
    if (has_bn) {
      const sf32x4 y = *reinterpret_cast<const sf32x4*>(P.Y + (size_t)r * N + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = sc[j] * (gv[j] - a0[j] - ((y[j] - mu[j]) * rs[j]) * a1[j]);
    }
    *reinterpret_cast<sf32x4*>(P.dY + (size_t)r * N + c) = o;
  }
}

}  // namespace

extern "C" int mmg_small_fwd_group(const mmg_small_fwd_t* probs, int n_probs, int N, int K, void* stream) {
  MMG_CHECK_ARG(probs && n_probs >= 1 && n_probs <= MMG_SMALL_MAX, "small_fwd_group: 1..%d problems", MMG_SMALL_MAX);
  MMG_CHECK_ARG(K == 64 || K == 128 || K == 256, "small_fwd_group: K=%d unsupported (64|128|256)", K);
  MMG_CHECK_ARG(N > 0 && N % 32 == 0, "small_fwd_group: N=%d must be a multiple of 32", N);
  FwdGroup g;
  g.n = n_probs;
  int64_t mmax = 0;
  for (int i = 0; i < n_probs; ++i) {
    const mmg_small_fwd_t& p = probs[i];
    MMG_CHECK_ARG(p.M >= 0 && p.M <= 4096, "small_fwd_group: problem %d has M=%lld (0..4096)", i, (long long)p.M);
    MMG_CHECK_ARG(p.M == 0 || (p.X && p.W && p.Y), "small_fwd_group: problem %d has a null buffer", i);
    MMG_CHECK_ARG((p.X2 == nullptr) == (p.W2 == nullptr), "small_fwd_group: X2 and W2 go together");
    MMG_CHECK_ARG((p.flags & ~(MMG_LIN_ACCUMULATE | MMG_LIN_W_KN)) == 0, "small_fwd_group: unknown flag bits");
    g.p[i] = p;
    if (p.M > mmax) mmax = p.M;
  }
  if (mmax == 0) return MMG_OK;
  dim3 grid((unsigned)(N / 32), (unsigned)((mmax + 31) / 32), (unsigned)n_probs);
  hipStream_t st = (hipStream_t)stream;
  if (K == 64) hipLaunchKernelGGL(k_small_fwd_group<64>, grid, dim3(256), 0, st, g, N);
  else if (K == 128) hipLaunchKernelGGL(k_small_fwd_group<128>, grid, dim3(256), 0, st, g, N);
  else hipLaunchKernelGGL(k_small_fwd_group<256>, grid, dim3(256), 0, st, g, N);
  MMG_CHECK_LAUNCH("small_fwd_group");
  return MMG_OK;
}

extern "C" int mmg_small_wgrad_group(const mmg_small_wgrad_t* probs, int n_probs, int N, int K, void* stream) {
  MMG_CHECK_ARG(probs && n_probs >= 1 && n_probs <= MMG_SMALL_MAX, "small_wgrad_group: 1..%d problems", MMG_SMALL_MAX);
  MMG_CHECK_ARG(N > 0 && N % 32 == 0 && K > 0 && K % 32 == 0, "small_wgrad_group: N=%d K=%d must be multiples of 32", N, K);
  WgradGroup g;
  g.n = n_probs;
  for (int i = 0; i < n_probs; ++i) {
    const mmg_small_wgrad_t& p = probs[i];
    MMG_CHECK_ARG(p.M >= 0 && p.M <= 4096, "small_wgrad_group: problem %d has M=%lld (0..4096)", i, (long long)p.M);
    MMG_CHECK_ARG(p.dW && (p.M == 0 || (p.dY && p.X)), "small_wgrad_group: problem %d has a null buffer", i);
    g.p[i] = p;
  }
  dim3 grid((unsigned)(K / 32), (unsigned)(N / 32), (unsigned)n_probs);
  hipLaunchKernelGGL(k_small_wgrad_group, grid, dim3(256), 0, (hipStream_t)stream, g, N, K);
  MMG_CHECK_LAUNCH("small_wgrad_group");
  return MMG_OK;
}

extern "C" int mmg_small_bn_act_group(const mmg_small_bn_t* probs, int n_probs, int N, float momentum, float eps,
                                      void* stream) {
  MMG_CHECK_ARG(probs && n_probs >= 1 && n_probs <= MMG_SMALL_MAX, "small_bn_act_group: 1..%d problems", MMG_SMALL_MAX);
  MMG_CHECK_ARG(N > 0 && N % 32 == 0, "small_bn_act_group: N=%d must be a multiple of 32", N);
  BnGroup g;
  g.n = n_probs;
  for (int i = 0; i < n_probs; ++i) {
    const mmg_small_bn_t& p = probs[i];
    MMG_CHECK_ARG(p.M >= 0 && p.M <= 4096, "small_bn_act_group: problem %d has M=%lld (0..4096)", i, (long long)p.M);
    MMG_CHECK_ARG(p.M == 0 || (p.Y && p.out), "small_bn_act_group: problem %d has a null buffer", i);
    MMG_CHECK_ARG(!p.gamma || (p.stats_out && p.running_mean && p.running_var), "small_bn_act_group: BatchNorm buffers");
    MMG_CHECK_ARG(!(p.gamma && p.training) || p.M > 1,
                  "Expected more than 1 value per channel when training (problem %d has %lld rows)", i, (long long)p.M);
    MMG_CHECK_ARG(p.drop_p >= 0.f && p.drop_p < 1.f && p.act >= 0 && p.act <= 3, "small_bn_act_group: bad activation / p");
    g.p[i] = p;
  }
  hipLaunchKernelGGL(k_small_bn_act_group, dim3((unsigned)(N / 32), (unsigned)n_probs), dim3(256), 0, (hipStream_t)stream, g, N,
                     momentum, eps);
  MMG_CHECK_LAUNCH("small_bn_act_group");
  return MMG_OK;
}

extern "C" int mmg_small_bn_bwd_group(const mmg_small_bn_bwd_t* probs, int n_probs, int N, void* stream) {
  MMG_CHECK_ARG(probs && n_probs >= 1 && n_probs <= MMG_SMALL_MAX, "small_bn_bwd_group: 1..%d problems", MMG_SMALL_MAX);
  MMG_CHECK_ARG(N > 0 && N % 32 == 0, "small_bn_bwd_group: N=%d must be a multiple of 32", N);
  BnBwdGroup g;
  g.n = n_probs;
  for (int i = 0; i < n_probs; ++i) {
    const mmg_small_bn_bwd_t& p = probs[i];
    MMG_CHECK_ARG(p.M >= 0 && p.M <= 4096, "small_bn_bwd_group: problem %d has M=%lld (0..4096)", i, (long long)p.M);
    MMG_CHECK_ARG(p.M == 0 || (p.G && p.Y && p.dY), "small_bn_bwd_group: problem %d has a null buffer", i);
    MMG_CHECK_ARG(!p.scale || (p.shift && p.mean && p.rstd), "small_bn_bwd_group: folded BatchNorm vectors go together");
    g.p[i] = p;
  }
  hipLaunchKernelGGL(k_small_bn_bwd_group, dim3((unsigned)(N / 32), (unsigned)n_probs), dim3(256), 0, (hipStream_t)stream, g, N);
  MMG_CHECK_LAUNCH("small_bn_bwd_group");
  return MMG_OK;
}
