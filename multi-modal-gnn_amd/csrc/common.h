// Shared device/host helpers for libmmgnn (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>
#include "../../include/mmgnn.h"

#define WAVE 64

void mmg_set_error(const char* fmt, ...);

#define MMG_CHECK_ARG(cond, ...)                   \
  do {                                             \
    if (!(cond)) {                                 \
      mmg_set_error(__VA_ARGS__);                  \
      return MMG_E_ARG;                            \
    }                                              \
  } while (0)

#define MMG_CHECK_LAUNCH(what)                                                    \
  do {                                                                            \
    hipError_t e__ = hipGetLastError();                                           \
    if (e__ != hipSuccess) {                                                      \
      mmg_set_error("%s: %s", what, hipGetErrorString(e__));                      \
      return MMG_E_LAUNCH;                                                        \
    }                                                                             \
  } while (0)

// every HIP runtime call that returns an error code goes through this
#define MMG_CHECK_HIP(expr, what)                                                 \
  do {                                                                            \
    hipError_t e__ = (expr);                                                      \
    if (e__ != hipSuccess) {                                                      \
      mmg_set_error("%s: %s", what, hipGetErrorString(e__));                      \
      return MMG_E_LAUNCH;                                                        \
    }                                                                             \
  } while (0)

// Dynamic-LDS limit of a kernel, raised once per (kernel, DEVICE): the attribute belongs to the device's code object, so
// a process that drives a second GPU has to raise it there too.  Only successes are remembered (a transient failure is
// retried by the next launch).
template <auto Kernel, int Bytes>
struct MmgMaxLds {
  static hipError_t set() {
    static std::atomic<uint64_t> done{0};
    int dev = 0;
    hipError_t rc = hipGetDevice(&dev);
    if (rc != hipSuccess) return rc;
    if (dev >= 0 && dev < 64 && ((done.load(std::memory_order_relaxed) >> dev) & 1ull)) return hipSuccess;
    rc = hipFuncSetAttribute((const void*)Kernel, hipFuncAttributeMaxDynamicSharedMemorySize, Bytes);
    if (rc == hipSuccess && dev >= 0 && dev < 64) done.fetch_or(1ull << dev, std::memory_order_relaxed);
    return rc;
  }
};

// Measurement hook (mmg_probe_arm / mmg_probe_read, include/mmgnn.h): while armed, a launch made through MMG_LAUNCH
// carries a HIP start / stop event pair on the kernel itself (hipExtLaunchKernelGGL: the kernel's own begin / end
// timestamps on its stream).  Process-wide and mutex-guarded (api.hip); unarmed cost: one relaxed atomic load.
// (the record also keeps what names the INSTANTIATED kernel: the kernel expression as written at the launch site and the
// __PRETTY_FUNCTION__ of the launcher, whose "[K = 128, WN = 4, ...]" suffix binds the template parameters it mentions)
bool mmg_probe_take(int tag, int64_t M, int N, int K, int flags, const char* kernel_text, const char* launcher,
                    hipEvent_t* e0, hipEvent_t* e1);
#define MMG_LAUNCH(tag, pM, pN, pK, pflags, kernel, grid, block, lds, st, ...)                         \
  do {                                                                                                \
    hipEvent_t e0__, e1__;                                                                            \
    if (mmg_probe_take(tag, pM, pN, pK, pflags, #kernel, __PRETTY_FUNCTION__, &e0__, &e1__))          \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, st, e0__, e1__, 0, __VA_ARGS__);                \
    else                                                                                              \
      hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                                  \
  } while (0)

static inline int mmg_valid_D(int D) { return D == 64 || D == 128 || D == 256; }

// Zero-fill on a stream, as a KERNEL -- never hipMemsetAsync.  Recorded into a hipGraph by stream capture, the memset node
// of this ROCm (the HIP runtime PyTorch 2.10 + rocm7.0 ships) replays, in about every second graph of a process, with a
// fill pattern that is not the recorded zero: 16-byte groups {n_dwords, 1, 0, 0} -- or whatever else lies where the pattern
// is fetched from -- land in the buffer (profiles/probes/hipgraph_memset_node.py).  As denormal floats they are invisible in
// a sum; as a large value in a buffer the step assumes zeroed (predictions outside the supervised pair lists, gradient
// accumulators) they surfaced as a non-finite loss once per few hundred captured steps.
static __global__ __launch_bounds__(256) void mmg_k_zero(unsigned char* __restrict__ p, size_t bytes) {
  const size_t mis = (size_t)((16u - (unsigned)((uintptr_t)p & 15u)) & 15u);
  const size_t head = mis < bytes ? mis : bytes;
  const size_t n16 = (bytes - head) >> 4, tail0 = head + (n16 << 4);
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)gridDim.x * 256;
  uint4* q = reinterpret_cast<uint4*>(p + head);
  for (size_t i = tid; i < n16; i += nth) q[i] = uint4{0u, 0u, 0u, 0u};
  if (tid < head) p[tid] = 0;
  if (tid < bytes - tail0) p[tail0 + tid] = 0;          // (fewer than 16 bytes)
}
static inline hipError_t mmg_zero_async(void* ptr, size_t bytes, hipStream_t st) {
  if (bytes == 0) return hipSuccess;
  size_t g = ((bytes >> 4) + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(mmg_k_zero, dim3((unsigned)g), dim3(256), 0, st, (unsigned char*)ptr, bytes);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// Counter-based dropout RNG: keep(seed, site, element) -- stateless, so the backward pass
// regenerates the forward mask instead of storing it.  Two rounds of a murmur3-style
// finaliser over (seed, site, 64-bit element index).
// ------------------------------------------------------------------------------------
__host__ __device__ static inline uint32_t mmg_mix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}
// kernel-invariant part: (seed, site) -> 32-bit key (the compiler hoists it out of every loop)
__host__ __device__ static inline uint32_t mmg_rng_key(uint64_t seed, uint32_t site) {
  uint32_t h = mmg_mix32((uint32_t)seed ^ 0x9e3779b9u);
  return mmg_mix32(h ^ (uint32_t)(seed >> 32) ^ (site * 0x632be5abu));
}
// 64 random bits for one aligned GROUP of four consecutive elements (grp = elem >> 2): one full mix for the first
// word, one multiply-xorshift step for the second (3 integer multiplies per 4 elements -- v_mul_lo_u32 is a
// quarter-rate instruction and was the bottleneck of the pair-head kernels).
__host__ __device__ static inline void mmg_rng_group(uint32_t key, uint64_t grp, uint32_t* w0, uint32_t* w1) {
  const uint32_t hi = (uint32_t)(grp >> 32);
  const uint32_t h = mmg_mix32(key ^ (uint32_t)grp ^ ((hi << 16) | (hi >> 16)) ^ hi);
  uint32_t t = (h ^ (h >> 15)) * 0x2c1b3c6du;
  t ^= t >> 13;
  *w0 = h; *w1 = t;
}
// 16 random bits of element `sub` (0..3) of a group
__host__ __device__ static inline uint32_t mmg_rng_field(uint32_t w0, uint32_t w1, uint32_t sub) {
  const uint32_t word = (sub & 2u) ? w1 : w0;
  return (word >> ((sub & 1u) * 16u)) & 0xFFFFu;
}
__host__ __device__ static inline uint32_t mmg_keep_threshold(float p) { return (uint32_t)(p * 65536.0f); }
// keep with probability 1-p.  One hash serves FOUR consecutive elements (16 random bits each, p quantised
// to 1/65536): callers walk elements in aligned groups of 4, so the compiler shares the hash across the group.
// element j (compile-time 0..3) of a group kept?  Written so that each use is ONE 16-bit sub-dword compare.
#define MMG_KEPT(w0, w1, j, thr) ((((j) & 2 ? (w1) : (w0)) >> (((j) & 1) * 16) & 0xFFFFu) >= (thr))
// dropout of the 4 consecutive elements e0 .. e0+3 (e0 % 4 == 0) held in v[0..3]: ONE hash, 4 compares, 4 selects
template <class V4>
__host__ __device__ static inline void mmg_drop4(V4& v, uint32_t key, uint64_t e0, uint32_t thr, float inv_keep) {
  uint32_t w0, w1;
  mmg_rng_group(key, e0 >> 2, &w0, &w1);
  v[0] = MMG_KEPT(w0, w1, 0, thr) ? v[0] * inv_keep : 0.f;
  v[1] = MMG_KEPT(w0, w1, 1, thr) ? v[1] * inv_keep : 0.f;
  v[2] = MMG_KEPT(w0, w1, 2, thr) ? v[2] * inv_keep : 0.f;
  v[3] = MMG_KEPT(w0, w1, 3, thr) ? v[3] * inv_keep : 0.f;
}
__host__ __device__ static inline bool mmg_keep(uint64_t seed, uint32_t site, uint64_t elem, float p) {
  uint32_t w0, w1;
  mmg_rng_group(mmg_rng_key(seed, site), elem >> 2, &w0, &w1);
  return mmg_rng_field(w0, w1, (uint32_t)elem & 3u) >= mmg_keep_threshold(p);
}

// Dropout keep-fields of a 32 x 32 tile held in the MFMA C layout (lane = column col0 + (lane & 31), register r = tile row
// (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) of a row-major [*, ncols] tensor: element (row, col) belongs to the RNG group
// (row * ncols + col) >> 2, shared by the four lanes of a quad.  For the four registers 4g .. 4g+3 quad lane j hashes the
// row of register 4g + j (ONE hash per lane instead of four) and the words travel inside the quad by DPP:
//   mmg_c_layout_hash(key, first row of the tile, ncols, col, lane, g, &w0, &w1);
//   field of register 4g + j  =  mmg_rng_field(mmg_quad_bcast(w0, j), mmg_quad_bcast(w1, j), col & 3)
__device__ __forceinline__ void mmg_c_layout_hash(uint32_t key, uint64_t row_first, uint32_t ncols, int col, int lane, int g,
                                                  uint32_t* w0, uint32_t* w1) {
  const uint64_t row = row_first + (uint64_t)((lane & 3) + 8 * g + 4 * (lane >> 5));
  mmg_rng_group(key, (row * (uint64_t)ncols + (uint64_t)col) >> 2, w0, w1);
}
__device__ __forceinline__ uint32_t mmg_quad_bcast(uint32_t v, int j) {     // j must fold to a constant (unrolled loops)
  switch (j & 3) {
    case 0: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x00, 0xF, 0xF, true);   // quad_perm: every lane reads quad lane j
    case 1: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x55, 0xF, 0xF, true);
    case 2: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xAA, 0xF, 0xF, true);
    default: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xFF, 0xF, 0xF, true);
  }
}

// activation codes of mmg_prologue_t::relu (include/mmgnn.h): the reference's HeteroRGCN accepts relu, elu and leaky_relu
// (src/model.py:145-152) for its conv layers.  act'(v) is taken at the PRE-activation value v, as autograd does.
__device__ static inline float mmg_act(int code, float v) {
  if (code == MMG_ACT_RELU) return fmaxf(v, 0.f);
  if (code == MMG_ACT_LEAKY_RELU) return v > 0.f ? v : 0.01f * v;
  if (code == MMG_ACT_ELU) return v > 0.f ? v : expm1f(v);
  return v;
}
__device__ static inline float mmg_act_grad(int code, float v) {
  if (code == MMG_ACT_RELU) return v > 0.f ? 1.f : 0.f;
  if (code == MMG_ACT_LEAKY_RELU) return v > 0.f ? 1.f : 0.01f;
  if (code == MMG_ACT_ELU) return v > 0.f ? 1.f : expf(v);
  return 1.f;
}

// folded prologue: dropout(act(x*scale+shift)); returns the transformed value
struct ProDev {
  const float* scale; const float* shift; int relu; float p; float inv_keep; uint64_t seed; uint32_t site;
  int64_t row_offset; const uint64_t* seed_ptr;
  uint32_t key, thr;       // derived at kernel entry: RNG key of (seed, site), keep threshold of p
  // resolve a device-resident seed (hipGraph replays) and derive the RNG constants; call once at kernel entry
  __device__ inline void resolve() {
    if (seed_ptr) seed = *seed_ptr;
    key = mmg_rng_key(seed, site);
    thr = mmg_keep_threshold(p);
  }
};
static inline ProDev mmg_pro_dev(const mmg_prologue_t* pro) {
  ProDev d;
  if (pro) {
    d.scale = pro->scale; d.shift = pro->shift; d.relu = pro->relu; d.p = pro->drop_p;
    d.inv_keep = pro->drop_p > 0.f ? 1.0f / (1.0f - pro->drop_p) : 1.0f;
    d.seed = pro->seed; d.site = pro->site; d.row_offset = pro->row_offset; d.seed_ptr = pro->seed_ptr;
  } else {
    d.scale = nullptr; d.shift = nullptr; d.relu = 0; d.p = 0.f; d.inv_keep = 1.f; d.seed = 0; d.site = 0;
    d.row_offset = 0; d.seed_ptr = nullptr;
  }
  d.key = 0; d.thr = 0;
  return d;
}
// four consecutive columns k0 .. k0+3 (k0 % 4 == 0, K % 4 == 0) of one row: the dropout hash is computed once
template <class V4>
__device__ static inline void mmg_pro_apply4(const ProDev& pr, V4& v, const V4& sc, const V4& sh, int64_t row, int k0,
                                             int K) {
  if (pr.scale) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = fmaf(v[j], sc[j], sh[j]);
  }
  if (pr.relu == MMG_ACT_RELU) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
  } else if (pr.relu) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = mmg_act(pr.relu, v[j]);
  }
  if (pr.p > 0.f) mmg_drop4(v, pr.key, (uint64_t)(pr.row_offset + row) * (uint64_t)K + (uint64_t)k0, pr.thr, pr.inv_keep);
}
__device__ static inline float mmg_pro_apply(const ProDev& pr, float x, float sc, float sh, int64_t row,
                                             int k, int K) {
  float v = pr.scale ? fmaf(x, sc, sh) : x;
  if (pr.relu) v = mmg_act(pr.relu, v);
  if (pr.p > 0.f) {
    uint64_t e = (uint64_t)(pr.row_offset + row) * (uint64_t)K + (uint64_t)k;
    v = mmg_keep(pr.seed, pr.site, e, pr.p) ? v * pr.inv_keep : 0.f;
  }
  return v;
}

// ---- mmg_next_bn_t: the column statistics of the BatchNorm backward that CONSUMES a kernel's output (the two sums of
// mmg_bn_bwd_stats), taken from the 32 x 32 output tile while it sits in the MFMA C layout.  The separate statistics pass
// read the [M, N] gradient and the [M, N] pre-BatchNorm activation again (k_col_reduce<1>: 34-55 us four times a step at
// the x100 shape); here the gradient never leaves the registers and only Y is read.  Sixteen rows are summed in fp32,
// tiles in fp64 (as the forward statistics of k_linear_fwd_x6 are), workgroups in a fixed order by mmg_partial_sum.
struct NextBnDev { const float* Y; const float* mean; const float* rstd; ProDev pr; };
struct NextBnCol { float sc, sh, mu, rs; bool relu, drop; };
static inline NextBnDev next_bn_none() {
  NextBnDev d;
  d.Y = nullptr; d.mean = nullptr; d.rstd = nullptr; d.pr = mmg_pro_dev(nullptr);
  return d;
}
__device__ __forceinline__ NextBnCol next_bn_col(NextBnDev& nb, int col) {
  nb.pr.resolve();
  NextBnCol c;
  c.sc = nb.pr.scale ? nb.pr.scale[col] : 1.f;
  c.sh = nb.pr.scale ? nb.pr.shift[col] : 0.f;
  c.mu = nb.mean[col]; c.rs = nb.rstd[col];
  c.relu = nb.pr.relu == MMG_ACT_RELU; c.drop = nb.pr.p > 0.f;
  return c;
}
__device__ __forceinline__ void next_bn_tile(const NextBnDev& nb, const NextBnCol& cc, const float* v, const float* yv, int rows,
                                             int64_t row0, int N, int col, int lane, double& cs1, double& cs2) {
  const int h4 = 4 * (lane >> 5);
  const uint32_t sub = (uint32_t)col & 3u;
  float t1 = 0.f, t2 = 0.f;
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {            // registers 4 g4 .. 4 g4 + 3: tile rows 8 g4 + 4 h + 0..3
    uint32_t w0 = 0u, w1 = 0u;
    if (cc.drop) mmg_c_layout_hash(nb.pr.key, (uint64_t)(nb.pr.row_offset + row0), (uint32_t)N, col, lane, g4, &w0, &w1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = 4 * g4 + j, r = j + 8 * g4 + h4;
      const float y = yv[i];
      float g = v[i];
      const float act = fmaf(y, cc.sc, cc.sh);
      if (cc.relu && !(act > 0.f)) g = 0.f;
      if (cc.drop) g = mmg_rng_field(mmg_quad_bcast(w0, j), mmg_quad_bcast(w1, j), sub) >= nb.pr.thr ? g * nb.pr.inv_keep : 0.f;
      g = r < rows ? g : 0.f;
      const float xh = (y - cc.mu) * cc.rs;
      t1 += g; t2 = fmaf(g, xh, t2);
    }
  }
  cs1 += (double)t1; cs2 += (double)t2;
}

__device__ static inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ static inline double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ------------------------------------------------------------------------------------
// Deterministic sum over partial slabs: out[i] = epi(sum_s slab[s][i]).  256 threads =
// 16 float4 elements x 16 split groups; every thread keeps 8 independent 16-B loads in flight,
// the 16 group partials are combined through LDS in fixed order.  Grid = ceil(n4 / 16).
// ------------------------------------------------------------------------------------
typedef float mmg_f4 __attribute__((ext_vector_type(4)));
template <class Epi>
__global__ __launch_bounds__(256) void mmg_k_reduce_slabs(const float* __restrict__ slab, int64_t n4, int n_split,
                                                          Epi epi) {
  __shared__ mmg_f4 part[16][16];
  const int e = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int64_t i4 = (int64_t)blockIdx.x * 16 + e;
  mmg_f4 acc = {0.f, 0.f, 0.f, 0.f};
  if (i4 < n4) {
    const mmg_f4* base = reinterpret_cast<const mmg_f4*>(slab) + i4;
    int s = g;
    for (; s + 7 * 16 < n_split; s += 8 * 16) {
      mmg_f4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = base[(size_t)(s + u * 16) * n4];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; s < n_split; s += 16) acc += base[(size_t)s * n4];
  }
  part[g][e] = acc;
  __syncthreads();
  if (g == 0 && i4 < n4) {
    mmg_f4 t = part[0][e];
#pragma unroll
    for (int q = 1; q < 16; ++q) t += part[q][e];
    epi(i4, t);
  }
}
