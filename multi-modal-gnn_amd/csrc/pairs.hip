// Degree-gated edge regression heads (src/model.py:305-333 and EdgeRegressionHead :342-396 of the
// reference).  The first Linear(2D,64) is applied per NODE by mmg_linear_fwd (A = x_P.W1[:, :D]^T,
// B = x_lab.W1[:, D:]^T + b1), so the per-pair work is a 64-wide gather-add + a 64x32 and a 32x1 layer.
//
// One launch serves ONE head over a compacted, patient-sorted list of pair positions (mmg_pair_select).
// k_pair_fwd_mfma / k_pair_bwd_duo: one wave (forward) / a front and a back wave (backward) per 32-pair tile on the
// matrix cores (the 64x32 layer as the exact six-term bf16 split in the forward; four chained products in the backward,
// which recomputes the forward -- nothing per pair is stored unless the caller asks for it, mmg_pair_saved_t), node rows
// and indices through a software pipeline of unconditional buffer loads.
// k_pair_bwd (thread-per-pair on the vector ALUs, LDS accumulators for dB) remains for lab vocabularies > 128 rows.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int PT = 256;          // pairs per workgroup tile == threads
constexpr uint32_t SITE_H1 = 64, SITE_H2 = 65;

struct HeadDev { const float *A, *B, *W2, *b2, *W3, *b3; };
struct HeadGradDev { float *dA, *dB, *dW2, *db2, *dW3, *db3; };

__device__ inline void load_row64(const float* __restrict__ p, float* dst) {
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p + q * 4);
    dst[q * 4 + 0] = v[0]; dst[q * 4 + 1] = v[1]; dst[q * 4 + 2] = v[2]; dst[q * 4 + 3] = v[3];
  }
}

// h1 = dropout(relu(A[pi] + B[li]))  (post-dropout values, kept in registers)
__device__ inline void head_layer1(const HeadDev& H, int p_i, int l_i, uint64_t pid, float drop_p, float inv_keep,
                                   uint64_t seed, float* h1) {
  float b[64];
  load_row64(H.A + (size_t)p_i * 64, h1);
  load_row64(H.B + (size_t)l_i * 64, b);
#pragma unroll
  for (int j = 0; j < 64; ++j) {
    float v = fmaxf(h1[j] + b[j], 0.f);
    if (drop_p > 0.f) v = mmg_keep(seed, SITE_H1, pid * 64ull + j, drop_p) ? v * inv_keep : 0.f;
    h1[j] = v;
  }
}
// pre-activation of unit i of layer 2 (weights broadcast from LDS)
__device__ inline float head_unit2(const float* W2s, const float* b2s, const float* h1, int i) {
  float s = b2s[i];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const f32x4 w = *reinterpret_cast<const f32x4*>(W2s + i * 64 + q * 4);
    s = fmaf(w[0], h1[q * 4 + 0], s); s = fmaf(w[1], h1[q * 4 + 1], s);
    s = fmaf(w[2], h1[q * 4 + 2], s); s = fmaf(w[3], h1[q * 4 + 3], s);
  }
  return s;
}

// ---------------------------------------------------------------------------- backward
// LDS layout (floats).  T1t/T2t are TRANSPOSED tiles [unit][pair] with row stride LDT so that
// thread-per-pair writes are conflict-free and the dW2 pass reads 4 pairs per ds_read_b128.
constexpr int LDT = PT + 4;
constexpr int LD4 = 65;                          // pair-major dh1 tile stride (aliases T1t)
constexpr int OFF_W2 = 0;                        // [32][64]
constexpr int OFF_B2 = OFF_W2 + 2048;            // [32]
constexpr int OFF_W3 = OFF_B2 + 32;              // [32]
constexpr int OFF_T1 = OFF_W3 + 32;              // T1t [64][LDT]  (>= PT*LD4 floats)
constexpr int OFF_T2 = OFF_T1 + 64 * LDT;        // T2t [32][LDT]
constexpr int OFF_PI = OFF_T2 + 32 * LDT;        // int [PT] patient id or -1
constexpr int OFF_LI = OFF_PI + PT;              // int [PT]
constexpr int OFF_RED = OFF_LI + PT;             // [4][68]: per-wave dW3[32] | db2[32] | db3
constexpr int OFF_DB = OFF_RED + 4 * 68;         // dB accumulators [n_labs][64] (when they fit)
static_assert(64 * LDT >= PT * LD4, "dh1 tile must fit in the T1t region");
constexpr size_t BWD_LDS_FIXED = (size_t)OFF_DB * 4;
constexpr size_t BWD_LDS_MAX = 160 * 1024;

__global__ __launch_bounds__(PT) void k_pair_bwd(HeadDev H, HeadGradDev Gd, const int32_t* __restrict__ pi,
                                                 const int32_t* __restrict__ li, const int32_t* __restrict__ deg, int thr,
                                                 int want_low, int64_t n, int64_t n_total, int n_pat, int n_labs, int lds_db,
                                                 float drop_p, uint64_t seed, const uint64_t* __restrict__ seed_ptr,
                                                 const int64_t* __restrict__ pair_id,
                                                 const float* __restrict__ dpred, const int32_t* __restrict__ sel,
                                                 const int32_t* __restrict__ n_sel, const int64_t* __restrict__ io) {
  if (seed_ptr) seed = *seed_ptr;
  if (sel) { const int64_t nl = *n_sel; n = nl < 0 ? 0 : (nl < n ? nl : n); }     // never beyond the launch bound
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* W2s = sm + OFF_W2; float* b2s = sm + OFF_B2; float* W3s = sm + OFF_W3;
  float* T1t = sm + OFF_T1; float* T2t = sm + OFF_T2;
  int* PIs = reinterpret_cast<int*>(sm + OFF_PI); int* LIs = reinterpret_cast<int*>(sm + OFF_LI);
  float* RED = sm + OFF_RED; float* DBs = sm + OFF_DB;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int i = tid; i < 2048; i += PT) W2s[i] = H.W2[i];
  if (tid < 32) { b2s[tid] = H.b2[tid]; W3s[tid] = H.W3[tid]; }
  for (int i = tid; i < 4 * 68; i += PT) RED[i] = 0.f;
  if (lds_db) for (int i = tid; i < n_labs * 64; i += PT) DBs[i] = 0.f;
  __syncthreads();
  const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;

  const int i0 = tid >> 3, j0 = (tid & 7) * 8;     // this thread owns dW2[i0][j0..j0+7]
  float w2acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  const int64_t n_tiles = (n + PT - 1) / PT;
  for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    int64_t k = t * PT + tid;
    int p_i = -1, l_i = 0;
    if (k < n) {
      if (sel) k = sel[k];
      if (k >= 0 && k < n_total) {                   // (an entry outside the pair arrays is not a pair)
        const int pp = pi[k], ll = li[k];
        if ((unsigned)pp < (unsigned)n_pat && (unsigned)ll < (unsigned)n_labs && ((int)(deg[pp] < thr)) == want_low) {
          p_i = pp; l_i = ll;
        }
      }
    }
    const bool active = p_i >= 0;
    // inactive lanes run the same code on row 0 with dout = 0: every contribution is then zero
    const uint64_t pid = active ? (pair_id ? (uint64_t)pair_id[k] : (uint64_t)k) : 0ull;
    const float dout = active ? dpred[io ? io[k] : k] : 0.f;
    float h1[64], dh1[64];
    head_layer1(H, active ? p_i : 0, l_i, pid, drop_p, inv_keep, seed, h1);
#pragma unroll
    for (int j = 0; j < 64; ++j) { T1t[j * LDT + tid] = h1[j]; dh1[j] = 0.f; }
    PIs[tid] = p_i; LIs[tid] = l_i;
#pragma unroll 2
    for (int i = 0; i < 32; ++i) {
      const float pre = head_unit2(W2s, b2s, h1, i);
      float m = (pre > 0.f) ? 1.f : 0.f;
      float post = fmaxf(pre, 0.f);
      if (drop_p > 0.f) {
        const bool kp = mmg_keep(seed, SITE_H2, pid * 32ull + i, drop_p);
        m = kp ? m * inv_keep : 0.f;
        post = kp ? post * inv_keep : 0.f;
      }
      const float d2 = dout * W3s[i] * m;        // grad wrt the layer-2 pre-activation
      T2t[i * LDT + tid] = d2;
      const float c3 = wave_sum(dout * post);    // dW3[i]
      const float c2 = wave_sum(d2);             // db2[i]
      if (lane == 0) { RED[wid * 68 + i] += c3; RED[wid * 68 + 32 + i] += c2; }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(W2s + i * 64 + q * 4);
        dh1[q * 4 + 0] = fmaf(w[0], d2, dh1[q * 4 + 0]); dh1[q * 4 + 1] = fmaf(w[1], d2, dh1[q * 4 + 1]);
        dh1[q * 4 + 2] = fmaf(w[2], d2, dh1[q * 4 + 2]); dh1[q * 4 + 3] = fmaf(w[3], d2, dh1[q * 4 + 3]);
      }
    }
    {
      const float c = wave_sum(dout);
      if (lane == 0) RED[wid * 68 + 64] += c;
    }
    // through dropout+relu of layer 1: h1 > 0  <=>  unit kept and positive
#pragma unroll
    for (int j = 0; j < 64; ++j) dh1[j] = (h1[j] > 0.f) ? dh1[j] * inv_keep : 0.f;
    __syncthreads();
    // ---- dW2[i0][j0..j0+7] += sum_p d2[p][i0] * h1[p][j0..]
#pragma unroll 2
    for (int p = 0; p < PT; p += 4) {
      const f32x4 d = *reinterpret_cast<const f32x4*>(T2t + i0 * LDT + p);
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const f32x4 hv = *reinterpret_cast<const f32x4*>(T1t + (j0 + jj) * LDT + p);
        w2acc[jj] = fmaf(d[0], hv[0], w2acc[jj]); w2acc[jj] = fmaf(d[1], hv[1], w2acc[jj]);
        w2acc[jj] = fmaf(d[2], hv[2], w2acc[jj]); w2acc[jj] = fmaf(d[3], hv[3], w2acc[jj]);
      }
    }
    __syncthreads();
    // ---- dh1 tile, pair-major (aliases T1t)
    float* T4 = T1t;
#pragma unroll
    for (int j = 0; j < 64; ++j) T4[tid * LD4 + j] = dh1[j];
    __syncthreads();
    // wave w flushes pairs [64w, 64w+64): lane j owns unit j; equal consecutive patients are pre-summed
    {
      float run = 0.f;
      int cur = -1;
      for (int q = 0; q < 64; ++q) {
        const int p = wid * 64 + q;
        const int pp = PIs[p];
        if (pp < 0) continue;
        const float v = T4[p * LD4 + lane];
        if (pp != cur) {
          if (cur >= 0) atomicAdd(Gd.dA + (size_t)cur * 64 + lane, run);
          cur = pp; run = 0.f;
        }
        run += v;
        const int ll = LIs[p];
        if (lds_db) atomicAdd(DBs + ll * 64 + lane, v);
        else atomicAdd(Gd.dB + (size_t)ll * 64 + lane, v);
      }
      if (cur >= 0) atomicAdd(Gd.dA + (size_t)cur * 64 + lane, run);
    }
    __syncthreads();
  }
  // ---- final flush of the workgroup accumulators
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) atomicAdd(Gd.dW2 + i0 * 64 + j0 + jj, w2acc[jj]);
  __syncthreads();
  if (tid < 65) {
    const float s = RED[tid] + RED[68 + tid] + RED[2 * 68 + tid] + RED[3 * 68 + tid];
    if (tid < 32) atomicAdd(Gd.dW3 + tid, s);
    else if (tid < 64) atomicAdd(Gd.db2 + (tid - 32), s);
    else atomicAdd(Gd.db3, s);
  }
  if (lds_db) for (int i = tid; i < n_labs * 64; i += PT) atomicAdd(Gd.dB + i, DBs[i]);
}

// ---------------------------------------------------------------------------- backward on MFMA
// A tile of 32 pairs; every contraction of the head backward runs on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32), the vector ALUs only do the gather-add, masks and epilogues:
//   (1) H2pre[pair,u]   = H1[pair,:] . W2[u,:]            32 MFMA   (A = h1 rows in registers)
//   (2) dW2[u,k]       += D2[pair,u] * H1[pair,k]          32 MFMA   (A = the C-layout of (1): no lane movement)
//   (3) dH1[pair,k]     = D2[pair,:] . W2[:,k]             32 MFMA   (A = D2 transposed through a 4 KB LDS tile)
//   (4) dB[lab,k]      += [li[pair]==lab] * dH1[pair,k]    12*LT bf16 MFMA (exact: one-hot x 3-way bf16 split of dH1)
// dA[pi] is flushed with run-length pre-reduction (pairs arrive sorted by patient).  Rounds 1-2 ran all of it in ONE wave
// per tile (k_pair_bwd_mfma: ~500 registers, one wave per SIMD); k_pair_bwd_duo below splits it between two.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 pbf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned pu32x2 __attribute__((ext_vector_type(2)));
typedef unsigned pu32x4 __attribute__((ext_vector_type(4)));
constexpr int TP = 32;                 // pairs per wave tile

// Every INDEXED access of the two matrix-core pair kernels goes through a buffer descriptor whose base and byte length
// are KERNEL ARGUMENTS (filled on the host from the array sizes; length 0 = the array is absent).  Two properties follow:
//   * an index that is stale, past the end of its list, or plain garbage can never become an address -- the load returns
//     0 and the store is dropped by the range check of the descriptor (round 2 built the io_perm / pair_id descriptors
//     with num_records = -1, i.e. unbounded: see DESIGN.md section 5, "the 12:45 fault");
//   * all four descriptor words are scalar (SGPR) values by construction, so the compiler never wraps a load in a
//     waterfall loop (its `base ? -1 : 0` select on the device was lowered to a per-lane v_cndmask, which made the
//     descriptor "divergent": 12 readfirstlane loops per tile, each behind a vmcnt(0)).
struct PairBufs {
  const void* pid; const void* io;          // pair_id / io_perm (int64 each) or any valid pointer when absent
  uint32_t pid_bytes, io_bytes;             // n_total * 8, or 0 when absent
  uint32_t pair_bytes;                      // n_total * 4: pi, li, pred / dpred
  uint32_t pat_bytes;                       // n_patients * 4: deg
  uint32_t a_bytes;                         // n_patients * 256: rows of A
  uint32_t b_bytes;                         // n_labs * 256: rows of B
  int32_t n_pat;
};
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pair_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ int pair_ld_i32(__amdgpu_buffer_rsrc_t d, unsigned byte_off) {
  return (int)__builtin_amdgcn_raw_buffer_load_b32(d, (int)byte_off, 0, 0);
}
__device__ __forceinline__ f32x4 pair_ld_f4(__amdgpu_buffer_rsrc_t d, unsigned byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(d, (int)byte_off, 0, 0));
}
constexpr int LDH = 68;                // H1 / dH1 tile row stride (floats)
constexpr int LDD = 36;                // D2 tile row stride

__device__ inline int crow(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// Layer-2 dropout keeps in the MFMA C layout (lane = unit u, register r = pair row crow(r, h)): the element
// (pair id, u) belongs to the RNG group (pair id * 8 + u / 4), shared by the four lanes of a quad.  Instead of every
// lane hashing all 16 of its rows (4 lanes computing the same 64 bits), quad lane j hashes the rows with r % 4 == j
// (4 hashes) and the words are broadcast inside the quad by DPP.  keep[r] = the 16-bit field of this lane's unit.
template <int R>
__device__ __forceinline__ uint32_t quad_bcast_field(const uint32_t* hw0, const uint32_t* hw1, uint32_t sub) {
  constexpr int j = R & 3, i = R >> 2;
  constexpr int ctrl = j | (j << 2) | (j << 4) | (j << 6);          // quad_perm: every lane reads quad lane j
  const uint32_t w0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)hw0[i], ctrl, 0xF, 0xF, true);
  const uint32_t w1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)hw1[i], ctrl, 0xF, 0xF, true);
  return mmg_rng_field(w0, w1, sub);
}
__device__ __forceinline__ void layer2_fields(uint32_t key, const unsigned* PLo, const unsigned* PHi, int h, int l31,
                                              uint32_t* bits /*[16]*/) {
  uint32_t hw0[4], hw1[4];
  const int j = l31 & 3;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = j + 8 * i + 4 * h;                              // = crow(j + 4 i, h)
    const uint64_t pd = ((uint64_t)PHi[row] << 32) | PLo[row];
    mmg_rng_group(key, pd * 8ull + (uint64_t)(l31 >> 2), &hw0[i], &hw1[i]);
  }
  const uint32_t sub = (uint32_t)l31 & 3u;
  bits[0] = quad_bcast_field<0>(hw0, hw1, sub);   bits[1] = quad_bcast_field<1>(hw0, hw1, sub);
  bits[2] = quad_bcast_field<2>(hw0, hw1, sub);   bits[3] = quad_bcast_field<3>(hw0, hw1, sub);
  bits[4] = quad_bcast_field<4>(hw0, hw1, sub);   bits[5] = quad_bcast_field<5>(hw0, hw1, sub);
  bits[6] = quad_bcast_field<6>(hw0, hw1, sub);   bits[7] = quad_bcast_field<7>(hw0, hw1, sub);
  bits[8] = quad_bcast_field<8>(hw0, hw1, sub);   bits[9] = quad_bcast_field<9>(hw0, hw1, sub);
  bits[10] = quad_bcast_field<10>(hw0, hw1, sub); bits[11] = quad_bcast_field<11>(hw0, hw1, sub);
  bits[12] = quad_bcast_field<12>(hw0, hw1, sub); bits[13] = quad_bcast_field<13>(hw0, hw1, sub);
  bits[14] = quad_bcast_field<14>(hw0, hw1, sub); bits[15] = quad_bcast_field<15>(hw0, hw1, sub);
}

__host__ __device__ constexpr int pair_slab_floats(int LT) { return 2048 + LT * 2048 + 68; }

// ---------------------------------------------------------------------------- backward on MFMA, two waves per tile
// One wave per tile needs ~500 registers: ONE wave per SIMD, so nothing issues while its fp32 matrix
// instructions run and its matrix pipe idles through every vector phase (measured: 35 % matrix-busy, the rest gather-add,
// RNG, masks, splits, LDS round trips, the run-length flush).  Here a tile is worked on by TWO waves of the same SIMD in
// turn, each under 256 registers:
//   front wave w (0..3): the load pipeline, h1 (gather-add, relu, dropout), (1) H2pre, the layer-2 epilogue -> D2, (2) dW2
//   back  wave w + 4   : (3) dH1, (4) dB on the bf16 matrix cores, the run-length flush of dA[pi]
// (64 fp32 matrix instructions + the RNG-heavy vector work in front, 32 fp32 + 24 bf16 matrix instructions + splits and
// the flush behind: about even.)  The front hands H1, D2 and the tile's patient / lab ids to its back wave through
// double-buffered LDS tiles; ONE workgroup barrier per tile orders both buffers: the front of tile t + 1 runs beside the
// back of tile t, one wave's vector instructions issue under the other's matrix instructions.  Same arithmetic, same
// summation order per wave and the same slab layout as the one-wave kernel had (dW2 and the three bias-like sums
// accumulate in the front waves, dB in the back waves).
constexpr int HAND_LDS = TP * LDH + TP * LDD;      // floats per hand-off buffer: H1 [32][LDH] | D2 [32][LDD]
constexpr int FRONT_LDS = 3 * TP;                  // floats private to a front wave: dout | pair id lo | hi

template <int LT, bool AUX>
__device__ __forceinline__ void pair_bwd_front(const HeadDev& H, const int32_t* __restrict__ pi, const int32_t* __restrict__ li,
                                               const int32_t* __restrict__ deg, int thr, int want_low, int64_t n, float drop_p,
                                               uint64_t seed, const PairBufs& pb, const float* __restrict__ dpred,
                                               const int32_t* __restrict__ sel, int n_iter, float* fl, const float* W2s,
                                               float (*HX)[4][HAND_LDS], int (*XP)[4][TP], int (*XL)[4][TP],
                                               float (*tail_red)[68], float* red) {
  const int tid = threadIdx.x, lane = tid & 63, w = (tid >> 6) & 3;
  const int h = lane >> 5, l31 = lane & 31;
  float* DOs = fl;                                  // [32] dout (0 for inactive)
  unsigned* PLo = reinterpret_cast<unsigned*>(DOs + TP);
  unsigned* PHi = PLo + TP;
  const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  const float b2v = H.b2[l31], w3v = H.W3[l31];
  float w3acc = 0.f, b2acc = 0.f, b3acc = 0.f;
  f32x16 accW2[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { accW2[0][i] = 0.f; accW2[1][i] = 0.f; }
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + w, n_waves = (int64_t)gridDim.x * 4;
  // Three-deep software pipeline over this wave's tiles, one dependent load per stage:  t+3: list position -> pair
  // index;  t+2: pair -> patient, lab, rng id, slot of its upstream gradient;  t+1: gate degree, upstream gradient, the
  // A / B row halves.  A stage only ISSUES loads; they are finalised one iteration later.  Every load is unconditional and
  // BOUNDED by a host-sized descriptor (a branch around a load makes the compiler's vmcnt waits conservative).
  struct Meta { int k; int p_i; int l_i; int o; uint64_t pid; };
  const __amdgpu_buffer_rsrc_t sel_d = pair_rsrc(sel ? sel : pi, sel ? (uint32_t)(n * 4) : 0u);
  const __amdgpu_buffer_rsrc_t io_d = pair_rsrc(pb.io, pb.io_bytes), pid_d = pair_rsrc(pb.pid, pb.pid_bytes);
  const __amdgpu_buffer_rsrc_t pi_d = pair_rsrc(pi, pb.pair_bytes), li_d = pair_rsrc(li, pb.pair_bytes);
  const __amdgpu_buffer_rsrc_t dp_d = pair_rsrc(dpred, pb.pair_bytes), deg_d = pair_rsrc(deg, pb.pat_bytes);
  const __amdgpu_buffer_rsrc_t A_d = pair_rsrc(H.A, pb.a_bytes), B_d = pair_rsrc(H.B, pb.b_bytes);
  const bool has_sel = sel != nullptr, has_io = pb.io_bytes != 0u, has_pid = pb.pid_bytes != 0u;
  struct RawMeta { int k, p, l; pu32x2 o2, d2; };
  auto issue_k = [&](int64_t t) {
    const int64_t idx = t * TP + l31;
    return pair_ld_i32(sel_d, (unsigned)(idx < n ? idx : 0) * 4u);
  };
  auto fin_k = [&](int kr, int64_t t) {
    const int64_t idx = t * TP + l31;
    return idx < n ? (has_sel ? kr : (int)idx) : -1;
  };
  auto issue_meta = [&](int k) {
    const unsigned kc = k >= 0 ? (unsigned)k : 0u;
    RawMeta r;
    r.k = k;
    r.p = pair_ld_i32(pi_d, kc * 4u);
    r.l = pair_ld_i32(li_d, kc * 4u);
    if (AUX) {
      r.o2 = __builtin_amdgcn_raw_buffer_load_b64(io_d, (int)(kc * 8u), 0, 0);
      r.d2 = __builtin_amdgcn_raw_buffer_load_b64(pid_d, (int)(kc * 8u), 0, 0);
    }
    return r;
  };
  auto fin_meta = [&](const RawMeta& r) {
    const int kc = r.k >= 0 ? r.k : 0;
    Meta m;
    m.k = r.k;
    m.p_i = ((unsigned)r.k < (pb.pair_bytes >> 2) && (unsigned)r.p < (unsigned)pb.n_pat) ? r.p : -1;
    m.l_i = r.l;
    m.o = (AUX && has_io) ? (int)r.o2[0] : kc;
    m.pid = (AUX && has_pid) ? ((uint64_t)r.d2[1] << 32 | r.d2[0]) : (uint64_t)kc;
    return m;
  };
  auto load_rows = [&](const Meta& m, f32x4* ra, f32x4* rb, int* dg, float* dv) {
    const unsigned pp = m.p_i >= 0 ? (unsigned)m.p_i : 0u;
    *dg = pair_ld_i32(deg_d, pp * 4u);
    *dv = __builtin_bit_cast(float, pair_ld_i32(dp_d, (unsigned)m.o * 4u));
    const unsigned ao = pp * 256u + 128u * h, bo = (unsigned)m.l_i * 256u + 128u * h;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      ra[q] = pair_ld_f4(A_d, ao + q * 16u);
      rb[q] = pair_ld_f4(B_d, bo + q * 16u);
    }
  };
  const int kr0 = issue_k(wave_id), kr1 = issue_k(wave_id + n_waves);
  int kr2 = issue_k(wave_id + 2 * n_waves);
  const RawMeta rm0 = issue_meta(fin_k(kr0, wave_id));
  RawMeta rm1 = issue_meta(fin_k(kr1, wave_id + n_waves));
  Meta m0 = fin_meta(rm0);
  f32x4 ra[8], rb[8];
  int dg0; float dv0;
  load_rows(m0, ra, rb, &dg0, &dv0);
  for (int it = 0; it <= n_iter; ++it) {
    const int par = it & 1;
    if (it < n_iter) {
      const int64_t t = wave_id + (int64_t)it * n_waves;
      float* H1s = HX[par][w];                      // [32][LDH]
      float* D2s = H1s + TP * LDH;                  // [32][LDD]
      const Meta mc = m0;
      const bool active = mc.p_i >= 0 && ((int)(dg0 < thr)) == want_low;
      const float dout = active ? dv0 : 0.f;
      f32x4 ca[8], cb[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) { ca[q] = ra[q]; cb[q] = rb[q]; }
      const Meta m1 = fin_meta(rm1);
      const int k2 = fin_k(kr2, t + 2 * n_waves);
      kr2 = issue_k(t + 3 * n_waves);
      rm1 = issue_meta(k2);
      load_rows(m1, ra, rb, &dg0, &dv0);           // next tile's rows: in flight during this tile's arithmetic
      __builtin_amdgcn_sched_barrier(0);
      m0 = m1;
      const int p_i = active ? mc.p_i : -1, l_i = mc.l_i;
      const uint64_t pid = mc.pid;
      if (h == 0) { XP[par][w][l31] = p_i; XL[par][w][l31] = l_i; }     // (an all -1 patient list = nothing to do for the back wave)
      if (__ballot(p_i >= 0) != 0ull) {
        if (h == 0) {
          DOs[l31] = dout;
          PLo[l31] = (unsigned)pid; PHi[l31] = (unsigned)(pid >> 32);
          b3acc += dout;
        }
        // ---- h1[pair=l31][k=32h+s]: gather-add, relu, dropout; kept in registers AND written to the hand-off tile
        float d2c[16];
        {
          float h1a[32];
          const uint32_t key1 = mmg_rng_key(seed, SITE_H1), thr1 = mmg_keep_threshold(drop_p);
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(ca[q][j] + cb[q][j], 0.f);
            if (drop_p > 0.f) mmg_drop4(v, key1, pid * 64ull + (uint64_t)(32 * h + q * 4), thr1, inv_keep);
#pragma unroll
            for (int j = 0; j < 4; ++j) h1a[q * 4 + j] = v[j];
            *reinterpret_cast<f32x4*>(H1s + l31 * LDH + 32 * h + q * 4) = v;
          }
          // ---- (1) H2pre = H1 . W2^T
          f32x16 acc1;
#pragma unroll
          for (int i = 0; i < 16; ++i) acc1[i] = 0.f;
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(W2s + l31 * LDH + 32 * h + q * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(h1a[q * 4 + j], wv[j], acc1, 0, 0, 0);
          }
          // ---- epilogue of layer 2 in the C layout: lane = unit u (l31), reg r = pair row crow(r,h)
          uint32_t kb[16];
          if (drop_p > 0.f) layer2_fields(mmg_rng_key(seed, SITE_H2), PLo, PHi, h, l31, kb);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = crow(r, h);
            const float pre = acc1[r] + b2v;
            float m = pre > 0.f ? 1.f : 0.f;
            float post = fmaxf(pre, 0.f);
            if (drop_p > 0.f) {
              const bool kp = kb[r] >= mmg_keep_threshold(drop_p);
              m = kp ? m * inv_keep : 0.f;
              post = kp ? post * inv_keep : 0.f;
            }
            const float dr = DOs[row];
            const float d2 = dr * w3v * m;
            d2c[r] = d2;
            w3acc = fmaf(dr, post, w3acc);
            b2acc += d2;
            D2s[row * LDD + l31] = d2;
          }
        }
        // ---- (2) dW2[u,k] += D2[pair,u] * H1[pair,k]   (A = d2c: already lane = u, step s = pair crow(s,h); B = H1 in
        //      the column layout, read back from the tile this wave wrote above)
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
          const float c0 = H1s[crow(s2, h) * LDH + l31], c1 = H1s[crow(s2, h) * LDH + 32 + l31];
          accW2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(d2c[s2], c0, accW2[0], 0, 0, 0);
          accW2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(d2c[s2], c1, accW2[1], 0, 0, 0);
        }
        // (with the saved state this wave has little vector work left, but giving it (3) dH1 as well -- a nearly pure
        //  matrix wave of 64 fp32 instructions beside a vector-only back wave -- measured 218 us against 153: the long
        //  matrix phase starves the other wave's issue; the two contractions stay split between the waves)
      }
    }
    __syncthreads();
  }
  // ---- final flush, front half: the four front waves' dW2 summed through LDS in wave order (the back waves add their dB
  //      slots in the same four rounds); the hand-off buffers are dead: every wave passed the last barrier of the loop
  for (int ww = 0; ww < 4; ++ww) {
    if (w == ww) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (ww == 0) { red[r * 64 + lane] = accW2[0][r]; red[(16 + r) * 64 + lane] = accW2[1][r]; }
        else { red[r * 64 + lane] += accW2[0][r]; red[(16 + r) * 64 + lane] += accW2[1][r]; }
      }
    }
    __syncthreads();
  }
  w3acc += __shfl_xor(w3acc, 32, 64);
  b2acc += __shfl_xor(b2acc, 32, 64);
  b3acc = wave_sum(b3acc);
  if (lane < 32) { tail_red[w][lane] = w3acc; tail_red[w][32 + lane] = b2acc; }
  if (lane == 0) tail_red[w][64] = b3acc;
}

template <int LT>
__device__ __forceinline__ void pair_bwd_back(float* __restrict__ dA, float drop_p, int n_iter, const float* W2s,
                                              float (*HX)[4][HAND_LDS], int (*XP)[4][TP], int (*XL)[4][TP], float* red) {
  const int tid = threadIdx.x, lane = tid & 63, w = (tid >> 6) & 3;
  const int h = lane >> 5, l31 = lane & 31;
  const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  f32x16 accB[LT][2];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int t = 0; t < LT; ++t) { accB[t][0][i] = 0.f; accB[t][1][i] = 0.f; }
  for (int it = 0; it <= n_iter; ++it) {
    if (it >= 1) {
      const int par = (it - 1) & 1;
      const int p_i = XP[par][w][l31];
      if (__ballot(p_i >= 0) != 0ull) {
        float* H1s = HX[par][w];                    // [32][LDH]  (later: the dH1 tile)
        const float* D2s = H1s + TP * LDH;          // [32][LDD]
        // ---- (3) dH1[pair,k] = D2[pair,:] . W2[:,k]     (A = D2[pair=l31][u=16h+s] from the LDS tile)
        float d2a[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(D2s + l31 * LDD + 16 * h + q * 4);
          d2a[q * 4 + 0] = v[0]; d2a[q * 4 + 1] = v[1]; d2a[q * 4 + 2] = v[2]; d2a[q * 4 + 3] = v[3];
        }
        f32x16 accH[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) { accH[0][i] = 0.f; accH[1][i] = 0.f; }
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
          const float w0 = W2s[(16 * h + s2) * LDH + l31], w1 = W2s[(16 * h + s2) * LDH + 32 + l31];
          accH[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(d2a[s2], w0, accH[0], 0, 0, 0);
          accH[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(d2a[s2], w1, accH[1], 0, 0, 0);
        }
        // through dropout + relu of layer 1 (h1 > 0 <=> kept and positive); C layout: lane = column, reg = pair row
        float dh[16][2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float c0 = H1s[crow(r, h) * LDH + l31], c1 = H1s[crow(r, h) * LDH + 32 + l31];   // H1 in the column layout
          dh[r][0] = c0 > 0.f ? accH[0][r] * inv_keep : 0.f;
          dh[r][1] = c1 > 0.f ? accH[1][r] * inv_keep : 0.f;
        }
        // ---- (4) dB[lab,k] += onehot(li[pair])[lab] * dH1[pair,k] on the bf16 matrix cores (exact products: see
        //      below); k index of step t2, lane half h, element j  <->  pair row crow(8 t2 + j, h): exactly the C-layout
        //      registers 8 t2 .. 8 t2 + 7.  The one-hot is exact in bf16 and dH1 splits exactly into three bf16 pieces, so the
        //      products are exact and the fp32 accumulation matches the fp32 path up to order -- at 1/5 of its matrix time.
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
          pbf16x8 bp[2][3];
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float v = dh[8 * t2 + j][ct];
              const __bf16 a = (__bf16)v;
              const float r1 = v - (float)a;
              const __bf16 b = (__bf16)r1;
              bp[ct][0][j] = a; bp[ct][1][j] = b; bp[ct][2][j] = (__bf16)(r1 - (float)b);
            }
          int labs[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) labs[j] = XL[par][w][crow(8 * t2 + j, h)];
#pragma unroll
          for (int lt = 0; lt < LT; ++lt) {
            pbf16x8 oh;
#pragma unroll
            for (int j = 0; j < 8; ++j) oh[j] = (labs[j] == lt * 32 + l31) ? (__bf16)1.0f : (__bf16)0.0f;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
              for (int p = 0; p < 3; ++p)
                accB[lt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(oh, bp[ct][p], accB[lt][ct], 0, 0, 0);
          }
        }
        // ---- dA[pi] += dH1: tile to LDS (aliases H1: every H1 read of this wave is done, the front wave writes the
        //      other buffer until the next barrier), then run-length flush (pairs arrive sorted by patient)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          H1s[crow(r, h) * LDH + l31] = dh[r][0];
          H1s[crow(r, h) * LDH + 32 + l31] = dh[r][1];
        }
        {
          float run = 0.f;
          int cur = -1;
#pragma unroll
          for (int bq = 0; bq < 2; ++bq) {
            float vq[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) vq[q] = H1s[(bq * 16 + q) * LDH + lane];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
              const int pp = __builtin_amdgcn_readlane(p_i, bq * 16 + q);
              if (pp < 0) continue;
              if (pp != cur) {
                if (cur >= 0) atomicAdd(dA + (size_t)cur * 64 + lane, run);
                cur = pp; run = 0.f;
              }
              run += vq[q];
            }
          }
          if (cur >= 0) atomicAdd(dA + (size_t)cur * 64 + lane, run);
        }
      }
    }
    __syncthreads();
  }
  // ---- final flush: the four back waves' accumulators summed through LDS in wave order (one slab per workgroup); the
  //      hand-off buffers are dead: every wave passed the last barrier of the tile loop
  for (int ww = 0; ww < 4; ++ww) {
    if (w == ww) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int lt = 0; lt < LT; ++lt) {
          if (ww == 0) {
            red[(32 + lt * 32 + r) * 64 + lane] = accB[lt][0][r];
            red[(48 + lt * 32 + r) * 64 + lane] = accB[lt][1][r];
          } else {
            red[(32 + lt * 32 + r) * 64 + lane] += accB[lt][0][r];
            red[(48 + lt * 32 + r) * 64 + lane] += accB[lt][1][r];
          }
        }
      }
    }
    __syncthreads();
  }
}

template <int LT, bool AUX>
__global__ __launch_bounds__(512) void k_pair_bwd_duo(HeadDev H, HeadGradDev Gd, const int32_t* __restrict__ pi,
                                                      const int32_t* __restrict__ li, const int32_t* __restrict__ deg,
                                                      int thr, int want_low, int64_t n, int n_labs, float drop_p,
                                                      uint64_t seed, const uint64_t* __restrict__ seed_ptr, PairBufs pb,
                                                      const float* __restrict__ dpred, const int32_t* __restrict__ sel,
                                                      const int32_t* __restrict__ n_sel, float* __restrict__ slab) {
  if (seed_ptr) seed = *seed_ptr;
  if (sel) { const int64_t nl = *n_sel; n = nl < 0 ? 0 : (nl < n ? nl : n); }
  __shared__ __attribute__((aligned(16))) float HX[2][4][HAND_LDS];      // hand-off tiles: [buffer][wave pair]
  __shared__ __attribute__((aligned(16))) float smf[4 * FRONT_LDS];
  __shared__ __attribute__((aligned(16))) float W2s[32 * LDH];
  __shared__ int XP[2][4][TP], XL[2][4][TP];
  __shared__ float tail_red[4][68];
  const int tid = threadIdx.x, wid = tid >> 6;
  for (int i = tid; i < 2048; i += 512) W2s[(i >> 6) * LDH + (i & 63)] = H.W2[i];
  __syncthreads();
  // tiles wave_id, wave_id + n_waves, ...: the first front wave of the workgroup has the most -- every wave runs that many
  // iterations (+ 1: the back waves trail by one) so that all of them meet at the same barriers
  const int64_t n_tiles = (n + TP - 1) / TP, n_waves = (int64_t)gridDim.x * 4, first = (int64_t)blockIdx.x * 4;
  const int n_iter = first < n_tiles ? (int)((n_tiles - first + n_waves - 1) / n_waves) : 0;
  float* red = &HX[0][0][0];
  constexpr int NR = (2 + 2 * LT) * 16;
  static_assert(NR * 64 <= 2 * 4 * HAND_LDS, "reduction tile must fit the hand-off buffers");
  if (wid < 4)
    pair_bwd_front<LT, AUX>(H, pi, li, deg, thr, want_low, n, drop_p, seed, pb, dpred, sel, n_iter, smf + (wid & 3) * FRONT_LDS,
                            W2s, HX, XP, XL, tail_red, red);
  else
    pair_bwd_back<LT>(Gd.dA, drop_p, n_iter, W2s, HX, XP, XL, red);
  // ---- one partial slab per workgroup (summed over the workgroups in fixed order by mmg_k_reduce_slabs)
  float* my = slab + (size_t)blockIdx.x * pair_slab_floats(LT);
  for (int e = tid; e < NR * 64; e += 512) {
    const int slot = e >> 6, ln = e & 63;
    const float v = red[e];
    const int hh = ln >> 5, c31 = ln & 31;
    if (slot < 32) {
      const int ct = slot >> 4, r = slot & 15;
      my[crow(r, hh) * 64 + ct * 32 + c31] = v;
    } else {
      const int q = slot - 32, lt = q >> 5, ct = (q >> 4) & 1, r = q & 15;
      my[2048 + (lt * 32 + crow(r, hh)) * 64 + ct * 32 + c31] = v;
    }
  }
  __syncthreads();                                   // the front waves' tail sums are in tail_red
  if (tid < 68) {
    const float t = tid < 65 ? ((tail_red[0][tid] + tail_red[1][tid]) + tail_red[2][tid]) + tail_red[3][tid] : 0.f;
    const int dst = tid < 32 ? 32 + tid : (tid < 64 ? tid - 32 : tid);
    my[2048 + LT * 2048 + dst] = t;
  }
}

// ---------------------------------------------------------------------------- backward, all contractions on the bf16 pipe
// The fp32 matrix instruction (v_mfma_f32_32x32x2_f32) shares the vector ALU's multipliers: while one runs, NO wave of
// the SIMD issues vector work, so the 96 of them per tile in k_pair_bwd_duo (6,144 cycles) simply ADD to the vector time
// of both waves -- measured: front alone 151 us, back alone 116 us, both 193 us; raising the matrix wave's priority
// (s_setprio) changes nothing.  The bf16 matrix pipe runs BESIDE the vector ALU.  Here (1), (2) and (3) use the exact
// six-term bf16 split of gemm.hip (fp32-grade: the three pieces of an fp32 value are exact, six of the nine cross terms
// are kept, fp32 accumulation), 24 matrix instructions of 32 cycles each instead of 32 of 64, and what the split costs in
// vector instructions issues in their shadow or in the other wave's:
//   front wave: load pipeline, h1 -> its three bf16 pieces (registers: A of (1); row-major LDS planes: B of (2) through
//               the transposing LDS read), (1) H2pre = H1 . W2^T in the FORWARD's own term order (the recomputed
//               pre-activation is the forward's bit for bit), layer-2 epilogue -> D2 (registers: A of (2); fp32 tile for the
//               back wave), (2) dW2 += D2^T . H1, the sign bits of h1 (64 per pair) for the back wave
//   back wave : (3) dH1 = D2 . W2 (A = D2 rows from the tile, split here; B = W2^T pieces held in registers), the layer-1
//               mask from the sign bits, (4) dB, the run-length flush of dA
// Hand-off per tile: D2 [32][36] fp32 + 2 words of sign bits + the patient / lab ids, double-buffered, one workgroup
// barrier per tile.  K index of a k-step of (2) and (4): lane half h, element j <-> pair row crow(8 t + j, h), i.e. the
// C-layout registers 8 t .. 8 t + 7 (A from registers) and two transposing reads at rows 16 t + 4 h and 16 t + 8 + 4 h (B).
constexpr int P1S = 80;                              // row stride of an h1 plane (bf16): 160 B, the 4 rows of a transposing
                                                     // read land on 4 disjoint 8-dword bank spans
constexpr int F6_LDS = 3 * TP * 4 + 3 * TP * P1S * 2 + TP * LDD * 4;   // bytes private to a front wave: dout | pair id lo | hi | 3 planes | saved h2 tile
constexpr int B6_LDS = 2 * TP * 4;                   // bytes private to a back wave: run id per pair | patient per run
constexpr int H6_LDS = TP * LDD * 4 + TP * 2 * 4;    // bytes per hand-off buffer: D2 tile | sign bits

__device__ __forceinline__ void psplit8(const float* v, pbf16x8& p0, pbf16x8& p1, pbf16x8& p2) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 a = (__bf16)v[j];
    const float r1 = v[j] - (float)a;
    const __bf16 b = (__bf16)r1;
    p0[j] = a; p1[j] = b; p2[j] = (__bf16)(r1 - (float)b);
  }
}
// six exact products, small terms first (the order of k_pair_fwd_mfma and gemm.hip)
#define MMG_X6(acc, a, b)                                                     \
  do {                                                                        \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);  \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);  \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);  \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);  \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);  \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);  \
  } while (0)

typedef short ps16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ pbf16x8 tr_pair(const __bf16* p_lo, const __bf16* p_hi) {   // 4 + 4 rows of this lane's column
  const ps16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ps16x4*)p_lo);
  const ps16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ps16x4*)p_hi);
  typedef short ps16x8 __attribute__((ext_vector_type(8)));
  const ps16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(pbf16x8, v);
}

// SAVED: the forward left the sign bits of h1 and the layer-2 activations of every pair it visited (mmg_pair_saved_t):
// no RNG, no (1) and no layer-2 epilogue arithmetic -- h1 = bit ? (A + B) / (1 - p) : 0, the layer-2 mask is the sign
// pattern of the saved activation.  Because (1) above reproduces the forward's pre-activation bit for bit, the saved and
// the recomputing variant return the SAME bits.
template <int LT, bool AUX, bool SAVED>
__device__ __forceinline__ void pair_bwd6_front(const HeadDev& H, const int32_t* __restrict__ pi, const int32_t* __restrict__ li,
                                                const int32_t* __restrict__ deg, int thr, int want_low, int64_t n, float drop_p,
                                                uint64_t seed, const PairBufs& pb, const float* __restrict__ dpred,
                                                const int32_t* __restrict__ sel, int n_iter, unsigned char* fl,
                                                unsigned char (*HX)[4][H6_LDS], int (*XP)[4][TP], int (*XL)[4][TP],
                                                float (*tail_red)[68], float* red, const uint32_t* __restrict__ sv_bits,
                                                const float* __restrict__ sv_h2, int sv_by_pos) {
  const int tid = threadIdx.x, lane = tid & 63, w = (tid >> 6) & 3;
  const int h = lane >> 5, l31 = lane & 31;
  float* DOs = reinterpret_cast<float*>(fl);        // [32] dout (0 for inactive)
  unsigned* PLo = reinterpret_cast<unsigned*>(DOs + TP);
  unsigned* PHi = PLo + TP;
  __bf16* P1 = reinterpret_cast<__bf16*>(PHi + TP);  // [3][32][P1S] the pieces of h1, row-major
  float* T2 = reinterpret_cast<float*>(P1 + 3 * TP * P1S);   // SAVED: [32][LDD] the tile's saved layer-2 activations
  const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  const uint32_t thr_keep = mmg_keep_threshold(drop_p);
  const uint32_t key1 = mmg_rng_key(seed, SITE_H1);
  const float b2v = H.b2[l31], w3v = H.W3[l31];
  // B of (1): W2[unit = l31][k = 16 ks + 8 h + j] as three exact bf16 pieces
  pbf16x8 w2p[SAVED ? 1 : 4][3];
  if constexpr (!SAVED) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = H.W2[l31 * 64 + 16 * ks + 8 * h + j];
      psplit8(v, w2p[ks][0], w2p[ks][1], w2p[ks][2]);
    }
  }
  float w3acc = 0.f, b2acc = 0.f, b3acc = 0.f;
  f32x16 accW2[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { accW2[0][i] = 0.f; accW2[1][i] = 0.f; }
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + w, n_waves = (int64_t)gridDim.x * 4;
  // the three-deep load pipeline of k_pair_bwd_duo; the row halves in the FORWARD's order: register q of a lane holds
  // k = 16 (q >> 1) + 8 h + 4 (q & 1) + 0..3, so that a k-step's eight values are lane-local
  pu32x2 sbw = {0u, 0u};                            // SAVED: the pair's 64 sign bits / this lane's 16 saved activations
  f32x4 sh2[SAVED ? 4 : 1];
  struct Meta { int k; int p_i; int l_i; int o; uint64_t pid; };
  const __amdgpu_buffer_rsrc_t sel_d = pair_rsrc(sel ? sel : pi, sel ? (uint32_t)(n * 4) : 0u);
  const __amdgpu_buffer_rsrc_t io_d = pair_rsrc(pb.io, pb.io_bytes), pid_d = pair_rsrc(pb.pid, pb.pid_bytes);
  const __amdgpu_buffer_rsrc_t pi_d = pair_rsrc(pi, pb.pair_bytes), li_d = pair_rsrc(li, pb.pair_bytes);
  const __amdgpu_buffer_rsrc_t dp_d = pair_rsrc(dpred, pb.pair_bytes), deg_d = pair_rsrc(deg, pb.pat_bytes);
  const __amdgpu_buffer_rsrc_t A_d = pair_rsrc(H.A, pb.a_bytes), B_d = pair_rsrc(H.B, pb.b_bytes);
  const bool has_sel = sel != nullptr, has_io = pb.io_bytes != 0u, has_pid = pb.pid_bytes != 0u;
  struct RawMeta { int k, p, l; pu32x2 o2, d2; };
  auto issue_k = [&](int64_t t) {
    const int64_t idx = t * TP + l31;
    return pair_ld_i32(sel_d, (unsigned)(idx < n ? idx : 0) * 4u);
  };
  auto fin_k = [&](int kr, int64_t t) {
    const int64_t idx = t * TP + l31;
    return idx < n ? (has_sel ? kr : (int)idx) : -1;
  };
  auto issue_meta = [&](int k) {
    const unsigned kc = k >= 0 ? (unsigned)k : 0u;
    RawMeta r;
    r.k = k;
    r.p = pair_ld_i32(pi_d, kc * 4u);
    r.l = pair_ld_i32(li_d, kc * 4u);
    if (AUX) {
      r.o2 = __builtin_amdgcn_raw_buffer_load_b64(io_d, (int)(kc * 8u), 0, 0);
      r.d2 = __builtin_amdgcn_raw_buffer_load_b64(pid_d, (int)(kc * 8u), 0, 0);
    }
    return r;
  };
  auto fin_meta = [&](const RawMeta& r) {
    const int kc = r.k >= 0 ? r.k : 0;
    Meta m;
    m.k = r.k;
    m.p_i = ((unsigned)r.k < (pb.pair_bytes >> 2) && (unsigned)r.p < (unsigned)pb.n_pat) ? r.p : -1;
    m.l_i = r.l;
    m.o = (AUX && has_io) ? (int)r.o2[0] : kc;
    m.pid = (AUX && has_pid) ? ((uint64_t)r.d2[1] << 32 | r.d2[0]) : (uint64_t)kc;
    return m;
  };
  auto load_rows = [&](const Meta& m, int64_t tile, f32x4* ra, f32x4* rb, int* dg, float* dv) {
    const unsigned pp = m.p_i >= 0 ? (unsigned)m.p_i : 0u;
    *dg = pair_ld_i32(deg_d, pp * 4u);
    *dv = __builtin_bit_cast(float, pair_ld_i32(dp_d, (unsigned)m.o * 4u));
    const unsigned ao = pp * 256u + 32u * h, bo = (unsigned)m.l_i * 256u + 32u * h;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      ra[q] = pair_ld_f4(A_d, ao + (q >> 1) * 64u + (q & 1) * 16u);
      rb[q] = pair_ld_f4(B_d, bo + (q >> 1) * 64u + (q & 1) * 16u);
    }
    if constexpr (SAVED) {          // (a pair that is not one -- p_i < 0 -- reads entry 0: always inside the buffers)
      const size_t kc = m.p_i >= 0 ? (size_t)(sv_by_pos ? tile * TP + l31 : (int64_t)m.k) : 0;
      sbw = *reinterpret_cast<const pu32x2*>(sv_bits + kc * 2);
#pragma unroll
      for (int q = 0; q < 4; ++q) sh2[q] = *reinterpret_cast<const f32x4*>(sv_h2 + kc * 32 + 16 * h + 4 * q);
    }
  };
  // transposing-read lane roles (see k_linear_wgrad_x6): group g = lane >> 4 supplies row q of columns 16 (g & 1) + 4 p
  const int trq = (lane >> 2) & 3, trc = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const int kr0 = issue_k(wave_id), kr1 = issue_k(wave_id + n_waves);
  int kr2 = issue_k(wave_id + 2 * n_waves);
  const RawMeta rm0 = issue_meta(fin_k(kr0, wave_id));
  RawMeta rm1 = issue_meta(fin_k(kr1, wave_id + n_waves));
  Meta m0 = fin_meta(rm0);
  f32x4 ra[8], rb[8];
  int dg0; float dv0;
  load_rows(m0, wave_id, ra, rb, &dg0, &dv0);
  for (int it = 0; it <= n_iter; ++it) {
    const int par = it & 1;
    if (it < n_iter) {
      const int64_t t = wave_id + (int64_t)it * n_waves;
      float* D2s = reinterpret_cast<float*>(HX[par][w]);                 // [32][LDD]
      unsigned* XB = reinterpret_cast<unsigned*>(D2s + TP * LDD);         // [32][2] sign bits of h1
      const Meta mc = m0;
      const bool active = mc.p_i >= 0 && ((int)(dg0 < thr)) == want_low;
      const float dout = active ? dv0 : 0.f;
      const Meta m1 = fin_meta(rm1);
      const int k2 = fin_k(kr2, t + 2 * n_waves);
      kr2 = issue_k(t + 3 * n_waves);
      rm1 = issue_meta(k2);
      m0 = m1;
      const int p_i = active ? mc.p_i : -1, l_i = mc.l_i;
      const uint64_t pid = mc.pid;
      if (h == 0) { XP[par][w][l31] = p_i; XL[par][w][l31] = l_i; }     // (an all -1 patient list = nothing to do for the back wave)
      const bool any = __ballot(p_i >= 0) != 0ull;                      // wave-uniform
      f32x16 acc1;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc1[i] = 0.f;
      if (any) {
        if (h == 0) {
          DOs[l31] = dout;
          PLo[l31] = (unsigned)pid; PHi[l31] = (unsigned)(pid >> 32);
          b3acc += dout;
        }
        if constexpr (SAVED) {
          // a pair of the tile that is not this head's was never written by the forward: its entry is arbitrary memory
          const pu32x2 bwv = active ? sbw : pu32x2{0u, 0u};
          const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(T2 + l31 * LDD + 16 * h + 4 * q) = active ? sh2[q] : z4;
          if (h == 0) *reinterpret_cast<pu32x2*>(XB + 2 * l31) = bwv;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const uint32_t wbits = (bwv[ks >> 1] >> (16 * (ks & 1) + 8 * h)) & 0xFFu;
            float x8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
              x8[j] = (wbits >> j) & 1u ? (ra[2 * ks + (j >> 2)][j & 3] + rb[2 * ks + (j >> 2)][j & 3]) * inv_keep : 0.f;
            pbf16x8 xp[3];
            psplit8(x8, xp[0], xp[1], xp[2]);
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
              *reinterpret_cast<pbf16x8*>(P1 + (pc * TP + l31) * P1S + 16 * ks + 8 * h) = xp[pc];
          }
        } else {
          // ---- h1 (gather-add, relu, dropout), k-step by k-step: pieces -> (1) and the planes, sign bits
          uint32_t bw[2] = {0u, 0u};
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            float x8[8];
#pragma unroll
            for (int c = 0; c < 2; ++c) {                // one aligned RNG group of 4 per chunk: one hash
              f32x4 x;
#pragma unroll
              for (int j = 0; j < 4; ++j) x[j] = fmaxf(ra[2 * ks + c][j] + rb[2 * ks + c][j], 0.f);
              if (drop_p > 0.f) mmg_drop4(x, key1, pid * 64ull + (uint64_t)(16 * ks + 8 * h + 4 * c), thr_keep, inv_keep);
#pragma unroll
              for (int j = 0; j < 4; ++j) x8[4 * c + j] = x[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) bw[ks >> 1] |= x8[j] > 0.f ? (1u << (16 * (ks & 1) + j)) << (8 * h) : 0u;
            pbf16x8 xp[3];
            psplit8(x8, xp[0], xp[1], xp[2]);
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
              *reinterpret_cast<pbf16x8*>(P1 + (pc * TP + l31) * P1S + 16 * ks + 8 * h) = xp[pc];
            // (1) C[pair rows, unit] = H1 . W2^T: the forward's six products in the forward's order (operands swapped)
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xp[0], w2p[ks][2], acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xp[2], w2p[ks][0], acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xp[1], w2p[ks][1], acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xp[0], w2p[ks][1], acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xp[1], w2p[ks][0], acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xp[0], w2p[ks][0], acc1, 0, 0, 0);
          }
          bw[0] |= (uint32_t)__shfl_xor((int)bw[0], 32, 64);
          bw[1] |= (uint32_t)__shfl_xor((int)bw[1], 32, 64);
          if (h == 0) *reinterpret_cast<pu32x2*>(XB + 2 * l31) = pu32x2{bw[0], bw[1]};
        }
      }
      // the row registers are consumed: the next tile's rows are requested now and have the rest of this tile to arrive
      __builtin_amdgcn_sched_barrier(0);
      load_rows(m1, t + n_waves, ra, rb, &dg0, &dv0);
      __builtin_amdgcn_sched_barrier(0);
      if (any) {
        // ---- epilogue of layer 2 in the C layout: lane = unit u (l31), reg r = pair row crow(r,h)
        float d2c[16];
        if constexpr (SAVED) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = crow(r, h);
            const float post = T2[row * LDD + l31];
            const float dr = DOs[row];
            float d2 = dr * w3v * (post > 0.f ? inv_keep : 0.f);
            asm volatile("" : "+v"(d2));               // d2 is ROUNDED here, in both variants (see the other one)
            d2c[r] = d2;
            w3acc = fmaf(dr, post, w3acc);
            b2acc += d2;
            D2s[row * LDD + l31] = d2;
          }
        } else {
          uint32_t kb[16];
          if (drop_p > 0.f) layer2_fields(mmg_rng_key(seed, SITE_H2), PLo, PHi, h, l31, kb);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = crow(r, h);
            const float pre = acc1[r] + b2v;
            float m = pre > 0.f ? 1.f : 0.f;
            float post = fmaxf(pre, 0.f);
            if (drop_p > 0.f) {
              const bool kp = kb[r] >= thr_keep;
              m = kp ? m * inv_keep : 0.f;
              post = kp ? post * inv_keep : 0.f;
            }
            const float dr = DOs[row];
            float d2 = dr * w3v * m;
            // d2 is ROUNDED here, in both variants: left to the compiler, the product is contracted into the additions and
            // subtractions that consume it (bias sum, piece residuals) in one variant and not in the other, and the saved and
            // the recomputing kernel stop agreeing bit for bit
            asm volatile("" : "+v"(d2));
            d2c[r] = d2;
            w3acc = fmaf(dr, post, w3acc);
            b2acc += d2;
            D2s[row * LDD + l31] = d2;
          }
        }
        // ---- (2) dW2[u,k] += D2[pair,u] * H1[pair,k]: A = d2c pieces (lane = u), B = h1 pieces of column ct * 32 + l31
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
          pbf16x8 ap[3];
          psplit8(d2c + 8 * t2, ap[0], ap[1], ap[2]);
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) {
            pbf16x8 bp[3];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
              const __bf16* base = P1 + (pc * TP + 16 * t2 + 4 * h + trq) * P1S + ct * 32 + trc;
              bp[pc] = tr_pair(base, base + 8 * P1S);
            }
            MMG_X6(accW2[ct], ap, bp);
          }
        }
      }
    }
    __syncthreads();
  }
  // ---- final flush, front half: the four front waves' dW2 summed through LDS in wave order (the back waves add their dB
  //      slots in the same four rounds); every wave passed the last barrier of the loop
  for (int ww = 0; ww < 4; ++ww) {
    if (w == ww) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (ww == 0) { red[r * 64 + lane] = accW2[0][r]; red[(16 + r) * 64 + lane] = accW2[1][r]; }
        else { red[r * 64 + lane] += accW2[0][r]; red[(16 + r) * 64 + lane] += accW2[1][r]; }
      }
    }
    __syncthreads();
  }
  w3acc += __shfl_xor(w3acc, 32, 64);
  b2acc += __shfl_xor(b2acc, 32, 64);
  b3acc = wave_sum(b3acc);
  if (lane < 32) { tail_red[w][lane] = w3acc; tail_red[w][32 + lane] = b2acc; }
  if (lane == 0) tail_red[w][64] = b3acc;
}

template <int LT>
__device__ __forceinline__ void pair_bwd6_back(const HeadDev& H, float* __restrict__ dA, float drop_p, int n_iter,
                                               unsigned char* bl, unsigned char (*HX)[4][H6_LDS], int (*XP)[4][TP],
                                               int (*XL)[4][TP], float* red) {
  const int tid = threadIdx.x, lane = tid & 63, w = (tid >> 6) & 3;
  const int h = lane >> 5, l31 = lane & 31;
  const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  int* RI = reinterpret_cast<int*>(bl);             // [32] run id of a pair (-1: not a pair of this head)
  int* RP = RI + TP;                                // [32] patient of a run
  // B of (3): W2[u = 16 kq + 8 h + j][k = ct * 32 + l31] as three exact bf16 pieces
  pbf16x8 w2t[2][2][3];
#pragma unroll
  for (int kq = 0; kq < 2; ++kq)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = H.W2[(16 * kq + 8 * h + j) * 64 + ct * 32 + l31];
      psplit8(v, w2t[kq][ct][0], w2t[kq][ct][1], w2t[kq][ct][2]);
    }
  f32x16 accB[LT][2];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int t = 0; t < LT; ++t) { accB[t][0][i] = 0.f; accB[t][1][i] = 0.f; }
  for (int it = 0; it <= n_iter; ++it) {
    if (it >= 1) {
      const int par = (it - 1) & 1;
      const int p_i = XP[par][w][l31];
      if (__ballot(p_i >= 0) != 0ull) {
        const float* D2s = reinterpret_cast<const float*>(HX[par][w]);    // [32][LDD]
        const unsigned* XB = reinterpret_cast<const unsigned*>(D2s + TP * LDD);
        // ---- runs of equal patients among the tile's pairs (sorted by patient; pairs of the other head lie between):
        //      pair j starts a run when the nearest pair of this head before it has another patient
        const unsigned am = (unsigned)__ballot(p_i >= 0);                  // (both lane halves hold the same 32 pairs)
        const unsigned below = am & ((1u << l31) - 1u);
        const int p_prev = __shfl(p_i, below ? 31 - __clz((int)below) : 0, 64);
        const bool start = p_i >= 0 && (below == 0u || p_prev != p_i);
        const unsigned sm = (unsigned)__ballot(start);
        const int rid = p_i >= 0 ? __popc(sm & ((2u << l31) - 1u)) - 1 : -1;
        const int n_runs = __popc(sm);
        if (h == 0) { RI[l31] = rid; if (start) RP[rid] = p_i; }
        f32x16 accR[2];                                                    // dA partial sums: [run][column]
#pragma unroll
        for (int i = 0; i < 16; ++i) { accR[0][i] = 0.f; accR[1][i] = 0.f; }
        // ---- (3) dH1[pair,k] = D2[pair,:] . W2[:,k]: A = the pair's D2 row (u = 16 kq + 8 h + j), split here
        f32x16 accH[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) { accH[0][i] = 0.f; accH[1][i] = 0.f; }
#pragma unroll
        for (int kq = 0; kq < 2; ++kq) {
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(D2s + l31 * LDD + 16 * kq + 8 * h);
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(D2s + l31 * LDD + 16 * kq + 8 * h + 4);
          const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          pbf16x8 ap[3];
          psplit8(v, ap[0], ap[1], ap[2]);
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) MMG_X6(accH[ct], ap, w2t[kq][ct]);
        }
        // through dropout + relu of layer 1: the sign bit of h1[pair crow(r,h)][ct * 32 + l31]
        float dh[16][2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const pu32x2 bwv = *reinterpret_cast<const pu32x2*>(XB + 2 * crow(r, h));
          dh[r][0] = (bwv[0] >> l31) & 1u ? accH[0][r] * inv_keep : 0.f;
          dh[r][1] = (bwv[1] >> l31) & 1u ? accH[1][r] * inv_keep : 0.f;
        }
        // ---- (4) dB[lab,k] += onehot(li[pair])[lab] * dH1[pair,k] (exact: one-hot x three bf16 pieces of dH1)
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
          pbf16x8 bp[2][3];
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = dh[8 * t2 + j][ct];
            psplit8(v, bp[ct][0], bp[ct][1], bp[ct][2]);
          }
          int labs[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) labs[j] = XL[par][w][crow(8 * t2 + j, h)];
#pragma unroll
          for (int lt = 0; lt < LT; ++lt) {
            pbf16x8 oh;
#pragma unroll
            for (int j = 0; j < 8; ++j) oh[j] = (labs[j] == lt * 32 + l31) ? (__bf16)1.0f : (__bf16)0.0f;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
              for (int p = 0; p < 3; ++p)
                accB[lt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(oh, bp[ct][p], accB[lt][ct], 0, 0, 0);
          }
          // ---- dA[pi] += dH1, run by run, the same way: [run == row]^T . dH1 (one-hot x the same three pieces)
          pbf16x8 ohr;
#pragma unroll
          for (int j = 0; j < 8; ++j) ohr[j] = (RI[crow(8 * t2 + j, h)] == l31) ? (__bf16)1.0f : (__bf16)0.0f;
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int p = 0; p < 3; ++p)
              accR[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ohr, bp[ct][p], accR[ct], 0, 0, 0);
        }
        // one atomic per (run, 32 columns): register r holds run row crow(r, h) of column ct * 32 + l31.  (A patient whose
        // pairs straddle two tiles receives two partial sums -- a + b in either order; more than two only beyond 32 pairs.)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (crow(r, 0) >= n_runs) continue;                              // wave-uniform: neither half has such a run
          const int row = crow(r, h);
          if (row < n_runs) {
            float* dst = dA + (size_t)RP[row] * 64 + l31;
            atomicAdd(dst, accR[0][r]);
            atomicAdd(dst + 32, accR[1][r]);
          }
        }
      }
    }
    __syncthreads();
  }
  for (int ww = 0; ww < 4; ++ww) {
    if (w == ww) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int lt = 0; lt < LT; ++lt) {
          if (ww == 0) {
            red[(32 + lt * 32 + r) * 64 + lane] = accB[lt][0][r];
            red[(48 + lt * 32 + r) * 64 + lane] = accB[lt][1][r];
          } else {
            red[(32 + lt * 32 + r) * 64 + lane] += accB[lt][0][r];
            red[(48 + lt * 32 + r) * 64 + lane] += accB[lt][1][r];
          }
        }
      }
    }
    __syncthreads();
  }
}

template <int LT, bool AUX, bool SAVED>
__global__ __launch_bounds__(512) void k_pair_bwd_duo6(HeadDev H, HeadGradDev Gd, const int32_t* __restrict__ pi,
                                                       const int32_t* __restrict__ li, const int32_t* __restrict__ deg,
                                                       int thr, int want_low, int64_t n, int n_labs, float drop_p,
                                                       uint64_t seed, const uint64_t* __restrict__ seed_ptr, PairBufs pb,
                                                       const float* __restrict__ dpred, const int32_t* __restrict__ sel,
                                                       const int32_t* __restrict__ n_sel, float* __restrict__ slab,
                                                       const uint32_t* __restrict__ sv_bits, const float* __restrict__ sv_h2,
                                                       int sv_by_pos) {
  if (seed_ptr) seed = *seed_ptr;
  if (sel) { const int64_t nl = *n_sel; n = nl < 0 ? 0 : (nl < n ? nl : n); }
  __shared__ __attribute__((aligned(16))) unsigned char smf[4][F6_LDS];      // front-private (later: the reduction tile)
  __shared__ __attribute__((aligned(16))) unsigned char smb[4][B6_LDS];      // back-private
  __shared__ __attribute__((aligned(16))) unsigned char HX[2][4][H6_LDS];    // hand-off: [buffer][wave pair]
  __shared__ int XP[2][4][TP], XL[2][4][TP];
  __shared__ float tail_red[4][68];
  const int tid = threadIdx.x, wid = tid >> 6;
  const int64_t n_tiles = (n + TP - 1) / TP, n_waves = (int64_t)gridDim.x * 4, first = (int64_t)blockIdx.x * 4;
  const int n_iter = first < n_tiles ? (int)((n_tiles - first + n_waves - 1) / n_waves) : 0;
  float* red = reinterpret_cast<float*>(&smf[0][0]);
  constexpr int NR = (2 + 2 * LT) * 16;
  static_assert(NR * 64 * 4 <= 4 * F6_LDS, "reduction tile must fit the front waves' buffers");
  if (wid < 4)
    pair_bwd6_front<LT, AUX, SAVED>(H, pi, li, deg, thr, want_low, n, drop_p, seed, pb, dpred, sel, n_iter, smf[wid & 3], HX,
                                    XP, XL, tail_red, red, sv_bits, sv_h2, sv_by_pos);
  else
    pair_bwd6_back<LT>(H, Gd.dA, drop_p, n_iter, smb[wid & 3], HX, XP, XL, red);
  float* my = slab + (size_t)blockIdx.x * pair_slab_floats(LT);
  for (int e = tid; e < NR * 64; e += 512) {
    const int slot = e >> 6, ln = e & 63;
    const float v = red[e];
    const int hh = ln >> 5, c31 = ln & 31;
    if (slot < 32) {
      const int ct = slot >> 4, r = slot & 15;
      my[crow(r, hh) * 64 + ct * 32 + c31] = v;
    } else {
      const int q = slot - 32, lt = q >> 5, ct = (q >> 4) & 1, r = q & 15;
      my[2048 + (lt * 32 + crow(r, hh)) * 64 + ct * 32 + c31] = v;
    }
  }
  __syncthreads();                                   // the front waves' tail sums are in tail_red
  if (tid < 68) {
    const float t = tid < 65 ? ((tail_red[0][tid] + tail_red[1][tid]) + tail_red[2][tid]) + tail_red[3][tid] : 0.f;
    const int dst = tid < 32 ? 32 + tid : (tid < 64 ? tid - 32 : tid);
    my[2048 + LT * 2048 + dst] = t;
  }
}

// adds the summed slab into the caller's gradient buffers (single writer per element: plain read-modify-write)
struct EpiPairFlush {
  float *dW2, *dB, *db2, *dW3, *db3;
  int n_labs, LT;
  __device__ void operator()(int64_t i4, mmg_f4 v) const {
    const int i = (int)i4 * 4;
    if (i < 2048) {
      mmg_f4* o = reinterpret_cast<mmg_f4*>(dW2 + i);
      *o = *o + v;
    } else if (i < 2048 + LT * 2048) {
      const int j = i - 2048;
      if (j / 64 < n_labs) {
        mmg_f4* o = reinterpret_cast<mmg_f4*>(dB + j);
        *o = *o + v;
      }
    } else {
      const int t = i - (2048 + LT * 2048);
      if (t < 32) { for (int q = 0; q < 4; ++q) db2[t + q] += v[q]; }
      else if (t < 64) { for (int q = 0; q < 4; ++q) dW3[t - 32 + q] += v[q]; }
      else if (t == 64) db3[0] += v[0];
    }
  }
};

// ---------------------------------------------------------------------------- forward on MFMA
// One wave per 32 pairs.  The layer-2 product is computed TRANSPOSED, H2pre^T[unit, pair] = W2 . H1^T (A = W2 rows in
// registers, B = the pair's own h1 values): the accumulator then has lane = pair and registers = 16 of the 32 units
// (the other 16 sit in lane + 32), so everything after the matrix product is lane-local -- bias, ReLU, the dropout
// mask (registers 4i .. 4i+3 are one aligned RNG group: one hash each), the 32 -> 1 output layer as 16 FMAs -- and
// one cross-half add finishes a prediction.  No LDS, no shuffles, no barrier.
constexpr int PF_LDB = 68;           // LDS row stride of the lab-side table (floats): rows land 4 banks apart
// SAVE (training): the forward leaves what the backward would otherwise recompute per pair k (mmg_pair_saved_t):
//   sv_bits[2 k + w]  bit j = [h1[32 w + j] > 0]  (kept by dropout AND positive: h1 = bit ? (A + B) / (1 - p) : 0 exactly)
//   sv_h2[32 k + u]   the layer-2 activation after ReLU and dropout (its sign pattern is the layer-2 mask)
// 136 B per pair instead of 12 RNG hashes, a 64 x 32 matrix product and its epilogue per pair in the backward.
template <bool B_LDS, bool SAVE>
__global__ __launch_bounds__(256, 2) void k_pair_fwd_mfma(HeadDev H, const int32_t* __restrict__ pi,
                                                          const int32_t* __restrict__ li, const int32_t* __restrict__ deg,
                                                          int thr, int want_low, int64_t n, float drop_p, uint64_t seed,
                                                          const uint64_t* __restrict__ seed_ptr,
                                                          PairBufs pb, float* __restrict__ pred,
                                                          const int32_t* __restrict__ sel,
                                                          const int32_t* __restrict__ n_sel,
                                                          int n_labs_lds, uint32_t* __restrict__ sv_bits,
                                                          float* __restrict__ sv_h2, int sv_by_pos) {
  if (seed_ptr) seed = *seed_ptr;
  if (sel) { const int64_t nl = *n_sel; n = nl < 0 ? 0 : (nl < n ? nl : n); }     // never beyond the list's capacity
  // the lab-side first-layer table B (a few dozen 256-B rows, read once per PAIR) lives in LDS when it fits:
  // 4.3 M pairs x 256 B would otherwise stream from L2
  extern __shared__ __attribute__((aligned(16))) float Bs[];
  if (B_LDS) {
    for (int i = threadIdx.x; i < n_labs_lds * 16; i += 256)
      *reinterpret_cast<f32x4*>(Bs + (i >> 4) * PF_LDB + (i & 15) * 4) = *reinterpret_cast<const f32x4*>(H.B + (size_t)i * 4);
    __syncthreads();
  }
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  const uint32_t thr_keep = mmg_keep_threshold(drop_p);
  const uint32_t key1 = mmg_rng_key(seed, SITE_H1), key2 = mmg_rng_key(seed, SITE_H2);
  // A operand: W2[unit = l31][k = 16 ks + 8 h + j] as three exact bf16 pieces (the 6-term split of gemm.hip: the
  // fp32 matrix instruction shares the vector ALU's multipliers -- its 2048 cycles per tile ADD to the VALU time of
  // this VALU-heavy kernel -- while the bf16 matrix pipe runs beside it)
  pbf16x8 w2p[4][3];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = H.W2[l31 * 64 + 16 * ks + 8 * h + j];
      const __bf16 a = (__bf16)v;
      const float r1 = v - (float)a;
      const __bf16 b = (__bf16)r1;
      w2p[ks][0][j] = a; w2p[ks][1][j] = b; w2p[ks][2][j] = (__bf16)(r1 - (float)b);
    }
  float b2r[16], w3r[16];            // this lane's 16 units: crow(r, h)
#pragma unroll
  for (int r = 0; r < 16; ++r) { b2r[r] = H.b2[crow(r, h)]; w3r[r] = H.W3[crow(r, h)]; }
  const float b3 = H.b3[0];

  const int64_t n_tiles = (n + TP - 1) / TP;
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + wid, n_waves = (int64_t)gridDim.x * 4;
  // Three-deep software pipeline over this wave's tiles, one dependent load per stage, so that no load is waited
  // for in the iteration that issues it:   tile t+3: list position -> pair index k   (sel[.])
  //                                        tile t+2: k -> patient, lab, rng id, output slot
  //                                        tile t+1: patient -> gate degree, A row;  lab -> B row
  // Every load of the pipeline is unconditional and BOUNDED: lists, index arrays, node tables and the output all go
  // through buffer descriptors sized on the host (PairBufs; zero-sized for an absent array: the load returns 0 and a
  // select substitutes the default).  A branch around a load makes the compiler's vmcnt waits conservative (minimum
  // over both paths) and the first version of this loop drained the whole pipeline twice per tile.
  struct Meta { int k; int p_i; int l_i; int o; uint64_t pid; };
  const __amdgpu_buffer_rsrc_t sel_d = pair_rsrc(sel ? sel : pi, sel ? (uint32_t)(n * 4) : 0u);
  const __amdgpu_buffer_rsrc_t io_d = pair_rsrc(pb.io, pb.io_bytes), pid_d = pair_rsrc(pb.pid, pb.pid_bytes);
  const __amdgpu_buffer_rsrc_t pi_d = pair_rsrc(pi, pb.pair_bytes), li_d = pair_rsrc(li, pb.pair_bytes);
  const __amdgpu_buffer_rsrc_t pred_d = pair_rsrc(pred, pb.pair_bytes), deg_d = pair_rsrc(deg, pb.pat_bytes);
  const __amdgpu_buffer_rsrc_t A_d = pair_rsrc(H.A, pb.a_bytes), B_d = pair_rsrc(H.B, pb.b_bytes);
  const bool has_sel = sel != nullptr, has_io = pb.io_bytes != 0u, has_pid = pb.pid_bytes != 0u;
  // A stage only ISSUES loads; what it loaded is finalised (selects, clamps) one iteration later by the next stage, so
  // nothing is waited for in the iteration that issued it.
  struct RawMeta { int k, p, l; pu32x2 o2, d2; };
  auto issue_k = [&](int64_t t) {                    // raw list entry of tile t (0 past the end / without a list)
    const int64_t idx = t * TP + l31;
    return pair_ld_i32(sel_d, (unsigned)(idx < n ? idx : 0) * 4u);
  };
  auto fin_k = [&](int kr, int64_t t) {
    const int64_t idx = t * TP + l31;
    return idx < n ? (has_sel ? kr : (int)idx) : -1;
  };
  auto issue_meta = [&](int k) {
    const unsigned kc = k >= 0 ? (unsigned)k : 0u;
    RawMeta r;
    r.k = k;
    r.p = pair_ld_i32(pi_d, kc * 4u);
    r.l = pair_ld_i32(li_d, kc * 4u);
    r.o2 = __builtin_amdgcn_raw_buffer_load_b64(io_d, (int)(kc * 8u), 0, 0);
    r.d2 = __builtin_amdgcn_raw_buffer_load_b64(pid_d, (int)(kc * 8u), 0, 0);
    return r;
  };
  auto fin_meta = [&](const RawMeta& r) {
    const int kc = r.k >= 0 ? r.k : 0;
    Meta m;
    m.k = r.k;
    // a list entry outside the pair arrays, or a row id outside the table, is not a pair (nothing is read or written for it)
    m.p_i = ((unsigned)r.k < (pb.pair_bytes >> 2) && (unsigned)r.p < (unsigned)pb.n_pat) ? r.p : -1;
    m.l_i = r.l;
    m.o = has_io ? (int)r.o2[0] : kc;
    m.pid = has_pid ? ((uint64_t)r.d2[1] << 32 | r.d2[0]) : (uint64_t)kc;
    return m;
  };
  auto load_rows = [&](const Meta& m, f32x4* ra, f32x4* rb, int* dg) {
    const unsigned pp = m.p_i >= 0 ? (unsigned)m.p_i : 0u;
    *dg = pair_ld_i32(deg_d, pp * 4u);
    const unsigned ao = pp * 256u + 32u * h, bo = (unsigned)m.l_i * 256u + 32u * h;
#pragma unroll
    for (int q = 0; q < 8; ++q) {              // q = 2 ks + half-chunk: floats 16 ks + 8 h + 4 (q & 1) ..
      ra[q] = pair_ld_f4(A_d, ao + (q >> 1) * 64u + (q & 1) * 16u);
      if (!B_LDS) rb[q] = pair_ld_f4(B_d, bo + (q >> 1) * 64u + (q & 1) * 16u);
    }
  };
  const int kr0 = issue_k(wave_id), kr1 = issue_k(wave_id + n_waves);
  int kr2 = issue_k(wave_id + 2 * n_waves);
  const RawMeta rm0 = issue_meta(fin_k(kr0, wave_id));
  RawMeta rm1 = issue_meta(fin_k(kr1, wave_id + n_waves));
  Meta m0 = fin_meta(rm0);
  f32x4 ra[8], rb[8];
  int dg0;
  load_rows(m0, ra, rb, &dg0);
  for (int64_t t = wave_id; t < n_tiles; t += n_waves) {
    const Meta mc = m0;
    const bool active = mc.p_i >= 0 && ((int)(dg0 < thr)) == want_low;
    f32x4 ca[8], cb[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { ca[q] = ra[q]; cb[q] = rb[q]; }
    const Meta m1 = fin_meta(rm1);                   // issued one iteration ago
    const int k2 = fin_k(kr2, t + 2 * n_waves);
    kr2 = issue_k(t + 3 * n_waves);
    rm1 = issue_meta(k2);
    load_rows(m1, ra, rb, &dg0);
    __builtin_amdgcn_sched_barrier(0);         // the three stages' loads stay ahead of this tile's arithmetic
    m0 = m1;
    if (__ballot(active) == 0ull) continue;
    if (B_LDS) {
      const int lc = (unsigned)mc.l_i < (unsigned)n_labs_lds ? mc.l_i : 0;       // (a lab id outside the staged table)
      const float* bl = Bs + lc * PF_LDB + 8 * h;
#pragma unroll
      for (int q = 0; q < 8; ++q) cb[q] = *reinterpret_cast<const f32x4*>(bl + (q >> 1) * 16 + (q & 1) * 4);
    }
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    uint32_t bw[2] = {0u, 0u};                       // SAVE: this lane's 2 x 16 sign bits of h1
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {                 // 16 h1 columns per k-step, this lane: 16 ks + 8 h + 0..7
      float x8[8];
#pragma unroll
      for (int c = 0; c < 2; ++c) {                  // one aligned RNG group of 4 per chunk: one hash
        f32x4 x;
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = fmaxf(ca[2 * ks + c][j] + cb[2 * ks + c][j], 0.f);
        if (drop_p > 0.f) mmg_drop4(x, key1, mc.pid * 64ull + (uint64_t)(16 * ks + 8 * h + 4 * c), thr_keep, inv_keep);
#pragma unroll
        for (int j = 0; j < 4; ++j) x8[4 * c + j] = x[j];
      }
      if constexpr (SAVE) {
#pragma unroll
        for (int j = 0; j < 8; ++j) bw[ks >> 1] |= x8[j] > 0.f ? (1u << (16 * (ks & 1) + j)) << (8 * h) : 0u;
      }
      pbf16x8 x1, x2, x3;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const __bf16 a = (__bf16)x8[j];
        const float r1 = x8[j] - (float)a;
        const __bf16 b = (__bf16)r1;
        x1[j] = a; x2[j] = b; x3[j] = (__bf16)(r1 - (float)b);
      }
      // C^T: lane = pair, reg = unit.  W2 . h1^T, six exact products, small terms first
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2p[ks][2], x1, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2p[ks][0], x3, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2p[ks][1], x2, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2p[ks][1], x1, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2p[ks][0], x2, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2p[ks][0], x1, acc, 0, 0, 0);
    }
    // lane = pair l31, register r = unit crow(r, h) = (r & 3) + 8 (r >> 2) + 4 h
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {                    // units 8 i + 4 h + 0..3: one aligned RNG group
      f32x4 post;
#pragma unroll
      for (int e = 0; e < 4; ++e) post[e] = fmaxf(acc[4 * i + e] + b2r[4 * i + e], 0.f);
      if (drop_p > 0.f) mmg_drop4(post, key2, mc.pid * 32ull + (uint64_t)(8 * i + 4 * h), thr_keep, inv_keep);
#pragma unroll
      for (int e = 0; e < 4; ++e) part = fmaf(w3r[4 * i + e], post[e], part);
      if constexpr (SAVE) {
        if (active) *reinterpret_cast<f32x4*>(sv_h2 + (size_t)(sv_by_pos ? t * TP + l31 : (int64_t)mc.k) * 32 + 8 * i + 4 * h) = post;
      }
    }
    if constexpr (SAVE) {
      bw[0] |= (uint32_t)__shfl_xor((int)bw[0], 32, 64);
      bw[1] |= (uint32_t)__shfl_xor((int)bw[1], 32, 64);
      if (h == 0 && active)
        *reinterpret_cast<pu32x2*>(sv_bits + (size_t)(sv_by_pos ? t * TP + l31 : (int64_t)mc.k) * 2) = pu32x2{bw[0], bw[1]};
    }
    part += __shfl_xor(part, 32, 64);
    if (h == 0 && active)          // (a slot past the end of pred is dropped by the descriptor's range check)
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, part + b3), pred_d, (int)((unsigned)mc.o * 4u), 0, 0);
  }
}

// ---------------------------------------------------------------------------- pair selection (stable compaction)
constexpr int SEL_PER = 8, SEL_CH = 256 * SEL_PER;      // pairs per thread / per workgroup

// packed per-thread counts: low list in bits 0..15, high list in bits 16..31 (a chunk holds 2048 pairs)
// dps (nullable): the upstream gradient in SORTED pair order.  The count pass fills it (the one random pass through
// io_perm); the write pass and the backward kernel then read it sequentially.
template <bool FILL>
__device__ inline unsigned sel_flags(const int32_t* __restrict__ pi, const int32_t* __restrict__ deg, int thr,
                                     const float* __restrict__ dpred, const int64_t* __restrict__ io,
                                     float* __restrict__ dps, int64_t n, int64_t k0, unsigned* bits) {
  unsigned cnt = 0, lowbits = 0, anybits = 0;
  // all loads of a thread's 8 pairs are issued before any is used (indices clamped, no branches in between):
  // the gather through io_perm is a chain of two dependent random loads per pair
  const int64_t kl = n - 1;
  float d[SEL_PER];
  int dg[SEL_PER];
  int64_t src[SEL_PER];
#pragma unroll
  for (int j = 0; j < SEL_PER; ++j) {
    const int64_t k = k0 + j < kl ? k0 + j : kl;
    src[j] = (dpred && io && (FILL || !dps)) ? io[k] : k;
  }
#pragma unroll
  for (int j = 0; j < SEL_PER; ++j) {
    const int64_t k = k0 + j < kl ? k0 + j : kl;
    d[j] = !dpred ? 1.f : ((FILL || !dps) ? dpred[src[j]] : dps[k]);
    dg[j] = deg[pi[k]];
  }
#pragma unroll
  for (int j = 0; j < SEL_PER; ++j) {
    const int64_t k = k0 + j;
    if (k < n && FILL && dpred && dps) dps[k] = d[j];
    if (k < n && d[j] != 0.f) {
      const bool low = dg[j] < thr;
      anybits |= 1u << j;
      if (low) { lowbits |= 1u << j; cnt += 1u; } else cnt += 1u << 16;
    }
  }
  *bits = anybits | (lowbits << 8);
  return cnt;
}

__device__ inline unsigned block_excl_scan(unsigned v, unsigned* total, unsigned* sm /*[4]*/) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  unsigned inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) sm[wid] = inc;
  __syncthreads();
  unsigned base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) { if (w < wid) base += sm[w]; tot += sm[w]; }
  *total = tot;
  return base + inc - v;
}

__global__ __launch_bounds__(256) void k_sel_count(const int32_t* __restrict__ pi, const int32_t* __restrict__ deg,
                                                   int thr, const float* __restrict__ dpred,
                                                   const int64_t* __restrict__ io, float* __restrict__ dps, int64_t n,
                                                   unsigned* __restrict__ cnt) {
  __shared__ unsigned sm[4];
  unsigned bits, tot;
  const unsigned c = sel_flags<true>(pi, deg, thr, dpred, io, dps, n,
                                     (int64_t)blockIdx.x * SEL_CH + threadIdx.x * SEL_PER, &bits);
  block_excl_scan(c, &tot, sm);
  if (threadIdx.x == 0) cnt[blockIdx.x] = tot;
}

// one workgroup: exclusive scan of the packed chunk counts (two 32-bit running sums), totals -> counts[2]
__global__ __launch_bounds__(1024) void k_sel_scan(const unsigned* __restrict__ cnt, int nb, int2* __restrict__ base,
                                                   int32_t* __restrict__ counts) {
  __shared__ int2 wsum[16];
  __shared__ int2 carry;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) carry = make_int2(0, 0);
  __syncthreads();
  for (int b0 = 0; b0 < nb; b0 += 1024) {
    const int b = b0 + tid;
    const unsigned c = b < nb ? cnt[b] : 0u;
    int lo = (int)(c & 0xFFFFu), hi = (int)(c >> 16);
    int ilo = lo, ihi = hi;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int tl = __shfl_up(ilo, o, 64), th = __shfl_up(ihi, o, 64);
      if (lane >= o) { ilo += tl; ihi += th; }
    }
    if (lane == 63) wsum[wid] = make_int2(ilo, ihi);
    __syncthreads();
    int bl = carry.x, bh = carry.y, tl = 0, th = 0;
    for (int w = 0; w < 16; ++w) {
      if (w < wid) { bl += wsum[w].x; bh += wsum[w].y; }
      tl += wsum[w].x; th += wsum[w].y;
    }
    if (b < nb) base[b] = make_int2(bl + ilo - lo, bh + ihi - hi);
    __syncthreads();
    if (tid == 0) { carry.x += tl; carry.y += th; }
    __syncthreads();
  }
  if (tid == 0) { counts[0] = carry.x; counts[1] = carry.y; }
}

__global__ __launch_bounds__(256) void k_sel_write(const int32_t* __restrict__ pi, const int32_t* __restrict__ deg,
                                                   int thr, const float* __restrict__ dpred,
                                                   const int64_t* __restrict__ io, float* __restrict__ dps, int64_t n,
                                                   const int2* __restrict__ base, int32_t* __restrict__ sel_low,
                                                   int32_t* __restrict__ sel_high) {
  __shared__ unsigned sm[4];
  unsigned bits, tot;
  const int64_t k0 = (int64_t)blockIdx.x * SEL_CH + threadIdx.x * SEL_PER;
  const unsigned c = sel_flags<false>(pi, deg, thr, dpred, io, dps, n, k0, &bits);
  const unsigned off = block_excl_scan(c, &tot, sm);
  const int2 b = base[blockIdx.x];
  int ol = b.x + (int)(off & 0xFFFFu), oh = b.y + (int)(off >> 16);
#pragma unroll
  for (int j = 0; j < SEL_PER; ++j) {
    if (bits & (1u << j)) {
      if (bits & (1u << (8 + j))) sel_low[ol++] = (int32_t)(k0 + j);
      else sel_high[oh++] = (int32_t)(k0 + j);
    }
  }
}

inline unsigned pair_grid(int64_t n) {
  int64_t t = (n + PT - 1) / PT;
  if (t > 1024) t = 1024;
  if (t < 1) t = 1;
  return (unsigned)t;
}

int check_head(const mmg_head_t* h, const char* what) {
  MMG_CHECK_ARG(h && h->A && h->B && h->W2 && h->b2 && h->W3 && h->b3, "%s: head has a null pointer", what);
  return MMG_OK;
}

}  // namespace

// host side of PairBufs: sizes are checked against the 32-bit byte offsets of the descriptors
static int fill_pair_bufs(PairBufs* pb, const int32_t* pi, const int64_t* pair_id, const int64_t* io_perm, int64_t n_total,
                          int64_t n_patients, int n_labs, const char* what) {
  MMG_CHECK_ARG(n_total >= 1 && n_total < (1ll << 29), "%s: %lld pairs outside [1, 2^29) (32-bit buffer offsets)", what,
                (long long)n_total);
  MMG_CHECK_ARG(n_patients >= 1 && n_patients < (1ll << 24), "%s: %lld patient rows outside [1, 2^24)", what,
                (long long)n_patients);
  MMG_CHECK_ARG(n_labs >= 1 && n_labs < (1 << 24), "%s: %d lab rows outside [1, 2^24)", what, n_labs);
  pb->pid = pair_id ? (const void*)pair_id : (const void*)pi;
  pb->io = io_perm ? (const void*)io_perm : (const void*)pi;
  pb->pid_bytes = pair_id ? (uint32_t)(n_total * 8) : 0u;
  pb->io_bytes = io_perm ? (uint32_t)(n_total * 8) : 0u;
  pb->pair_bytes = (uint32_t)(n_total * 4);
  pb->pat_bytes = (uint32_t)(n_patients * 4);
  pb->a_bytes = (uint32_t)(n_patients * 256);
  pb->b_bytes = (uint32_t)n_labs * 256u;
  pb->n_pat = (int32_t)n_patients;
  return MMG_OK;
}

extern "C" int mmg_pair_head_fwd_save(const mmg_head_t* head, const int32_t* pi, const int32_t* li, const int32_t* deg,
                                      int degree_threshold, int want_low, int64_t n_pairs, int64_t n_total,
                                      int64_t n_patients, int n_labs, float drop_p, uint64_t seed, const uint64_t* seed_ptr,
                                      const int64_t* pair_id, float* pred, const int32_t* sel, const int32_t* n_sel,
                                      const int64_t* io_perm, const mmg_pair_saved_t* saved, void* stream);

extern "C" int mmg_pair_head_fwd(const mmg_head_t* head, const int32_t* pi, const int32_t* li, const int32_t* deg,
                                 int degree_threshold, int want_low, int64_t n_pairs, int64_t n_total, int64_t n_patients,
                                 int n_labs, float drop_p, uint64_t seed, const uint64_t* seed_ptr, const int64_t* pair_id,
                                 float* pred, const int32_t* sel, const int32_t* n_sel, const int64_t* io_perm, void* stream) {
  return mmg_pair_head_fwd_save(head, pi, li, deg, degree_threshold, want_low, n_pairs, n_total, n_patients, n_labs, drop_p,
                                seed, seed_ptr, pair_id, pred, sel, n_sel, io_perm, nullptr, stream);
}

extern "C" int mmg_pair_head_fwd_save(const mmg_head_t* head, const int32_t* pi, const int32_t* li, const int32_t* deg,
                                      int degree_threshold, int want_low, int64_t n_pairs, int64_t n_total,
                                      int64_t n_patients, int n_labs, float drop_p, uint64_t seed, const uint64_t* seed_ptr,
                                      const int64_t* pair_id, float* pred, const int32_t* sel, const int32_t* n_sel,
                                      const int64_t* io_perm, const mmg_pair_saved_t* saved, void* stream) {
  MMG_CHECK_ARG(!saved || (saved->h1_bits && saved->h2), "pair_head_fwd_save: saved buffers with a null pointer");
  MMG_CHECK_ARG(n_pairs >= 0, "pair_head_fwd: n_pairs < 0");
  MMG_CHECK_ARG((sel == nullptr) == (n_sel == nullptr), "pair_head_fwd: sel and n_sel go together");
  if (n_pairs == 0) return MMG_OK;
  int rc = check_head(head, "pair_head_fwd");
  if (rc) return rc;
  MMG_CHECK_ARG(pi && li && deg && pred, "pair_head_fwd: null buffer");
  MMG_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "pair_head_fwd: drop_p out of range");
  MMG_CHECK_ARG(n_pairs <= n_total, "pair_head_fwd: %lld listed pairs of %lld", (long long)n_pairs, (long long)n_total);
  PairBufs pb;
  rc = fill_pair_bufs(&pb, pi, pair_id, io_perm, n_total, n_patients, n_labs, "pair_head_fwd");
  if (rc) return rc;
  HeadDev H{head->A, head->B, head->W2, head->b2, head->W3, head->b3};
  int64_t g = ((n_pairs + TP - 1) / TP + 3) / 4;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  const int n_labs_lds = n_labs <= 256 ? n_labs : 0;       // 256 rows x 272 B = 68 KB: two workgroups per CU
  const size_t lds = (size_t)n_labs_lds * PF_LDB * sizeof(float);
  uint32_t* svb = saved ? saved->h1_bits : nullptr;
  float* svh = saved ? saved->h2 : nullptr;
  const int sv_pos = saved && saved->by_position ? 1 : 0;
  MMG_CHECK_ARG(!sv_pos || sel, "pair_head_fwd_save: by_position needs a pair list");
  MMG_CHECK_ARG(!saved || saved->n_entries >= (sv_pos ? n_pairs : n_total),
                "pair_head_fwd_save: the saved buffers are shorter than the entries this launch can write");
  constexpr int lds_max = 256 * PF_LDB * (int)sizeof(float);
#define MMG_LAUNCH_PFWD(BL_, SV_, LDS_, NL_)                                                                           \
  MMG_LAUNCH(MMG_PROBE_PAIR_FWD, n_pairs, 0, 0, (want_low ? 2 : 0) | (SV_ ? 512 : 0), (k_pair_fwd_mfma<BL_, SV_>),      \
             dim3((unsigned)g), dim3(256), LDS_, (hipStream_t)stream, H, pi, li, deg, degree_threshold, want_low ? 1 : 0, \
             n_pairs, drop_p, seed, seed_ptr, pb, pred, sel, n_sel, NL_, svb, svh, sv_pos)
  if (n_labs_lds) {
    if (saved) {
      MMG_CHECK_HIP((MmgMaxLds<&k_pair_fwd_mfma<true, true>, lds_max>::set()), "pair_head_fwd(attr)");
      MMG_LAUNCH_PFWD(true, true, lds, n_labs_lds);
    } else {
      MMG_CHECK_HIP((MmgMaxLds<&k_pair_fwd_mfma<true, false>, lds_max>::set()), "pair_head_fwd(attr)");
      MMG_LAUNCH_PFWD(true, false, lds, n_labs_lds);
    }
  } else {
    if (saved) MMG_LAUNCH_PFWD(false, true, 0, 0); else MMG_LAUNCH_PFWD(false, false, 0, 0);
  }
#undef MMG_LAUNCH_PFWD
  MMG_CHECK_LAUNCH("pair_head_fwd");
  return MMG_OK;
}

extern "C" size_t mmg_pair_head_bwd_ws_bytes(int64_t n_pairs, int n_labs) {
  if (n_pairs <= 0 || n_labs < 0 || n_labs > 128) return 256;
  return (size_t)256 * pair_slab_floats(n_labs <= 64 ? 2 : 4) * sizeof(float) + 256;
}

extern "C" int mmg_pair_head_bwd_saved(const mmg_head_t* head, const mmg_head_grad_t* grad, const int32_t* pi,
                                       const int32_t* li, const int32_t* deg, int degree_threshold, int want_low,
                                       int64_t n_pairs, int64_t n_total, int64_t n_patients, int n_labs, float drop_p,
                                       uint64_t seed, const uint64_t* seed_ptr, const int64_t* pair_id, const float* dpred,
                                       const int32_t* sel, const int32_t* n_sel, const int64_t* io_perm,
                                       const mmg_pair_saved_t* saved, void* ws, size_t ws_bytes, void* stream);

extern "C" int mmg_pair_head_bwd(const mmg_head_t* head, const mmg_head_grad_t* grad, const int32_t* pi, const int32_t* li,
                                 const int32_t* deg, int degree_threshold, int want_low, int64_t n_pairs, int64_t n_total,
                                 int64_t n_patients, int n_labs, float drop_p, uint64_t seed, const uint64_t* seed_ptr,
                                 const int64_t* pair_id, const float* dpred, const int32_t* sel, const int32_t* n_sel,
                                 const int64_t* io_perm, void* ws, size_t ws_bytes, void* stream) {
  return mmg_pair_head_bwd_saved(head, grad, pi, li, deg, degree_threshold, want_low, n_pairs, n_total, n_patients, n_labs,
                                 drop_p, seed, seed_ptr, pair_id, dpred, sel, n_sel, io_perm, nullptr, ws, ws_bytes, stream);
}

extern "C" int mmg_pair_head_bwd_saved(const mmg_head_t* head, const mmg_head_grad_t* grad, const int32_t* pi,
                                       const int32_t* li, const int32_t* deg, int degree_threshold, int want_low,
                                       int64_t n_pairs, int64_t n_total, int64_t n_patients, int n_labs, float drop_p,
                                       uint64_t seed, const uint64_t* seed_ptr, const int64_t* pair_id, const float* dpred,
                                       const int32_t* sel, const int32_t* n_sel, const int64_t* io_perm,
                                       const mmg_pair_saved_t* saved, void* ws, size_t ws_bytes, void* stream) {
  MMG_CHECK_ARG(!saved || (saved->h1_bits && saved->h2), "pair_head_bwd_saved: saved buffers with a null pointer");
  MMG_CHECK_ARG(n_pairs >= 0 && n_labs >= 0, "pair_head_bwd: negative size");
  MMG_CHECK_ARG((sel == nullptr) == (n_sel == nullptr), "pair_head_bwd: sel and n_sel go together");
  if (n_pairs == 0) return MMG_OK;
  int rc = check_head(head, "pair_head_bwd");
  if (rc) return rc;
  MMG_CHECK_ARG(grad && grad->dA && grad->dB && grad->dW2 && grad->db2 && grad->dW3 && grad->db3,
                "pair_head_bwd: grad has a null pointer");
  MMG_CHECK_ARG(pi && li && deg && dpred, "pair_head_bwd: null buffer");
  MMG_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "pair_head_bwd: drop_p out of range");
  MMG_CHECK_ARG(n_pairs <= n_total, "pair_head_bwd: %lld listed pairs of %lld", (long long)n_pairs, (long long)n_total);
  PairBufs pb;
  rc = fill_pair_bufs(&pb, pi, pair_id, io_perm, n_total, n_patients, n_labs, "pair_head_bwd");
  if (rc) return rc;
  HeadDev H{head->A, head->B, head->W2, head->b2, head->W3, head->b3};
  HeadGradDev G{grad->dA, grad->dB, grad->dW2, grad->db2, grad->dW3, grad->db3};
  hipStream_t st = (hipStream_t)stream;
  if (n_labs <= 128) {
    // MFMA path: a front and a back wave per 32-pair tile, 4 + 4 waves per workgroup, persistent grid
    int64_t g = ((n_pairs + TP - 1) / TP + 3) / 4;
    if (g > 256) g = 256;                // ONE workgroup is resident per CU; fewer workgroups = fewer partial slabs
    if (g < 1) g = 1;
    MMG_CHECK_ARG(ws && ws_bytes >= mmg_pair_head_bwd_ws_bytes(n_pairs, n_labs), "pair_head_bwd: workspace too small");
    float* slab = (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    // a front and a back wave per 32-pair tile (k_pair_bwd_duo), two or four lab tiles of dB accumulators
#define MMG_LAUNCH_PBWD(KERNEL_, NT_, ...)                                                                            \
  MMG_LAUNCH(MMG_PROBE_PAIR_BWD, n_pairs, 0, 0, want_low ? 2 : 0, KERNEL_, dim3((unsigned)g),                          \
             dim3(NT_), 0, st, H, G, pi, li, deg, degree_threshold, want_low ? 1 : 0, n_pairs, n_labs, drop_p, seed,     \
             seed_ptr, pb, dpred, sel, n_sel, slab, ##__VA_ARGS__)
    const bool aux = pair_id != nullptr || io_perm != nullptr;
    const uint32_t* svb = saved ? saved->h1_bits : nullptr;
    const float* svh = saved ? saved->h2 : nullptr;
    const int sv_pos = saved && saved->by_position ? 1 : 0;
    MMG_CHECK_ARG(!sv_pos || sel, "pair_head_bwd_saved: by_position needs the pair list the forward ran over");
    MMG_CHECK_ARG(!saved || saved->n_entries >= (sv_pos ? n_pairs : n_total),
                  "pair_head_bwd_saved: the saved buffers are shorter than the entries this launch can read");
    if (n_labs <= 64) {
      if (saved) {
        if (aux) MMG_LAUNCH_PBWD((k_pair_bwd_duo6<2, true, true>), 512, svb, svh, sv_pos);
        else MMG_LAUNCH_PBWD((k_pair_bwd_duo6<2, false, true>), 512, svb, svh, sv_pos);
      } else {
        if (aux) MMG_LAUNCH_PBWD((k_pair_bwd_duo6<2, true, false>), 512, svb, svh, 0);
        else MMG_LAUNCH_PBWD((k_pair_bwd_duo6<2, false, false>), 512, svb, svh, 0);
      }
    } else {                                   // 65 .. 128 labs: four lab tiles of dB in the back wave (the saved state is not used)
      if (aux) MMG_LAUNCH_PBWD((k_pair_bwd_duo<4, true>), 512);
      else MMG_LAUNCH_PBWD((k_pair_bwd_duo<4, false>), 512);
    }
#undef MMG_LAUNCH_PBWD
    const int LT = n_labs <= 64 ? 2 : 4;
    const int64_t n4 = pair_slab_floats(LT) / 4;
    hipLaunchKernelGGL((mmg_k_reduce_slabs<EpiPairFlush>), dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0, st, slab, n4,
                       (int)g, EpiPairFlush{G.dW2, G.dB, G.db2, G.dW3, G.db3, n_labs, LT});
  } else {
    size_t lds = BWD_LDS_FIXED;
    int lds_db = 0;
    if (lds + (size_t)n_labs * 64 * 4 <= BWD_LDS_MAX) { lds += (size_t)n_labs * 64 * 4; lds_db = 1; }
    MMG_CHECK_HIP((MmgMaxLds<&k_pair_bwd, (int)BWD_LDS_MAX>::set()), "pair_head_bwd(attr)");
    int64_t g = (n_pairs + PT - 1) / PT;
    if (g > 512) g = 512;
    hipLaunchKernelGGL(k_pair_bwd, dim3((unsigned)g), dim3(PT), lds, st, H, G, pi, li, deg, degree_threshold,
                       want_low ? 1 : 0, n_pairs, n_total, (int)n_patients, n_labs, lds_db, drop_p, seed, seed_ptr, pair_id,
                       dpred, sel, n_sel, io_perm);
  }
  MMG_CHECK_LAUNCH("pair_head_bwd");
  return MMG_OK;
}

extern "C" size_t mmg_pair_select_ws_bytes(int64_t n_pairs) {
  const int64_t nb = (n_pairs + SEL_CH - 1) / SEL_CH;
  return (size_t)(nb > 0 ? nb : 1) * (sizeof(unsigned) + sizeof(int2)) + 64;
}

extern "C" int mmg_pair_select(const int32_t* pi, const int32_t* deg, int degree_threshold, const float* dpred,
                               const int64_t* io_perm, float* dpred_sorted, int64_t n_pairs, int32_t* sel_low,
                               int32_t* sel_high, int32_t* counts, void* ws, size_t ws_bytes, void* stream) {
  MMG_CHECK_ARG(n_pairs >= 0 && n_pairs < (int64_t)INT32_MAX, "pair_select: n_pairs out of range");
  MMG_CHECK_ARG(counts, "pair_select: counts is null");
  hipStream_t st = (hipStream_t)stream;
  if (n_pairs == 0) {
    MMG_CHECK_HIP(mmg_zero_async(counts, 2 * sizeof(int32_t), st), "pair_select(memset)");
    return MMG_OK;
  }
  MMG_CHECK_ARG(pi && deg && sel_low && sel_high, "pair_select: null buffer");
  MMG_CHECK_ARG(ws && ws_bytes >= mmg_pair_select_ws_bytes(n_pairs), "pair_select: workspace too small");
  const int nb = (int)((n_pairs + SEL_CH - 1) / SEL_CH);
  int2* base = reinterpret_cast<int2*>(ws);                                  // 8-byte aligned: first in the workspace
  unsigned* cnt = reinterpret_cast<unsigned*>(base + nb);
  hipLaunchKernelGGL(k_sel_count, dim3(nb), dim3(256), 0, st, pi, deg, degree_threshold, dpred, io_perm, dpred_sorted,
                     n_pairs, cnt);
  hipLaunchKernelGGL(k_sel_scan, dim3(1), dim3(1024), 0, st, cnt, nb, base, counts);
  hipLaunchKernelGGL(k_sel_write, dim3(nb), dim3(256), 0, st, pi, deg, degree_threshold, dpred, io_perm, dpred_sorted,
                     n_pairs, base, sel_low, sel_high);
  MMG_CHECK_LAUNCH("pair_select");
  return MMG_OK;
}
