// Sparse aggregates over CSR-by-patient (replace PyG SAGEConv's gather + scatter-mean and their
// backward; call site src/model.py:125-131,256 of the reference).  HBM-bound byte work: no MFMA.
//
//  gather : one wave per patient row, lane owns D/64 contiguous floats (a 64-lane row read is one
//           coalesced 256..1024-B segment); the source tables are the tiny vocab tables (L2-resident).
//  scatter: patient-major streaming; every workgroup owns a contiguous row chunk and multiplies the
//           32-row indicator tile (built in LDS from the CSR) with the feature tile on the fp32 matrix
//           cores, keeping the [V, D] accumulators in registers; ONE partial slab per workgroup, summed
//           by a second kernel in fixed order (bitwise reproducible).
#include "common.h"
#include <stdlib.h>

namespace {

typedef float f32x4s __attribute__((ext_vector_type(4)));

struct RelDev {
  const int32_t* rowptr; const int32_t* col; const float* rowscale; const float* colscale;
  const float* table; float* out; int32_t n_cols; int32_t acc_off;   // acc_off: first accumulator row
  uint32_t flags;
};
struct RelPack { RelDev r[MMG_MAX_REL]; int n; };

// ------------------------------------------------------------------------------ gather
template <int VEC>
__global__ __launch_bounds__(256) void k_gather(RelPack rp, int64_t n_rows, float* __restrict__ out, int accumulate) {
  constexpr int D = VEC * 64;
  const int lane = threadIdx.x & 63;
  const int64_t row = __builtin_amdgcn_readfirstlane((int)(((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6));
  if (row >= n_rows) return;
  float tot[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) tot[v] = 0.f;

  for (int r = 0; r < rp.n; ++r) {
    const RelDev& R = rp.r[r];
    const int beg = R.rowptr[row], end = R.rowptr[row + 1];
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    for (int base = beg; base < end; base += 64) {
      const int cnt = min(64, end - base);
      // 64 column ids (and their scales) in one coalesced load, then broadcast lane by lane
      int cidx = 0; float cs = 1.f;
      if (lane < cnt) {
        cidx = R.col[base + lane];
        if (R.colscale) cs = R.colscale[cidx];
      }
      int j = 0;
      for (; j + 4 <= cnt; j += 4) {
        float t[4][VEC]; float w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int c = __builtin_amdgcn_readlane(cidx, j + u);
          w[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cs), j + u));
          const float* src = R.table + (size_t)c * D + lane * VEC;
#pragma unroll
          for (int v = 0; v < VEC; ++v) t[u][v] = src[v];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int v = 0; v < VEC; ++v) acc[v] = fmaf(w[u], t[u][v], acc[v]);
      }
      for (; j < cnt; ++j) {
        const int c = __builtin_amdgcn_readlane(cidx, j);
        const float w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cs), j));
        const float* src = R.table + (size_t)c * D + lane * VEC;
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = fmaf(w, src[v], acc[v]);
      }
    }
    const float rs = R.rowscale ? R.rowscale[row] : 1.f;
#pragma unroll
    for (int v = 0; v < VEC; ++v) tot[v] = fmaf(rs, acc[v], tot[v]);
  }
  float* dst = out + (size_t)row * D + lane * VEC;
#pragma unroll
  for (int v = 0; v < VEC; ++v) dst[v] = accumulate ? dst[v] + tot[v] : tot[v];
}

// gather with the vocab tables resident in LDS (they are tiny: 264 x 128 x 4 B = 135 KB at the eICU
// shape).  Every edge then costs one conflict-free ds_read of the row slice instead of an L2 round trip
// (the L2-served version above tops out at the L2 gather rate, ~16 TB/s of row reads for 0.8 TB/s of
// algorithmic traffic).  1024 threads (16 waves) per CU hide the rowptr -> col -> LDS dependency chain.
constexpr int GL_THREADS = 1024;
constexpr size_t GL_LDS_BUDGET = 150 * 1024;

template <int VECC>   // floats per lane inside the column chunk: DC = 64*VECC
__global__ __launch_bounds__(GL_THREADS) void k_gather_lds(RelPack rp, int64_t n_rows, int64_t rows_per_blk, int D,
                                                           float* __restrict__ out, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) float tab[];
  constexpr int DC = VECC * 64;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nw = GL_THREADS / 64;
  const int d0 = blockIdx.y * DC;
  // ---- stage the table slices [acc_off + c][DC] (pre-multiplied by colscale)
  for (int r = 0; r < rp.n; ++r) {
    const RelDev& R = rp.r[r];
    const int n4 = R.n_cols * (DC / 4);
    for (int i = tid; i < n4; i += GL_THREADS) {
      const int c = i / (DC / 4), q = i - c * (DC / 4);
      f32x4s v = *reinterpret_cast<const f32x4s*>(R.table + (size_t)c * D + d0 + q * 4);
      if (R.colscale) v *= R.colscale[c];
      *reinterpret_cast<f32x4s*>(tab + (size_t)(R.acc_off + c) * DC + q * 4) = v;
    }
  }
  __syncthreads();
  const int64_t r_beg = (int64_t)blockIdx.x * rows_per_blk, r_end = min(n_rows, r_beg + rows_per_blk);
  // Software pipeline over this wave's rows (row_i = r_beg + wid + i*nw): the rowptr pair of row i+2 and
  // the column ids of row i+1 are in flight while row i is reduced out of LDS, so the dependent
  // rowptr -> col -> LDS chain never stalls the wave.
  int b0[MMG_MAX_REL], e0[MMG_MAX_REL], b1[MMG_MAX_REL], e1[MMG_MAX_REL], c0[MMG_MAX_REL];
  auto meta = [&](int64_t row, int* bb, int* ee) {
    const int64_t rr = row < r_end ? row : (r_end - 1);
#pragma unroll
    for (int r = 0; r < MMG_MAX_REL; ++r) {
      bb[r] = 0; ee[r] = 0;
      if (r < rp.n && row < r_end) { bb[r] = rp.r[r].rowptr[rr]; ee[r] = rp.r[r].rowptr[rr + 1]; }
    }
  };
  auto cols = [&](const int* bb, const int* ee, int* cc) {
#pragma unroll
    for (int r = 0; r < MMG_MAX_REL; ++r) {
      cc[r] = 0;
      if (r < rp.n && bb[r] + lane < ee[r]) cc[r] = rp.r[r].col[bb[r] + lane];
    }
  };
  int64_t row = r_beg + wid;
  if (row >= r_end) return;
  meta(row, b0, e0);
  meta(row + nw, b1, e1);
  cols(b0, e0, c0);
  for (; row < r_end; row += nw) {
    int b2[MMG_MAX_REL], e2[MMG_MAX_REL], c1[MMG_MAX_REL];
    meta(row + 2 * nw, b2, e2);
    cols(b1, e1, c1);
    float rsv[MMG_MAX_REL];        // issued now, consumed after the LDS reduction of each relation
#pragma unroll
    for (int r = 0; r < MMG_MAX_REL; ++r) rsv[r] = (r < rp.n && rp.r[r].rowscale) ? rp.r[r].rowscale[row] : 1.f;
    float prev[VECC];
    float* dst = out + (size_t)row * D + d0 + lane * VECC;
#pragma unroll
    for (int v = 0; v < VECC; ++v) prev[v] = accumulate ? dst[v] : 0.f;
    float tot[VECC];
#pragma unroll
    for (int v = 0; v < VECC; ++v) tot[v] = 0.f;
#pragma unroll
    for (int r = 0; r < MMG_MAX_REL; ++r) {
      if (r >= rp.n || b0[r] == e0[r]) continue;
      const RelDev& R = rp.r[r];
      float acc[VECC];
#pragma unroll
      for (int v = 0; v < VECC; ++v) acc[v] = 0.f;
      const float* tr = tab + (size_t)R.acc_off * DC + lane * VECC;
      int cidx = c0[r];
      for (int base = b0[r]; base < e0[r]; base += 64) {
        const int cnt = min(64, e0[r] - base);
        if (base != b0[r]) cidx = (lane < cnt) ? R.col[base + lane] : 0;   // rows with > 64 edges (rare)
        int j = 0;
        for (; j + 4 <= cnt; j += 4) {
          float t[4][VECC];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int c = __builtin_amdgcn_readlane(cidx, j + u);
#pragma unroll
            for (int v = 0; v < VECC; ++v) t[u][v] = tr[(size_t)c * DC + v];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < VECC; ++v) acc[v] += t[u][v];
        }
        for (; j < cnt; ++j) {
          const int c = __builtin_amdgcn_readlane(cidx, j);
#pragma unroll
          for (int v = 0; v < VECC; ++v) acc[v] += tr[(size_t)c * DC + v];
        }
      }
#pragma unroll
      for (int v = 0; v < VECC; ++v) tot[v] = fmaf(rsv[r], acc[v], tot[v]);
    }
#pragma unroll
    for (int v = 0; v < VECC; ++v) dst[v] = prev[v] + tot[v];
#pragma unroll
    for (int r = 0; r < MMG_MAX_REL; ++r) { b0[r] = b1[r]; e0[r] = e1[r]; b1[r] = b2[r]; e1[r] = e2[r]; c0[r] = c1[r]; }
  }
}

// ------------------------------------------------------------------------------ gather on bf16 matrix cores
// For simple graphs (MMG_REL_SIMPLE) the per-row indicator is 0/1, which bf16 holds exactly, and an fp32
// table value splits exactly into three bf16 pieces (hi + mid + lo, 8 significant bits each).  So
//     out[i, :] = sum_r rs_r[i] * sum_v Ind_r[i, v] * T'_r[v, :]        (T' = colscale * T)
// is three v_mfma_f32_32x32x16_bf16 per 16 vocab columns with EXACT products and fp32 accumulation: the
// fp32 result up to summation order, at 16x the fp32 matrix rate.  The three split tables stay in LDS for
// the whole workgroup ([piece][d][v], v contiguous = the B-fragment order), the 32-row indicator tile is
// rebuilt per stage ([row][v]); one accumulator per relation keeps the per-relation mean scaling exact.
// Column ids of the next stage and row bounds of the stage after are prefetched into registers.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float gf32x16 __attribute__((ext_vector_type(16)));
constexpr int GB_ROWS = 32, GB_DH = 64, GB_PF = 8;
constexpr size_t GB_LDS_MAX = 156 * 1024;

__device__ inline void split3(float v, __bf16& a, __bf16& b, __bf16& c) {
  a = (__bf16)v;
  const float r1 = v - (float)a;
  b = (__bf16)r1;
  c = (__bf16)(r1 - (float)b);
}

__global__ __launch_bounds__(256) void k_gather_bf16(RelPack rp, int64_t n_rows, int64_t rows_per_blk, int D, int vp,
                                                     float* __restrict__ out, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int ldv = vp + 8;                                   // bf16 elements per LDS row (16-B multiple)
  __bf16* TT = reinterpret_cast<__bf16*>(lds_raw);          // [3][GB_DH][ldv]
  __bf16* CT = TT + 3 * GB_DH * ldv;                        // [GB_ROWS][ldv]
  float* RS = reinterpret_cast<float*>(CT + GB_ROWS * ldv); // [MMG_MAX_REL][GB_ROWS]
  float* PART = RS + MMG_MAX_REL * GB_ROWS;                 // [2][32][32]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int dt = wid & 1, kh = wid >> 1;
  const int d0 = blockIdx.y * GB_DH;
  const int nks = vp / 16, kmid = nks / 2;

  // ---- stage the split tables once
  {
    const mmg_f4 z = {0.f, 0.f, 0.f, 0.f};
    mmg_f4* t4 = reinterpret_cast<mmg_f4*>(TT);
    for (int i = tid; i < 3 * GB_DH * ldv / 8; i += 256) t4[i] = z;
  }
  __syncthreads();
  for (int r = 0; r < rp.n; ++r) {
    const RelDev& R = rp.r[r];
    for (int i = tid; i < R.n_cols * GB_DH; i += 256) {
      const int c = i / GB_DH, dd = i - c * GB_DH;
      float v = R.table[(size_t)c * D + d0 + dd];
      if (R.colscale) v *= R.colscale[c];
      __bf16 a, b, cc;
      split3(v, a, b, cc);
      TT[(0 * GB_DH + dd) * ldv + R.acc_off + c] = a;
      TT[(1 * GB_DH + dd) * ldv + R.acc_off + c] = b;
      TT[(2 * GB_DH + dd) * ldv + R.acc_off + c] = cc;
    }
  }

  const int64_t r_beg = (int64_t)blockIdx.x * rows_per_blk, r_end = min(n_rows, r_beg + rows_per_blk);
  const int m = tid >> 3, q = tid & 7;                      // 8 lanes per indicator row
  // software pipeline state: bounds of the next stage's row, prefetched column ids of the current one
  int cb[MMG_MAX_REL], ce[MMG_MAX_REL], nb[MMG_MAX_REL], ne[MMG_MAX_REL], cc[MMG_MAX_REL][GB_PF];
  auto bounds = [&](int64_t row, int* bb, int* ee) {
#pragma unroll
    for (int r = 0; r < MMG_MAX_REL; ++r) {
      bb[r] = 0; ee[r] = 0;
      if (r < rp.n && row < r_end) { bb[r] = rp.r[r].rowptr[row]; ee[r] = rp.r[r].rowptr[row + 1]; }
    }
  };
  auto fetch_cols = [&](const int* bb, const int* ee, int (*dst)[GB_PF]) {
#pragma unroll
    for (int r = 0; r < MMG_MAX_REL; ++r)
#pragma unroll
      for (int i = 0; i < GB_PF; ++i) {
        const int k = bb[r] + q + 8 * i;
        dst[r][i] = (r < rp.n && k < ee[r]) ? rp.r[r].col[k] : -1;
      }
  };
  bounds(r_beg + m, cb, ce);
  bounds(r_beg + GB_ROWS + m, nb, ne);
  fetch_cols(cb, ce, cc);

  gf32x16 acc[MMG_MAX_REL];
  for (int64_t r0 = r_beg; r0 < r_end; r0 += GB_ROWS) {
    __syncthreads();                                        // previous stage done with CT / PART / RS
    {
      const mmg_f4 z = {0.f, 0.f, 0.f, 0.f};
      mmg_f4* c4 = reinterpret_cast<mmg_f4*>(CT);
      for (int i = tid; i < GB_ROWS * ldv / 8; i += 256) c4[i] = z;
      if (tid < MMG_MAX_REL * GB_ROWS) {
        const int r = tid / GB_ROWS, mm = tid - r * GB_ROWS;
        float v = 1.f;
        if (r < rp.n && rp.r[r].rowscale && r0 + mm < r_end) v = rp.r[r].rowscale[r0 + mm];
        RS[tid] = v;
      }
    }
    float prev[16];
    if (kh == 0) {                                          // accumulate: issue the old values early
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int64_t gr = r0 + (i & 3) + 8 * (i >> 2) + 4 * h;
        prev[i] = (accumulate && gr < r_end) ? out[(size_t)gr * D + d0 + dt * 32 + l31] : 0.f;
      }
    }
    __syncthreads();                                        // zero-fill visible before the ones land
    {
      unsigned short* c16 = reinterpret_cast<unsigned short*>(CT) + m * ldv;
#pragma unroll
      for (int r = 0; r < MMG_MAX_REL; ++r) {
        if (r >= rp.n) continue;
        const int off = rp.r[r].acc_off;
#pragma unroll
        for (int i = 0; i < GB_PF; ++i)
          if (cc[r][i] >= 0) c16[off + cc[r][i]] = 0x3F80;                    // bf16 1.0
        for (int k = cb[r] + q + 8 * GB_PF; k < ce[r]; k += 8) c16[off + rp.r[r].col[k]] = 0x3F80;   // > 64 edges (rare)
      }
    }
    // prefetch: column ids for the next stage (bounds known), bounds for the one after
    int tb[MMG_MAX_REL], te[MMG_MAX_REL], tc[MMG_MAX_REL][GB_PF];
    fetch_cols(nb, ne, tc);
    bounds(r0 + 2 * GB_ROWS + m, tb, te);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < MMG_MAX_REL; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
    const int kb = kh ? kmid : 0, ke = kh ? nks : kmid;
#pragma unroll
    for (int r = 0; r < MMG_MAX_REL; ++r) {
      if (r >= rp.n) continue;
      const int k0 = rp.r[r].acc_off / 16;
      const int k1 = k0 + ((rp.r[r].n_cols + 15) >> 4);
      for (int ks = max(k0, kb); ks < min(k1, ke); ++ks) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(CT + l31 * ldv + 16 * ks + 8 * h);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          const bf16x8 b = *reinterpret_cast<const bf16x8*>(TT + (p * GB_DH + dt * 32 + l31) * ldv + 16 * ks + 8 * h);
          acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[r], 0, 0, 0);
        }
      }
    }
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < MMG_MAX_REL; ++r)
        if (r < rp.n) t = fmaf(RS[r * GB_ROWS + row], acc[r][i], t);
      v[i] = t;
      if (kh == 1) PART[(dt * 32 + row) * 32 + l31] = t;
    }
    __syncthreads();
    if (kh == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        const int64_t gr = r0 + row;
        if (gr < r_end) out[(size_t)gr * D + d0 + dt * 32 + l31] = prev[i] + v[i] + PART[(dt * 32 + row) * 32 + l31];
      }
    }
#pragma unroll
    for (int r = 0; r < MMG_MAX_REL; ++r) {
      cb[r] = nb[r]; ce[r] = ne[r]; nb[r] = tb[r]; ne[r] = te[r];
#pragma unroll
      for (int i = 0; i < GB_PF; ++i) cc[r][i] = tc[r][i];
    }
  }
}

// ------------------------------------------------------------------------------ scatter
// out[v, :] = sum_rows Ind[row, v] * x[row, :]  is a tall-skinny product  Ind^T [V x P] . x [P x D].
// LDS float atomics (ds_add_f32) were measured at ~0.3 adds/clk/CU on gfx950 -- 40x off the HBM
// time of this pass -- so the per-row indicator tile (rowscale at the edge positions, else 0) is
// materialised in LDS for 32 patient rows at a time and multiplied on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: 1.0*x and rs*x products are exact, accumulation is fp32).  has_lab is 67 %
// dense, so this is the dense formulation of a dense block, not a reshaped sparse op; the sparse
// relations ride along in the same pass so that x is read from HBM once.
// Each workgroup owns a contiguous row chunk and ALL vocab tiles (accumulators stay in registers for
// the whole chunk), writes one partial slab; k_scatter_reduce sums the slabs in fixed order.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int SC_ROWS = 32;     // patient rows per LDS stage (= MFMA K extent per stage)
constexpr int SC_MAX_NT = 16;   // 32-row vocab tiles per launch (512 padded vocab rows)

struct ScatterPlan {
  int nt;              // padded vocab tiles (template instance)
  int total_pad;       // nt * 32
  int dc;              // feature columns per workgroup (64 | 128)
  int n_dchunks;
  int n_split;
  int64_t rows_per_split;
  bool ok;
};

int pad32(int n) { return (n + 31) & ~31; }

ScatterPlan plan_scatter(const mmg_rel_t* rels, int n_rel, int64_t n_rows, int D) {
  ScatterPlan p{};
  int tiles = 0;
  for (int r = 0; r < n_rel; ++r) tiles += pad32(rels[r].n_cols) / 32;
  const int inst[] = {2, 4, 6, 8, 10, 12, 16};
  p.ok = false;
  for (int i = 0; i < 7; ++i)
    if (tiles <= inst[i]) { p.nt = inst[i]; p.ok = true; break; }
  p.total_pad = p.nt * 32;
  p.dc = D >= 128 ? 128 : 64;
  p.n_dchunks = D / p.dc;
  int64_t max_split = (n_rows + SC_ROWS - 1) / SC_ROWS;
  if (max_split < 1) max_split = 1;
  int64_t want = 512 / p.n_dchunks;      // two workgroups per CU: one stages while the other multiplies
  if (want < 1) want = 1;
  p.n_split = (int)(want < max_split ? want : max_split);
  int64_t rps = (n_rows + p.n_split - 1) / p.n_split;
  rps = (rps + SC_ROWS - 1) / SC_ROWS * SC_ROWS;
  p.rows_per_split = rps;
  p.n_split = (int)((n_rows + rps - 1) / rps);
  if (p.n_split < 1) p.n_split = 1;
  return p;
}

// slab layout: [n_split][NT*32][D]
// The indicator tile holds integer edge COUNTS (ds_add_u32: multi-edges add up, exactly like
// scatter_add); the rowscale of the tile's relation is applied when the A fragment is read.
template <int NT, int KT, bool HAS_RS>   // KT = 32-column tiles per workgroup (dc = 32*KT)
__global__ __launch_bounds__(256, 2) void k_scatter_mfma(RelPack rp, int64_t n_rows, int64_t rows_per_split, int D,
                                                         const float* __restrict__ x, float* __restrict__ slab) {
  constexpr int NTOT = NT * 32, DC = KT * 32;
  constexpr int NSPLIT = 4 / KT, NTW = NT / NSPLIT;
  __shared__ __attribute__((aligned(16))) int Cs[SC_ROWS][NTOT];
  __shared__ __attribute__((aligned(16))) float Xs[SC_ROWS][DC];
  __shared__ float RSt[NT][SC_ROWS];   // rowscale of the relation that owns each vocab tile, per stage row
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int kt = wid % KT, nt0 = (wid / KT) * NTW;
  const int h = lane >> 5, l31 = lane & 31;
  const int d0 = blockIdx.y * DC;
  const int64_t r_beg = (int64_t)blockIdx.x * rows_per_split;
  const int64_t r_end = min(n_rows, r_beg + rows_per_split);

  f32x16 acc[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  for (int64_t r0 = r_beg; r0 < r_end; r0 += SC_ROWS) {
    __syncthreads();                       // previous stage's MFMA reads are done
    const f32x4s z = {0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < SC_ROWS * NTOT / 4; i += 256) reinterpret_cast<f32x4s*>(&Cs[0][0])[i] = z;
    for (int i = tid; i < SC_ROWS * DC / 4; i += 256) {
      const int r = i / (DC / 4), c4 = i - r * (DC / 4);
      f32x4s v = z;
      if (r0 + r < r_end) v = *reinterpret_cast<const f32x4s*>(x + (size_t)(r0 + r) * D + d0 + c4 * 4);
      *reinterpret_cast<f32x4s*>(&Xs[r][c4 * 4]) = v;
    }
    if (HAS_RS) {
      for (int i = tid; i < NT * SC_ROWS; i += 256) {
        const int t = i / SC_ROWS, m = i - t * SC_ROWS;
        int rel = 0;                       // tiles never straddle relations (acc_off is a multiple of 32)
        for (int r = 1; r < rp.n; ++r)
          if (t * 32 >= rp.r[r].acc_off) rel = r;
        float v = 1.f;
        if (rp.r[rel].rowscale && r0 + m < r_end) v = rp.r[rel].rowscale[r0 + m];
        RSt[t][m] = v;
      }
    }
    __syncthreads();                       // zero-fill complete before the counts land
    {   // 8 lanes per row: every thread issues its (independent) rowptr and col loads at once
      const int m = tid >> 3, q = tid & 7;
      const int64_t row = r0 + m;
      if (row < r_end) {
        int beg[MMG_MAX_REL], end[MMG_MAX_REL];
#pragma unroll
        for (int r = 0; r < MMG_MAX_REL; ++r) {
          beg[r] = 0; end[r] = 0;
          if (r < rp.n) { beg[r] = rp.r[r].rowptr[row]; end[r] = rp.r[r].rowptr[row + 1]; }
        }
#pragma unroll
        for (int r = 0; r < MMG_MAX_REL; ++r) {
          if (r < rp.n)
            for (int k = beg[r] + q; k < end[r]; k += 8) atomicAdd(&Cs[m][rp.r[r].acc_off + rp.r[r].col[k]], 1);
        }
      }
    }
    __syncthreads();
    // A[i=v][k=m] = cnt[m][v] * rs[m],  B[k=m][j=d] = Xs[m][d];  lane half h takes m = 16h + s
#pragma unroll 4
    for (int s = 0; s < SC_ROWS / 2; ++s) {
      const int m = h * (SC_ROWS / 2) + s;
      const float b = Xs[m][kt * 32 + l31];
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        float a = (float)Cs[m][(nt0 + t) * 32 + l31];
        if (HAS_RS) a *= RSt[nt0 + t][m];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
      }
    }
  }
  float* dst = slab + (size_t)blockIdx.x * NTOT * D;
#pragma unroll
  for (int t = 0; t < NTW; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int v = (nt0 + t) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
      dst[(size_t)v * D + d0 + kt * 32 + l31] = acc[t][i];
    }
}

// epilogue of the slab sum: padded accumulator row -> (relation, vocab row), times colscale
struct EpiScatter {
  RelPack rp; int D;
  __device__ void operator()(int64_t i4, mmg_f4 v) const {
    const int64_t i = i4 * 4;
    const int c = (int)(i / D), d = (int)(i - (int64_t)c * D);
    for (int r = 0; r < rp.n; ++r) {
      const RelDev& R = rp.r[r];
      if (c >= R.acc_off && c < R.acc_off + R.n_cols) {
        const int j = c - R.acc_off;
        const float cs = R.colscale ? R.colscale[j] : 1.f;
        *reinterpret_cast<mmg_f4*>(R.out + (size_t)j * D + d) = v * cs;
        return;
      }
    }
  }
};

template <int NT>
void launch_scatter_mfma(const ScatterPlan& p, const RelPack& rp, int64_t n_rows, int D, const float* x, float* slab,
                         hipStream_t st) {
  dim3 grid((unsigned)p.n_split, (unsigned)p.n_dchunks);
  bool has_rs = false;
  for (int r = 0; r < rp.n; ++r) has_rs |= rp.r[r].rowscale != nullptr;
  if (p.dc == 128) {
    if (has_rs) hipLaunchKernelGGL((k_scatter_mfma<NT, 4, true>), grid, dim3(256), 0, st, rp, n_rows, p.rows_per_split, D, x, slab);
    else hipLaunchKernelGGL((k_scatter_mfma<NT, 4, false>), grid, dim3(256), 0, st, rp, n_rows, p.rows_per_split, D, x, slab);
  } else {
    if (has_rs) hipLaunchKernelGGL((k_scatter_mfma<NT, 2, true>), grid, dim3(256), 0, st, rp, n_rows, p.rows_per_split, D, x, slab);
    else hipLaunchKernelGGL((k_scatter_mfma<NT, 2, false>), grid, dim3(256), 0, st, rp, n_rows, p.rows_per_split, D, x, slab);
  }
}

// fallback when the accumulators do not fit LDS: global float atomics (contiguous 256-B per wave op)
template <int VEC>
__global__ __launch_bounds__(256) void k_scatter_atomic(RelPack rp, int64_t n_rows, const float* __restrict__ x) {
  constexpr int D = VEC * 64;
  const int lane = threadIdx.x & 63;
  const int64_t row = __builtin_amdgcn_readfirstlane((int)(((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6));
  if (row >= n_rows) return;
  float xv[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) xv[v] = x[(size_t)row * D + lane * VEC + v];
  for (int r = 0; r < rp.n; ++r) {
    const RelDev& R = rp.r[r];
    const int beg = R.rowptr[row], end = R.rowptr[row + 1];
    const float rs = R.rowscale ? R.rowscale[row] : 1.f;
    for (int k = beg; k < end; ++k) {
      const int c = R.col[k];
      const float cs = R.colscale ? R.colscale[c] : 1.f;
#pragma unroll
      for (int v = 0; v < VEC; ++v) atomicAdd(R.out + (size_t)c * D + lane * VEC + v, xv[v] * rs * cs);
    }
  }
}

int pack(const mmg_rel_t* rels, int n_rel, RelPack* rp, bool need_table, bool need_out, bool pad_cols = false) {
  MMG_CHECK_ARG(rels && n_rel >= 1 && n_rel <= MMG_MAX_REL, "aggregate: n_rel must be 1..%d", MMG_MAX_REL);
  rp->n = n_rel;
  int off = 0;
  for (int r = 0; r < n_rel; ++r) {
    MMG_CHECK_ARG(rels[r].rowptr && rels[r].n_cols >= 0, "aggregate: relation %d has null rowptr", r);
    MMG_CHECK_ARG(!need_table || rels[r].table, "aggregate: relation %d has null table", r);
    MMG_CHECK_ARG(!need_out || rels[r].out, "aggregate: relation %d has null out", r);
    rp->r[r] = RelDev{rels[r].rowptr, rels[r].col, rels[r].rowscale, rels[r].colscale, rels[r].table,
                      rels[r].out, rels[r].n_cols, off, rels[r].flags};
    off += pad_cols ? ((rels[r].n_cols + 31) & ~31) : rels[r].n_cols;
  }
  return MMG_OK;
}

}  // namespace

extern "C" int mmg_gather_rows(const mmg_rel_t* rels, int n_rel, int64_t n_rows, int D, float* out, int accumulate,
                               void* stream) {
  MMG_CHECK_ARG(mmg_valid_D(D), "gather_rows: D=%d unsupported (64|128|256)", D);
  MMG_CHECK_ARG(n_rows >= 0 && n_rows < 2147483647LL / 64, "gather_rows: n_rows out of range");
  MMG_CHECK_ARG(out || n_rows == 0, "gather_rows: out is null");
  RelPack rp;
  int rc = pack(rels, n_rel, &rp, true, false);
  if (rc) return rc;
  if (n_rows == 0) return MMG_OK;
  hipStream_t st = (hipStream_t)stream;
  int total_cols = 0;
  for (int r = 0; r < n_rel; ++r) total_cols += rels[r].n_cols;
  // bf16-split matrix-core path: simple relations, tables + indicator tile fit LDS, D a multiple of 64
  {
    bool simple = true;
    int vp = 0;
    for (int r = 0; r < n_rel; ++r) { simple &= (rels[r].flags & MMG_REL_SIMPLE) != 0; vp += (rels[r].n_cols + 15) & ~15; }
    // Measured on MI355X (x100, D=128): 0.53 ms per launch vs 0.16 ms for the LDS-table kernel below -- the
    // 45 four-barrier stages per workgroup expose one global round trip each at one workgroup per CU.
    // Kept (parity-tested, opt-in) as the base for a deeper-pipelined version: MMG_AGG_BF16=1.
    static const int use_bf16 = [] { const char* e = getenv("MMG_AGG_BF16"); return e ? atoi(e) : 0; }();
    const size_t lds = (size_t)(3 * GB_DH + GB_ROWS) * (vp + 8) * 2 + (size_t)MMG_MAX_REL * GB_ROWS * 4 + 2 * 32 * 32 * 4;
    if (use_bf16 && simple && vp > 0 && lds <= GB_LDS_MAX && n_rows >= 256) {
      RelPack rb = rp;
      int off = 0;
      for (int r = 0; r < n_rel; ++r) { rb.r[r].acc_off = off; off += (rels[r].n_cols + 15) & ~15; }
      const int n_dh = D / GB_DH;
      int64_t nblk = 256 / n_dh;
      if (nblk < 1) nblk = 1;
      const int64_t max_blk = (n_rows + 4 * GB_ROWS - 1) / (4 * GB_ROWS);
      if (nblk > max_blk) nblk = max_blk;
      int64_t rows_per_blk = (n_rows + nblk - 1) / nblk;
      rows_per_blk = (rows_per_blk + GB_ROWS - 1) / GB_ROWS * GB_ROWS;
      nblk = (n_rows + rows_per_blk - 1) / rows_per_blk;
      (void)hipFuncSetAttribute((const void*)k_gather_bf16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GB_LDS_MAX);
      hipLaunchKernelGGL(k_gather_bf16, dim3((unsigned)nblk, (unsigned)n_dh), dim3(256), lds, st, rb, n_rows, rows_per_blk, D,
                         vp, out, accumulate);
      MMG_CHECK_LAUNCH("gather_rows(bf16)");
      return MMG_OK;
    }
  }
  // LDS-resident tables when a column chunk of every table fits; else the L2-served kernel
  int dc = D >= 128 ? 128 : 64;
  while (dc > 64 && (size_t)total_cols * dc * 4 > GL_LDS_BUDGET) dc >>= 1;
  if (total_cols > 0 && (size_t)total_cols * dc * 4 <= GL_LDS_BUDGET && n_rows >= 64) {
    const size_t lds = (size_t)total_cols * dc * 4;
    const int n_dchunks = D / dc;
    int64_t nblk = 256 / n_dchunks;
    if (nblk < 1) nblk = 1;
    const int64_t max_blk = (n_rows + 15) / 16;
    if (nblk > max_blk) nblk = max_blk;
    const int64_t rows_per_blk = (n_rows + nblk - 1) / nblk;
    nblk = (n_rows + rows_per_blk - 1) / rows_per_blk;
    dim3 grid((unsigned)nblk, (unsigned)n_dchunks);
    if (dc == 128) {
      (void)hipFuncSetAttribute((const void*)k_gather_lds<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GL_LDS_BUDGET);
      hipLaunchKernelGGL(k_gather_lds<2>, grid, dim3(GL_THREADS), lds, st, rp, n_rows, rows_per_blk, D, out, accumulate);
    } else {
      (void)hipFuncSetAttribute((const void*)k_gather_lds<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GL_LDS_BUDGET);
      hipLaunchKernelGGL(k_gather_lds<1>, grid, dim3(GL_THREADS), lds, st, rp, n_rows, rows_per_blk, D, out, accumulate);
    }
    MMG_CHECK_LAUNCH("gather_rows(lds)");
    return MMG_OK;
  }
  const unsigned nb = (unsigned)((n_rows + 3) / 4);
  if (D == 64) hipLaunchKernelGGL(k_gather<1>, dim3(nb), dim3(256), 0, st, rp, n_rows, out, accumulate);
  else if (D == 128) hipLaunchKernelGGL(k_gather<2>, dim3(nb), dim3(256), 0, st, rp, n_rows, out, accumulate);
  else hipLaunchKernelGGL(k_gather<4>, dim3(nb), dim3(256), 0, st, rp, n_rows, out, accumulate);
  MMG_CHECK_LAUNCH("gather_rows");
  return MMG_OK;
}

extern "C" size_t mmg_scatter_rows_ws_bytes(const mmg_rel_t* rels, int n_rel, int64_t n_rows, int D) {
  if (!rels || n_rel < 1 || n_rel > MMG_MAX_REL || !mmg_valid_D(D) || n_rows < 0) return 0;
  ScatterPlan p = plan_scatter(rels, n_rel, n_rows, D);
  if (!p.ok) return 256;
  return (size_t)p.n_split * p.total_pad * D * 4 + 256;
}

extern "C" int mmg_scatter_rows(const mmg_rel_t* rels, int n_rel, int64_t n_rows, int D, const float* x, void* ws,
                                size_t ws_bytes, void* stream) {
  MMG_CHECK_ARG(mmg_valid_D(D), "scatter_rows: D=%d unsupported (64|128|256)", D);
  MMG_CHECK_ARG(n_rows >= 0 && n_rows < 2147483647LL / 64, "scatter_rows: n_rows out of range");
  RelPack rp;
  int rc = pack(rels, n_rel, &rp, false, true, true);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  ScatterPlan p = plan_scatter(rels, n_rel, n_rows, D);
  int total_cols = 0;
  for (int r = 0; r < n_rel; ++r) total_cols += rels[r].n_cols;
  if (total_cols == 0) return MMG_OK;
  if (n_rows == 0 || !p.ok) {
    // empty input, or more than 512 padded vocab rows: zero + global float atomics
    rc = pack(rels, n_rel, &rp, false, true, false);
    if (rc) return rc;
    for (int r = 0; r < n_rel; ++r)
      if (rels[r].n_cols > 0) (void)hipMemsetAsync(rels[r].out, 0, (size_t)rels[r].n_cols * D * 4, st);
    if (n_rows == 0) return MMG_OK;
    MMG_CHECK_ARG(x, "scatter_rows: x is null");
    const unsigned nb = (unsigned)((n_rows + 3) / 4);
    if (D == 64) hipLaunchKernelGGL(k_scatter_atomic<1>, dim3(nb), dim3(256), 0, st, rp, n_rows, x);
    else if (D == 128) hipLaunchKernelGGL(k_scatter_atomic<2>, dim3(nb), dim3(256), 0, st, rp, n_rows, x);
    else hipLaunchKernelGGL(k_scatter_atomic<4>, dim3(nb), dim3(256), 0, st, rp, n_rows, x);
    MMG_CHECK_LAUNCH("scatter_rows(atomic)");
    return MMG_OK;
  }
  MMG_CHECK_ARG(x && ws, "scatter_rows: null buffer");
  const size_t need = mmg_scatter_rows_ws_bytes(rels, n_rel, n_rows, D);
  if (ws_bytes < need) {
    mmg_set_error("scatter_rows: workspace %zu < %zu", ws_bytes, need);
    return MMG_E_WS;
  }
  float* slab = (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  switch (p.nt) {
    case 2: launch_scatter_mfma<2>(p, rp, n_rows, D, x, slab, st); break;
    case 4: launch_scatter_mfma<4>(p, rp, n_rows, D, x, slab, st); break;
    case 6: launch_scatter_mfma<6>(p, rp, n_rows, D, x, slab, st); break;
    case 8: launch_scatter_mfma<8>(p, rp, n_rows, D, x, slab, st); break;
    case 10: launch_scatter_mfma<10>(p, rp, n_rows, D, x, slab, st); break;
    case 12: launch_scatter_mfma<12>(p, rp, n_rows, D, x, slab, st); break;
    default: launch_scatter_mfma<16>(p, rp, n_rows, D, x, slab, st); break;
  }
  const int64_t n = (int64_t)p.total_pad * D;
  hipLaunchKernelGGL((mmg_k_reduce_slabs<EpiScatter>), dim3((unsigned)((n / 4 + 15) / 16)), dim3(256), 0, st, slab, n / 4,
                     p.n_split, EpiScatter{rp, D});
  MMG_CHECK_LAUNCH("scatter_rows");
  return MMG_OK;
}
