// Sparse aggregates over CSR-by-patient (replace PyG SAGEConv's gather + scatter-mean and their
// backward; call site src/model.py:125-131,256 of the reference).
//
//  Simple relations (no repeated (patient, item) pair -- the reference's frames are de-duplicated) have a 0/1
//  indicator, exact in bf16, and an fp32 feature is the exact sum of three bf16 pieces: Ind . x is three
//  v_mfma_f32_32x32x16_bf16 with EXACT products and fp32 accumulation.  The adjacency is kept as bit planes
//  (mmg_rel_mask_build) and expanded into operand fragments in the kernel -- no rowptr -> col chase in the loop.
//    k_gather_bits  : vocab -> patient (forward; backward of the scatter), item tables split once per workgroup
//    k_scatter_bits : patient -> vocab (forward; backward of the gather), patient-major streaming of x, all item
//                     tiles' accumulators in registers, one partial slab per workgroup summed in fixed order
//  Multigraphs / vocabularies without a bit-plane instance: k_gather_lds (tables resident in LDS, one wave per
//  patient row), k_gather (L2-served), k_scatter_mfma (fp32 matrix cores on an integer count tile),
//  k_scatter_atomic (> 512 padded vocab rows).
#include "common.h"

namespace {

typedef float f32x4s __attribute__((ext_vector_type(4)));

struct RelDev {
  const int32_t* rowptr; const int32_t* col; const float* rowscale; const float* colscale;
  const float* table; float* out; int32_t n_cols; int32_t acc_off;   // acc_off: first accumulator row
  uint32_t flags;
  const uint64_t* mask;      // bit planes (simple relations) or null
  const uint16_t* mask_r;    // row-major fields (gather side) or null
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
      if (r >= rp.n) continue;
      // row bounds are wave-uniform: make that explicit so the edge loops run on the scalar unit
      const int rb_ = __builtin_amdgcn_readfirstlane(b0[r]), re_ = __builtin_amdgcn_readfirstlane(e0[r]);
      if (rb_ == re_) continue;
      const RelDev& R = rp.r[r];
      float acc[VECC], acc2[VECC];
#pragma unroll
      for (int v = 0; v < VECC; ++v) { acc[v] = 0.f; acc2[v] = 0.f; }
      const float* tr = tab + (size_t)R.acc_off * DC + lane * VECC;
      int cidx = c0[r];
      for (int base = rb_; base < re_; base += 64) {
        const int cnt = min(64, re_ - base);
        if (base != rb_) cidx = (lane < cnt) ? R.col[base + lane] : 0;     // rows with > 64 edges (rare)
        int j = 0;
        for (; j + 8 <= cnt; j += 8) {             // 8 LDS reads in flight, two independent add chains
          float t[8][VECC];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int c = __builtin_amdgcn_readlane(cidx, j + u);
#pragma unroll
            for (int v = 0; v < VECC; ++v) t[u][v] = tr[(size_t)c * DC + v];
          }
#pragma unroll
          for (int u = 0; u < 8; u += 2)
#pragma unroll
            for (int v = 0; v < VECC; ++v) { acc[v] += t[u][v]; acc2[v] += t[u + 1][v]; }
        }
        for (; j < cnt; ++j) {
          const int c = __builtin_amdgcn_readlane(cidx, j);
#pragma unroll
          for (int v = 0; v < VECC; ++v) acc[v] += tr[(size_t)c * DC + v];
        }
      }
#pragma unroll
      for (int v = 0; v < VECC; ++v) acc[v] += acc2[v];
#pragma unroll
      for (int v = 0; v < VECC; ++v) tot[v] = fmaf(rsv[r], acc[v], tot[v]);
    }
#pragma unroll
    for (int v = 0; v < VECC; ++v) dst[v] = prev[v] + tot[v];
#pragma unroll
    for (int r = 0; r < MMG_MAX_REL; ++r) { b0[r] = b1[r]; e0[r] = e1[r]; b1[r] = b2[r]; e1[r] = e2[r]; c0[r] = c1[r]; }
  }
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

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
__global__ __launch_bounds__(256) void k_scatter_mfma(RelPack rp, int64_t n_rows, int64_t rows_per_split, int D,
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


// ------------------------------------------------------------------------------ scatter on the bf16 matrix cores
// Simple relations (MMG_REL_SIMPLE: no repeated (patient, item) pair -- the reference's frames are de-duplicated)
// have a 0/1 indicator, exact in bf16, and an fp32 feature splits exactly into three bf16 pieces (8 significant
// bits each).  Ind^T . x is then three v_mfma_f32_32x32x16_bf16 per 16 patients with EXACT products and fp32
// accumulation: the fp32 result up to summation order at 16x the fp32 matrix rate (12.4 GFLOP of fp32 MFMA work,
// 79 us at peak, becomes 15 us).
// The indicator never exists in memory: the relation's adjacency is kept as bit planes (mmg_rel_mask_build; two
// words per [64-row stage][item] = which of the 64 patients have the item, 80 B per patient instead of 180 B of
// CSR), a lane holds the word of ITS item row and patient half, and the A fragment of a k-step (8 patients) is
// one field of that word expanded through a 256-entry LDS table.  No index chasing, no tile build, no barrier: every wave streams x straight from
// HBM into registers (lane = feature column, 8 patient rows per k-step, three k-steps of loads in flight), splits,
// and feeds the matrix cores; accumulators stay in registers for the whole row range of the workgroup.
// A rowscale (backward of the mean gather) multiplies x per relation before the split.
// Measured on MI355X (x100 eICU shape, 183,400 rows x 128, 320 padded items): 46 us without / 57 us with rowscale
// (the fp32-MFMA indicator kernel above: 192 / 216 us).  Ablations: no x loads 45 us, no split and no LUT reads
// 42 us -- the kernel runs at the pace of its 1.38 M matrix instructions, not of HBM (94 MB, 2.0 TB/s); by the issue-rate
// probe (profiles/probes/mfma_rate.hip: 32 clocks per MFMA per wave) their pure issue time is ~21 us.
constexpr int SB_SR = 64;                    // patient rows per stage = bits per mask word (4 k-steps of 16)

__device__ inline void split8(const float* v, bf16x8& p0, bf16x8& p1, bf16x8& p2) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 a = (__bf16)v[j];
    const float r1 = v[j] - (float)a;
    const __bf16 b = (__bf16)r1;
    p0[j] = a; p1[j] = b; p2[j] = (__bf16)(r1 - (float)b);
  }
}

typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
typedef unsigned u32x2s __attribute__((ext_vector_type(2)));
// the three bf16 pieces of 8 values by TRUNCATION (upper 16 bits of a, of a - hi, of a - hi - mid: each exactly a bf16,
// their sum is a): v_perm / v_and / v_sub only.  Element j of a piece sits in half j & 1 of dword j / 2 = the MFMA
// operand order.
__device__ __forceinline__ void split8_tr(const float* v, u32x4s& p0, u32x4s& p1, u32x4s& p2) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned a0 = __builtin_bit_cast(unsigned, v[2 * j]), a1 = __builtin_bit_cast(unsigned, v[2 * j + 1]);
    p0[j] = __builtin_amdgcn_perm(a1, a0, 0x07060302u);
    const float r0 = v[2 * j] - __builtin_bit_cast(float, a0 & 0xFFFF0000u);
    const float r1 = v[2 * j + 1] - __builtin_bit_cast(float, a1 & 0xFFFF0000u);
    const unsigned b0 = __builtin_bit_cast(unsigned, r0), b1 = __builtin_bit_cast(unsigned, r1);
    p1[j] = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
    const float s0 = r0 - __builtin_bit_cast(float, b0 & 0xFFFF0000u);
    const float s1 = r1 - __builtin_bit_cast(float, b1 & 0xFFFF0000u);
    p2[j] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, s1), __builtin_bit_cast(unsigned, s0), 0x07060302u);
  }
}

// (the scatter kernels' split since round 3: 44 plain vector instructions per 8 values instead of 56 with conversions;
//  same-box A/B 44.9 -> 44.4 us without, 55.7 -> 54.4 us with a rowscale)
__device__ inline void split8x(const float* v, bf16x8& p0, bf16x8& p1, bf16x8& p2) {
  u32x4s q0, q1, q2;
  split8_tr(v, q0, q1, q2);
  p0 = __builtin_bit_cast(bf16x8, q0); p1 = __builtin_bit_cast(bf16x8, q1); p2 = __builtin_bit_cast(bf16x8, q2);
}

// ---- two f16 pieces of x * 2^e (round 4): 18 instead of 27 matrix instructions per 16 patients x 9 item tiles.
// hi = f16(X), lo = f16(X - hi) with X = x * 2^e: 22 significant bits, relative error <= 2^-22 of X while 2^-3 <= |X| < 65504;
// below that the residual is an f16 denormal (absolute error 2^-25 in units of X).  The scale is a power of two chosen BY THE
// WAVE from the data it streams (block floating point over the wave's row range): e starts at whatever brings the first
// 16 x 32 block's largest magnitude to [2^12, 2^13); a later block with an element beyond 2^15 / 2^e lowers e, the
// accumulators are multiplied by 2^(e_new - e_old) and that block is split again -- so every piece is finite for finite x,
// whatever its range, and an element's error is <= 2^-22 of itself or 2^-37 of the largest magnitude seen so far in the
// strip, whichever is larger: the error of an fp32 running sum, not of a format with fewer bits.  A NaN or an infinity in x
// gives NaN in the sums it enters (fp32 index_add_: NaN, or +-inf for an infinity alone).
typedef _Float16 f16x8s __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2s __attribute__((ext_vector_type(2)));
typedef float f32x2s __attribute__((ext_vector_type(2)));
constexpr int H2_E_INIT = 120, H2_E_MIN = -110;

__device__ __forceinline__ void split8_h2(const float* v, float scale, f16x8s& p0, f16x8s& p1) {
  u32x4s q0, q1;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x2s a = {v[2 * j] * scale, v[2 * j + 1] * scale};
    const f16x2s hh = __builtin_convertvector(a, f16x2s);
    const f32x2s r = {a[0] - (float)hh[0], a[1] - (float)hh[1]};
    const f16x2s ll = __builtin_convertvector(r, f16x2s);
    q0[j] = __builtin_bit_cast(unsigned, hh); q1[j] = __builtin_bit_cast(unsigned, ll);
  }
  p0 = __builtin_bit_cast(f16x8s, q0); p1 = __builtin_bit_cast(f16x8s, q1);
}

__device__ __forceinline__ float absmax8(const float* v) {      // (fmaxf drops a NaN operand: a NaN never moves the scale)
  float m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fabsf(v[2]));
  m = fmaxf(fmaxf(m, fabsf(v[3])), fabsf(v[4]));
  m = fmaxf(fmaxf(m, fabsf(v[5])), fabsf(v[6]));
  return fmaxf(m, fabsf(v[7]));
}

// (rare path only) the same over the FINITE values: an infinity takes part in no scale decision -- it poisons the sums
// of its own column, as it must, and leaves the scale of the 31 other columns of the strip alone
__device__ __forceinline__ float absmax8_finite(const float* v) {
  float m = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(v[j]) < __builtin_inff() ? fabsf(v[j]) : 0.f);
  return m;
}

// The wave's scale.  e: the pieces are those of x * 2^e.  The first block with a finite non-zero value sets e so that its
// largest magnitude lands in [2^12, 2^13) and anchors e_floor = e - 10; from then on a block that does not fit
// (an element beyond 2^15 / 2^e) either lowers e (not below e_floor: the data grew, the accumulators are multiplied by the
// power of two, nothing is lost) or -- an outlier more than ~2^12 above anything seen before -- is multiplied exactly on its own.
// Every term is therefore kept to 2^-22 of itself while it is within [2^-16, 2^2] of the wave's reference magnitude 2^(13 - e)
// (at least 2^-6 of the first block's maximum), exactly if it is an outlier above, and to an absolute 2^-25 * 2^-e below.
struct H2Scale {
  int e, e_floor, seen;
  float sc, lim;                                       // 2^e, 2^(15 - e): wave-uniform
  __device__ __forceinline__ void set(int en) { e = en; sc = h2_pow2_(en); lim = h2_pow2_(15 - en); }
  __device__ __forceinline__ void init() { seen = 0; e_floor = H2_E_MIN; set(H2_E_INIT); }
  static __device__ __forceinline__ float h2_pow2_(int k) { return __builtin_bit_cast(float, (unsigned)(k + 127) << 23); }
};
constexpr int H2_E_DROP = 10;

// m = a lane's finite magnitude maximum of the block.  Returns 1: multiply this block exactly (bf16 pieces), scale unchanged;
// 0: split it as f16 pieces at the (possibly lowered) scale after multiplying the accumulators by 2^d.
__device__ __forceinline__ int h2_decide(float m, H2Scale& hs, int& d) {
#pragma unroll
  for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  // (scalar from here on: every decision below is a uniform branch and the wave's scale stays in scalar registers -- with
  //  the maximum left in a vector register the compiler treats the whole state as divergent and masks the main loop)
  m = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, m)));
  d = 0;
  if (!(m > 0.f)) return 0;                            // nothing finite and non-zero: the block cannot move the scale
  const int fe = __builtin_amdgcn_frexp_expf(m);       // floor(log2 m) + 1
  int en = 13 - fe;
  en = en < H2_E_MIN ? H2_E_MIN : en;
  en = en > hs.e ? hs.e : en;
  if (!hs.seen) { hs.seen = 1; hs.e_floor = en - H2_E_DROP; hs.set(en); return 0; }   // (the accumulators are still zero)
  if (en >= hs.e_floor) { d = en - hs.e; hs.set(en); return 0; }
  if (fe + hs.e >= 100) {                              // x * 2^e would leave the fp32 range: re-anchor
    d = en - hs.e; hs.e_floor = en - H2_E_DROP; hs.set(en); return 0;
  }
  return 1;
}
__device__ __forceinline__ float h2_pow2(int e) {          // 2^e for |e| <= 126 (wave-uniform)
  return __builtin_bit_cast(float, (unsigned)(e + 127) << 23);
}

// The strip layout (round 2): a workgroup owns a 32-column STRIP of x for a range of rows; four row quarters stream it and
// their partial accumulators are summed through LDS in fixed order, so ONE [padded items x 32] strip per workgroup goes to the
// partial slabs: 256 / (D / 32) row ranges -> 10.5 MB at the eICU vocabulary instead of 42 MB.  The relations' items are
// packed back to back (264 items = 9 tiles; a tile may straddle two relations: every lane carries its own bit-plane row).
// The operands of k-step q+1 are produced in the shadow of the matrix instructions of k-step q: the vector work is dealt
// out between the MFMAs explicitly (sched_group_barrier) instead of left in clumps.
// k_scatter_strip2<NA, NB>: the same strip, its item tiles dealt over TWO waves per SIMD.  k_scatter_strip<9> holds 392
// registers (144 accumulators + operand ring): ONE wave per SIMD, whose matrix pipe is busy 52 % of its life -- every LUT
// round trip, every x load it waits for and whatever vector work does not co-issue is exposed.  Here wave w (tiles
// 0 .. NA-1) and wave w + 4 (tiles NA .. NA+NB-1) of a 512-thread workgroup sit on the same SIMD (profiles/probes/
// wave_simd) and stream the SAME quarter of the rows, each under 256 registers: while one waits or expands fragments the
// other multiplies.  Both load and split x themselves (the second reader hits the cache; two splits per SIMD still fit
// under the matrix time, three -- k_scatter_units at this vocabulary -- did not).  Same arithmetic, same slab layout.
// strip_main_h: the strip's main loop on the f16 matrix instructions -- two pieces of x * 2^e (split8_h2) instead of three
// bf16 pieces; the scale exponent e belongs to the wave (H2Scale).  The magnitude check of a k-step's block rides in the
// shadow of the matrix instructions (four v_max3 and a compare); a block with an element beyond 2^15 / 2^e is found at the top
// of ITS k-step -- before anything multiplied it -- and handled there (h2_decide): e is lowered and the accumulators rescaled
// (exact: a power of two), or, for an outlier, the block alone is multiplied as three exact bf16 pieces at the unchanged
// scale.  One 64-row stage per loop trip.  On return the accumulators are in units of 2^-e_out.
template <int NT>
__device__ __forceinline__ void strip_main_h(const RelPack& rp, int64_t n_rows, int n_stage_total, int D, const float* __restrict__ x,
                                             const unsigned (*lut)[4], const unsigned (*lutb)[4], int t_first, int q_id, int n_q,
                                             f32x16* acc, int& e_out) {
  constexpr int RING = 4, AHEAD = 3;
  const int lane = threadIdx.x & 63, h = lane >> 5, l31 = lane & 31;
  const int d0 = blockIdx.y * 32;
  const int s_beg = (int)((int64_t)q_id * n_stage_total / n_q);
  const int s_end = (int)((int64_t)(q_id + 1) * n_stage_total / n_q);
  const int ns = s_end - s_beg;
  const int64_t r_beg = (int64_t)s_beg * SB_SR;
  const int64_t rows_here = ns <= 0 ? 0 : ((n_rows - r_beg) < (int64_t)ns * SB_SR ? (n_rows - r_beg) : (int64_t)ns * SB_SR);
  const float* xw = x + (size_t)r_beg * D + d0;
  const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(xw), 0, rows_here > 0 ? (int)((rows_here * D - d0) * 4) : 0, 0x00020000);
  const unsigned row_bytes = (unsigned)D * 4u;
  const unsigned voff0 = (unsigned)(8 * h) * row_bytes + (unsigned)l31 * 4u;
  float xq[RING][8];
  auto loadx = [&](int kg, float* dst) {
    const unsigned vo = voff0 + (unsigned)kg * 16u * row_bytes;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      dst[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xsrc, vo, j * row_bytes, 0));
  };
#pragma unroll
  for (int q = 0; q < AHEAD; ++q) loadx(q, xq[q]);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  e_out = 0;
  if (ns <= 0) return;
  const uint64_t* mb[NT];
  unsigned ms[NT], lm[NT];
  const uint64_t* any_mask = nullptr;
#pragma unroll
  for (int r = 0; r < MMG_MAX_REL; ++r)
    if (r < rp.n && rp.r[r].mask && !any_mask) any_mask = rp.r[r].mask;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    mb[t] = any_mask; ms[t] = 0; lm[t] = 0u;
    const int it = (t_first + t) * 32 + l31;
#pragma unroll
    for (int r = 0; r < MMG_MAX_REL; ++r) {
      if (r >= rp.n) continue;
      const int padc = (rp.r[r].n_cols + 31) & ~31;
      if (rp.r[r].mask && it >= rp.r[r].acc_off && it < rp.r[r].acc_off + rp.r[r].n_cols) {
        mb[t] = rp.r[r].mask + ((size_t)s_beg * padc + (it - rp.r[r].acc_off)) * 2 + h;
        ms[t] = 2u * (unsigned)padc; lm[t] = 0xFF0u;
      }
    }
  }
  auto loadm = [&](int s, uint64_t* dst) {             // past the end: re-read the last stage (its x reads as 0)
    const int sc = s < ns ? s : ns - 1;
#pragma unroll
    for (int t = 0; t < NT; ++t) dst[t] = mb[t][(size_t)sc * ms[t]];
  };
  // field kq (k-step inside the 64-row stage) of a mask word = (8 patient bits) << 4 = byte offset of the table entry
  auto frag_off = [&](const uint64_t* mw, int kq, int t) -> unsigned {
    const unsigned w = (kq & 2) ? (unsigned)(mw[t] >> 32) : (unsigned)mw[t];
    return (kq & 1) ? ((w >> 16) & lm[t]) : (w & lm[t]);
  };
  auto make_af = [&](const uint64_t* mw, int kq, f16x8s* af) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
      af[t] = *reinterpret_cast<const f16x8s*>(reinterpret_cast<const unsigned char*>(&lut[0][0]) + frag_off(mw, kq, t));
  };
  uint64_t mc[NT], mn[NT];
  loadm(0, mc);
  __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): the loop is entered with no load in flight
  f16x8s afc[NT], bc[2];
  make_af(mc, 0, afc);
  H2Scale hs;
  hs.init();
  unsigned long long over = ~0ull;                     // block 0 goes through the decision like any block that does not fit
  bc[0] = bc[1] = f16x8s{0, 0, 0, 0, 0, 0, 0, 0};
  for (int u = 0; u < ns; ++u) {
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
      if (__builtin_expect(over != 0ull, 0)) {
        // the block of this k-step (raw in xq[kq], nothing has multiplied it yet) does not fit the scale, or is the first
        const float* xb = xq[kq];
        int d;
        if (h2_decide(absmax8_finite(xb), hs, d)) {
          // an outlier: this block alone as three exact bf16 pieces of x * 2^e (any range), fragments from the bf16 table
          float xs[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) xs[j] = xb[j] * hs.sc;
          bf16x8 p0, p1, p2;
          split8x(xs, p0, p1, p2);
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const bf16x8 ab = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const unsigned char*>(&lutb[0][0]) + frag_off(mc, kq, t));
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, p0, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, p1, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, p2, acc[t], 0, 0, 0);
          }
          bc[0] = bc[1] = f16x8s{0, 0, 0, 0, 0, 0, 0, 0};      // (the f16 products of this k-step add nothing)
        } else {
          if (d != 0) {
            const float f = h2_pow2(d < -126 ? -126 : d);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
              for (int i = 0; i < 16; ++i) acc[t][i] *= f;
          }
          split8_h2(xb, hs.sc, bc[0], bc[1]);
        }
      }
      if (kq == 0) loadm(u + 1, mn);
      loadx(u * 4 + kq + AHEAD, xq[(kq + AHEAD) & (RING - 1)]);
      __builtin_amdgcn_sched_barrier(0);
      f16x8s afn[NT], bn[2];
      make_af(kq == 3 ? mn : mc, (kq + 1) & 3, afn);
      const float* xn = xq[(kq + 1) & (RING - 1)];
      split8_h2(xn, hs.sc, bn[0], bn[1]);
      over = __builtin_amdgcn_ballot_w64(absmax8(xn) > hs.lim);
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afc[t], bc[p], acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) afc[t] = afn[t];
      bc[0] = bn[0]; bc[1] = bn[1];
#pragma unroll
      for (int i = 0; i < 2 * NT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                      // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, (30 + 2 * NT + 2 * NT - 1) / (2 * NT), 0);   // a slice of the split / expansion work
        if (i < NT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);          // one LUT read
      }
      // (the pieces are pinned HERE: left alone, the compiler sinks the split behind the branch at the top of the next
      //  k-step -- the rare path replaces them -- and out of the shadow of the matrix instructions above)
      asm volatile("" :: "v"(bc[0]), "v"(bc[1]));
      if (kq == 3) {
#pragma unroll
        for (int t = 0; t < NT; ++t) mc[t] = mn[t];
      }
    }
  }
  e_out = hs.e;
}

template <int NA, int NB>
__global__ __launch_bounds__(512) void k_scatter_strip2(RelPack rp, int64_t n_rows, int n_stage_total, int D, int total_pad,
                                                        const float* __restrict__ x, float* __restrict__ slab) {
  constexpr int NP = NA > NB ? NA : NB;                                    // tiles parked per phase
  __shared__ __attribute__((aligned(16))) unsigned lut[256][4];            // byte -> 8 f16 in {0, 1}
  __shared__ __attribute__((aligned(16))) unsigned lutb[256][4];           // ... as bf16: the exact path of an outlier block
  extern __shared__ __attribute__((aligned(16))) float st_red[];           // [4 row quarters][NP * 16][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31, w4 = wid & 3;
  const int d0 = blockIdx.y * 32;
  {
    unsigned (*tb)[4] = tid < 256 ? lut : lutb;
    const unsigned one_lo = tid < 256 ? 0x3C00u : 0x3F80u, one_hi = one_lo << 16, b = tid & 255;
#pragma unroll
    for (int q = 0; q < 4; ++q) tb[b][q] = ((b >> (2 * q)) & 1 ? one_lo : 0u) | ((b >> (2 * q + 1)) & 1 ? one_hi : 0u);
  }
  __syncthreads();
  const int q_id = blockIdx.x * 4 + w4, n_q = gridDim.x * 4;
  float* dst = slab + (size_t)blockIdx.x * total_pad * D + d0 + l31;
  // fixed-order sum over the four row quarters of the tiles [t_first, t_first + ntp) parked in st_red: every one of the
  // eight waves sums an eighth of the registers (quarter 0 + 1 + 2 + 3) and stores it
  auto sum_store = [&](int t_first, int ntp) {
    const int nreg = ntp * 16;
    for (int idx = wid * nreg / 8; idx < (wid + 1) * nreg / 8; ++idx) {
      float v = st_red[((size_t)0 * NP * 16 + idx) * 64 + lane];
      v += st_red[((size_t)1 * NP * 16 + idx) * 64 + lane];
      v += st_red[((size_t)2 * NP * 16 + idx) * 64 + lane];
      v += st_red[((size_t)3 * NP * 16 + idx) * 64 + lane];
      const int i = idx & 15;
      const int vrow = (t_first + (idx >> 4)) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
      dst[(size_t)vrow * D] = v;
    }
  };
  int e;
  if (wid < 4) {
    f32x16 acc[NA];
    strip_main_h<NA>(rp, n_rows, n_stage_total, D, x, lut, lutb, 0, q_id, n_q, acc, e);
    const float un = h2_pow2(-e);                      // the accumulators are in units of 2^-e
#pragma unroll
    for (int t = 0; t < NA; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) st_red[((size_t)w4 * NP * 16 + t * 16 + i) * 64 + lane] = acc[t][i] * un;
    __syncthreads();
    sum_store(0, NA);
    __syncthreads();                                   // the first phase's reads are done
    __syncthreads();                                   // (the other waves parked theirs)
    sum_store(NA, NB);
  } else {
    f32x16 acc[NB];
    strip_main_h<NB>(rp, n_rows, n_stage_total, D, x, lut, lutb, NA, q_id, n_q, acc, e);
    const float un = h2_pow2(-e);
    __syncthreads();
    sum_store(0, NA);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NB; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) st_red[((size_t)w4 * NP * 16 + t * 16 + i) * 64 + lane] = acc[t][i] * un;
    __syncthreads();
    sum_store(NA, NB);
  }
}

template <int NA, int NB>
int launch_scatter_strip2(const RelPack& rp, int64_t n_rows, int D, int n_ranges, int total_pad, const float* x, float* slab,
                          hipStream_t st) {
  constexpr int lds = 4 * (NA > NB ? NA : NB) * 16 * 64 * 4;
  MMG_CHECK_HIP((MmgMaxLds<&k_scatter_strip2<NA, NB>, lds>::set()), "scatter_rows(attr)");
  const int nst = (int)((n_rows + SB_SR - 1) / SB_SR);
  MMG_LAUNCH(MMG_PROBE_SCATTER, n_rows, D, total_pad, 0, (k_scatter_strip2<NA, NB>),
             dim3((unsigned)n_ranges, (unsigned)(D / 32)), dim3(512), lds, st, rp, n_rows, nst, D, total_pad, x, slab);
  return MMG_OK;
}


// strip plan: instance (padded tile count) and number of row ranges; ok = false -> k_scatter_units / fp32 kernels
struct StripPlan { bool ok; int nt; int n_ranges; int total_pad; };
StripPlan plan_strip(const mmg_rel_t* rels, int n_rel, int64_t n_rows, int D) {
  StripPlan sp{};
  sp.ok = false;
  int items = 0;
  bool has_rs = false;
  for (int r = 0; r < n_rel; ++r) {
    if (rels[r].n_cols == 0) continue;
    if ((rels[r].flags & MMG_REL_SIMPLE) == 0 || rels[r].mask_t == nullptr) return sp;
    items += rels[r].n_cols;                         // relations packed back to back: tiles may straddle them
    has_rs |= rels[r].rowscale != nullptr;
  }
  const int tiles = (items + 31) / 32;
  // (one wave holds every tile's accumulators + a double-buffered operand set: 10 tiles fill the 512 registers)
  if (tiles == 0 || tiles > 11 || n_rows < SB_SR) return sp;      // (12+ tiles: the unit-per-wave kernel)
  if (has_rs) return sp;
  // tiles of the instance: <4,4>, <5,4> (eICU vocabulary), <5,5>, <6,5>.  (No instance below four tiles per wave: a <2,2>
  // build of the same code returned wrong sums in columns 16..31 of a strip in some launches -- only four matrix
  // instructions per k-step between the rewrites of the piece registers; 40 launches of each instance kept are clean --
  // so a small vocabulary runs the <4,4> instance with empty tiles.)
  const int inst[] = {8, 9, 10, 11};
  for (int i = 0; i < 4; ++i)
    if (tiles <= inst[i]) { sp.nt = inst[i]; break; }
  sp.total_pad = sp.nt * 32;
  const int strips = D / 32;
  int64_t nr = 256 / strips;
  const int64_t nst = (n_rows + SB_SR - 1) / SB_SR;
  if (nr * 4 > nst) nr = (nst + 3) / 4;
  if (nr < 1) nr = 1;
  sp.n_ranges = (int)nr;
  sp.ok = true;
  return sp;
}

// ------------------------------------------------------------------------------ scatter, unit-per-wave layout
// Same arithmetic as k_scatter_bits (0/1 indicator fragments x three exact bf16 pieces of x, fp32 accumulation), laid
// out for the two things the counters said that kernel lost its time to: (1) with ALL item tiles in one wave (160
// accumulator registers) only one wave fits a SIMD, and a lone wave serialises its vector work (piece split, fragment
// expansion) with its matrix instructions -- the matrix pipe idled ~45 % of the loop; (2) 256 partial slabs of
// [320 x 128] were written and read back (84 MB of traffic beside 94 MB of x).
//   * A wave owns ONE "unit": up to SU_NT consecutive 32-item tiles of ONE relation (64 accumulator registers), so a
//     rowscale is one scale vector per wave for ANY vocabulary layout (no compile-time tile -> relation map), and three
//     waves share a SIMD: one multiplies while the others split / expand / wait for memory.
//   * A workgroup owns a 32-column strip of x for a range of rows: its U units x R row sub-ranges (U x R <= 12 waves)
//     stream that strip; the R partial accumulators of a unit are summed through LDS in fixed order, so ONE
//     [padded items x 32] strip leaves the workgroup: 256 / (D / 32) row ranges -> 10.5 MB of partial slabs instead of 42.
//   * The bf16 split uses v_dot2c_f32_bf16 for the residuals (x - hi as ONE instruction: hi.lo * -1 + hi.hi * 0 + x,
//     exact): 28 instead of 52 vector instructions per 8 values.
constexpr int SU_NT = 4;          // tiles per unit
constexpr int SU_MAXU = 6;        // units per launch (<= 24 tiles = 768 padded items)
constexpr int SU_MAXW = 12;       // waves per workgroup (three per SIMD at <= 168 registers)
constexpr int SU_AH = 3;          // k-steps of x in flight per wave (ring of 4)

struct ScUnits {
  const uint64_t* mask[SU_MAXU];  // bit planes of the unit's relation
  int padc[SU_MAXU];              // padded items of that relation (mask row stride)
  int tile0[SU_MAXU];             // first tile of the unit inside its relation
  int ntiles[SU_MAXU];            // 1 .. SU_NT
  int rel[SU_MAXU];               // relation index (rowscale)
  int out_tile[SU_MAXU];          // first tile of the unit in the slab
  int n_units, R, n_waves;
  unsigned char wave_unit[SU_MAXW], wave_rq[SU_MAXW];   // wave -> (unit, row sub-range): balanced over the four SIMDs on the host
};

template <int NT, bool RS>
__device__ __forceinline__ void scatter_unit_body(const ScUnits& su, const RelPack& rp, int u, int rq, int64_t n_rows,
                                                  int n_stage_total, int D, const float* __restrict__ x,
                                                  const unsigned (*lut)[4], float* rss, f32x16* acc) {
  const int lane = threadIdx.x & 63, h = lane >> 5, l31 = lane & 31;
  const int d0 = blockIdx.y * 32;
  // stages of this workgroup, then of this wave
  const int S0 = (int)((int64_t)blockIdx.x * n_stage_total / gridDim.x);
  const int S1 = (int)((int64_t)(blockIdx.x + 1) * n_stage_total / gridDim.x);
  const int s_beg = S0 + (int)((int64_t)rq * (S1 - S0) / su.R), s_end = S0 + (int)((int64_t)(rq + 1) * (S1 - S0) / su.R);
  const int ns = s_end - s_beg;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  if (ns <= 0) return;
  const int64_t r_beg = (int64_t)s_beg * SB_SR;
  const int64_t rows_here = (n_rows - r_beg) < (int64_t)ns * SB_SR ? (n_rows - r_beg) : (int64_t)ns * SB_SR;
  // bit-plane words: lane = (item row of the tile, patient half)
  const int padc = su.padc[u];
  const uint64_t* mb = su.mask[u] + ((size_t)s_beg * padc + (size_t)su.tile0[u] * 32 + l31) * 2 + h;
  const size_t ms = (size_t)2 * padc;
  // x through a buffer descriptor over exactly this wave's rows: rows past the end (and the run-ahead past the last
  // k-step) read 0, no clamps and no exec-masked regions in the loop
  const float* xw = x + (size_t)r_beg * D + d0;
  const __amdgpu_buffer_rsrc_t xsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xw), 0, (int)((rows_here * D - d0) * 4), 0x00020000);
  const unsigned row_bytes = (unsigned)D * 4u;
  const unsigned voff0 = (unsigned)(8 * h) * row_bytes + (unsigned)l31 * 4u;
  const float* rsp = RS ? rp.r[su.rel[u]].rowscale : nullptr;
  float xq[4][8];
  auto loadx = [&](int kg, float* dst) {
    const unsigned vo = voff0 + (unsigned)kg * 16u * row_bytes;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      dst[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xsrc, vo, j * row_bytes, 0));
  };
  auto loadm = [&](int s, uint64_t* dst) {
    const int sc = s < ns ? s : ns - 1;
#pragma unroll
    for (int t = 0; t < NT; ++t) dst[t] = mb[(size_t)sc * ms + (size_t)t * 64];
  };
  auto loadrs = [&](int s) -> float {                 // lane = patient of the stage
    int64_t lr = (int64_t)(s < ns ? s : ns - 1) * SB_SR + lane;
    if (lr > rows_here - 1) lr = rows_here - 1;
    return rsp ? rsp[r_beg + lr] : 1.f;
  };
  uint64_t mc[NT], mn[NT];
  float rsn = 1.f;
  loadm(0, mc);
  if (RS) { rss[lane] = loadrs(0); rsn = loadrs(1); }
#pragma unroll
  for (int q = 0; q < SU_AH; ++q) loadx(q, xq[q]);
  for (int s = 0; s < ns; ++s) {
    loadm(s + 1, mn);
    float* rs_cur = rss + (s & 1) * SB_SR;
    if (RS) { rss[((s + 1) & 1) * SB_SR + lane] = rsn; rsn = loadrs(s + 2); }
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
      loadx(s * 4 + kq + SU_AH, xq[(kq + SU_AH) & 3]);
      bf16x8 b0, b1, b2;
      if (RS) {
        const f32x4s s0 = *reinterpret_cast<const f32x4s*>(rs_cur + kq * 16 + 8 * h);
        const f32x4s s1 = *reinterpret_cast<const f32x4s*>(rs_cur + kq * 16 + 8 * h + 4);
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = xq[kq][j] * s0[j]; v[4 + j] = xq[kq][4 + j] * s1[j]; }
        split8x(v, b0, b1, b2);
      } else {
        split8x(xq[kq], b0, b1, b2);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const unsigned w = (kq & 2) ? (unsigned)(mc[t] >> 32) : (unsigned)mc[t];
        const unsigned off = __builtin_amdgcn_ubfe(w, 16u * (kq & 1), 12u);
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const unsigned char*>(&lut[0][0]) + off);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b0, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b1, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b2, acc[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) mc[t] = mn[t];
  }
}

template <bool RS>
__global__ __launch_bounds__(64 * SU_MAXW) void k_scatter_units(ScUnits su, RelPack rp, int64_t n_rows, int n_stage_total,
                                                                 int D, int total_pad, const float* __restrict__ x,
                                                                 float* __restrict__ slab) {
  extern __shared__ __attribute__((aligned(16))) unsigned char su_lds[];
  unsigned (*lut)[4] = reinterpret_cast<unsigned (*)[4]>(su_lds);                       // [256][4]: byte -> 8 bf16 {0,1}
  float* rss_all = reinterpret_cast<float*>(su_lds + 4096);                              // [waves][2][64] row scales
  float* red = reinterpret_cast<float*>(su_lds + 4096 + SU_MAXW * 2 * SB_SR * 4);        // [units][SU_NT * 16][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: everything derived from it stays scalar
  const int h = lane >> 5, l31 = lane & 31;
  if (tid < 256) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      lut[tid][q] = ((tid >> (2 * q)) & 1 ? 0x3F80u : 0u) | ((tid >> (2 * q + 1)) & 1 ? 0x3F800000u : 0u);
  }
  __syncthreads();
  const int u = su.wave_unit[wid], rq = su.wave_rq[wid];
  const int nt = su.ntiles[u];
  f32x16 acc[SU_NT];
  float* rss = rss_all + wid * 2 * SB_SR;
  switch (nt) {                                     // wave-uniform: one static tile count per code path
    case 1: scatter_unit_body<1, RS>(su, rp, u, rq, n_rows, n_stage_total, D, x, lut, rss, acc); break;
    case 2: scatter_unit_body<2, RS>(su, rp, u, rq, n_rows, n_stage_total, D, x, lut, rss, acc); break;
    case 3: scatter_unit_body<3, RS>(su, rp, u, rq, n_rows, n_stage_total, D, x, lut, rss, acc); break;
    default: scatter_unit_body<4, RS>(su, rp, u, rq, n_rows, n_stage_total, D, x, lut, rss, acc); break;
  }
  // fixed-order sum of the R row sub-ranges of every unit through LDS, then ONE strip per workgroup
  float* mine = red + (size_t)u * (SU_NT * 16 * 64);
  for (int rnd = 1; rnd < su.R; ++rnd) {
    if (rq == rnd) {
#pragma unroll
      for (int t = 0; t < SU_NT; ++t)               // (static register indices; the tile count is wave-uniform)
        if (t < nt) {
#pragma unroll
          for (int i = 0; i < 16; ++i) mine[(t * 16 + i) * 64 + lane] = acc[t][i];
        }
    }
    __syncthreads();
    if (rq == 0) {
#pragma unroll
      for (int t = 0; t < SU_NT; ++t)
        if (t < nt) {
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[t][i] += mine[(t * 16 + i) * 64 + lane];
        }
    }
    __syncthreads();
  }
  if (rq == 0) {
    float* dst = slab + ((size_t)blockIdx.x * total_pad + (size_t)su.out_tile[u] * 32) * D + blockIdx.y * 32 + l31;
#pragma unroll
    for (int t = 0; t < SU_NT; ++t)
      if (t < nt) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int v = t * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
          dst[(size_t)v * D] = acc[t][i];
        }
      }
  }
}

// host side: units (<= SU_NT tiles of one relation each), row sub-ranges, and a wave -> (unit, sub-range) table that
// spreads every SIMD's (wave id % 4) matrix work evenly
struct UnitsPlan { ScUnits su; int n_ranges; int total_pad; size_t lds; bool ok; };

UnitsPlan plan_units(const mmg_rel_t* rels, int n_rel, int64_t n_rows, int D) {
  UnitsPlan up{};
  up.ok = false;
  ScUnits& su = up.su;
  int U = 0, tile_off = 0;
  for (int r = 0; r < n_rel; ++r) {
    const int tiles = pad32(rels[r].n_cols) / 32;
    if (tiles == 0) continue;
    if ((rels[r].flags & MMG_REL_SIMPLE) == 0 || rels[r].mask_t == nullptr) return up;
    const int chunks = (tiles + SU_NT - 1) / SU_NT;
    int t0 = 0;
    for (int c = 0; c < chunks; ++c) {
      const int nt = (tiles - t0 + (chunks - c) - 1) / (chunks - c);      // balanced chunks: 7 -> 4 + 3
      if (U >= SU_MAXU) return up;
      su.mask[U] = rels[r].mask_t; su.padc[U] = tiles * 32; su.tile0[U] = t0; su.ntiles[U] = nt; su.rel[U] = r;
      su.out_tile[U] = tile_off + t0;
      ++U; t0 += nt;
    }
    tile_off += tiles;
  }
  if (U == 0 || n_rows < SB_SR) return up;
  su.n_units = U;
  su.R = U <= 3 ? 4 : (U == 4 ? 3 : 2);
  su.n_waves = U * su.R;
  up.total_pad = tile_off * 32;
  // greedy: heaviest (unit, sub-range) first onto the SIMD with the least tiles that still has a free slot
  int load[4] = {0, 0, 0, 0}, used[4] = {0, 0, 0, 0};
  const int slots = (su.n_waves + 3) / 4;
  int order[SU_MAXU];
  for (int i = 0; i < U; ++i) order[i] = i;
  for (int i = 0; i < U; ++i)
    for (int j = i + 1; j < U; ++j)
      if (su.ntiles[order[j]] > su.ntiles[order[i]]) { int t = order[i]; order[i] = order[j]; order[j] = t; }
  for (int i = 0; i < U; ++i)
    for (int rq = 0; rq < su.R; ++rq) {
      int best = -1;
      for (int sd = 0; sd < 4; ++sd) {
        if (used[sd] >= slots || sd + 4 * used[sd] >= su.n_waves) continue;
        if (best < 0 || load[sd] < load[best]) best = sd;
      }
      if (best < 0) return up;
      const int w = best + 4 * used[best];
      su.wave_unit[w] = (unsigned char)order[i]; su.wave_rq[w] = (unsigned char)rq;
      load[best] += su.ntiles[order[i]]; ++used[best];
    }
  const int strips = D / 32;
  int64_t nr = 256 / strips;
  const int64_t nst = (n_rows + SB_SR - 1) / SB_SR;
  if (nr > nst) nr = nst;
  if (nr < 1) nr = 1;
  up.n_ranges = (int)nr;
  up.lds = 4096 + (size_t)SU_MAXW * 2 * SB_SR * 4 + (size_t)U * SU_NT * 16 * 64 * 4;
  up.ok = true;
  return up;
}

// bit planes of a CSR-by-row relation.  Word [row / 64][col][half] (half = bit 3 of the row inside its 64-row
// stage) carries four 16-bit fields, one per k-step of 16 rows: field ks = (8 patient bits of rows
// 16 ks + 8 half + 0..7) << 4, i.e. the byte offset of the LUT entry that expands them.
__global__ __launch_bounds__(256) void k_mask_build(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                    int64_t n_rows, int padc, unsigned long long* __restrict__ mask) {
  const int lane = threadIdx.x & 63;
  const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  if (row >= n_rows) return;
  const int b = rowptr[row], e = rowptr[row + 1];
  const int p = (int)(row & 63);
  unsigned long long* mrow = mask + ((size_t)(row >> 6) * padc) * 2 + ((p >> 3) & 1);
  const unsigned long long bit = 1ull << (16 * (p >> 4) + 4 + (p & 7));
  for (int k = b + lane; k < e; k += 64) atomicOr(mrow + (size_t)col[k] * 2, bit);
}

// ------------------------------------------------------------------------------ gather on the bf16 matrix cores
// out[i, :] (+)= sum_r rs_r[i] * sum_v Ind_r[i, v] * (cs_r[v] T_r[v, :])  as  Ind [32 patients x items] . T' pieces.
// Mirror image of k_scatter_bits: the 0/1 indicator fragment of a k-step (16 items) is one 12-bit field of the
// patient's row-major bit planes expanded through the 256-entry LDS table, and the (tiny) item tables are split
// ONCE per workgroup into three exact bf16 pieces that stay in registers for every patient tile.  Eight waves:
// wave (ft, kh) owns feature tile ft and HALF of the k-steps (10 x 3 pieces x 4 VGPRs per lane, so two waves per
// SIMD fit); each relation accumulates separately so that its mean scale applies exactly; the kh = 1 waves hand
// their scaled partial sums to their kh = 0 partner through a double-buffered LDS tile (one barrier per patient
// tile), which adds, accumulates and stores.  The loop body is LUT reads + MFMAs; the masks / scales / previous
// output of the next tile are prefetched, and out goes through a buffer descriptor (no exec-masked tails).
// Relation r owns the k-steps [K(r), K(r+1)), K = {0, K1, K2, NK} at compile time.
// NBN: stat_partial receives the statistics of the BatchNorm backward that consumes `out` (NextBnDev, common.h) instead
// of the forward ones -- the last layer's backward hands its patient gradient to the BatchNorm of the layer below.
template <int NK, int K1, int K2, bool ACCUM, int KH, bool NBN>
__device__ __forceinline__ void gather_bits_body(const RelPack& rp, int64_t n_rows, int n_tile_total, int D,
                                                 float* __restrict__ out, const unsigned (*lut)[4],
                                                 float (*rss)[3][32], float (*xch)[4][16][64],
                                                 double* __restrict__ stat_partial, NextBnDev nb) {
  double cs1 = 0.0, cs2 = 0.0;       // column sums / sums of squares of the final output (lane = feature column)
  constexpr int KB = KH * (NK / 2), KE = KB + NK / 2;               // this wave's k-steps
  constexpr bool USE0 = KB < K1, USE1 = KB < K2 && KE > K1, USE2 = KE > K2;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31, ft = wid & 3;
  const int dcol = blockIdx.y * 128 + ft * 32 + l31;
  NextBnCol nbc = {};
  if constexpr (NBN && KH == 0) nbc = next_bn_col(nb, dcol);
  // ---- table pieces: B[k = item 16 ks + 8 h + j][n = this lane's feature column]
  // (every load is issued before any is used: indices clamped, no branch in between -- a conditional load per element
  //  made this prologue a chain of 2 x 80 dependent round trips, ~25 us of a 67 us launch)
  bf16x8 tb[NK / 2][3];
  {
    float tv[NK / 2][8], cv[NK / 2][8];
#pragma unroll
    for (int q = 0; q < NK / 2; ++q) {
      const int ks = KB + q;
      const int r = ks < K1 ? 0 : (ks < K2 ? 1 : 2);
      const int kr = ks - (r == 0 ? 0 : (r == 1 ? K1 : K2));
      const RelDev& R = rp.r[r];
      const float* cs = R.colscale ? R.colscale : R.table;      // (any readable address when there is no scale)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int item = kr * 16 + 8 * h + j;
        const int ic = item < R.n_cols ? item : 0;
        tv[q][j] = R.table[(size_t)ic * D + dcol];
        cv[q][j] = cs[ic];
      }
    }
#pragma unroll
    for (int q = 0; q < NK / 2; ++q) {
      const int ks = KB + q;
      const int r = ks < K1 ? 0 : (ks < K2 ? 1 : 2);
      const int kr = ks - (r == 0 ? 0 : (r == 1 ? K1 : K2));
      const RelDev& R = rp.r[r];
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int item = kr * 16 + 8 * h + j;
        const float t = R.colscale ? tv[q][j] * cv[q][j] : tv[q][j];
        v[j] = item < R.n_cols ? t : 0.f;
      }
      split8(v, tb[q][0], tb[q][1], tb[q][2]);
    }
  }
  constexpr int NF0 = K1, NF1 = K2 - K1, NF2 = NK - K2;             // 16-bit fields per (row, half)
  static_assert(NF0 % 2 == 0 && NF1 % 2 == 0 && NF2 % 2 == 0 && K1 % 2 == 0 && K2 % 2 == 0 && NK % 4 == 0,
                "fields are loaded as dwords");
  // 32-patient tiles are dealt round-robin: workgroup b takes tiles b, b + G, ... (the grid then works on one contiguous
  // window of `out` and of the bit planes, which spreads over every HBM channel; a contiguous range per workgroup makes
  // G read-modify-write streams advance a fixed stride apart).  `out` is addressed through one descriptor over the
  // whole tensor (the launcher checks that it is below 4 GB).
  const int t_beg = blockIdx.x, t_step = gridDim.x, t_end = n_tile_total;
  const int64_t last_row = n_rows - 1;
  const __amdgpu_buffer_rsrc_t osrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)(unsigned)(n_rows * D * 4), 0x00020000);
  const unsigned row_bytes = (unsigned)D * 4u;
  unsigned mcur[NK / 4], mnxt[NK / 4];                              // fields KB..KE-1 as dwords
  float rsn[3], prev[16];
  auto loadm = [&](int tile, unsigned* dst) {                       // this lane's patient and item half
    int64_t row = (int64_t)tile * 32 + l31;
    if (row > last_row) row = last_row;
#pragma unroll
    for (int i = 0; i < NK / 4; ++i) {
      const int ks = KB + 2 * i;                                    // static: relation and offset inside it
      const int r = ks < K1 ? 0 : (ks < K2 ? 1 : 2);
      const int kr = ks - (r == 0 ? 0 : (r == 1 ? K1 : K2));
      const int nf = r == 0 ? NF0 : (r == 1 ? NF1 : NF2);
      dst[i] = reinterpret_cast<const unsigned*>(rp.r[r].mask_r)[((size_t)row * 2 + h) * (nf / 2) + kr / 2];
    }
  };
  auto loadrs = [&](int tile, float* dst) {
    int64_t row = (int64_t)tile * 32 + l31;
    if (row > last_row) row = last_row;
#pragma unroll
    for (int r = 0; r < 3; ++r) dst[r] = (r < rp.n && rp.r[r].rowscale) ? rp.r[r].rowscale[row] : 1.f;
  };
  auto loadprev = [&](int tile, float* dst) {
    const unsigned vo = ((unsigned)(tile * 32 + 4 * h) * (unsigned)D + (unsigned)dcol) * 4u;
#pragma unroll
    for (int i = 0; i < 16; ++i)
      dst[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(osrc, vo, ((i & 3) + 8 * (i >> 2)) * row_bytes, 0));
  };
  // next-BatchNorm statistics: the pre-BatchNorm activation of the layer below, same shape and addressing as `out`; one
  // tile ahead where registers allow (few table pieces), else in flight under this tile's products
  constexpr bool NB_ON = NBN && KH == 0, NB_AHEAD = NB_ON && NK <= 8;
  const __amdgpu_buffer_rsrc_t ysrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(NB_ON ? nb.Y : out), 0,
                                                                        (int)(unsigned)(n_rows * D * 4), 0x00020000);
  float ynx[NB_ON ? 16 : 1];
  auto loady = [&](int tile, float* dst) {
    const unsigned vo = ((unsigned)(tile * 32 + 4 * h) * (unsigned)D + (unsigned)dcol) * 4u;
#pragma unroll
    for (int i = 0; i < 16; ++i)
      dst[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ysrc, vo, ((i & 3) + 8 * (i >> 2)) * row_bytes, 0));
  };
  if (t_beg < t_end) {
    loadm(t_beg, mcur); loadrs(t_beg, rsn);
    if (ACCUM && KH == 0) loadprev(t_beg, prev);
    if constexpr (NB_AHEAD) loady(t_beg, ynx);
  }

  // The tile body, called once before the loop (peeled): the loop is then entered with the same loads / stores in flight
  // as on its back edge, so the vmcnt waits the compiler derives (the minimum over both edges) are the steady-state ones
  // and the stores of a tile are not waited for at the top of the next one.
  auto tile_body = [&](int tile, int par) {
    const int tn = tile + t_step < t_end ? tile + t_step : tile;
    loadm(tn, mnxt);
    if (h == 0) {
#pragma unroll
      for (int r = 0; r < 3; ++r) rss[wid][r][l31] = rsn[r];        // private to this wave
    }
    loadrs(tn, rsn);
    float pc[16];
    if (ACCUM && KH == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) pc[i] = prev[i];
      loadprev(tn, prev);                                           // consumed one tile later
    }
    float yc[NB_ON ? 16 : 1];
    if constexpr (NB_AHEAD) {
#pragma unroll
      for (int i = 0; i < 16; ++i) yc[i] = ynx[i];
      loady(tn, ynx);
    } else if constexpr (NB_ON) {
      loady(tile, yc);
    }
    f32x16 acc[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
#pragma unroll
    for (int q = 0; q < NK / 2; ++q) {
      const int ks = KB + q;
      const int r = ks < K1 ? 0 : (ks < K2 ? 1 : 2);
      const unsigned off = __builtin_amdgcn_ubfe(mcur[q >> 1], 16u * (q & 1), 12u);
      const bf16x8 af = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const unsigned char*>(&lut[0][0]) + off);
      acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, tb[q][0], acc[r], 0, 0, 0);
      acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, tb[q][1], acc[r], 0, 0, 0);
      acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, tb[q][2], acc[r], 0, 0, 0);
    }
    // scaled partial in the C layout: register i <-> patient row (i & 3) + 8 (i >> 2) + 4 h, lane <-> feature column
    float v[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4s s0, s1, s2;
      if (USE0) s0 = *reinterpret_cast<const f32x4s*>(&rss[wid][0][8 * q + 4 * h]);
      if (USE1) s1 = *reinterpret_cast<const f32x4s*>(&rss[wid][1][8 * q + 4 * h]);
      if (USE2) s2 = *reinterpret_cast<const f32x4s*>(&rss[wid][2][8 * q + 4 * h]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = 4 * q + e;
        float t = 0.f;
        if (USE0) t = s0[e] * acc[0][i];
        if (USE1) t = fmaf(s1[e], acc[1][i], t);
        if (USE2) t = fmaf(s2[e], acc[2][i], t);
        v[i] = t;
      }
    }
    float (*xb)[64] = xch[par][ft];
    if (KH == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) xb[i][lane] = v[i];
    }
    __syncthreads();
    if (KH == 0) {
      const unsigned vo = ((unsigned)(tile * 32 + 4 * h) * (unsigned)D + (unsigned)dcol) * 4u;
      if constexpr (NBN) {
        float tv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float t = v[i] + xb[i][lane];
          if (ACCUM) t += pc[i];
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, t), osrc, vo, ((i & 3) + 8 * (i >> 2)) * row_bytes, 0);
          tv[i] = t;
        }
        const int64_t left = n_rows - (int64_t)tile * 32;
        next_bn_tile(nb, nbc, tv, yc, left < 32 ? (int)left : 32, (int64_t)tile * 32, D, dcol, lane, cs1, cs2);
      } else {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float t = v[i] + xb[i][lane];
          if (ACCUM) t += pc[i];
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, t), osrc, vo, ((i & 3) + 8 * (i >> 2)) * row_bytes, 0);
          if ((int64_t)tile * 32 + (i & 3) + 8 * (i >> 2) + 4 * h < n_rows) { t1 += t; t2 = fmaf(t, t, t2); }
        }
        if (stat_partial) { cs1 += (double)t1; cs2 += (double)t2; }
      }
    }
#pragma unroll
    for (int i = 0; i < NK / 4; ++i) mcur[i] = mnxt[i];
  };
  if (t_beg < t_end) {
    tile_body(t_beg, 0);
    int par = 1;
    for (int tile = t_beg + t_step; tile < t_end; tile += t_step, par ^= 1) tile_body(tile, par);
  }
  if (stat_partial) {                    // (a workgroup-uniform branch: both wave halves take it)
    __syncthreads();                     // the exchange tiles are dead
    double* red = reinterpret_cast<double*>(&xch[0][0][0][0]);      // [h][which][128]
    if (KH == 0) {
      red[(h * 2 + 0) * 128 + ft * 32 + l31] = cs1;
      red[(h * 2 + 1) * 128 + ft * 32 + l31] = cs2;
    }
    __syncthreads();
    if (KH == 0) {
      const int which = tid >> 7, c = tid & 127;
      stat_partial[((size_t)blockIdx.x * 2 + which) * D + blockIdx.y * 128 + c] = red[which * 128 + c] + red[(2 + which) * 128 + c];
    }
  }
}

template <int NK, int K1, int K2, bool ACCUM, bool NBN = false>
__global__ __launch_bounds__(512) void k_gather_bits(RelPack rp, int64_t n_rows, int n_tile_total, int D,
                                                     float* __restrict__ out, double* __restrict__ stat_partial,
                                                     NextBnDev nb) {
  __shared__ __attribute__((aligned(16))) unsigned lut[256][4];
  __shared__ __attribute__((aligned(16))) float rss[8][3][32];
  __shared__ __attribute__((aligned(16))) float xch[2][4][16][64];
  const int tid = threadIdx.x;
  if (tid < 256) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      lut[tid][q] = ((tid >> (2 * q)) & 1 ? 0x3F80u : 0u) | ((tid >> (2 * q + 1)) & 1 ? 0x3F800000u : 0u);
  }
  __syncthreads();
  if ((tid >> 8) == 0) gather_bits_body<NK, K1, K2, ACCUM, 0, NBN>(rp, n_rows, n_tile_total, D, out, lut, rss, xch, stat_partial, nb);
  else gather_bits_body<NK, K1, K2, ACCUM, 1, NBN>(rp, n_rows, n_tile_total, D, out, lut, rss, xch, stat_partial, nb);
}

// ------------------------------------------------------------------------------ gather, unit-per-wave layout
// k_gather_bits above needs its relation boundaries at compile time (the eICU vocabulary).  Every other layout of simple
// relations up to 768 padded items -- the MIMIC-III schema's 50 / 200 / 100 (conf/config.yaml:70,101,112 of the
// reference) among them -- takes this kernel: a wave owns one feature tile and ONE unit = up to GU_KU k-steps (16 items
// each) of ONE relation, keeps that unit's three bf16 table pieces in registers, and its relation's mean scale is one
// vector per wave; the units' scaled partial tiles are summed through LDS in fixed order (unit 0 + 1 + 2 ...) by the
// unit-0 wave, which also accumulates into `out` and takes the BatchNorm column sums.  Same arithmetic as k_gather_bits.
constexpr int GU_KU = 8;          // k-steps per unit (128 items): 96 registers of table pieces
constexpr int GU_MAXU = 6;        // units per launch
#ifndef MMG_GU_DEPTH
#define MMG_GU_DEPTH 3
#endif
constexpr int GU_DEPTH = MMG_GU_DEPTH;   // tiles of look-ahead of the inputs
struct GaUnits {
  int rel[GU_MAXU], ks0[GU_MAXU], nks[GU_MAXU];   // relation, first k-step inside it (even), k-steps (even, <= GU_KU)
  int nf[GU_MAXU];                                // uint16 fields per (row, half) of that relation = padded items / 16
  int n_units, FT;                                // feature tiles (32 columns) per workgroup: waves = FT * n_units
};

template <bool ACCUM>
__global__ __launch_bounds__(512) void k_gather_units(GaUnits gu, RelPack rp, int64_t n_rows, int n_tile_total, int D,
                                                      float* __restrict__ out, double* __restrict__ stat_partial) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gu_lds[];
  unsigned (*lut)[4] = reinterpret_cast<unsigned (*)[4]>(gu_lds);                      // [256][4]
  float* rss_all = reinterpret_cast<float*>(gu_lds + 4096);                             // [waves <= 8][32]
  float* xch = reinterpret_cast<float*>(gu_lds + 4096 + 8 * 32 * 4);                    // [2][FT][U - 1][16][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const int FT = gu.FT, U = gu.n_units;
  const int ft = wid % FT, u = wid / FT;
  for (int e = tid; e < 256; e += blockDim.x) {         // (a workgroup can be as small as two waves)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      lut[e][q] = ((e >> (2 * q)) & 1 ? 0x3F80u : 0u) | ((e >> (2 * q + 1)) & 1 ? 0x3F800000u : 0u);
  }
  __syncthreads();
  const int dcol = (blockIdx.y * FT + ft) * 32 + l31;
  const RelDev& R = rp.r[gu.rel[u]];
  const int ks0 = gu.ks0[u], nks = gu.nks[u], nf = gu.nf[u];
  // ---- table pieces of this unit: B[k = item 16 (ks0 + q) + 8 h + j][n = this lane's feature column], pre-scaled by
  // colscale; every load is issued before any is used (indices clamped, no branch in between)
  bf16x8 tb[GU_KU][3];
#pragma unroll
  for (int q0 = 0; q0 < GU_KU; q0 += 2) {           // two k-steps at a time: 32 loads in flight, 32 registers of staging
    float tv[2][8], cv[2][8];
    const float* cs = R.colscale ? R.colscale : R.table;
#pragma unroll
    for (int qq = 0; qq < 2; ++qq)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int q = q0 + qq, item = (ks0 + q) * 16 + 8 * h + j;
        const int ic = (q < nks && item < R.n_cols) ? item : 0;
        tv[qq][j] = R.table[(size_t)ic * D + dcol];
        cv[qq][j] = cs[ic];
      }
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
      const int q = q0 + qq;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int item = (ks0 + q) * 16 + 8 * h + j;
        const float t = R.colscale ? tv[qq][j] * cv[qq][j] : tv[qq][j];
        v[j] = (q < nks && item < R.n_cols) ? t : 0.f;
      }
      split8(v, tb[q][0], tb[q][1], tb[q][2]);
    }
  }
  const int t_step = gridDim.x;
  const int64_t last_row = n_rows - 1;
  const __amdgpu_buffer_rsrc_t osrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)(unsigned)(n_rows * D * 4), 0x00020000);
  const unsigned row_bytes = (unsigned)D * 4u;
  const unsigned* mrow = reinterpret_cast<const unsigned*>(R.mask_r);
  // Inputs run GU_DEPTH tiles ahead in as many register sets (iteration i uses set i % GU_DEPTH): a tile takes less time
  // than a loaded HBM round trip, so with one tile of look-ahead the loop ran at one round trip per tile whatever the tile
  // held.  A set is refilled right after its last use in an iteration (the masks after the indicator reads, the previous
  // output after the store), which needs no copy: two sets take the registers that one set + its copy took before.
  unsigned mS_[GU_DEPTH][GU_KU / 2];
  float rsS_[GU_DEPTH], pS_[GU_DEPTH][16];
  auto loadm = [&](int tile, unsigned* dst) {                       // this lane's patient (l31) and item half (h)
    int64_t row = (int64_t)tile * 32 + l31;
    if (row > last_row) row = last_row;
    const size_t base = ((size_t)row * 2 + h) * (nf / 2) + ks0 / 2;
#pragma unroll
    for (int i = 0; i < GU_KU / 2; ++i) dst[i] = mrow[base + (2 * i < nks ? i : 0)];
  };
  auto loadrs = [&](int tile) -> float {
    int64_t row = (int64_t)tile * 32 + l31;
    if (row > last_row) row = last_row;
    return R.rowscale ? R.rowscale[row] : 1.f;
  };
  auto loadprev = [&](int tile, float* dst) {
    const unsigned vo = ((unsigned)(tile * 32 + 4 * h) * (unsigned)D + (unsigned)dcol) * 4u;
#pragma unroll
    for (int i = 0; i < 16; ++i)
      dst[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(osrc, vo, ((i & 3) + 8 * (i >> 2)) * row_bytes, 0));
  };
  float* rss = rss_all + wid * 32;
  double cs1 = 0.0, cs2 = 0.0;
  const int t_beg = blockIdx.x;
  if (t_beg < n_tile_total) {
#pragma unroll
    for (int d = 0; d < GU_DEPTH; ++d) {
      const int td = t_beg + d * t_step < n_tile_total ? t_beg + d * t_step : t_beg;
      loadm(td, mS_[d]); rsS_[d] = loadrs(td);
      if (ACCUM && u == 0) loadprev(td, pS_[d]);
    }
  }
  auto tile_body = [&](int tile, int par, unsigned* mS, float& rsS, float* pS) __attribute__((always_inline)) {
    const int tn = tile + GU_DEPTH * t_step < n_tile_total ? tile + GU_DEPTH * t_step : tile;     // (any valid tile: loaded, never used)
    if (h == 0) rss[l31] = rsS;                                      // private to this wave
    rsS = loadrs(tn);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int q = 0; q < GU_KU; ++q)
      if (q < nks) {                                                 // wave-uniform
        const unsigned off = __builtin_amdgcn_ubfe(mS[q >> 1], 16u * (q & 1), 12u);
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const unsigned char*>(&lut[0][0]) + off);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, tb[q][0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, tb[q][1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, tb[q][2], acc, 0, 0, 0);
      }
    loadm(tn, mS);
    // scaled partial in the C layout: register i <-> patient row (i & 3) + 8 (i >> 2) + 4 h, lane <-> feature column
    float v[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4s sc = *reinterpret_cast<const f32x4s*>(&rss[8 * q + 4 * h]);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[4 * q + e] = sc[e] * acc[4 * q + e];
    }
    float* xb = xch + ((size_t)(par * FT + ft) * (U - 1)) * (16 * 64);
    if (u > 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) xb[((size_t)(u - 1) * 16 + i) * 64 + lane] = v[i];
    }
    __syncthreads();
    if (u == 0) {
      const unsigned vo = ((unsigned)(tile * 32 + 4 * h) * (unsigned)D + (unsigned)dcol) * 4u;
      const int rows_left = (int)(n_rows - (int64_t)tile * 32 < 32 ? n_rows - (int64_t)tile * 32 : 32);
      float tt[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) tt[i] = v[i];
#pragma unroll
      for (int uu = 1; uu < GU_MAXU; ++uu)                    // fixed order: unit 0 + 1 + 2 + ...
        if (uu < U) {                                          // (workgroup-uniform)
#pragma unroll
          for (int i = 0; i < 16; ++i) tt[i] += xb[((size_t)(uu - 1) * 16 + i) * 64 + lane];
        }
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float t = tt[i];
        if (ACCUM) t += pS[i];
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, t), osrc, vo, ((i & 3) + 8 * (i >> 2)) * row_bytes, 0);
        if ((i & 3) + 8 * (i >> 2) + 4 * h < rows_left) { t1 += t; t2 = fmaf(t, t, t2); }
      }
      if (stat_partial) { cs1 += (double)t1; cs2 += (double)t2; }
      if (ACCUM) loadprev(tn, pS);
    }
  };
  // (peeling the first round, as k_gather_bits does, made this loop slower: 113 us against 94 at the MIMIC-III shape)
  int par = 0;
  for (int tile = t_beg; tile < n_tile_total; tile += GU_DEPTH * t_step) {
#pragma unroll
    for (int d = 0; d < GU_DEPTH; ++d)
      if (tile + d * t_step < n_tile_total) {                                          // (workgroup-uniform)
        tile_body(tile + d * t_step, par, mS_[d], rsS_[d], pS_[d]);
        par ^= 1;
      }
  }
  if (stat_partial && u == 0) {               // the two lane halves hold different rows of the same column
    cs1 += __shfl_xor(cs1, 32, 64);
    cs2 += __shfl_xor(cs2, 32, 64);
    if (h == 0) {
      stat_partial[((size_t)blockIdx.x * 2 + 0) * D + dcol] = cs1;
      stat_partial[((size_t)blockIdx.x * 2 + 1) * D + dcol] = cs2;
    }
  }
}

struct GaPlan { GaUnits gu; bool ok; int waves; size_t lds; };
GaPlan plan_gather_units(const mmg_rel_t* rels, int n_rel, int64_t n_rows, int D) {
  GaPlan gp{};
  gp.ok = false;
  if (D % 32 || n_rows < 32 || (uint64_t)n_rows * (uint64_t)D * 4u >= (1ull << 31)) return gp;
  int U = 0;
  for (int r = 0; r < n_rel; ++r) {
    if (rels[r].n_cols == 0) continue;
    if ((rels[r].flags & MMG_REL_SIMPLE) == 0 || rels[r].mask_r == nullptr) return gp;
    const int padc = pad32(rels[r].n_cols), nk = padc / 16;
    const int chunks = (nk + GU_KU - 1) / GU_KU;
    int k0 = 0;
    for (int c = 0; c < chunks; ++c) {
      int n = (nk - k0 + (chunks - c) - 1) / (chunks - c);
      n = (n + 1) & ~1;                                  // even: fields are read as dwords
      if (n > nk - k0) n = nk - k0;
      if (U >= GU_MAXU) return gp;
      gp.gu.rel[U] = r; gp.gu.ks0[U] = k0; gp.gu.nks[U] = n; gp.gu.nf[U] = padc / 16;
      ++U; k0 += n;
    }
  }
  if (U == 0) return gp;
  gp.gu.n_units = U;
  // two waves per SIMD (<= 256 registers: 96 of table pieces, the accumulator, the previous output tile, ...)
  const int tiles_d = D / 32;
  int FT = 4;
  while (FT > 1 && (FT * U > 8 || tiles_d % FT)) FT >>= 1;
  gp.gu.FT = FT;
  gp.waves = FT * U;
  gp.lds = 4096 + 8 * 32 * 4 + (size_t)2 * FT * (U > 1 ? U - 1 : 1) * 16 * 64 * 4;
  gp.ok = gp.waves <= 8;
  return gp;
}

// row-major bit planes: field [row][half][col / 16] (uint16) |= 1 << (4 + col % 8), half = (col % 16) / 8
__global__ __launch_bounds__(256) void k_mask_build_rows(const int32_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ col, int64_t n_rows, int nf,
                                                         unsigned* __restrict__ mask) {
  const int lane = threadIdx.x & 63;
  const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  if (row >= n_rows) return;
  const int b = rowptr[row], e = rowptr[row + 1];
  for (int k = b + lane; k < e; k += 64) {
    const int c = col[k];
    const size_t f = ((size_t)row * 2 + ((c >> 3) & 1)) * nf + (c >> 4);       // uint16 index
    atomicOr(mask + (f >> 1), 1u << (4 + (c & 7) + 16 * (int)(f & 1)));
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
    if (has_rs) MMG_LAUNCH(MMG_PROBE_SCATTER, n_rows, D, NT * 32, 8 | 16, (k_scatter_mfma<NT, 4, true>), grid, dim3(256), 0, st, rp, n_rows, p.rows_per_split, D, x, slab);
    else MMG_LAUNCH(MMG_PROBE_SCATTER, n_rows, D, NT * 32, 16, (k_scatter_mfma<NT, 4, false>), grid, dim3(256), 0, st, rp, n_rows, p.rows_per_split, D, x, slab);
  } else {
    if (has_rs) MMG_LAUNCH(MMG_PROBE_SCATTER, n_rows, D, NT * 32, 8 | 16, (k_scatter_mfma<NT, 2, true>), grid, dim3(256), 0, st, rp, n_rows, p.rows_per_split, D, x, slab);
    else MMG_LAUNCH(MMG_PROBE_SCATTER, n_rows, D, NT * 32, 16, (k_scatter_mfma<NT, 2, false>), grid, dim3(256), 0, st, rp, n_rows, p.rows_per_split, D, x, slab);
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
                      rels[r].out, rels[r].n_cols, off, rels[r].flags, rels[r].mask_t, rels[r].mask_r};
    off += pad_cols ? ((rels[r].n_cols + 31) & ~31) : rels[r].n_cols;
  }
  return MMG_OK;
}

}  // namespace

extern "C" int mmg_partial_sum(const double* partial, double* out, int n, int n_rows, void* stream);
extern "C" int mmg_partial_sum_bn(const double* partial, double* col_sums, int N, int n_rows, const mmg_bn_fin_t* fin, void* stream);
extern "C" int mmg_bn_finalize(const double* sums, int64_t count, const float* gamma, const float* beta, float* running_mean,
                               float* running_var, int training, int n_updates, float momentum, float eps, float* scale,
                               float* shift, float* mean, float* rstd, int N, void* stream);
extern "C" int mmg_gather_rows_stats(const mmg_rel_t* rels, int n_rel, int64_t n_rows, int D, float* out, int accumulate,
                                     double* col_sums, void* ws, size_t ws_bytes, void* stream);

extern "C" size_t mmg_gather_rows_stats_ws_bytes(int64_t n_rows, int D) {
  if (n_rows < 0 || D <= 0) return 0;
  const size_t a = (size_t)256 * 2 * D * sizeof(double) + 256, b = mmg_col_reduce2_ws_bytes(n_rows, D);
  return a > b ? a : b;
}

extern "C" int mmg_gather_rows(const mmg_rel_t* rels, int n_rel, int64_t n_rows, int D, float* out, int accumulate,
                               void* stream) {
  return mmg_gather_rows_stats(rels, n_rel, n_rows, D, out, accumulate, nullptr, nullptr, 0, stream);
}

extern "C" int mmg_next_bn_dev(const mmg_next_bn_t* next, int64_t M, int N, const char* what, NextBnDev* d, double** partial);
extern "C" int mmg_next_bn_finish(const mmg_next_bn_t* next, const double* partial, int N, int rows, void* stream);
extern "C" int mmg_next_bn_fallback(const float* G, int64_t M, int N, const mmg_next_bn_t* next, const char* what, void* stream);

// next (nullable, exclusive with col_sums): the statistics of the BatchNorm backward that consumes `out` (mmg_next_bn_t)
static int gather_rows_stats_impl(const mmg_rel_t* rels, int n_rel, int64_t n_rows, int D, float* out, int accumulate,
                                  double* col_sums, void* ws, size_t ws_bytes, void* stream, const mmg_bn_fin_t* fin,
                                  const mmg_next_bn_t* next = nullptr) {
  MMG_CHECK_ARG(!(next && col_sums), "gather_rows: forward statistics and next-BatchNorm statistics are exclusive");
  NextBnDev nbd = next_bn_none();
  double* nb_partial = nullptr;
  const bool nb_fused = next && next->pro && (next->pro->relu == MMG_ACT_NONE || next->pro->relu == MMG_ACT_RELU);
  if (next) {
    int rcn = mmg_next_bn_dev(next, n_rows, D, "gather_rows_next_bn", &nbd, &nb_partial);
    if (rcn) return rcn;
  }
  if (col_sums) {
    MMG_CHECK_ARG(n_rows > 0 && ws && ws_bytes >= mmg_gather_rows_stats_ws_bytes(n_rows, D),
                  "gather_rows_stats: workspace too small");
  }
  double* partial = col_sums ? (double*)(((uintptr_t)ws + 255) & ~(uintptr_t)255) : nullptr;
#define MMG_GATHER_TAIL(what)                                                                                 \
  do {                                                                                                        \
    MMG_CHECK_LAUNCH(what);                                                                                   \
    if (next) return mmg_next_bn_fallback(out, n_rows, D, next, "gather_rows_next_bn", stream);               \
    if (col_sums) {                                                                                           \
      int rc_ = mmg_col_reduce2(out, nullptr, col_sums, n_rows, D, ws, ws_bytes, stream);                     \
      if (rc_ || !fin) return rc_;                                                                            \
      return mmg_bn_finalize(col_sums, fin->count, fin->gamma, fin->beta, fin->running_mean, fin->running_var, 1, \
                             fin->n_updates, fin->momentum, fin->eps, fin->scale, fin->shift, fin->mean, fin->rstd, D, stream); \
    }                                                                                                         \
    return MMG_OK;                                                                                            \
  } while (0)
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
  // bit-plane matrix-core kernel: simple relations in a layout with a static instance -- the eICU vocabulary
  // (64 | 128 | 128 padded items) or its first relation alone (the last layer's backward only reaches the labs)
  {
    bool okb = (n_rel == 3 || n_rel == 1) && D >= 128 && n_rows >= 32 &&
               (uint64_t)n_rows * (uint64_t)D * 4u < (1ull << 32);     // `out` sits behind one 32-bit buffer descriptor
    for (int r = 0; r < n_rel && okb; ++r) okb = (rels[r].flags & MMG_REL_SIMPLE) != 0 && rels[r].mask_r != nullptr;
    okb = okb && ((rels[0].n_cols + 31) & ~31) == 64;
    if (n_rel == 3) okb = okb && ((rels[1].n_cols + 31) & ~31) == 128 && ((rels[2].n_cols + 31) & ~31) == 128;
    if (okb && n_rel == 1) {
      const int n_tiles = (int)((n_rows + 31) / 32);
      const int n_dchunks = D / 128;
      int g = 256 / n_dchunks;
      if (g > n_tiles) g = n_tiles;
      dim3 grid((unsigned)g, (unsigned)n_dchunks);
      if (nb_fused) {
        if (accumulate) MMG_LAUNCH(MMG_PROBE_GATHER, n_rows, D, total_cols, 1 | 512, (k_gather_bits<4, 4, 4, true, true>), grid, dim3(512), 0, st, rp, n_rows, n_tiles, D, out, nb_partial, nbd);
        else MMG_LAUNCH(MMG_PROBE_GATHER, n_rows, D, total_cols, 512, (k_gather_bits<4, 4, 4, false, true>), grid, dim3(512), 0, st, rp, n_rows, n_tiles, D, out, nb_partial, nbd);
        MMG_CHECK_LAUNCH("gather_rows(bits)");
        return mmg_next_bn_finish(next, nb_partial, D, g, stream);
      }
      if (accumulate) MMG_LAUNCH(MMG_PROBE_GATHER, n_rows, D, total_cols, 1, (k_gather_bits<4, 4, 4, true, false>), grid, dim3(512), 0, st, rp, n_rows, n_tiles, D, out, partial, next_bn_none());
      else MMG_LAUNCH(MMG_PROBE_GATHER, n_rows, D, total_cols, 0, (k_gather_bits<4, 4, 4, false, false>), grid, dim3(512), 0, st, rp, n_rows, n_tiles, D, out, partial, next_bn_none());
      MMG_CHECK_LAUNCH("gather_rows(bits)");
      if (next) return mmg_next_bn_fallback(out, n_rows, D, next, "gather_rows_next_bn", stream);
      if (col_sums) return mmg_partial_sum_bn(partial, col_sums, D, g, fin, stream);
      return MMG_OK;
    }
    if (okb) {
      const int n_tiles = (int)((n_rows + 31) / 32);
      const int n_dchunks = D / 128;
      int g = 256 / n_dchunks;
      if (g > n_tiles) g = n_tiles;
      dim3 grid((unsigned)g, (unsigned)n_dchunks);
      if (nb_fused) {
        if (accumulate) MMG_LAUNCH(MMG_PROBE_GATHER, n_rows, D, total_cols, 1 | 512, (k_gather_bits<20, 4, 12, true, true>), grid, dim3(512), 0, st, rp, n_rows, n_tiles, D, out, nb_partial, nbd);
        else MMG_LAUNCH(MMG_PROBE_GATHER, n_rows, D, total_cols, 512, (k_gather_bits<20, 4, 12, false, true>), grid, dim3(512), 0, st, rp, n_rows, n_tiles, D, out, nb_partial, nbd);
        MMG_CHECK_LAUNCH("gather_rows(bits)");
        return mmg_next_bn_finish(next, nb_partial, D, g, stream);
      }
      if (accumulate) MMG_LAUNCH(MMG_PROBE_GATHER, n_rows, D, total_cols, 1, (k_gather_bits<20, 4, 12, true, false>), grid, dim3(512), 0, st, rp, n_rows, n_tiles, D, out, partial, next_bn_none());
      else MMG_LAUNCH(MMG_PROBE_GATHER, n_rows, D, total_cols, 0, (k_gather_bits<20, 4, 12, false, false>), grid, dim3(512), 0, st, rp, n_rows, n_tiles, D, out, partial, next_bn_none());
      MMG_CHECK_LAUNCH("gather_rows(bits)");
      if (next) return mmg_next_bn_fallback(out, n_rows, D, next, "gather_rows_next_bn", stream);
      if (col_sums) return mmg_partial_sum_bn(partial, col_sums, D, g, fin, stream);
      return MMG_OK;
    }
  }
  // any other layout of simple relations with bit planes: one unit (<= 128 items of one relation) per wave
  {
    const GaPlan gp = plan_gather_units(rels, n_rel, n_rows, D);
    if (gp.ok) {
      const int n_tiles = (int)((n_rows + 31) / 32);
      const int gy = (D / 32) / gp.gu.FT;
      int g = 256 / gy;
      if (g < 1) g = 1;
      if (g > n_tiles) g = n_tiles;
      if (g > 256) g = 256;                              // <= 256 partial statistic rows (workspace)
      dim3 grid((unsigned)g, (unsigned)gy);
      constexpr int lds_max = 4096 + 8 * 32 * 4 + 2 * 6 * 16 * 64 * 4;     // FT * (U - 1) <= 6 exchange tiles, double-buffered
      if (accumulate) {
        MMG_CHECK_HIP((MmgMaxLds<&k_gather_units<true>, lds_max>::set()), "gather_rows(attr)");
        MMG_LAUNCH(MMG_PROBE_GATHER, n_rows, D, total_cols, 1 | 32, k_gather_units<true>, grid, dim3(64 * gp.waves), gp.lds, st,
                   gp.gu, rp, n_rows, n_tiles, D, out, partial);
      } else {
        MMG_CHECK_HIP((MmgMaxLds<&k_gather_units<false>, lds_max>::set()), "gather_rows(attr)");
        MMG_LAUNCH(MMG_PROBE_GATHER, n_rows, D, total_cols, 32, k_gather_units<false>, grid, dim3(64 * gp.waves), gp.lds, st,
                   gp.gu, rp, n_rows, n_tiles, D, out, partial);
      }
      MMG_CHECK_LAUNCH("gather_rows(units)");
      if (next) return mmg_next_bn_fallback(out, n_rows, D, next, "gather_rows_next_bn", stream);
      if (col_sums) return mmg_partial_sum_bn(partial, col_sums, D, g, fin, stream);
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
      MMG_CHECK_HIP((MmgMaxLds<&k_gather_lds<2>, (int)GL_LDS_BUDGET>::set()), "gather_rows(attr)");
      MMG_LAUNCH(MMG_PROBE_GATHER, n_rows, D, total_cols, accumulate | 16, k_gather_lds<2>, grid, dim3(GL_THREADS), lds, st, rp, n_rows, rows_per_blk, D, out, accumulate);
    } else {
      MMG_CHECK_HIP((MmgMaxLds<&k_gather_lds<1>, (int)GL_LDS_BUDGET>::set()), "gather_rows(attr)");
      MMG_LAUNCH(MMG_PROBE_GATHER, n_rows, D, total_cols, accumulate | 16, k_gather_lds<1>, grid, dim3(GL_THREADS), lds, st, rp, n_rows, rows_per_blk, D, out, accumulate);
    }
    MMG_GATHER_TAIL("gather_rows(lds)");
  }
  const unsigned nb = (unsigned)((n_rows + 3) / 4);
  if (D == 64) hipLaunchKernelGGL(k_gather<1>, dim3(nb), dim3(256), 0, st, rp, n_rows, out, accumulate);
  else if (D == 128) hipLaunchKernelGGL(k_gather<2>, dim3(nb), dim3(256), 0, st, rp, n_rows, out, accumulate);
  else hipLaunchKernelGGL(k_gather<4>, dim3(nb), dim3(256), 0, st, rp, n_rows, out, accumulate);
  MMG_GATHER_TAIL("gather_rows");
#undef MMG_GATHER_TAIL
}

extern "C" int mmg_gather_rows_next_bn(const mmg_rel_t* rels, int n_rel, int64_t n_rows, int D, float* out, int accumulate,
                                       const mmg_next_bn_t* next, void* stream) {
  return gather_rows_stats_impl(rels, n_rel, n_rows, D, out, accumulate, nullptr, nullptr, 0, stream, nullptr, next);
}

extern "C" int mmg_gather_rows_stats(const mmg_rel_t* rels, int n_rel, int64_t n_rows, int D, float* out, int accumulate,
                                     double* col_sums, void* ws, size_t ws_bytes, void* stream) {
  return gather_rows_stats_impl(rels, n_rel, n_rows, D, out, accumulate, col_sums, ws, ws_bytes, stream, nullptr);
}

extern "C" int mmg_gather_rows_stats_bn(const mmg_rel_t* rels, int n_rel, int64_t n_rows, int D, float* out, int accumulate,
                                        double* col_sums, void* ws, size_t ws_bytes, const mmg_bn_fin_t* fin, void* stream) {
  MMG_CHECK_ARG(col_sums && fin && fin->count > 0 && fin->scale && fin->shift, "gather_rows_stats_bn: col_sums and a fold descriptor are required");
  return gather_rows_stats_impl(rels, n_rel, n_rows, D, out, accumulate, col_sums, ws, ws_bytes, stream, fin);
}

extern "C" size_t mmg_scatter_rows_ws_bytes(const mmg_rel_t* rels, int n_rel, int64_t n_rows, int D) {
  if (!rels || n_rel < 1 || n_rel > MMG_MAX_REL || !mmg_valid_D(D) || n_rows < 0) return 0;
  const StripPlan sp = plan_strip(rels, n_rel, n_rows, D);
  if (sp.ok) return (size_t)sp.n_ranges * sp.total_pad * D * 4 + 256;
  const UnitsPlan up = plan_units(rels, n_rel, n_rows, D);
  if (up.ok) return (size_t)up.n_ranges * up.total_pad * D * 4 + 256;
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
      if (rels[r].n_cols > 0)
        MMG_CHECK_HIP(mmg_zero_async(rels[r].out, (size_t)rels[r].n_cols * D * 4, st), "scatter_rows(memset)");
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
  // 1. bit-plane strip kernel: simple relations, <= 10 tiles, no rowscale
  const StripPlan sp = plan_strip(rels, n_rel, n_rows, D);
  if (sp.ok) {
    int rc2 = MMG_OK;
    RelPack rq;                                       // packed accumulator rows: acc_off = items in front of the relation
    int rc3 = pack(rels, n_rel, &rq, false, true, false);
    if (rc3) return rc3;
    if (sp.nt == 8) rc2 = launch_scatter_strip2<4, 4>(rq, n_rows, D, sp.n_ranges, sp.total_pad, x, slab, st);
    else if (sp.nt == 9) rc2 = launch_scatter_strip2<5, 4>(rq, n_rows, D, sp.n_ranges, sp.total_pad, x, slab, st);
    else if (sp.nt == 10) rc2 = launch_scatter_strip2<5, 5>(rq, n_rows, D, sp.n_ranges, sp.total_pad, x, slab, st);
    else rc2 = launch_scatter_strip2<6, 5>(rq, n_rows, D, sp.n_ranges, sp.total_pad, x, slab, st);
    if (rc2) return rc2;
    const int64_t n = (int64_t)sp.total_pad * D;
    MMG_LAUNCH(MMG_PROBE_SCATTER_REDUCE, n_rows, D, sp.total_pad, 0, (mmg_k_reduce_slabs<EpiScatter>),
               dim3((unsigned)((n / 4 + 15) / 16)), dim3(256), 0, st, slab, n / 4, sp.n_ranges, EpiScatter{rq, D});
    MMG_CHECK_LAUNCH("scatter_rows(strip)");
    return MMG_OK;
  }
  // 2. unit-per-wave kernel: any vocabulary layout of simple relations up to 24 tiles, with or without a rowscale
  const UnitsPlan up = plan_units(rels, n_rel, n_rows, D);
  if (up.ok) {
    bool has_rs = false;
    for (int r = 0; r < n_rel; ++r) has_rs |= rels[r].rowscale != nullptr;
    constexpr int lds_max = 4096 + SU_MAXW * 2 * SB_SR * 4 + SU_MAXU * SU_NT * 16 * 64 * 4;
    const int nst = (int)((n_rows + SB_SR - 1) / SB_SR);
    dim3 grid((unsigned)up.n_ranges, (unsigned)(D / 32));
    if (has_rs) {
      MMG_CHECK_HIP((MmgMaxLds<&k_scatter_units<true>, lds_max>::set()), "scatter_rows(attr)");
      MMG_LAUNCH(MMG_PROBE_SCATTER, n_rows, D, up.total_pad, 8 | 32, k_scatter_units<true>, grid, dim3(64 * up.su.n_waves),
                 up.lds, st, up.su, rp, n_rows, nst, D, up.total_pad, x, slab);
    } else {
      MMG_CHECK_HIP((MmgMaxLds<&k_scatter_units<false>, lds_max>::set()), "scatter_rows(attr)");
      MMG_LAUNCH(MMG_PROBE_SCATTER, n_rows, D, up.total_pad, 32, k_scatter_units<false>, grid, dim3(64 * up.su.n_waves),
                 up.lds, st, up.su, rp, n_rows, nst, D, up.total_pad, x, slab);
    }
    const int64_t n = (int64_t)up.total_pad * D;
    MMG_LAUNCH(MMG_PROBE_SCATTER_REDUCE, n_rows, D, up.total_pad, 0, (mmg_k_reduce_slabs<EpiScatter>),
               dim3((unsigned)((n / 4 + 15) / 16)), dim3(256), 0, st, slab, n / 4, up.n_ranges, EpiScatter{rp, D});
    MMG_CHECK_LAUNCH("scatter_rows(units)");
    return MMG_OK;
  }
  // 3. multigraphs (repeated (patient, item) pairs): integer count tile on the fp32 matrix cores
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
  MMG_LAUNCH(MMG_PROBE_SCATTER_REDUCE, n_rows, D, p.total_pad, 0, (mmg_k_reduce_slabs<EpiScatter>),
             dim3((unsigned)((n / 4 + 15) / 16)), dim3(256), 0, st, slab, n / 4, p.n_split, EpiScatter{rp, D});
  MMG_CHECK_LAUNCH("scatter_rows");
  return MMG_OK;
}


extern "C" size_t mmg_rel_mask_words(int64_t n_rows, int32_t n_cols) {
  if (n_rows <= 0 || n_cols <= 0) return 0;
  return (size_t)((n_rows + 63) / 64) * (size_t)((n_cols + 31) & ~31) * 2;
}

extern "C" int mmg_rel_mask_build(const int32_t* rowptr, const int32_t* col, int64_t n_rows, int32_t n_cols,
                                  uint64_t* mask_t, uint16_t* mask_r, void* stream) {
  MMG_CHECK_ARG(n_rows >= 0 && n_rows < 2147483647LL / 64 && n_cols >= 0, "rel_mask_build: size out of range");
  const size_t words = mmg_rel_mask_words(n_rows, n_cols);
  if (words == 0) return MMG_OK;
  MMG_CHECK_ARG(rowptr && (mask_t || mask_r), "rel_mask_build: null buffer");
  hipStream_t st = (hipStream_t)stream;
  const int padc = (n_cols + 31) & ~31;
  if (mask_t) {
    MMG_CHECK_HIP(mmg_zero_async(mask_t, words * sizeof(uint64_t), st), "rel_mask_build(memset)");
    hipLaunchKernelGGL(k_mask_build, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, st, rowptr, col, n_rows, padc,
                       reinterpret_cast<unsigned long long*>(mask_t));
  }
  if (mask_r) {
    MMG_CHECK_HIP(mmg_zero_async(mask_r, words * sizeof(uint64_t), st), "rel_mask_build(memset)");
    hipLaunchKernelGGL(k_mask_build_rows, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, st, rowptr, col, n_rows,
                       padc / 16, reinterpret_cast<unsigned*>(mask_r));
  }
  MMG_CHECK_LAUNCH("rel_mask_build");
  return MMG_OK;
}
