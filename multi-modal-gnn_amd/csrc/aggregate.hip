// Sparse aggregates over CSR-by-patient (replace PyG SAGEConv's gather + scatter-mean and their
// backward; call site src/model.py:125-131,256 of the reference).  HBM-bound byte work: no MFMA.
//
//  gather : one wave per patient row, lane owns D/64 contiguous floats (a 64-lane row read is one
//           coalesced 256..1024-B segment); the source tables are the tiny vocab tables (L2-resident).
//  scatter: patient-major streaming; every workgroup owns a contiguous row chunk, accumulates into
//           LDS-resident vocab accumulators [n_cols, DC] with ds_add_f32, then writes ONE partial
//           slab; a second kernel sums the slabs in fixed order (deterministic across launches for a
//           fixed grid except for the intra-workgroup LDS add order).
#include "common.h"

namespace {

struct RelDev {
  const int32_t* rowptr; const int32_t* col; const float* rowscale; const float* colscale;
  const float* table; float* out; int32_t n_cols; int32_t acc_off;   // acc_off: first accumulator row
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

// ------------------------------------------------------------------------------ scatter
constexpr int SC_THREADS = 512;
constexpr size_t SC_LDS_BUDGET = 144 * 1024;

struct ScatterPlan {
  int dc;            // accumulator column-chunk width (64 | 128 | 256)
  int n_dchunks;     // D / dc
  int total_cols;    // sum of n_cols
  int n_rowchunks;   // workgroups along the row axis
  int64_t rows_per_chunk;
  size_t lds_bytes;
  bool lds_ok;
};

ScatterPlan plan_scatter(const mmg_rel_t* rels, int n_rel, int64_t n_rows, int D) {
  ScatterPlan p{};
  p.total_cols = 0;
  for (int r = 0; r < n_rel; ++r) p.total_cols += rels[r].n_cols;
  p.dc = D;
  while ((size_t)p.total_cols * p.dc * 4 > SC_LDS_BUDGET && p.dc > 64) p.dc >>= 1;
  p.lds_ok = (size_t)p.total_cols * p.dc * 4 <= SC_LDS_BUDGET;
  p.n_dchunks = D / p.dc;
  p.lds_bytes = (size_t)p.total_cols * p.dc * 4;
  // one workgroup per CU and column chunk; at least 32 rows per workgroup
  int64_t want = 256 / p.n_dchunks;
  if (want < 1) want = 1;
  int64_t maxc = (n_rows + 31) / 32;
  if (maxc < 1) maxc = 1;
  p.n_rowchunks = (int)(want < maxc ? want : maxc);
  p.rows_per_chunk = (n_rows + p.n_rowchunks - 1) / p.n_rowchunks;
  return p;
}

// slab layout: [n_rowchunks][total_cols][D]
template <int VECC>   // floats per lane inside the column chunk: dc = 64*VECC
__global__ __launch_bounds__(SC_THREADS) void k_scatter_lds(RelPack rp, int64_t n_rows, int64_t rows_per_chunk, int D,
                                                            int total_cols, const float* __restrict__ x,
                                                            float* __restrict__ slab) {
  extern __shared__ __attribute__((aligned(16))) float acc[];
  constexpr int DC = VECC * 64;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = SC_THREADS / 64;
  const int d0 = blockIdx.y * DC;
  const int n_acc = total_cols * DC;
  for (int i = threadIdx.x; i < n_acc; i += SC_THREADS) acc[i] = 0.f;
  __syncthreads();

  const int64_t r0 = (int64_t)blockIdx.x * rows_per_chunk;
  const int64_t r1 = min(n_rows, r0 + rows_per_chunk);
  for (int64_t row = r0 + wid; row < r1; row += nw) {
    float xv[VECC];
    const float* src = x + (size_t)row * D + d0 + lane * VECC;
#pragma unroll
    for (int v = 0; v < VECC; ++v) xv[v] = src[v];
    for (int r = 0; r < rp.n; ++r) {
      const RelDev& R = rp.r[r];
      const int beg = R.rowptr[row], end = R.rowptr[row + 1];
      if (beg == end) continue;
      const float rs = R.rowscale ? R.rowscale[row] : 1.f;
      float sv[VECC];
#pragma unroll
      for (int v = 0; v < VECC; ++v) sv[v] = xv[v] * rs;
      for (int base = beg; base < end; base += 64) {
        const int cnt = min(64, end - base);
        const int cidx = (lane < cnt) ? R.col[base + lane] : 0;
        for (int j = 0; j < cnt; ++j) {
          const int c = __builtin_amdgcn_readlane(cidx, j);
          float* a = acc + (size_t)(R.acc_off + c) * DC + lane * VECC;
#pragma unroll
          for (int v = 0; v < VECC; ++v) atomicAdd(a + v, sv[v]);   // ds_add_f32
        }
      }
    }
  }
  __syncthreads();
  // flush: acc[c][0..DC) -> slab[blockIdx.x][c][d0..d0+DC)
  float* dst = slab + (size_t)blockIdx.x * total_cols * D;
  for (int i = threadIdx.x; i < n_acc; i += SC_THREADS) {
    const int c = i / DC, d = i - c * DC;
    dst[(size_t)c * D + d0 + d] = acc[i];
  }
}

__global__ __launch_bounds__(256) void k_scatter_reduce(RelPack rp, int D, int total_cols, int n_chunks,
                                                        const float* __restrict__ slab) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n = (int64_t)total_cols * D;
  if (i >= n) return;
  const int c = (int)(i / D), d = (int)(i - (int64_t)c * D);
  float s = 0.f;
  for (int b = 0; b < n_chunks; ++b) s += slab[(size_t)b * n + i];
  for (int r = 0; r < rp.n; ++r) {
    const RelDev& R = rp.r[r];
    if (c >= R.acc_off && c < R.acc_off + R.n_cols) {
      const int j = c - R.acc_off;
      const float cs = R.colscale ? R.colscale[j] : 1.f;
      R.out[(size_t)j * D + d] = s * cs;
    }
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

int pack(const mmg_rel_t* rels, int n_rel, RelPack* rp, bool need_table, bool need_out) {
  MMG_CHECK_ARG(rels && n_rel >= 1 && n_rel <= MMG_MAX_REL, "aggregate: n_rel must be 1..%d", MMG_MAX_REL);
  rp->n = n_rel;
  int off = 0;
  for (int r = 0; r < n_rel; ++r) {
    MMG_CHECK_ARG(rels[r].rowptr && rels[r].n_cols >= 0, "aggregate: relation %d has null rowptr", r);
    MMG_CHECK_ARG(!need_table || rels[r].table, "aggregate: relation %d has null table", r);
    MMG_CHECK_ARG(!need_out || rels[r].out, "aggregate: relation %d has null out", r);
    rp->r[r] = RelDev{rels[r].rowptr, rels[r].col, rels[r].rowscale, rels[r].colscale, rels[r].table,
                      rels[r].out, rels[r].n_cols, off};
    off += rels[r].n_cols;
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
  const unsigned nb = (unsigned)((n_rows + 3) / 4);
  hipStream_t st = (hipStream_t)stream;
  if (D == 64) hipLaunchKernelGGL(k_gather<1>, dim3(nb), dim3(256), 0, st, rp, n_rows, out, accumulate);
  else if (D == 128) hipLaunchKernelGGL(k_gather<2>, dim3(nb), dim3(256), 0, st, rp, n_rows, out, accumulate);
  else hipLaunchKernelGGL(k_gather<4>, dim3(nb), dim3(256), 0, st, rp, n_rows, out, accumulate);
  MMG_CHECK_LAUNCH("gather_rows");
  return MMG_OK;
}

extern "C" size_t mmg_scatter_rows_ws_bytes(const mmg_rel_t* rels, int n_rel, int64_t n_rows, int D) {
  if (!rels || n_rel < 1 || n_rel > MMG_MAX_REL || !mmg_valid_D(D) || n_rows < 0) return 0;
  ScatterPlan p = plan_scatter(rels, n_rel, n_rows, D);
  if (!p.lds_ok) return 256;
  return (size_t)p.n_rowchunks * p.total_cols * D * 4 + 256;
}

extern "C" int mmg_scatter_rows(const mmg_rel_t* rels, int n_rel, int64_t n_rows, int D, const float* x, void* ws,
                                size_t ws_bytes, void* stream) {
  MMG_CHECK_ARG(mmg_valid_D(D), "scatter_rows: D=%d unsupported (64|128|256)", D);
  MMG_CHECK_ARG(n_rows >= 0 && n_rows < 2147483647LL / 64, "scatter_rows: n_rows out of range");
  RelPack rp;
  int rc = pack(rels, n_rel, &rp, false, true);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  ScatterPlan p = plan_scatter(rels, n_rel, n_rows, D);
  if (p.total_cols == 0) return MMG_OK;
  if (n_rows == 0 || !p.lds_ok) {
    for (int r = 0; r < n_rel; ++r)
      if (rels[r].n_cols > 0) hipMemsetAsync(rels[r].out, 0, (size_t)rels[r].n_cols * D * 4, st);
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
  dim3 grid((unsigned)p.n_rowchunks, (unsigned)p.n_dchunks);
  if (p.dc == 64) {
    hipFuncSetAttribute((const void*)k_scatter_lds<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
    hipLaunchKernelGGL(k_scatter_lds<1>, grid, dim3(SC_THREADS), p.lds_bytes, st, rp, n_rows, p.rows_per_chunk, D,
                       p.total_cols, x, slab);
  } else if (p.dc == 128) {
    hipFuncSetAttribute((const void*)k_scatter_lds<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
    hipLaunchKernelGGL(k_scatter_lds<2>, grid, dim3(SC_THREADS), p.lds_bytes, st, rp, n_rows, p.rows_per_chunk, D,
                       p.total_cols, x, slab);
  } else {
    hipFuncSetAttribute((const void*)k_scatter_lds<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
    hipLaunchKernelGGL(k_scatter_lds<4>, grid, dim3(SC_THREADS), p.lds_bytes, st, rp, n_rows, p.rows_per_chunk, D,
                       p.total_cols, x, slab);
  }
  const int64_t n = (int64_t)p.total_cols * D;
  hipLaunchKernelGGL(k_scatter_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, rp, D, p.total_cols,
                     p.n_rowchunks, slab);
  MMG_CHECK_LAUNCH("scatter_rows");
  return MMG_OK;
}
