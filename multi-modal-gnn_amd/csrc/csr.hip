// CSR construction for the edge_index contract of src/graph_build.py (reference) --
// SURVEY.md section 8 row a2.  Integer-exact with torch.sort(stable=True)+bincount+cumsum.
//
// Stable LSD radix sort (8-bit digits) of (key = edge_index[sort_row][e], value = e):
// no atomics on the data path, so the result is deterministic; work is O(E * ceil(bits/8)).
//   per pass:  tile histogram -> exclusive scan (digit-major) -> stable scatter
//   in-tile stable rank: wave-level match-any by ballots + per-group digit counts in LDS.
#include "common.h"

namespace {

constexpr int TILE = 1024;      // items per workgroup tile (256 threads x 4)
constexpr int NTHR = 256;
constexpr int GROUPS = TILE / WAVE;   // 16 groups of 64 consecutive items

// ---------------------------------------------------------------- exclusive scan (uint32)
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_CHUNK = NTHR * SCAN_ITEMS;

__global__ __launch_bounds__(NTHR) void k_scan_local(uint32_t* data, uint32_t* block_sums, int64_t n) {
  __shared__ uint32_t s_wave[NTHR / WAVE];
  const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS];
  uint32_t tsum = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    v[i] = (base + i < n) ? data[base + i] : 0u;
    tsum += v[i];
  }
  // inclusive scan of tsum across the wave
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint32_t inc = tsum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) s_wave[wid] = inc;
  __syncthreads();
  uint32_t woff = 0;
  for (int w = 0; w < wid; ++w) woff += s_wave[w];
  uint32_t run = woff + inc - tsum;   // exclusive prefix of this thread
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    if (base + i < n) data[base + i] = run;
    run += v[i];
  }
  if (threadIdx.x == NTHR - 1 && block_sums) block_sums[blockIdx.x] = run;
}

__global__ __launch_bounds__(NTHR) void k_scan_add(uint32_t* data, const uint32_t* block_prefix, int64_t n) {
  const uint32_t add = block_prefix[blockIdx.x];
  const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_ITEMS;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i)
    if (base + i < n) data[base + i] += add;
}

// scratch needed for the block-sum levels of a scan over n items (in uint32 elements)
size_t scan_scratch_elems(int64_t n) {
  size_t tot = 0;
  while (n > SCAN_CHUNK) {
    n = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    tot += (size_t)n;
  }
  return tot + 1;
}

void exclusive_scan_u32(uint32_t* data, int64_t n, uint32_t* scratch, hipStream_t st) {
  if (n <= 0) return;
  const int64_t nb = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
  if (nb == 1) {
    hipLaunchKernelGGL(k_scan_local, dim3(1), dim3(NTHR), 0, st, data, (uint32_t*)nullptr, n);
    return;
  }
  hipLaunchKernelGGL(k_scan_local, dim3((unsigned)nb), dim3(NTHR), 0, st, data, scratch, n);
  exclusive_scan_u32(scratch, nb, scratch + nb, st);
  hipLaunchKernelGGL(k_scan_add, dim3((unsigned)nb), dim3(NTHR), 0, st, data, scratch, n);
}

// ---------------------------------------------------------------- radix sort passes
// A key outside [0, n_rows) is never used as an address: it is filed under the sentinel bucket n_rows, i.e. it sorts
// behind the last row and rowptr[n_rows] < n_edges tells the caller (include/mmgnn.h).
__global__ __launch_bounds__(NTHR) void k_prep(const int64_t* __restrict__ key_src, uint32_t* keys,
                                               int32_t* vals, uint32_t* counts, int64_t n, int64_t n_rows) {
  const int64_t e = (int64_t)blockIdx.x * NTHR + threadIdx.x;
  if (e < n) {
    const int64_t k64 = key_src[e];
    const uint32_t k = (k64 < 0 || k64 >= n_rows) ? (uint32_t)n_rows : (uint32_t)k64;
    keys[e] = k;
    vals[e] = (int32_t)e;
    atomicAdd(&counts[k], 1u);   // integer histogram -> rowptr (order-independent result)
  }
}

__global__ __launch_bounds__(NTHR) void k_hist(const uint32_t* __restrict__ keys, uint32_t* tile_hist,
                                               int64_t n, int shift, int64_t n_tiles) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * TILE;
#pragma unroll
  for (int i = 0; i < TILE / NTHR; ++i) {
    const int64_t e = base + i * NTHR + threadIdx.x;
    if (e < n) atomicAdd(&h[(keys[e] >> shift) & 255u], 1u);
  }
  __syncthreads();
  tile_hist[(int64_t)threadIdx.x * n_tiles + blockIdx.x] = h[threadIdx.x];
}

__global__ __launch_bounds__(NTHR) void k_scatter(const uint32_t* __restrict__ keys_in,
                                                  const int32_t* __restrict__ vals_in,
                                                  uint32_t* __restrict__ keys_out, int32_t* __restrict__ vals_out,
                                                  const uint32_t* __restrict__ tile_off, int64_t n, int shift,
                                                  int64_t n_tiles) {
  __shared__ uint32_t gcnt[GROUPS][256];   // per 64-item group: count of each digit -> exclusive offset
  for (int i = threadIdx.x; i < GROUPS * 256; i += NTHR) (&gcnt[0][0])[i] = 0;
  __syncthreads();

  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t base = (int64_t)blockIdx.x * TILE;
  uint32_t key[TILE / NTHR];
  int32_t val[TILE / NTHR];
  uint32_t rank[TILE / NTHR];
  // wave w owns groups 4w .. 4w+3 (consecutive 64-item runs) => item order is preserved
#pragma unroll
  for (int i = 0; i < TILE / NTHR; ++i) {
    const int g = wid * (TILE / NTHR) + i;
    const int64_t e = base + (int64_t)g * 64 + lane;
    const bool valid = e < n;
    key[i] = valid ? keys_in[e] : 0u;
    val[i] = valid ? vals_in[e] : 0;
    const uint32_t d = (key[i] >> shift) & 255u;
    unsigned long long m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1u;
      const unsigned long long bal = __ballot(bit);
      m &= bit ? bal : ~bal;
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    rank[i] = (uint32_t)__popcll(m & lt);
    if (valid && rank[i] == 0) gcnt[g][d] = (uint32_t)__popcll(m);
  }
  __syncthreads();
  {  // thread d: exclusive scan of digit d over the 16 groups, plus the tile's global offset
    const int d = threadIdx.x;
    uint32_t run = tile_off[(int64_t)d * n_tiles + blockIdx.x];
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) {
      const uint32_t c = gcnt[g][d];
      gcnt[g][d] = run;
      run += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < TILE / NTHR; ++i) {
    const int g = wid * (TILE / NTHR) + i;
    const int64_t e = base + (int64_t)g * 64 + lane;
    if (e < n) {
      const uint32_t d = (key[i] >> shift) & 255u;
      const uint32_t pos = gcnt[g][d] + rank[i];
      keys_out[pos] = key[i];
      vals_out[pos] = val[i];
    }
  }
}

__global__ __launch_bounds__(NTHR) void k_finish(const int64_t* __restrict__ other, const int32_t* __restrict__ perm,
                                                 int32_t* __restrict__ col, int64_t n) {
  const int64_t k = (int64_t)blockIdx.x * NTHR + threadIdx.x;
  if (k < n) col[k] = (int32_t)other[perm[k]];
}

__global__ __launch_bounds__(NTHR) void k_row_degree(const int32_t* __restrict__ rowptr, int64_t n, int32_t* deg,
                                                     float* inv) {
  const int64_t i = (int64_t)blockIdx.x * NTHR + threadIdx.x;
  if (i < n) {
    const int32_t d = rowptr[i + 1] - rowptr[i];
    if (deg) deg[i] = d;
    if (inv) inv[i] = 1.0f / (float)(d > 1 ? d : 1);
  }
}
// column histogram: per-workgroup counts in LDS (the vocabularies have 50..200 entries: global atomics onto so few
// counters serialise -- 2.9 ms for 6 M edges), one global add per non-empty bin and workgroup at the end
constexpr int CC_BINS = 4096;
__global__ __launch_bounds__(NTHR) void k_col_count(const int32_t* __restrict__ col, int64_t n, int32_t* cnt,
                                                    int n_cols) {
  __shared__ int h[CC_BINS];
  const bool local = n_cols <= CC_BINS;
  if (local) {
    for (int i = threadIdx.x; i < n_cols; i += NTHR) h[i] = 0;
    __syncthreads();
  }
  for (int64_t k = (int64_t)blockIdx.x * NTHR + threadIdx.x; k < n; k += (int64_t)gridDim.x * NTHR) {
    const int c = col[k];
    if ((unsigned)c >= (unsigned)n_cols) continue;           // never index outside the table
    if (local) atomicAdd(&h[c], 1);
    else atomicAdd(&cnt[c], 1);
  }
  if (local) {
    __syncthreads();
    for (int i = threadIdx.x; i < n_cols; i += NTHR)
      if (h[i]) atomicAdd(&cnt[i], h[i]);
  }
}
__global__ __launch_bounds__(NTHR) void k_inv_count(const int32_t* __restrict__ cnt, int64_t n, float* inv) {
  const int64_t i = (int64_t)blockIdx.x * NTHR + threadIdx.x;
  if (i < n) inv[i] = 1.0f / (float)(cnt[i] > 1 ? cnt[i] : 1);
}

inline int key_passes(int64_t n_rows) {
  int bits = 1;
  while (((int64_t)1 << bits) < n_rows) ++bits;
  return (bits + 7) / 8;
}
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" size_t mmg_csr_build_ws_bytes(int64_t n_edges, int64_t n_rows) {
  if (n_edges < 0 || n_rows < 0) return 0;
  const int64_t n_tiles = (n_edges + TILE - 1) / TILE;
  size_t b = 0;
  b += 2 * align256((size_t)n_edges * 4);           // keysA, keysB
  b += align256((size_t)n_edges * 4);               // valsB
  b += align256((size_t)(256 * (n_tiles > 0 ? n_tiles : 1)) * 4);   // tile histograms
  b += align256(scan_scratch_elems(256 * (n_tiles > 0 ? n_tiles : 1)) * 4);
  b += align256(scan_scratch_elems(n_rows + 1) * 4);
  return b + 256;
}

extern "C" int mmg_csr_build(const int64_t* edge_index, int64_t n_edges, int64_t n_rows, int sort_row,
                             int32_t* rowptr, int32_t* col, int32_t* perm, void* ws, size_t ws_bytes,
                             void* stream) {
  MMG_CHECK_ARG(n_edges >= 0 && n_rows >= 0, "csr_build: negative size");
  MMG_CHECK_ARG(n_edges < 2147483647LL && n_rows < 2147483647LL, "csr_build: int32 index range exceeded");
  MMG_CHECK_ARG(sort_row == 0 || sort_row == 1, "csr_build: sort_row must be 0 or 1");
  MMG_CHECK_ARG(rowptr, "csr_build: rowptr is null");
  MMG_CHECK_ARG(n_edges == 0 || (edge_index && col && perm && ws), "csr_build: null buffer");
  if (ws_bytes < mmg_csr_build_ws_bytes(n_edges, n_rows)) {
    mmg_set_error("csr_build: workspace %zu < %zu", ws_bytes, mmg_csr_build_ws_bytes(n_edges, n_rows));
    return MMG_E_WS;
  }
  hipStream_t st = (hipStream_t)stream;
  uint32_t* cnt = (uint32_t*)rowptr;   // histogram is built in place, then scanned into rowptr
  MMG_CHECK_HIP(mmg_zero_async(cnt, (size_t)(n_rows + 1) * 4, st), "csr_build(memset)");
  if (n_edges == 0) {
    MMG_CHECK_LAUNCH("csr_build(memset)");
    return MMG_OK;
  }
  const int64_t n_tiles = (n_edges + TILE - 1) / TILE;
  char* p = (char*)ws;
  p = (char*)(((uintptr_t)p + 255) & ~(uintptr_t)255);
  uint32_t* keysA = (uint32_t*)p; p += align256((size_t)n_edges * 4);
  uint32_t* keysB = (uint32_t*)p; p += align256((size_t)n_edges * 4);
  int32_t* valsB = (int32_t*)p;   p += align256((size_t)n_edges * 4);
  uint32_t* thist = (uint32_t*)p; p += align256((size_t)(256 * n_tiles) * 4);
  uint32_t* scr1 = (uint32_t*)p;  p += align256(scan_scratch_elems(256 * n_tiles) * 4);
  uint32_t* scr2 = (uint32_t*)p;

  const int passes = key_passes(n_rows + 1);        // + the sentinel bucket of out-of-range keys
  const unsigned eb = (unsigned)((n_edges + NTHR - 1) / NTHR);
  const int64_t* key_src = edge_index + (int64_t)sort_row * n_edges;
  const int64_t* oth_src = edge_index + (int64_t)(1 - sort_row) * n_edges;

  // the last pass must land in `perm`: odd pass count starts from valsB, even from perm
  int32_t* vals_cur = (passes & 1) ? valsB : perm;
  int32_t* vals_nxt = (passes & 1) ? perm : valsB;
  uint32_t* keys_cur = keysA;
  uint32_t* keys_nxt = keysB;

  hipLaunchKernelGGL(k_prep, dim3(eb), dim3(NTHR), 0, st, key_src, keys_cur, vals_cur, cnt, n_edges, n_rows);
  exclusive_scan_u32(cnt, n_rows + 1, scr2, st);   // rowptr = exclusive scan of the row histogram

  for (int ps = 0; ps < passes; ++ps) {
    const int shift = 8 * ps;
    hipLaunchKernelGGL(k_hist, dim3((unsigned)n_tiles), dim3(NTHR), 0, st, keys_cur, thist, n_edges, shift, n_tiles);
    exclusive_scan_u32(thist, 256 * n_tiles, scr1, st);
    hipLaunchKernelGGL(k_scatter, dim3((unsigned)n_tiles), dim3(NTHR), 0, st, keys_cur, vals_cur, keys_nxt,
                       vals_nxt, thist, n_edges, shift, n_tiles);
    uint32_t* tk = keys_cur; keys_cur = keys_nxt; keys_nxt = tk;
    int32_t* tv = vals_cur; vals_cur = vals_nxt; vals_nxt = tv;
  }
  // vals_cur == perm here
  hipLaunchKernelGGL(k_finish, dim3(eb), dim3(NTHR), 0, st, oth_src, perm, col, n_edges);
  MMG_CHECK_LAUNCH("csr_build");
  return MMG_OK;
}

extern "C" int mmg_row_degree(const int32_t* rowptr, int64_t n_rows, int32_t* deg, float* inv_deg, void* stream) {
  MMG_CHECK_ARG(n_rows >= 0 && rowptr, "row_degree: bad args");
  if (n_rows == 0) return MMG_OK;
  hipLaunchKernelGGL(k_row_degree, dim3((unsigned)((n_rows + NTHR - 1) / NTHR)), dim3(NTHR), 0,
                     (hipStream_t)stream, rowptr, n_rows, deg, inv_deg);
  MMG_CHECK_LAUNCH("row_degree");
  return MMG_OK;
}

extern "C" int mmg_col_degree(const int32_t* col, int64_t n_edges, int64_t n_cols, int32_t* cnt, float* inv_cnt,
                              void* stream) {
  MMG_CHECK_ARG(n_edges >= 0 && n_cols >= 0 && n_cols < 2147483647LL && cnt, "col_degree: bad args");
  hipStream_t st = (hipStream_t)stream;
  if (n_cols > 0) MMG_CHECK_HIP(mmg_zero_async(cnt, (size_t)n_cols * 4, st), "col_degree(memset)");
  if (n_edges > 0 && n_cols > 0) {
    int64_t nb = (n_edges + NTHR * 8 - 1) / (NTHR * 8);      // >= 8 edges per thread: the flush is amortised
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(k_col_count, dim3((unsigned)nb), dim3(NTHR), 0, st, col, n_edges, cnt, (int)n_cols);
  }
  if (n_cols > 0 && inv_cnt)
    hipLaunchKernelGGL(k_inv_count, dim3((unsigned)((n_cols + NTHR - 1) / NTHR)), dim3(NTHR), 0, st, cnt, n_cols, inv_cnt);
  MMG_CHECK_LAUNCH("col_degree");
  return MMG_OK;
}
