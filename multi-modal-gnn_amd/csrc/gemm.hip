// Dense layers.  Replaces nn.Linear (+ the BatchNorm/ReLU/Dropout in front of it) of src/model.py:93-105 and the
// lin_l / lin_r of PyG SAGEConv (call site src/model.py:125-131).
//
// fp32 products on the bf16 matrix cores, exactly: an fp32 value is the sum of three bf16 pieces (8 significant bits
// each) and x.w = x1w1 + (x1w2 + x2w1) + (x2w2 + x1w3 + x3w1) + O(2^-24 |x||w|) -- six v_mfma_f32_32x32x16_bf16 with
// fp32 accumulation per product tile (k_linear_fwd_x6, k_linear_wgrad_x6, k_linear_wgrad_ws).
// 6/16 of the fp32 matrix time turns the [P,128] x [128,128] layers from matrix-bound into HBM-bound.
//
// linear_fwd : Y[M,N] = prologue(X)[M,K] . W[N,K]^T + b.  The W pieces stay resident in registers; X tiles are
//              prologue'd and split ONCE while staged to three bf16 LDS planes (double-buffered, one barrier per
//              32-row tile).  k_linear_small serves the vocab-side tables (M <= 512).
// linear_wgrad: dW[N,K] = dY^T . prologue(X): contraction over the rows; every workgroup reduces a set of 32-row
//              stages into a [TN,TK] register tile, writes one partial slab, a second kernel sums the slabs in
//              fixed order (bitwise reproducible).  The bias gradient (column sums of dY) rides in the same pass.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------- fp32 GEMM on the bf16 matrix cores
// An fp32 value is EXACTLY hi + mid + lo with three bf16 pieces (8 significant bits each), so
//     x . w = x1 w1 + (x1 w2 + x2 w1) + (x2 w2 + x1 w3 + x3 w1) + O(2^-24 |x||w|)
// Six v_mfma_f32_32x32x16_bf16 products (each exact, accumulated in fp32) reproduce the fp32 product to within
// its own rounding error at 6/16 of the fp32 matrix time: the dropped terms x2 w3 + x3 w2 + x3 w3 are below
// 3 * 2^-25 |x||w| -- less than the rounding of the fp32 FMA chain they replace.
// Layout: the three W pieces stay in registers for the whole kernel (lane (n = l&31, h = l>>5) holds
// W[n][16 ks + 8 h + 0..7]); every 64-row X tile is prologue'd AND split once while it is staged to LDS (three
// bf16 planes, double-buffered: tile t+1 is staged while tile t is multiplied, one barrier per tile), and an A
// fragment is one conflict-free ds_read_b128 per piece.
typedef __bf16 xbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 xbf16x4 __attribute__((ext_vector_type(4)));

// XCD-aware block -> (tile, row-set) map for 2-D grids whose x index selects an output tile (a column slice of the
// forward, an [TN, TK] tile of the weight gradient) and whose y index selects the rows.  Every tile of one row-set reads
// the same rows of the streamed operand(s); workgroups are dealt round-robin over the 8 XCDs (linear id b and b + 8 share
// an XCD and its 4 MB L2, MI355X_MICROARCH.md), so with the natural map (x fastest) the tiles of a row-set land on
// DIFFERENT XCDs and every one of them fetches its rows from HBM.  Here ids 8 apart -- same XCD, dispatched together --
// become the tiles of one row-set: the first reader's lines are in L2 for the others.  Speed only: a bijection of the
// grid for gridDim.y % 8 == 0, the identity otherwise.
__device__ __forceinline__ void xcd_tile_map(int* bx, int* by) {
  const unsigned nx = gridDim.x, ny = gridDim.y;
  *bx = (int)blockIdx.x; *by = (int)blockIdx.y;
  if (nx > 1u && (ny & 7u) == 0u) {
    const unsigned L = blockIdx.x + nx * blockIdx.y, j = L >> 3;
    *bx = (int)(j % nx);
    *by = (int)((j / nx) * 8u + (L & 7u));
  }
}


// Memory pipeline: every global access goes through a per-tile buffer descriptor (rows past M and tiles past the end
// read 0 / are dropped by the range check), so the tile loop is straight-line code without a branch around a load or a
// store and the compiler's vmcnt waits are exact: X runs two tiles ahead in registers, the epilogue's stores are never
// waited for (gfx9 counts stores in vmcnt, in order: a conservative wait exposes the full store-acknowledge latency
// once per tile, which is what the first version of this kernel did), and in accumulate mode the old output tile is
// loaded INTO the accumulator before the next tile is staged, so its latency hides under the split.
typedef unsigned xu32x4 __attribute__((ext_vector_type(4)));
// cache policy of the streamed operands (aux = 2: the `nt` bit -- X is read once, Y written once: -1.3 % on the step)
#ifndef MMG_NT_LD
#define MMG_NT_LD 2
#endif
#ifndef MMG_NT_ST
#define MMG_NT_ST 2
#endif
#ifndef MMG_NT_WG
#define MMG_NT_WG 0
#endif

// (NextBnDev and its epilogue arithmetic: common.h)
// the Y values of this lane's 16 tile elements; vo = byte offset of (row 4h, this lane's column) in the tile
__device__ __forceinline__ void next_bn_load(const NextBnDev& nb, int64_t tile, int rows, int N, int c0, int vo, float* yv) {
  const size_t off = (size_t)(rows ? tile : 0) * 32 * N + c0;
  const int bytes = rows ? (rows * N - c0) * 4 : 0;
  const __amdgpu_buffer_rsrc_t ysrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(nb.Y) + off, 0, bytes, 0x00020000);
#pragma unroll
  for (int i = 0; i < 16; ++i)
    yv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ysrc, vo, ((i & 3) + 8 * (i >> 2)) * N * 4, MMG_NT_LD));
}
// L2: the row L2 normalisation that follows the layer (F.normalize, src/model.py:232) in the epilogue -- the workgroup
// holds whole rows (N == 32 * WN), so Y receives  y / max(|y|, eps)  and rn_out[row] = 1 / max(|y|, eps); the separate
// pass read and wrote the [M, N] tensor once more.
// NBN: stat_partial receives the statistics of the NEXT BatchNorm backward (nb, see NextBnDev) instead of the forward ones.
template <int K, int WN, bool PRO, bool ACC, bool L2 = false, bool NBN = false>   // WN waves along N (32 columns each) over a 32-row tile: 64 * WN threads
__global__ __launch_bounds__(64 * WN, (WN == 4 && K <= 128) ? 2 : 1) void k_linear_fwd_x6(
    const float* __restrict__ X, ProDev pr, const float* __restrict__ W, const float* __restrict__ bias,
    float* __restrict__ Y, int64_t M, int N, int flags, double* __restrict__ stat_partial, float* __restrict__ rn_out,
    float l2_eps, NextBnDev nb) {
  static_assert(!(NBN && L2), "one epilogue at a time");
  __shared__ float l2_part[L2 ? WN * 32 : 1], l2_rn[L2 ? 32 : 1];
  if (PRO) pr.resolve();
  double cs1 = 0.0, cs2 = 0.0;          // column statistics of the output (BatchNorm batch stats) ride along: lane = column
  constexpr int LDP = K + 8;            // plane row stride in bf16 (K*2 + 16 bytes: fragment reads hit 64 distinct banks)
  constexpr int BN = 32 * WN, NK = K / 16, NTHR = 64 * WN, BM = 32;
  extern __shared__ __attribute__((aligned(16))) __bf16 planes[];     // [2 buffers][3 pieces][BM][LDP]
  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  int bx, by;
  xcd_tile_map(&bx, &by);                 // (column slice, row-set): the slices of one row-set share an XCD's L2
  const int col = bx * BN + wn * 32 + l31;

  xbf16x8 wb[NK][3];
  {
    const bool wkn = (flags & MMG_LIN_W_KN) != 0;
    const float* wp = wkn ? W + (size_t)(8 * h) * N + col : W + (size_t)col * K + 8 * h;
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
      f32x4 w0, w1;
      if (wkn) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { w0[j] = wp[(size_t)(ks * 16 + j) * N]; w1[j] = wp[(size_t)(ks * 16 + 4 + j) * N]; }
      } else {
        w0 = *reinterpret_cast<const f32x4*>(wp + ks * 16); w1 = *reinterpret_cast<const f32x4*>(wp + ks * 16 + 4);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = j < 4 ? w0[j] : w1[j - 4];
        const __bf16 a = (__bf16)v;
        const float r1 = v - (float)a;
        const __bf16 b = (__bf16)r1;
        wb[ks][0][j] = a; wb[ks][1][j] = b; wb[ks][2][j] = (__bf16)(r1 - (float)b);
      }
    }
  }
  const float bv = bias ? bias[col] : 0.f;
  NextBnCol nbc = {};
  if constexpr (NBN) nbc = next_bn_col(nb, col);

  constexpr int K4 = K / 4;
  const int kc4 = tid % K4;                 // float4 column: this thread always touches the same 4 k's
  constexpr int ROWS_PER_PASS = NTHR / K4;
  constexpr int NP = BM / ROWS_PER_PASS;    // 16-B loads per thread per tile
  const int prow = tid / K4;
  // prologue constants, loaded once (identity when a part is absent: x*1+0, max(x,-inf))
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  float floor_v = -__builtin_inff();
  bool drop = false;
  if (PRO) {
    if (pr.scale) {
      sc = *reinterpret_cast<const f32x4*>(pr.scale + kc4 * 4);
      sh = *reinterpret_cast<const f32x4*>(pr.shift + kc4 * 4);
    }
    if (pr.relu) floor_v = 0.f;
    drop = pr.p > 0.f;
  }
  const int64_t n_tiles = (M + BM - 1) / BM;
  const int64_t G = gridDim.y, t0 = by;
  if (t0 >= n_tiles) return;
  const int n_my = (int)((n_tiles - t0 + G - 1) / G);      // tiles t0, t0+G, ... of this workgroup
  auto rows_of = [&](int64_t tile) -> int {                // valid rows of a tile (0 past the end)
    const int64_t r = M - tile * BM;
    return r <= 0 ? 0 : (r < BM ? (int)r : BM);
  };
  f32x4 nxa[NP], nxb[NP];                   // two tiles of X in flight: tiles alternate between them
  const int xvo = (prow * K + kc4 * 4) * 4;
  auto fetch = [&](int64_t tile, f32x4* nx) {
    const int rows = rows_of(tile);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(X) + (size_t)(rows ? tile : 0) * BM * K, 0, rows * K * 4, 0x00020000);
#pragma unroll
    for (int p = 0; p < NP; ++p)
      nx[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, xvo, p * ROWS_PER_PASS * K * 4, MMG_NT_LD));
  };
  auto stage_pass = [&](int64_t tile, int buf, const f32x4* nx, int p) __attribute__((always_inline)) {   // one 16-B element per thread
    const int64_t row0 = tile * BM;
    __bf16* pb = planes + (size_t)buf * 3 * BM * LDP;
    {
      const int r = p * ROWS_PER_PASS + prow;
      f32x4 v = nx[p];
      if (PRO) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(fmaf(v[j], sc[j], sh[j]), floor_v);
        if (drop)
          mmg_drop4(v, pr.key, (uint64_t)(pr.row_offset + row0 + r) * (uint64_t)K + (uint64_t)(kc4 * 4), pr.thr, pr.inv_keep);
      }
      xbf16x4 q0, q1, q2;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const __bf16 a = (__bf16)v[j];
        const float r1 = v[j] - (float)a;
        const __bf16 b = (__bf16)r1;
        q0[j] = a; q1[j] = b; q2[j] = (__bf16)(r1 - (float)b);
      }
      *reinterpret_cast<xbf16x4*>(pb + (0 * BM + r) * LDP + kc4 * 4) = q0;
      *reinterpret_cast<xbf16x4*>(pb + (1 * BM + r) * LDP + kc4 * 4) = q1;
      *reinterpret_cast<xbf16x4*>(pb + (2 * BM + r) * LDP + kc4 * 4) = q2;
    }
  };
  auto stage = [&](int64_t tile, int buf, const f32x4* nx) __attribute__((always_inline)) {  // prologue + split + three plane writes
#pragma unroll
    for (int p = 0; p < NP; ++p) stage_pass(tile, buf, nx, p);
  };
  // K = 256 keeps ONE workgroup per CU (192 registers of W pieces per wave, 101 KB of planes): no second workgroup stages
  // its tile while this one multiplies, so the staging of tile t+1 is dealt out between the k-steps of tile t here (the
  // matrix pipe runs a 32 x 32 x 16 product for 32 clocks; the vector and LDS instructions of a staging pass issue in
  // its shadow).  The two tiles use different plane buffers, the barrier at the end of the tile orders both.
  constexpr bool WEAVE = (K >= 128) && !L2 && (NP <= NK) && (NK % NP == 0);
  fetch(t0, nxa);
  stage(t0, 0, nxa);
  fetch(t0 + G, nxa);
  fetch(t0 + 2 * G, nxb);
  __syncthreads();
  const int yvo = ((4 * h) * N + wn * 32 + l31) * 4;     // C/D map: col = lane&31, row = (i&3) + 8*(i>>2) + 4*(lane>>5)
  const int c0 = bx * BN;
  auto tile_body = [&](int64_t tt, int buf, f32x4* nx) {
    // nx holds tile tt+G (fetched two tiles ago); after staging it, the registers take tile tt+3G
    const int rows = rows_of(tt);
    const __amdgpu_buffer_rsrc_t ys = __builtin_amdgcn_make_buffer_rsrc(
        Y + (size_t)(rows ? tt : 0) * BM * N + c0, 0, rows ? (rows * N - c0) * 4 : 0, 0x00020000);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i)
      acc[i] = ACC ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ys, yvo, ((i & 3) + 8 * (i >> 2)) * N * 4, 0))
                   : 0.f;
    float nby[NBN ? 16 : 1];
    if constexpr (NBN) next_bn_load(nb, tt, rows, N, c0, yvo, nby);   // in flight under the products
    if constexpr (!WEAVE) {
      stage(tt + G, buf ^ 1, nx);              // the other buffer: its readers passed the barrier of the last tile
      fetch(tt + 3 * G, nx);
      __builtin_amdgcn_sched_barrier(0);       // keep the fetch ahead of the matrix loop (the scheduler sinks it otherwise)
    }
    const __bf16* ap = planes + (size_t)buf * 3 * BM * LDP + l31 * LDP + 8 * h;
    if constexpr (!WEAVE) {
#pragma unroll
      for (int ks = 0; ks < NK; ++ks) {
        const xbf16x8 a1 = *reinterpret_cast<const xbf16x8*>(ap + ks * 16);
        const xbf16x8 a2 = *reinterpret_cast<const xbf16x8*>(ap + BM * LDP + ks * 16);
        const xbf16x8 a3 = *reinterpret_cast<const xbf16x8*>(ap + 2 * BM * LDP + ks * 16);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, wb[ks][0], acc, 0, 0, 0);   // small terms first
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, wb[ks][2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, wb[ks][1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, wb[ks][0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, wb[ks][1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, wb[ks][0], acc, 0, 0, 0);
      }
    } else {
      // A wave that is alone on its SIMD hides nothing by itself: the fragments of k-step ks+1 are fetched before the
      // products of k-step ks (their LDS latency then passes under six matrix instructions), and one staging pass of the
      // NEXT tile is dealt out between the products of every KS_PER_PASS-th k-step (sched_group_barrier: a dependent
      // MFMA chain issues one instruction per 32 clocks, the vector / LDS work of the pass issues in between).
      constexpr int KS_PER_PASS = NK / NP;
      xbf16x8 fr[2][3];
      auto ldfrag = [&](int ks, xbf16x8* f) __attribute__((always_inline)) {
        f[0] = *reinterpret_cast<const xbf16x8*>(ap + ks * 16);
        f[1] = *reinterpret_cast<const xbf16x8*>(ap + BM * LDP + ks * 16);
        f[2] = *reinterpret_cast<const xbf16x8*>(ap + 2 * BM * LDP + ks * 16);
      };
      ldfrag(0, fr[0]);
#pragma unroll
      for (int pg = 0; pg < NP; ++pg) {          // one scheduling region = the k-steps that carry one staging pass
#pragma unroll
        for (int kk = 0; kk < KS_PER_PASS; ++kk) {
          const int ks = pg * KS_PER_PASS + kk;
          if (ks + 1 < NK) ldfrag(ks + 1, fr[(ks + 1) & 1]);
          const xbf16x8* f = fr[ks & 1];
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[2], wb[ks][0], acc, 0, 0, 0);   // small terms first
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[0], wb[ks][2], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[1], wb[ks][1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[1], wb[ks][0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[0], wb[ks][1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[0], wb[ks][0], acc, 0, 0, 0);
        }
        stage_pass(tt + G, buf ^ 1, nx, pg);
        // a lone wave hides 2-3 vector instructions behind a matrix instruction (profiles/probes/mfma_shadow): the pass
        // (~25 of them, ~45 with a prologue) is dealt out over all 6 * KS_PER_PASS products of its k-steps
        constexpr int NM = 6 * KS_PER_PASS, VPER = ((PRO ? 48 : 26) + NM - 1) / NM;
#pragma unroll
        for (int i = 0; i < NM; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                        // one MFMA
          if (i % 6 < 3) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);         // one fragment read of the next k-step
          __builtin_amdgcn_sched_group_barrier(0x002, VPER, 0);                     // a slice of the staging pass
          if (i >= NM - 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);       // its plane writes at the end
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (WEAVE) fetch(tt + 3 * G, nx);  // (its registers are free only now: two tiles stay in flight all the same)
    if constexpr (L2) {
      // row sums of squares: 32 lanes of a half-wave hold the 32 columns of this wave for 16 rows each; the WN waves'
      // partials meet in LDS, 32 threads turn them into 1 / max(norm, eps), every lane scales its 16 values
      float vv[16], ss[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) { vv[i] = acc[i] + bv; ss[i] = vv[i] * vv[i]; }
      // 32-lane sums on the vector ALU's data-parallel primitives (five v_add_f32 with a DPP operand per value; the
      // first version went through 80 ds_bpermute round trips per tile): xor 1, xor 2 inside a quad, mirror inside 8,
      // mirror inside 16, then the row total of lanes 0..15 broadcast onto lanes 16..31 -- those hold the half's sum
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float v = ss[i];
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false)); // row_bcast:15 -> rows 1, 3
        ss[i] = v;
      }
      if (l31 == 31) {
#pragma unroll
        for (int i = 0; i < 16; ++i) l2_part[wn * 32 + (i & 3) + 8 * (i >> 2) + 4 * h] = ss[i];
      }
      __syncthreads();                       // partials complete; also: buf fully read, buf^1 fully written
      if (tid < 32) {
        float tot = l2_part[tid];
#pragma unroll
        for (int w = 1; w < WN; ++w) tot += l2_part[w * 32 + tid];
        const float rinv = 1.0f / fmaxf(sqrtf(tot), l2_eps);
        l2_rn[tid] = rinv;
        if (tid < rows) rn_out[tt * BM + tid] = rinv;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int r = (i & 3) + 8 * (i >> 2);
        const float v = vv[i] * l2_rn[r + 4 * h];
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ys, yvo, r * N * 4, MMG_NT_ST);
      }
      return;
    }
    if constexpr (NBN) {
      float vv[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        vv[i] = acc[i] + bv;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vv[i]), ys, yvo, ((i & 3) + 8 * (i >> 2)) * N * 4, MMG_NT_ST);
      }
      next_bn_tile(nb, nbc, vv, nby, rows, tt * BM, N, col, lane, cs1, cs2);
    } else {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int r = (i & 3) + 8 * (i >> 2);
        const float v = acc[i] + bv;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ys, yvo, r * N * 4, MMG_NT_ST);
        const float vs = r + 4 * h < rows ? v : 0.f;
        t1 += vs; t2 = fmaf(vs, vs, t2);
      }
      if (stat_partial) { cs1 += (double)t1; cs2 += (double)t2; }     // 16 rows in fp32, tiles in fp64
    }
    __syncthreads();                         // buf fully read, buf^1 fully written
  };
  // A tile index past the end is harmless (zero-sized descriptors).  The first pair is peeled: the loop is then entered
  // with the same loads / stores in flight as on its back edge, and the wait counts the compiler derives (the minimum
  // over both edges) are the steady-state ones.
  tile_body(t0, 0, nxa);
  tile_body(t0 + G, 1, nxb);
  for (int i = 2; i < n_my; i += 2) {
    tile_body(t0 + (int64_t)i * G, 0, nxa);
    tile_body(t0 + (int64_t)(i + 1) * G, 1, nxb);
  }
  if (stat_partial) {
    // two partials per column (the lane halves) -> one: partial[row-set][2][N], fixed order
    double* red = reinterpret_cast<double*>(planes);          // the planes are dead: every wave passed the last barrier
    red[(h * 2 + 0) * BN + wn * 32 + l31] = cs1;
    red[(h * 2 + 1) * BN + wn * 32 + l31] = cs2;
    __syncthreads();
    for (int e = tid; e < 2 * BN; e += NTHR) {
      const int which = e / BN, c = e % BN;
      stat_partial[((size_t)by * 2 + which) * N + bx * BN + c] = red[which * BN + c] + red[(2 + which) * BN + c];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_linear_fwd_h3_k256 (round 4): the same layer on THREE f16 products instead of six bf16 ones -- half the matrix
// instructions of a shape that is matrix-bound.  Every row of X and every row of W (an output column) gets its own power
// of two: X = x * 2^ex with the row's largest magnitude in [2^13, 2^14), likewise W = w * 2^ew; each is split into
//   hi = f16(V),  lo = f16((V - hi) * 2^11)          (22 significant bits; the low piece is scaled up by 2^11 so that it
// is a NORMAL f16 wherever hi is: |V| >= 2^-14, i.e. within 2^-27 of the row's largest magnitude), and
//   x . w * 2^(ex + ew) = Xh Wh + 2^-11 (Xh Wl + Xl Wh) + O(2^-22 |X||W|):
// the first product goes to one fp32 accumulator, the two cross terms to a second one, combined and unscaled in the
// epilogue.  The dropped Xl Wl is 2^-22 of a product, as are the two roundings of the low pieces: ~3 * 2^-22 = 7e-7 per
// product against 2^-24 for the six-term bf16 split -- tests/test_ops_gpu.py holds the layer to 2e-6 of an fp64 reference
// like the other kernels.  A row's scale comes from the row itself (wave = row while a tile is staged: one DPP reduction),
// so the operand range is the fp32 range; an element more than 2^27 below its row's maximum loses relative precision, but
// what it can add to a dot product with that row is below the rounding of the fp32 sum.  inf / NaN in a row of X: NaN in that
// row of Y (as the bf16 kernels).  One workgroup of eight waves owns all 256 output columns (a wave: one 32-column tile, its W
// pieces -- 128 registers -- resident; two waves per SIMD) and persists over 32-row tiles of X dealt round-robin: a tile is
// staged ONCE per CU (prologue, row scale, split, two f16 planes in LDS, double-buffered), the staging of tile t+1 is woven
// between the products of tile t, two tiles of X are in flight in registers.  Round 3's six-term kernel for this shape (four
// waves, two column tiles and 384 registers of W pieces per wave) measured 147 us per [183400, 256] x [256, 256] call, this
// one 102-112 (179 -> 122 with BatchNorm fold + ReLU + dropout in the prologue); its parts add up -- staging and epilogue
// alone 48 us, + matrix instructions 32, + memory 29 -- the two waves of a SIMD run the same phase between the per-tile barriers.
typedef _Float16 xf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 xf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 xf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float wave_max_dpp(float x) {      // -> the maximum over the 64 lanes, in a scalar register
#define MMG_DPP_MAX(ctrl, rmask)                                                                                             \
  x = fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, x), __builtin_bit_cast(int, x),   \
                                                                     ctrl, rmask, 0xF, false)))
  MMG_DPP_MAX(0xB1, 0xF);        // quad_perm [1,0,3,2]
  MMG_DPP_MAX(0x4E, 0xF);        // quad_perm [2,3,0,1]
  MMG_DPP_MAX(0x141, 0xF);       // row_half_mirror
  MMG_DPP_MAX(0x140, 0xF);       // row_mirror
  MMG_DPP_MAX(0x142, 0xA);       // row_bcast:15 -> rows 1, 3
  MMG_DPP_MAX(0x143, 0xC);       // row_bcast:31 -> rows 2, 3
#undef MMG_DPP_MAX
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}
// exponent e with  m * 2^e in [2^13, 2^14)  for a normal m (0 for zero / denormal / inf / NaN), from the exponent field of m:
// integer arithmetic only, so that a wave-uniform m (wave_max_dpp) keeps all of it on the scalar unit
__device__ __forceinline__ int h3_exponent(float m) {
  const int E = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xFFu);     // m = 1.f * 2^(E - 127)
  const int e = 140 - E;
  return (E == 0 || E == 255) ? 0 : (e > 126 ? 126 : e);
}
__device__ __forceinline__ float h3_pow2(int e) { return __builtin_bit_cast(float, (unsigned)(e + 127) << 23); }

template <int PRO, bool ACC, bool STATS, int WN>      // WN = 4: two column tiles per wave; 8: one (two waves per SIMD)
__global__ __launch_bounds__(64 * WN, 1) void k_linear_fwd_h3_k256(
    const float* __restrict__ X, ProDev pr, const float* __restrict__ W, const float* __restrict__ bias,
    float* __restrict__ Y, int64_t M, int N, int flags, double* __restrict__ stat_partial) {
  constexpr int K = 256, CT = 8 / WN, LDP = K + 8, BN = 256, NK = K / 16, NTHR = 64 * WN, BM = 32;
  constexpr int PLANE = BM * LDP;            // one piece of one buffer, in f16 elements
  if (PRO) pr.resolve();
  extern __shared__ __attribute__((aligned(16))) _Float16 hplanes[];     // [2 buffers][2 pieces][BM][LDP], then float ex2[2][BM]
  float* rowf = reinterpret_cast<float*>(hplanes + 2 * 2 * PLANE);       // 2^-ex of the staged rows, per buffer
  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const int bx = blockIdx.x, by = blockIdx.y;
  const int c0 = bx * BN;
  xf16x8 wh[CT][NK], wl[CT][NK];
  float bv[CT], cf[CT];                      // bias and 2^-ew of this lane's two columns
  {
    const bool wkn = (flags & MMG_LIN_W_KN) != 0;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int col = c0 + (wn * CT + ct) * 32 + l31;
      bv[ct] = bias ? bias[col] : 0.f;
      const float* wp = wkn ? W + (size_t)(8 * h) * N + col : W + (size_t)col * K + 8 * h;
      float wv[NK][8];
      float m = 0.f;
#pragma unroll
      for (int ks = 0; ks < NK; ++ks) {
        if (wkn) {
#pragma unroll
          for (int j = 0; j < 8; ++j) wv[ks][j] = wp[(size_t)(ks * 16 + j) * N];
        } else {
          const f32x4 w0 = *reinterpret_cast<const f32x4*>(wp + ks * 16), w1 = *reinterpret_cast<const f32x4*>(wp + ks * 16 + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { wv[ks][j] = w0[j]; wv[ks][4 + j] = w1[j]; }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float a = fabsf(wv[ks][j]);
          m = fmaxf(m, a < __builtin_inff() ? a : 0.f);
        }
      }
      m = fmaxf(m, __shfl_xor(m, 32, 64));             // the other half of the column's K (lane ^ 32)
      const int ew = h3_exponent(m);
      const float sw = h3_pow2(ew > 126 ? 126 : ew);
      cf[ct] = h3_pow2(-(ew > 126 ? 126 : ew));
#pragma unroll
      for (int ks = 0; ks < NK; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = wv[ks][j] * sw;
          const _Float16 a = (_Float16)v;
          wh[ct][ks][j] = a;
          wl[ct][ks][j] = (_Float16)((v - (float)a) * 2048.f);
        }
    }
  }
  constexpr int K4 = K / 4, ROWS_PER_PASS = NTHR / K4, NP = BM / ROWS_PER_PASS;      // 64 quads, 4 rows per pass, 8 passes
  constexpr int KPP = NK / NP;               // k-steps per staging pass (2 | 4)
  static_assert(NK % NP == 0 && K4 == 64, "a staging pass every KPP k-steps; a wave stages one row per pass");
  const int kc4 = lane, prow = wn;           // (tid % K4, tid / K4)
  float floor_v = -__builtin_inff();
  bool drop = false;
  if (PRO) {
    if (pr.relu) floor_v = 0.f;
    drop = pr.p > 0.f;
  }
  const int64_t n_tiles = (M + BM - 1) / BM;
  const int64_t G = gridDim.y, t0 = by;
  if (t0 >= n_tiles) return;
  const int n_my = (int)((n_tiles - t0 + G - 1) / G);
  auto rows_of = [&](int64_t tile) -> int {
    const int64_t r = M - tile * BM;
    return r <= 0 ? 0 : (r < BM ? (int)r : BM);
  };
  auto x_rsrc = [&](int64_t tile) __attribute__((always_inline)) {
    const int rows = rows_of(tile);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X) + (size_t)(rows ? tile : 0) * BM * K, 0, rows * K * 4,
                                             0x00020000);
  };
  const int xvo = (prow * K + kc4 * 4) * 4;
  // TWO tiles of X in registers (the six-term kernel had room for one): a workgroup per CU with one 32 KB tile in flight
  // is 8 MB on the chip -- at ~2 us of loaded HBM latency a ceiling of ~4 TB/s; the freed W-piece registers hold the second
  f32x4 nx[2][NP];
  auto fetch_pass = [&](__amdgpu_buffer_rsrc_t rs, int slot, int p) __attribute__((always_inline)) {
    nx[slot][p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, xvo, p * ROWS_PER_PASS * K * 4, MMG_NT_LD));
  };
  auto stage_pass = [&](int64_t tile, int buf, int slot, int p) __attribute__((always_inline)) {
    const int64_t row0 = tile * BM;
    _Float16* pb = hplanes + (size_t)buf * 2 * PLANE;
    const int r = p * ROWS_PER_PASS + prow;
    f32x4 v = nx[slot][p];
    if (PRO) {
      f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
      if (PRO == 2 || pr.scale) {
        sc = *reinterpret_cast<const f32x4*>(pr.scale + kc4 * 4);
        sh = *reinterpret_cast<const f32x4*>(pr.shift + kc4 * 4);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaxf(fmaf(v[j], sc[j], sh[j]), floor_v);
      if (PRO == 2 || drop)
        mmg_drop4(v, pr.key, (uint64_t)(pr.row_offset + row0 + r) * (uint64_t)K + (uint64_t)(kc4 * 4), pr.thr, pr.inv_keep);
    }
    // the row's scale: this wave holds the whole row (64 quads)
    float m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
    m = wave_max_dpp(m);
    const int ex = h3_exponent(m);
    const float sx = h3_pow2(ex);
    xf16x4 qh, ql;
#pragma unroll
    for (int j = 0; j < 4; j += 2) {
      const f32x2 a = {v[j] * sx, v[j + 1] * sx};
      const xf16x2 hh = __builtin_convertvector(a, xf16x2);
      const f32x2 rr = {(a[0] - (float)hh[0]) * 2048.f, (a[1] - (float)hh[1]) * 2048.f};
      const xf16x2 ll = __builtin_convertvector(rr, xf16x2);
      qh[j] = hh[0]; qh[j + 1] = hh[1]; ql[j] = ll[0]; ql[j + 1] = ll[1];
    }
    *reinterpret_cast<xf16x4*>(pb + (0 * BM + r) * LDP + kc4 * 4) = qh;
    *reinterpret_cast<xf16x4*>(pb + (1 * BM + r) * LDP + kc4 * 4) = ql;
    if (lane == 0) rowf[buf * BM + r] = h3_pow2(-ex);
  };
  {
    const __amdgpu_buffer_rsrc_t r0 = x_rsrc(t0), r1 = x_rsrc(t0 + G), r2 = x_rsrc(t0 + 2 * G);
#pragma unroll
    for (int p = 0; p < NP; ++p) fetch_pass(r0, 0, p);
#pragma unroll
    for (int p = 0; p < NP; ++p) fetch_pass(r1, 1, p);
#pragma unroll
    for (int p = 0; p < NP; ++p) { stage_pass(t0, 0, 0, p); fetch_pass(r2, 0, p); }
  }
  __syncthreads();
  double cs1[CT] = {}, cs2[CT] = {};
  // (two tiles per trip: the register slot of a tile is its parity, a compile-time index)
  for (int i2 = 0; i2 < n_my; i2 += 2) {
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const int i = i2 + par;
    if (i >= n_my) break;
    const int64_t tt = t0 + (int64_t)i * G;
    const int buf = par;
    const int rows = rows_of(tt);
    const __amdgpu_buffer_rsrc_t ys = __builtin_amdgcn_make_buffer_rsrc(
        Y + (size_t)(rows ? tt : 0) * BM * N + c0, 0, rows ? (rows * N - c0) * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t xs3 = x_rsrc(tt + 3 * G);     // refills: three tiles ahead (slot of tile t + 1)
    f32x16 ah[CT], al[CT];
    float yold[ACC ? CT : 1][16];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int yvo = ((4 * h) * N + (wn * CT + ct) * 32 + l31) * 4;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        ah[ct][r] = 0.f; al[ct][r] = 0.f;
        if (ACC) yold[ct][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ys, yvo, ((r & 3) + 8 * (r >> 2)) * N * 4, 0));
      }
    }
    const _Float16* ap = hplanes + (size_t)buf * 2 * PLANE + l31 * LDP + 8 * h;
    xf16x8 fr[2][2];
    auto ldfrag = [&](int ks, xf16x8* f) __attribute__((always_inline)) {
      f[0] = *reinterpret_cast<const xf16x8*>(ap + ks * 16);
      f[1] = *reinterpret_cast<const xf16x8*>(ap + PLANE + ks * 16);
    };
    ldfrag(0, fr[0]);
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
      const bool pass = (ks % KPP) == KPP - 1;
      if (ks + 1 < NK) ldfrag(ks + 1, fr[(ks + 1) & 1]);
      const xf16x8* f = fr[ks & 1];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        al[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[1], wh[ct][ks], al[ct], 0, 0, 0);   // Xl Wh
        al[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[0], wl[ct][ks], al[ct], 0, 0, 0);   // Xh Wl
        ah[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[0], wh[ct][ks], ah[ct], 0, 0, 0);   // Xh Wh

      }
      if (pass) { stage_pass(tt + G, buf ^ 1, par ^ 1, ks / KPP); fetch_pass(xs3, par ^ 1, ks / KPP); }
#pragma unroll
      for (int q = 0; q < 3 * CT; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                          // one MFMA
        if (q < 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);               // a fragment read of the next k-step
        if (pass) __builtin_amdgcn_sched_group_barrier(0x002, (PRO ? 72 : 42) / (3 * CT), 0);   // a slice of the staging pass
        if (pass && q >= 3 * CT - 3 && q < 3 * CT - 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // its plane writes
        if (pass && q == 3 * CT - 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // the refill load
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // epilogue: y = (ah + 2^-11 al) * 2^-ex[row] * 2^-ew[col] + bias (+ the old tile)
    const float* rf = rowf + buf * BM + 4 * h;
    f32x4 rf4[4];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) rf4[g4] = *reinterpret_cast<const f32x4*>(rf + 8 * g4);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int yvo = ((4 * h) * N + (wn * CT + ct) * 32 + l31) * 4;
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int r16 = 0; r16 < 16; ++r16) {
        const int r = (r16 & 3) + 8 * (r16 >> 2);
        float v = fmaf(al[ct][r16], 1.f / 2048.f, ah[ct][r16]) * (rf4[r16 >> 2][r16 & 3] * cf[ct]) + bv[ct];
        if (ACC) v += yold[ct][r16];
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ys, yvo, r * N * 4, MMG_NT_ST);
        if (STATS) {
          const float vs = r + 4 * h < rows ? v : 0.f;
          t1 += vs; t2 = fmaf(vs, vs, t2);
        }
      }
      if (STATS) { cs1[ct] += (double)t1; cs2[ct] += (double)t2; }
    }
    __syncthreads();                         // buf fully read, buf^1 fully written
  }
  }
  if (STATS) {
    double* red = reinterpret_cast<double*>(hplanes);         // the planes are dead: every wave passed the last barrier
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      red[(h * 2 + 0) * BN + (wn * CT + ct) * 32 + l31] = cs1[ct];
      red[(h * 2 + 1) * BN + (wn * CT + ct) * 32 + l31] = cs2[ct];
    }
    __syncthreads();
    for (int e = tid; e < 2 * BN; e += NTHR) {
      const int which = e / BN, c = e % BN;
      stat_partial[((size_t)by * 2 + which) * N + c0 + c] = red[which * BN + c] + red[(2 + which) * BN + c];
    }
  }
}

template <int PRO, bool ACC, bool STATS>
int launch_fwd_h3_k256_vs(const float* X, const ProDev& pr, const float* W, const float* bias, float* Y, int64_t M, int N,
                          int flags, hipStream_t st, double* stat_partial, int64_t gy) {
  constexpr int lds = 2 * 2 * 32 * (256 + 8) * 2 + 2 * 32 * 4;
  // eight waves (WN = 4 -- two column tiles per wave, 424+ registers -- measured 122 us against 108); the literal is
  // written out so that the probe records the instantiated symbol
  MMG_CHECK_HIP((MmgMaxLds<&k_linear_fwd_h3_k256<PRO, ACC, STATS, 8>, lds>::set()), "linear_fwd(attr)");
  MMG_LAUNCH(MMG_PROBE_LINEAR_FWD, M, N, 256, (flags & MMG_LIN_ACCUMULATE) | (PRO ? 4 : 0),
             (k_linear_fwd_h3_k256<PRO, ACC, STATS, 8>), dim3((unsigned)(N / 256), (unsigned)gy), dim3(64 * 8), lds, st, X, pr,
             W, bias, Y, M, N, flags, stat_partial);
  return 0;
}
template <int PRO, bool ACC>
int launch_fwd_h3_k256_v(const float* X, const ProDev& pr, const float* W, const float* bias, float* Y, int64_t M, int N,
                         int flags, hipStream_t st, double* stat_partial, int64_t gy) {
  return stat_partial ? launch_fwd_h3_k256_vs<PRO, ACC, true>(X, pr, W, bias, Y, M, N, flags, st, stat_partial, gy)
                      : launch_fwd_h3_k256_vs<PRO, ACC, false>(X, pr, W, bias, Y, M, N, flags, st, stat_partial, gy);
}

// rows of partial statistics = grid.y of the K = 256, 256-column kernel
inline int64_t fwd_k256_rows(int64_t M, int N) {
  const int64_t n_tiles = (M + 31) / 32;
  int64_t gy = 256 / (N / 256);
  if (gy < 1) gy = 1;
  return gy > n_tiles ? n_tiles : gy;
}

int launch_fwd_h3_k256(const float* X, const ProDev& pr, const float* W, const float* bias, float* Y, int64_t M, int N,
                       int flags, hipStream_t st, double* stat_partial) {
  const bool pro = pr.scale || pr.relu || pr.p > 0.f, acc = (flags & MMG_LIN_ACCUMULATE) != 0;
  const int64_t gy = fwd_k256_rows(M, N);
  if (pro && pr.scale && pr.p > 0.f)
    return acc ? launch_fwd_h3_k256_v<2, true>(X, pr, W, bias, Y, M, N, flags, st, stat_partial, gy)
               : launch_fwd_h3_k256_v<2, false>(X, pr, W, bias, Y, M, N, flags, st, stat_partial, gy);
  if (pro) return acc ? launch_fwd_h3_k256_v<1, true>(X, pr, W, bias, Y, M, N, flags, st, stat_partial, gy)
                      : launch_fwd_h3_k256_v<1, false>(X, pr, W, bias, Y, M, N, flags, st, stat_partial, gy);
  return acc ? launch_fwd_h3_k256_v<0, true>(X, pr, W, bias, Y, M, N, flags, st, stat_partial, gy)
             : launch_fwd_h3_k256_v<0, false>(X, pr, W, bias, Y, M, N, flags, st, stat_partial, gy);
}

inline int64_t fwd_x6_rows(int64_t M, int N, int BN, int K = 128) {   // grid.y of the bf16-split forward (= partial stat rows)
  const int n_slices = N / BN;
  const int64_t n_tiles = (M + 31) / 32;
  // persistent workgroups: two per CU (52 KB of LDS, <= 256 registers) up to K = 128; K = 256 keeps 192 registers of
  // W pieces per wave and 101 KB of planes: one per CU
  // (a 64-column workgroup is two waves: three of them fit a CU -- 52 KB of planes, <= 256 registers -- and a layer with
  //  N = 64 has a single slice, so 768 workgroups instead of 512 put six waves on every CU)
  int64_t gy = (K <= 128 ? (BN == 64 ? 768 : 512) : 256) / n_slices;
  if (gy < 1) gy = 1;
  if (gy > n_tiles) gy = n_tiles;
  return gy;
}

template <int K, int WN, bool PRO, bool ACC, bool L2 = false, bool NBN = false>
int launch_fwd_x6_v(const float* X, const ProDev& pr, const float* W, const float* bias, float* Y, int64_t M, int N,
                    int flags, hipStream_t st, double* stat_partial, float* rn_out = nullptr, float l2_eps = 0.f,
                    const NextBnDev& nb = next_bn_none()) {
  constexpr int BN = 32 * WN;
  const int n_slices = N / BN;
  const int64_t gy = fwd_x6_rows(M, N, BN, K);
  constexpr int lds = 2 * 3 * 32 * (K + 8) * 2;
  MMG_CHECK_HIP((MmgMaxLds<&k_linear_fwd_x6<K, WN, PRO, ACC, L2, NBN>, lds>::set()), "linear_fwd(attr)");
  MMG_LAUNCH(MMG_PROBE_LINEAR_FWD, M, N, K, (flags & MMG_LIN_ACCUMULATE) | (PRO ? 4 : 0) | (L2 ? 32 : 0) | (NBN ? 512 : 0),
             (k_linear_fwd_x6<K, WN, PRO, ACC, L2, NBN>), dim3((unsigned)n_slices, (unsigned)gy), dim3(64 * WN), lds, st, X,
             pr, W, bias, Y, M, N, flags, stat_partial, rn_out, l2_eps, nb);
  return 0;
}

template <int K, int WN>
int launch_fwd_x6(const float* X, const ProDev& pr, const float* W, const float* bias, float* Y, int64_t M, int N,
                  int flags, hipStream_t st, double* stat_partial = nullptr) {
  const bool pro = pr.scale || pr.relu || pr.p > 0.f, acc = (flags & MMG_LIN_ACCUMULATE) != 0;
  if (pro) return acc ? launch_fwd_x6_v<K, WN, true, true>(X, pr, W, bias, Y, M, N, flags, st, stat_partial)
                      : launch_fwd_x6_v<K, WN, true, false>(X, pr, W, bias, Y, M, N, flags, st, stat_partial);
  return acc ? launch_fwd_x6_v<K, WN, false, true>(X, pr, W, bias, Y, M, N, flags, st, stat_partial)
             : launch_fwd_x6_v<K, WN, false, false>(X, pr, W, bias, Y, M, N, flags, st, stat_partial);
}

// ---------------------------------------------------------------------------- BatchNorm backward inside the data-gradient GEMM
// dX = dZ . W  with  dZ = the backward of  dropout(relu(BN(y)))  at the upstream gradient G  (mmg_bn_bwd_apply's
// arithmetic), computed WHILE the tile is staged: the kernel reads G and Y once, writes dZ once (the weight gradient of
// the same layer reads it afterwards) and dX -- the separate apply pass read G and Y, wrote dZ, and this GEMM read dZ
// again: 94 MB less per layer at the x100 shape.  Same skeleton as k_linear_fwd_x6 (W pieces in registers, three bf16
// planes double-buffered in LDS, per-tile buffer descriptors, peeled tile pair); both inputs run ONE tile ahead in
// registers (two tiles of two tensors would not fit 256 registers beside the W pieces).  One workgroup covers all N
// output columns, so every dZ tile is produced exactly once.
// MODE 1: the same skeleton with the backward of the row L2 normalisation in the staging phase (mmg_l2norm_bwd's
// arithmetic: dZ = rn * (G - out * <G, out>), the row dot product over the K / 4 lanes that hold a row): Y = the
// normalised rows, mean = rn [M].
// MODE 2: MODE 0 with TWO upstream gradients through the same BatchNorm + ReLU, each with its own dropout mask
// (mmg_bn_bwd_apply2: the two encode_nodes passes of a training step share their first layer).
// MODE 3: MODE 0 with an upstream gradient that is ZERO outside a short list of rows (mmg_bn_bwd_apply with G = NULL +
// mmg_bn_bwd_apply_rows): G holds the listed rows back to back, row_pos[row] = position in that list or -1.
struct BnBwdDev {
  const float* Y; const float* mean; const float* rstd; const double* sums; double inv_count;
  float* dZ; float* dbeta; float* dgamma; float l2_eps; const float* G2; const int32_t* row_pos; int64_t n_sel;
};

// NBN: the statistics of the NEXT BatchNorm backward -- the one that consumes DX -- from the epilogue (NextBnDev),
// partial[row-set][2][N] -> mmg_partial_sum.
template <int K, int WN, int MODE = 0, bool NBN = false>
__global__ __launch_bounds__(64 * WN, WN == 4 ? 2 : 1) void k_linear_bnbwd_x6(const float* __restrict__ G, BnBwdDev bb, ProDev pr,
                                                                const float* __restrict__ W, float* __restrict__ DX,
                                                                int64_t M, ProDev pr2, NextBnDev nb,
                                                                double* __restrict__ stat_partial) {
  if (MODE != 1) pr.resolve();
  if (MODE == 2) pr2.resolve();
  constexpr int LDP = K + 8, N = 32 * WN, NK = K / 16, NTHR = 64 * WN, BM = 32;
  extern __shared__ __attribute__((aligned(16))) __bf16 planes[];     // [2 buffers][3 pieces][BM][LDP]
  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int col = wn * 32 + l31;
  xbf16x8 wb[NK][3];                       // W stored [K, N] (the forward weight, read in place)
  {
    const float* wp = W + (size_t)(8 * h) * N + col;
#pragma unroll
    for (int ks = 0; ks < NK; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = wp[(size_t)(ks * 16 + j) * N];
        const __bf16 a = (__bf16)v;
        const float r1 = v - (float)a;
        const __bf16 b = (__bf16)r1;
        wb[ks][0][j] = a; wb[ks][1][j] = b; wb[ks][2][j] = (__bf16)(r1 - (float)b);
      }
  }
  NextBnCol nbc = {};
  double cs1 = 0.0, cs2 = 0.0;
  if constexpr (NBN) nbc = next_bn_col(nb, col);
  constexpr int K4 = K / 4;
  const int kc4 = tid % K4, c = kc4 * 4;    // this thread always touches the same 4 columns of G / Y / dZ
  constexpr int ROWS_PER_PASS = NTHR / K4;
  constexpr int NP = BM / ROWS_PER_PASS;
  const int prow = tid / K4;
  const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 sc = one, sh = zero, mu = zero, rs = one, a0 = zero, a1 = zero;
  if (MODE != 1 && pr.scale) {
    sc = *reinterpret_cast<const f32x4*>(pr.scale + c); sh = *reinterpret_cast<const f32x4*>(pr.shift + c);
    mu = *reinterpret_cast<const f32x4*>(bb.mean + c); rs = *reinterpret_cast<const f32x4*>(bb.rstd + c);
    if (bb.sums) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a0[j] = (float)(bb.sums[c + j] * bb.inv_count);
        a1[j] = (float)(bb.sums[K + c + j] * bb.inv_count);
      }
    }
  }
  // (the row-set index through readfirstlane: behind the lane-dependent branch below the compiler otherwise carries
  //  blockIdx.y -- known to be 0 inside it -- in a VECTOR register, every tile index and buffer descriptor derived from it
  //  becomes "divergent", and each of the ~135 buffer accesses of the kernel is wrapped in a waterfall loop)
  const int by_u = __builtin_amdgcn_readfirstlane((int)blockIdx.y);
  if (MODE != 1 && bb.sums && by_u == 0 && tid < K4) {          // d beta / d gamma ride along
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (bb.dbeta) bb.dbeta[c + j] = (float)bb.sums[c + j];
      if (bb.dgamma) bb.dgamma[c + j] = (float)bb.sums[K + c + j];
    }
  }
  const bool relu = MODE != 1 && pr.relu == MMG_ACT_RELU, drop = MODE != 1 && pr.p > 0.f, has_bn = MODE != 1 && pr.scale != nullptr;
  const bool drop2 = MODE == 2 && pr2.p > 0.f;
  const int64_t n_tiles = (M + BM - 1) / BM;
  const int64_t GY = gridDim.y, t0 = by_u;
  if (t0 >= n_tiles) return;
  const int n_my = (int)((n_tiles - t0 + GY - 1) / GY);
  auto rows_of = [&](int64_t tile) __attribute__((always_inline)) -> int {
    const int64_t r = M - tile * BM;
    return r <= 0 ? 0 : (r < BM ? (int)r : BM);
  };
  f32x4 ng[NP], ny[NP], ng2[MODE == 2 ? NP : 1];      // the NEXT tile of G and Y (and G2)
  const int xvo = (prow * K + kc4 * 4) * 4;
  int ridx[MODE == 3 ? NP : 1];             // MODE 3: list positions of the rows of the tile the NEXT fetch serves
  auto fetch_idx = [&](int64_t tile) __attribute__((always_inline)) {
    const int rows = rows_of(tile);
    const __amdgpu_buffer_rsrc_t ps = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<int32_t*>(bb.row_pos) + (size_t)(rows ? tile : 0) * BM, 0, rows * 4, 0x00020000);
#pragma unroll
    for (int p = 0; p < NP; ++p) ridx[p] = rows ? (int)__builtin_amdgcn_raw_buffer_load_b32(ps, (p * ROWS_PER_PASS + prow) * 4, 0, 0) : -1;
  };
  auto fetch = [&](int64_t tile) __attribute__((always_inline)) {
    const int rows = rows_of(tile);
    const size_t off = (size_t)(rows ? tile : 0) * BM * K;
    const __amdgpu_buffer_rsrc_t ysrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bb.Y) + off, 0, rows * K * 4, 0x00020000);
    if constexpr (MODE == 3) {
      // the listed rows through one descriptor over the whole list: a row that is not listed (or past the end) gets an
      // offset behind the list and reads 0
      const __amdgpu_buffer_rsrc_t gs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(G), 0, (int)(bb.n_sel * K * 4), 0x00020000);
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const unsigned vo = ridx[p] >= 0 && p * ROWS_PER_PASS + prow < rows ? (unsigned)ridx[p] * (unsigned)(K * 4) + (unsigned)(kc4 * 16) : 0xFFFFFF00u;
        ng[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(gs, vo, 0, 0));
        ny[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ysrc, xvo, p * ROWS_PER_PASS * K * 4, MMG_NT_LD));
      }
      fetch_idx(tile + GY);
      return;
    }
    const __amdgpu_buffer_rsrc_t gs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(G) + off, 0, rows * K * 4, 0x00020000);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      ng[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(gs, xvo, p * ROWS_PER_PASS * K * 4, MMG_NT_LD));
      ny[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ysrc, xvo, p * ROWS_PER_PASS * K * 4, MMG_NT_LD));
    }
    if constexpr (MODE == 2) {
      const __amdgpu_buffer_rsrc_t g2s = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bb.G2) + off, 0, rows * K * 4, 0x00020000);
#pragma unroll
      for (int p = 0; p < NP; ++p)
        ng2[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(g2s, xvo, p * ROWS_PER_PASS * K * 4, MMG_NT_LD));
    }
  };
  auto stage = [&](int64_t tile, int buf) __attribute__((always_inline)) {   // BatchNorm backward of the tile in ng / ny, dZ out, split, three plane writes
    const int64_t row0 = tile * BM;
    const int rows = rows_of(tile);
    const __amdgpu_buffer_rsrc_t zs = __builtin_amdgcn_make_buffer_rsrc(
        bb.dZ + (size_t)(rows ? tile : 0) * BM * K, 0, rows * K * 4, 0x00020000);
    __bf16* pb = planes + (size_t)buf * 3 * BM * LDP;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int r = p * ROWS_PER_PASS + prow;
      const f32x4 y4 = ny[p];
      f32x4 gm = ng[p], v;
      if constexpr (MODE == 1) {
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) dot = fmaf(gm[j], y4[j], dot);
#pragma unroll
        for (int o = 1; o < K4; o <<= 1) dot += __shfl_xor(dot, o, 64);      // the K / 4 lanes of a row are adjacent
        const int64_t grow = row0 + r;
        const float rr = grow < M ? bb.mean[grow] : 0.f;
        if (!(rr * bb.l2_eps < 1.0f)) dot = 0.f;           // ||z|| <= eps: the denominator was the constant eps
#pragma unroll
        for (int j = 0; j < 4; ++j) gm[j] = rr * (gm[j] - y4[j] * dot);
      }
      f32x4 gm2 = zero;
      if constexpr (MODE == 2) gm2 = ng2[p];
      if (relu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float act = has_bn ? fmaf(y4[j], sc[j], sh[j]) : y4[j];
          if (!(act > 0.f)) { gm[j] = 0.f; gm2[j] = 0.f; }
        }
      }
      if (drop)
        mmg_drop4(gm, pr.key, (uint64_t)(pr.row_offset + row0 + r) * (uint64_t)K + (uint64_t)c, pr.thr, pr.inv_keep);
      if constexpr (MODE == 2) {
        if (drop2)
          mmg_drop4(gm2, pr2.key, (uint64_t)(pr2.row_offset + row0 + r) * (uint64_t)K + (uint64_t)c, pr2.thr, pr2.inv_keep);
        gm += gm2;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float g = gm[j];
        if (has_bn) {
          const float xh = (y4[j] - mu[j]) * rs[j];
          g = sc[j] * (g - a0[j] - xh * a1[j]);
        }
        v[j] = g;
      }
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(xu32x4, v), zs, xvo, p * ROWS_PER_PASS * K * 4, 0);
      xbf16x4 q0, q1, q2;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const __bf16 a = (__bf16)v[j];
        const float r1 = v[j] - (float)a;
        const __bf16 b = (__bf16)r1;
        q0[j] = a; q1[j] = b; q2[j] = (__bf16)(r1 - (float)b);
      }
      *reinterpret_cast<xbf16x4*>(pb + (0 * BM + r) * LDP + kc4 * 4) = q0;
      *reinterpret_cast<xbf16x4*>(pb + (1 * BM + r) * LDP + kc4 * 4) = q1;
      *reinterpret_cast<xbf16x4*>(pb + (2 * BM + r) * LDP + kc4 * 4) = q2;
    }
  };
  if constexpr (MODE == 3) fetch_idx(t0);
  fetch(t0);
  stage(t0, 0);
  fetch(t0 + GY);
  __syncthreads();
  const int yvo = ((4 * h) * N + wn * 32 + l31) * 4;     // C/D map: col = lane&31, row = (i&3) + 8*(i>>2) + 4*(lane>>5)
  auto tile_body = [&](int64_t tt, int buf) __attribute__((always_inline)) {
    const int rows = rows_of(tt);
    const __amdgpu_buffer_rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc(
        DX + (size_t)(rows ? tt : 0) * BM * N, 0, rows * N * 4, 0x00020000);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    stage(tt + GY, buf ^ 1);                   // the other buffer: its readers passed the barrier of the last tile
    fetch(tt + 2 * GY);
    float nby[NBN ? 16 : 1];
    if constexpr (NBN) next_bn_load(nb, tt, rows, N, 0, yvo, nby);   // in flight under the products
    __builtin_amdgcn_sched_barrier(0);         // keep the fetch ahead of the matrix loop
    const __bf16* ap = planes + (size_t)buf * 3 * BM * LDP + l31 * LDP + 8 * h;
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
      const xbf16x8 a1f = *reinterpret_cast<const xbf16x8*>(ap + ks * 16);
      const xbf16x8 a2f = *reinterpret_cast<const xbf16x8*>(ap + BM * LDP + ks * 16);
      const xbf16x8 a3f = *reinterpret_cast<const xbf16x8*>(ap + 2 * BM * LDP + ks * 16);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3f, wb[ks][0], acc, 0, 0, 0);   // small terms first
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1f, wb[ks][2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2f, wb[ks][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2f, wb[ks][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1f, wb[ks][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1f, wb[ks][0], acc, 0, 0, 0);
    }
    float vv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      vv[i] = acc[i];                        // (a bit_cast straight from the vector element stored element 0 sixteen times)
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vv[i]), xs, yvo, ((i & 3) + 8 * (i >> 2)) * N * 4,
                                            MMG_NT_ST);
    }
    if constexpr (NBN) next_bn_tile(nb, nbc, vv, nby, rows, tt * BM, N, col, lane, cs1, cs2);
    __syncthreads();                         // buf fully read, buf^1 fully written
  };
  tile_body(t0, 0);
  tile_body(t0 + GY, 1);
  for (int i = 2; i < n_my; i += 2) {
    tile_body(t0 + (int64_t)i * GY, 0);
    tile_body(t0 + (int64_t)(i + 1) * GY, 1);
  }
  if constexpr (NBN) {
    // two partials per column (the lane halves) -> one: partial[row-set][2][N], fixed order
    double* red = reinterpret_cast<double*>(planes);          // the planes are dead: every wave passed the last barrier
    red[(h * 2 + 0) * N + col] = cs1;
    red[(h * 2 + 1) * N + col] = cs2;
    __syncthreads();
    for (int e = tid; e < 2 * N; e += NTHR) {
      const int which = e / N, cc = e % N;
      stat_partial[((size_t)by_u * 2 + which) * N + cc] = red[which * N + cc] + red[(2 + which) * N + cc];
    }
  }
}

inline int64_t bnbwd_x6_rows(int64_t M, int K) {       // grid.y of k_linear_bnbwd_x6 (= rows of its partial statistics)
  // one workgroup spans all N columns; two workgroups per CU whatever the width (register-bound), one at K = 256
  const int64_t n_tiles_ = (M + 31) / 32, want_ = K <= 128 ? 512 : 256;
  return want_ < n_tiles_ ? want_ : (n_tiles_ > 0 ? n_tiles_ : 1);
}

template <int K, int WN, int MODE = 0, bool NBN = false>
int launch_bnbwd_x6(const float* G, const BnBwdDev& bb, const ProDev& pr, const float* W, float* DX, int64_t M, hipStream_t st,
                    const ProDev& pr2 = mmg_pro_dev(nullptr), const NextBnDev& nb = next_bn_none(),
                    double* stat_partial = nullptr) {
  constexpr int N = 32 * WN;
  const int64_t gy = bnbwd_x6_rows(M, K);
  constexpr int lds = 2 * 3 * 32 * (K + 8) * 2;
  MMG_CHECK_HIP((MmgMaxLds<&k_linear_bnbwd_x6<K, WN, MODE, NBN>, lds>::set()), "linear_bnbwd(attr)");
  MMG_LAUNCH(MMG_PROBE_LINEAR_FWD, M, N, K, (MODE == 1 ? 64 : (MODE == 2 ? 16 | 128 : (MODE == 3 ? 16 | 256 : 16))) | (NBN ? 512 : 0),
             (k_linear_bnbwd_x6<K, WN, MODE, NBN>), dim3(1u, (unsigned)gy),
             dim3(64 * WN), lds, st, G, bb, pr, W, DX, M, pr2, nb, stat_partial);
  return 0;
}

// Small M (the vocab-side tables: 50..200 rows): one 256-thread workgroup per 32x32 output tile, the k axis split
// over its four waves (and over the lane halves inside a wave), every operand load issued up front straight from
// L2 into registers, the four partial tiles summed through LDS.  These launches are pure latency (a dependent chain
// load -> MFMAs -> store): a quarter of the chain per wave, one load round trip.
template <int K>
__global__ __launch_bounds__(256) void k_linear_small(const float* __restrict__ X, ProDev pr,
                                                      const float* __restrict__ W, const float* __restrict__ bias,
                                                      float* __restrict__ Y, int64_t M, int N, int flags) {
  pr.resolve();
  __shared__ float part[4][16][64];
  const int accumulate = flags & MMG_LIN_ACCUMULATE;
  const bool wkn = (flags & MMG_LIN_W_KN) != 0;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, h = lane >> 5, l31 = lane & 31;
  constexpr int KW = K / 8;                       // k values per (wave, lane half)
  const int kb = wid * (K / 4) + h * KW;
  const int n0 = blockIdx.x * 32;
  const int64_t row0 = (int64_t)blockIdx.y * 32;
  const int64_t ar = row0 + l31;
  const float* xp = X + (size_t)(ar < M ? ar : 0) * K + kb;
  const float* wp = wkn ? W + (size_t)kb * N + n0 + l31 : W + (size_t)(n0 + l31) * K + kb;
  f32x4 a[KW / 4], w[KW / 4];
#pragma unroll
  for (int q = 0; q < KW / 4; ++q) {
    a[q] = *reinterpret_cast<const f32x4*>(xp + q * 4);
    if (wkn) {
#pragma unroll
      for (int j = 0; j < 4; ++j) w[q][j] = wp[(size_t)(q * 4 + j) * N];
    } else {
      w[q] = *reinterpret_cast<const f32x4*>(wp + q * 4);
    }
  }
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
  for (int q = 0; q < KW / 4; ++q) {
    if (pr.scale || pr.relu || pr.p > 0.f) {
      const int k0 = kb + q * 4;
      f32x4 s4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f};
      if (pr.scale) { s4 = *reinterpret_cast<const f32x4*>(pr.scale + k0); sh4 = *reinterpret_cast<const f32x4*>(pr.shift + k0); }
      mmg_pro_apply4(pr, a[q], s4, sh4, ar, k0, K);
    }
    if (ar >= M) a[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][j], w[q][j], acc, 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) part[wid][i][lane] = acc[i];
  __syncthreads();
  // wave w finishes accumulator registers 4w .. 4w+3 (rows (i & 3) + 8 w + 4 h)
  const int col = n0 + l31;
  const float bv = bias ? bias[col] : 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int i = 4 * wid + e;
    const float v0 = part[0][i][lane] + part[1][i][lane] + part[2][i][lane] + part[3][i][lane];
    const int64_t gr = row0 + (i & 3) + 8 * (i >> 2) + 4 * h;
    if (gr < M) {
      float* dst = Y + (size_t)gr * N + col;
      float v = v0 + bv;
      if (accumulate) v += *dst;
      *dst = v;
    }
  }
}

// ---------------------------------------------------------------------------------- wgrad
constexpr int WG_ROWS = 32;   // rows reduced per LDS stage

// wgrad on the bf16 matrix cores: dW[n, k] = sum_m dY[m, n] * pro(X)[m, k] with the same exact 6-term split.  Both
// operands are contracted over their ROW index, so the fragments are column slices of the row-major tiles: the
// three bf16 planes of dY and X are staged row-major (coalesced 16-B loads, one split per element) and read back
// through the gfx950 transposing LDS read (ds_read_b64_tr_b16: a 16-lane group fetches 4 rows x 16 columns and
// each lane receives 4 consecutive rows of ITS column).  Row stride = tile bytes + 64: the 4 rows of a group then
// sit on 4 disjoint 16-bank spans (conflict-free).  32-row stages, double-buffered, one barrier per stage.
typedef short xs16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ xbf16x8 tr_frag(const __bf16* p, int row_stride) {   // rows +0..3 and +4..7 of this lane's column
  const xs16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) xs16x4*)p);
  const xs16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) xs16x4*)(p + 4 * row_stride));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(xbf16x8, v);
}

template <int TN, int TK, int WNN, int WNK, bool PRO>      // WNN x WNK waves over the [TN, TK] output tile
__global__ __launch_bounds__(64 * WNN * WNK) void k_linear_wgrad_x6(const float* __restrict__ dY, const float* __restrict__ X,
                                                         ProDev pr, float* __restrict__ slab, int64_t M, int N, int K,
                                                         int64_t rows_per_split, int direct_accumulate,
                                                         int64_t slab_stride, float* __restrict__ dbias) {
  if (PRO) pr.resolve();
  constexpr int NTHR = 64 * WNN * WNK;
  constexpr int MT = TN / (32 * WNN), KT = TK / (32 * WNK);      // 32x32 tiles per wave along n and k
  static_assert(MT >= 1 && KT >= 1, "tile too small for the wave grid");
  constexpr int SY = TN + 32, SX = TK + 32;      // plane row strides in bf16 (+64 B)
  constexpr int PY = 3 * WG_ROWS * SY, PX = 3 * WG_ROWS * SX;
  extern __shared__ __attribute__((aligned(16))) __bf16 wplanes[];   // [2][ dY: 3 x 32 x SY | X: 3 x 32 x SX ]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wn = wid / WNK, wk = wid % WNK;
  const int tiles_k = K / TK;
  int bx, by;
  xcd_tile_map(&bx, &by);                 // (output tile, stage set): the tiles that stream the same rows share an XCD's L2
  const int tn0 = (bx / tiles_k) * TN, tk0 = (bx % tiles_k) * TK;
  // 32-row stages are dealt round-robin: workgroup y takes stages y, y + G, y + 2G, ...  At any moment the grid reads
  // one contiguous window of dY and X (G x 16 KB each), which spreads over every HBM channel; a contiguous chunk per
  // workgroup makes G streams advance in lockstep a fixed (power-of-two-ish) stride apart and pile onto few channels.
  (void)rows_per_split;
  const int64_t total_st = (M + WG_ROWS - 1) / WG_ROWS;
  const int64_t G = gridDim.y;
  const int n_st = (int64_t)by < total_st ? (int)((total_st - by + G - 1) / G) : 0;

  f32x16 acc[MT][KT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < KT; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  // Memory pipeline: a ring of RING 32-row stages in registers on top of the double-buffered LDS planes (the stage
  // time is far below the HBM latency: with one stage in flight the kernel ran at the pace of one round trip per 32
  // rows).  All loads go through per-stage buffer descriptors (rows past the end read 0, stages past the end have a
  // zero-sized descriptor), so the loop has no branch around a load and the vmcnt waits are exact.
  constexpr int RING = 4;
  constexpr int NY = WG_ROWS * (TN / 4) / NTHR, NX = WG_ROWS * (TK / 4) / NTHR;   // 16-B loads per thread per stage
  static_assert(NY >= 1 && NX >= 1, "stage smaller than the workgroup");
  static_assert(NTHR % (TN / 4) == 0 && NTHR % (TK / 4) == 0, "a thread keeps its column quad across passes");
  constexpr int RY = NTHR / (TN / 4), RX = NTHR / (TK / 4);      // rows covered per pass
  const int yr = tid / (TN / 4), yc4 = tid % (TN / 4), xr = tid / (TK / 4), xc4 = tid % (TK / 4);
  const int yvo = (yr * N + yc4 * 4) * 4, xvo = (xr * K + xc4 * 4) * 4;
  f32x4 ny[RING][NY], nxr[RING][NX];
  auto fetch = [&](int st, f32x4* fy, f32x4* fx) {
    const int64_t r0 = ((int64_t)by + (int64_t)st * G) * WG_ROWS;
    const int64_t left = M - r0;
    const int rows = st < n_st ? (left < WG_ROWS ? (int)left : WG_ROWS) : 0;
    const int64_t rb = rows ? r0 : 0;
    const __amdgpu_buffer_rsrc_t ys = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(dY) + (size_t)rb * N + tn0, 0, rows ? (rows * N - tn0) * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(X) + (size_t)rb * K + tk0, 0, rows ? (rows * K - tk0) * 4 : 0, 0x00020000);
#pragma unroll
    for (int u = 0; u < NY; ++u)
      fy[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ys, yvo, u * RY * N * 4, MMG_NT_WG));
#pragma unroll
    for (int u = 0; u < NX; ++u)
      fx[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xs, xvo, u * RX * K * 4, MMG_NT_WG));
  };
  auto split_store = [&](f32x4 v, __bf16* plane0, int plane_elems, int off) {
    xbf16x4 q0, q1, q2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const __bf16 a = (__bf16)v[j];
      const float r1 = v[j] - (float)a;
      const __bf16 b = (__bf16)r1;
      q0[j] = a; q1[j] = b; q2[j] = (__bf16)(r1 - (float)b);
    }
    *reinterpret_cast<xbf16x4*>(plane0 + off) = q0;
    *reinterpret_cast<xbf16x4*>(plane0 + plane_elems + off) = q1;
    *reinterpret_cast<xbf16x4*>(plane0 + 2 * plane_elems + off) = q2;
  };
  // prologue constants of this thread's column quad (identity when a part is absent: x*1+0, max(x,-inf))
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  float floor_v = -__builtin_inff();
  bool drop = false;
  if (PRO) {
    if (pr.scale) {
      sc = *reinterpret_cast<const f32x4*>(pr.scale + tk0 + xc4 * 4);
      sh = *reinterpret_cast<const f32x4*>(pr.shift + tk0 + xc4 * 4);
    }
    if (pr.relu) floor_v = 0.f;
    drop = pr.p > 0.f;
  }
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  auto stage = [&](int st, int buf, const f32x4* fy, const f32x4* fx) {
    __bf16* yb = wplanes + (size_t)buf * (PY + PX);
    __bf16* xb = yb + PY;
    const int64_t r0 = ((int64_t)by + (int64_t)st * G) * WG_ROWS;
#pragma unroll
    for (int u = 0; u < NY; ++u) {
      const f32x4 v = fy[u];                     // rows past the end arrive as zeros
      bsum += v;                                 // column sums of dY (the bias gradient) ride along: the quad is fixed per thread
      split_store(v, yb, WG_ROWS * SY, (yr + u * RY) * SY + yc4 * 4);
    }
#pragma unroll
    for (int u = 0; u < NX; ++u) {
      f32x4 v = fx[u];
      if (PRO) {
        const int64_t gr = r0 + xr + u * RX;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(fmaf(v[j], sc[j], sh[j]), floor_v);
        if (drop)
          mmg_drop4(v, pr.key, (uint64_t)(pr.row_offset + gr) * (uint64_t)K + (uint64_t)(tk0 + xc4 * 4), pr.thr, pr.inv_keep);
        // no zeroing of the rows past the end: their dY rows are zero, so they contribute 0 to every product
      }
      split_store(v, xb, WG_ROWS * SX, (xr + u * RX) * SX + xc4 * 4);
    }
  };
  // transposing-read lane roles: group g = lane >> 4 fetches rows 8 (g >> 1) + q, columns 16 (g & 1) + 4 p
  const int g = lane >> 4, q = (lane >> 2) & 3, p4 = lane & 3;
  const int tr_row = 8 * (g >> 1) + q, tr_col = 16 * (g & 1) + 4 * p4;
  const int h = lane >> 5, l31 = lane & 31;

  // stage s lives in ring slot s % RING and LDS buffer s & 1
#pragma unroll
  for (int j = 0; j < RING; ++j) fetch(j, ny[j], nxr[j]);
  stage(0, 0, ny[0], nxr[0]);
  fetch(RING, ny[0], nxr[0]);
  __syncthreads();
  auto body = [&](int st, int j) {               // j = st % RING (compile-time after unrolling), RING is even: buf = j & 1
    constexpr int dummy = 0; (void)dummy;
    const int jn = (j + 1) % RING;
    stage(st + 1, (j + 1) & 1, ny[jn], nxr[jn]);
    fetch(st + 1 + RING, ny[jn], nxr[jn]);
    __builtin_amdgcn_sched_barrier(0);           // keep the fetch ahead of the matrix loop
    const int buf = j & 1;
    const __bf16* yb = wplanes + (size_t)buf * (PY + PX) + tr_row * SY + wn * (TN / WNN) + tr_col;
    const __bf16* xb = wplanes + (size_t)buf * (PY + PX) + PY + tr_row * SX + wk * (TK / WNK) + tr_col;
#pragma unroll
    for (int ks = 0; ks < WG_ROWS / 16; ++ks) {
      xbf16x8 a[MT][3], b[KT][3];
#pragma unroll
      for (int x = 0; x < MT; ++x)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) a[x][pc] = tr_frag(yb + pc * WG_ROWS * SY + ks * 16 * SY + x * 32, SY);
#pragma unroll
      for (int x = 0; x < KT; ++x)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) b[x][pc] = tr_frag(xb + pc * WG_ROWS * SX + ks * 16 * SX + x * 32, SX);
#pragma unroll
      for (int x = 0; x < MT; ++x)
#pragma unroll
        for (int y = 0; y < KT; ++y) {
          acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[x][2], b[y][0], acc[x][y], 0, 0, 0);   // small terms first
          acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[x][0], b[y][2], acc[x][y], 0, 0, 0);
          acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[x][1], b[y][1], acc[x][y], 0, 0, 0);
          acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[x][1], b[y][0], acc[x][y], 0, 0, 0);
          acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[x][0], b[y][1], acc[x][y], 0, 0, 0);
          acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[x][0], b[y][0], acc[x][y], 0, 0, 0);
        }
    }
    __syncthreads();
  };
  static_assert(RING % 2 == 0, "the LDS buffer parity must follow the ring slot");
  for (int s = 0; s < n_st; s += RING) {         // stages past the end multiply zeros (at most RING - 1 per workgroup)
#pragma unroll
    for (int j = 0; j < RING; ++j) body(s + j, j);
  }
  // slab[split][N*K (+N bias sums)]; with a single split `slab` is dW itself (direct_accumulate: 1 = overwrite, 2 = add)
  float* dst = slab + (size_t)by * slab_stride;
#pragma unroll
  for (int x = 0; x < MT; ++x)
#pragma unroll
    for (int y = 0; y < KT; ++y)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int n = tn0 + wn * (TN / WNN) + x * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        const int k = tk0 + wk * (TK / WNK) + y * 32 + l31;
        float v = acc[x][y][i];
        if (direct_accumulate == 2) v += dst[(size_t)n * K + k];
        dst[(size_t)n * K + k] = v;
      }
  if (dbias && tk0 == 0) {                       // one k-tile column of workgroups owns the bias sums of its TN columns
    float* red = reinterpret_cast<float*>(wplanes);          // the planes are dead: every wave passed the last barrier
    constexpr int GROUPS = NTHR / (TN / 4);
    const int c4 = tid % (TN / 4), grp = tid / (TN / 4);
    *reinterpret_cast<f32x4*>(red + grp * TN + c4 * 4) = bsum;
    __syncthreads();
    if (tid < TN) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < GROUPS; ++q) t += red[q * TN + tid];
      float* bdst = direct_accumulate ? dbias : dst + (size_t)N * K;
      if (direct_accumulate == 2) t += bdst[tn0 + tid];
      bdst[tn0 + tid] = t;
    }
  }
}

// Role-specialised variant for the plain (no prologue) 128 x 128 tile.  Waves 0..3 only multiply (a 64 x 64 quadrant
// each, one wave per SIMD), waves 4..7 only stage (loads -> 3-way split -> LDS for stage s+1 while the multipliers work on
// stage s).  The stagers transpose in registers: a stager thread owns an 8-row x 4-column block (8 coalesced 16-B loads),
// splits it and writes, per piece and column, the 8 rows as ONE 16-B entry of a fragment-ordered plane
// [operand][piece][8-row group][column (swizzled)][8 rows]; a multiplier lane fetches a whole MFMA operand fragment with
// one conflict-free ds_read_b128: 24 reads per 48 MFMAs instead of 96 ds_read_b64_tr_b16 (which move 64 B/clk).
// Ablations at [183400 x 128]^T [183400 x 128] (us): empty skeleton 12, loads only 31-33, stagers only 35, multipliers
// only 33 (1075 MFMAs per wave: 42 clocks each against 32 issue-bound), staging + multiplying without loads 46, all 50.
// Staging and multiplying still do not overlap fully (they share the VALU issue port and the LDS), and the fixed
// part (launch, ring priming, slab write, tail) is a quarter of the kernel; the HBM floor of the 188 MB is ~31 us.
__global__ __launch_bounds__(512) void k_linear_wgrad_ws(const float* __restrict__ dY, const float* __restrict__ X,
                                                         float* __restrict__ slab, int64_t M, int N, int K,
                                                         int direct_accumulate, int64_t slab_stride,
                                                         float* __restrict__ dbias) {
  constexpr int TN = 128, TK = 128;
  constexpr int GRP = 128 * 8;                   // one (operand, piece, row group): 128 columns x 8 rows (bf16 elements)
  constexpr int BUF = 2 * 3 * 4 * GRP;           // one stage: 2 operands x 3 pieces x 4 row groups = 48 KB
  extern __shared__ __attribute__((aligned(16))) __bf16 wplanes[];   // [2 buffers][BUF]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_k = K / TK;
  int bx, by;
  xcd_tile_map(&bx, &by);                 // (output tile, stage set): see k_linear_wgrad_x6
  const int tn0 = (bx / tiles_k) * TN, tk0 = (bx % tiles_k) * TK;
  const int64_t total_st = (M + WG_ROWS - 1) / WG_ROWS;
  const int64_t G = gridDim.y;                   // stages are dealt round-robin (see k_linear_wgrad_x6)
  const int n_st = (int64_t)by < total_st ? (int)((total_st - by + G - 1) / G) : 0;
  constexpr int RING = 4;
  const int n_pad = (n_st + RING - 1) / RING * RING;
  float* dst = slab + (size_t)by * slab_stride;

  if (wid >= 4) {
    // ------------------------------------------------------------------ stagers: waves 4,5 stage dY, waves 6,7 stage X
    const int op = wid >> 1 & 1;                             // wave-uniform
    const int t = tid & 127, g = t >> 5, c4 = t & 31;        // 8-row group, column quad
    const float* src = op ? X : dY;
    const int ld = op ? K : N, c0 = op ? tk0 : tn0;
    const int vo = (8 * g * ld + c4 * 4) * 4;
    f32x4 ring[RING][8];
    auto fetch = [&](int st, f32x4* f) {
      const int64_t r0 = ((int64_t)by + (int64_t)st * G) * WG_ROWS;
      const int64_t left = M - r0;
      const int rows = st < n_st ? (left < WG_ROWS ? (int)left : WG_ROWS) : 0;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(src) + (size_t)(rows ? r0 : 0) * ld + c0, 0, rows ? (rows * ld - c0) * 4 : 0, 0x00020000);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        f[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, j * ld * 4, MMG_NT_WG));
    };
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    auto stage = [&](int st, int buf, const f32x4* f) {
      xbf16x8 q[3][4];                             // [piece][column of the quad] = 8 rows
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const f32x4 v = f[j];                      // rows past the end arrive as zeros
        bsum += v;                                 // column sums of dY (the bias gradient); ignored by the X waves
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const __bf16 a = (__bf16)v[e];
          const float r1 = v[e] - (float)a;
          const __bf16 b = (__bf16)r1;
          q[0][e][j] = a; q[1][e][j] = b; q[2][e][j] = (__bf16)(r1 - (float)b);
        }
      }
      // column c of a group sits at entry (c & ~15) | ((c / 4 + 4 (c % 4)) & 15): 16 consecutive lanes then touch 16
      // distinct 16-B bank groups both here (lane = column quad, fixed e) and in the fragment reads (lane = column)
      __bf16* base = wplanes + (size_t)buf * BUF + (size_t)(op * 3 * 4 + g) * GRP + (c4 >> 2) * 128;
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          *reinterpret_cast<xbf16x8*>(base + pc * 4 * GRP + ((c4 + 4 * e) & 15) * 8) = q[pc][e];
    };
#pragma unroll
    for (int j = 0; j < RING; ++j) fetch(j, ring[j]);
    stage(0, 0, ring[0]);
    fetch(RING, ring[0]);
    __syncthreads();
    for (int s = 0; s < n_pad; s += RING) {
#pragma unroll
      for (int j = 0; j < RING; ++j) {             // stage s+j+1 lives in ring slot (j+1) % RING, LDS buffer (j+1) & 1
        constexpr int dummy = 0; (void)dummy;
        const int jn = (j + 1) % RING;
        stage(s + j + 1, (j + 1) & 1, ring[jn]);
        __builtin_amdgcn_sched_barrier(0);
        fetch(s + j + 1 + RING, ring[jn]);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
      }
    }
    const bool own_bias = dbias && tk0 == 0;       // one k-tile column of workgroups owns the bias sums of its TN columns
    if (own_bias && !op) {
      float* red = reinterpret_cast<float*>(wplanes);        // the planes are dead: every wave passed the last barrier
      *reinterpret_cast<f32x4*>(red + g * TN + c4 * 4) = bsum;
    }
    __syncthreads();
    if (own_bias && !op) {
      const float* red = reinterpret_cast<const float*>(wplanes);
      const float v = red[t] + red[TN + t] + red[2 * TN + t] + red[3 * TN + t];
      float* bdst = direct_accumulate ? dbias : dst + (size_t)N * K;
      bdst[tn0 + t] = direct_accumulate == 2 ? bdst[tn0 + t] + v : v;
    }
  } else {
    // ------------------------------------------------------------------ multipliers: wave w owns quadrant (w >> 1, w & 1)
    const int wn = wid >> 1, wk = wid & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    const int h = lane >> 5, l31 = lane & 31;
    auto entry = [](int c) { return (c & ~15) | (((c >> 2) + 4 * (c & 3)) & 15); };     // the stagers' column swizzle
    int ya[2], xa[2];                              // dY / X entries: row group 2 ks + h, column x * 32 + l31 of the quadrant
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      ya[x] = h * GRP + entry(wn * 64 + x * 32 + l31) * 8;
      xa[x] = 3 * 4 * GRP + h * GRP + entry(wk * 64 + x * 32 + l31) * 8;
    }
    __syncthreads();
    for (int s = 0; s < n_pad; s += 2) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (s + j < n_st) {
          const __bf16* pb = wplanes + (size_t)j * BUF;
#pragma unroll
          for (int ks = 0; ks < WG_ROWS / 16; ++ks) {
            xbf16x8 a[2][3], b[2][3];
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
              for (int pc = 0; pc < 3; ++pc) {
                a[x][pc] = *reinterpret_cast<const xbf16x8*>(pb + ya[x] + (pc * 4 + 2 * ks) * GRP);
                b[x][pc] = *reinterpret_cast<const xbf16x8*>(pb + xa[x] + (pc * 4 + 2 * ks) * GRP);
              }
            // small terms first; consecutive MFMAs go to different accumulators
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int tm = 0; tm < 6; ++tm)
#pragma unroll
              for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y)
                  acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[x][PA[tm]], b[y][PB[tm]], acc[x][y], 0, 0, 0);
          }
        }
        __syncthreads();
      }
    }
    // slab[split][N*K (+N bias sums)]; with a single split `slab` is dW itself (direct_accumulate: 1 = overwrite, 2 = add)
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int n = tn0 + wn * 64 + x * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
          const int k = tk0 + wk * 64 + y * 32 + l31;
          float v = acc[x][y][i];
          if (direct_accumulate == 2) v += dst[(size_t)n * K + k];
          dst[(size_t)n * K + k] = v;
        }
    __syncthreads();                               // pairs with the stagers' bias-sum barrier
  }
}

int launch_wgrad_ws(dim3 grid, hipStream_t st, const float* dY, const float* X, float* target, int64_t M, int N, int K,
                    int direct, int64_t slab_stride, float* dbias) {
  constexpr int lds = 2 * 2 * 3 * 4 * 128 * 8 * 2;             // two stages of fragment-ordered planes: 96 KB
  MMG_CHECK_HIP((MmgMaxLds<&k_linear_wgrad_ws, lds>::set()), "linear_wgrad(attr)");
  MMG_LAUNCH(MMG_PROBE_LINEAR_WGRAD, M, N, K, 0, k_linear_wgrad_ws, grid, dim3(512), lds, st, dY, X, target, M, N, K,
             direct, slab_stride, dbias);
  return MMG_OK;
}

template <int TN, int TK, int WNN, int WNK, bool PRO>
int launch_wgrad_x6_v(dim3 grid, hipStream_t st, const float* dY, const float* X, const ProDev& pr, float* target, int64_t M,
                      int N, int K, int64_t rps, int direct, int64_t slab_stride, float* dbias) {
  constexpr int lds = 2 * 3 * WG_ROWS * ((TN + 32) + (TK + 32)) * 2;
  MMG_CHECK_HIP((MmgMaxLds<&k_linear_wgrad_x6<TN, TK, WNN, WNK, PRO>, lds>::set()), "linear_wgrad(attr)");
  MMG_LAUNCH(MMG_PROBE_LINEAR_WGRAD, M, N, K, PRO ? 4 : 0, (k_linear_wgrad_x6<TN, TK, WNN, WNK, PRO>), grid,
             dim3(64 * WNN * WNK), lds, st, dY, X, pr, target, M, N, K, rps, direct, slab_stride, dbias);
  return MMG_OK;
}

template <int TN, int TK, int WNN, int WNK>
int launch_wgrad_x6(dim3 grid, hipStream_t st, const float* dY, const float* X, const ProDev& pr, float* target, int64_t M,
                    int N, int K, int64_t rps, int direct, int64_t slab_stride, float* dbias) {
  if (pr.scale || pr.relu || pr.p > 0.f)
    return launch_wgrad_x6_v<TN, TK, WNN, WNK, true>(grid, st, dY, X, pr, target, M, N, K, rps, direct, slab_stride, dbias);
  return launch_wgrad_x6_v<TN, TK, WNN, WNK, false>(grid, st, dY, X, pr, target, M, N, K, rps, direct, slab_stride, dbias);
}

struct EpiStore {
  float* out; int accumulate;
  float* out2; int64_t n4_first;          // elements past n4_first float4s go to out2 (the bias sums behind a dW slab)
  __device__ void operator()(int64_t i4, mmg_f4 v) const {
    mmg_f4* o = i4 < n4_first ? reinterpret_cast<mmg_f4*>(out) + i4 : reinterpret_cast<mmg_f4*>(out2) + (i4 - n4_first);
    *o = accumulate ? (*o + v) : v;
  }
};

struct WgradPlan { int TN, TK, n_tiles, n_split; int64_t rows_per_split; };
WgradPlan plan_wgrad(int64_t M, int N, int K) {
  WgradPlan p;
  p.TN = (N % 128 == 0) ? 128 : 64;
  p.TK = (K % 128 == 0) ? 128 : 64;
  p.n_tiles = (N / p.TN) * (K / p.TK);
  int64_t max_split = (M + WG_ROWS - 1) / WG_ROWS;
  if (max_split < 1) max_split = 1;
  int64_t want = 256 / p.n_tiles;         // one workgroup per CU
  if (want < 1) want = 1;
  p.n_split = (int)(want < max_split ? want : max_split);
  if (M <= 256) p.n_split = 1;            // vocab-side tables: one workgroup per output tile, direct write
  int64_t rps = (M + p.n_split - 1) / p.n_split;
  rps = (rps + WG_ROWS - 1) / WG_ROWS * WG_ROWS;
  if (rps < WG_ROWS) rps = WG_ROWS;
  p.rows_per_split = rps;
  p.n_split = (int)((M + rps - 1) / rps);
  if (p.n_split < 1) p.n_split = 1;
  return p;
}

template <int K>
int launch_small(const float* X, const ProDev& pr, const float* W, const float* bias, float* Y, int64_t M, int N,
                 int accumulate, hipStream_t st) {
  hipLaunchKernelGGL((k_linear_small<K>), dim3((unsigned)(N / 32), (unsigned)((M + 31) / 32)), dim3(256), 0, st, X, pr, W,
                     bias, Y, M, N, accumulate);
  return 0;
}

}  // namespace

extern "C" int mmg_col_reduce2(const float* A, const float* B, double* out, int64_t M, int N, void* ws, size_t ws_bytes,
                               void* stream);
extern "C" int mmg_partial_sum(const double* partial, double* out, int n, int n_rows, void* stream);
extern "C" int mmg_partial_sum_bn(const double* partial, double* col_sums, int N, int n_rows, const mmg_bn_fin_t* fin, void* stream);
extern "C" int mmg_bn_finalize(const double* sums, int64_t count, const float* gamma, const float* beta, float* running_mean,
                               float* running_var, int training, int n_updates, float momentum, float eps, float* scale,
                               float* shift, float* mean, float* rstd, int N, void* stream);

// ---- mmg_next_bn_t (include/mmgnn.h): host side shared by the producers (this file and aggregate.hip)
extern "C" int mmg_partial_sum_add(const double* partial, double* out, int n, int n_rows, const double* add, void* stream);
extern "C" int mmg_bn_bwd_stats(const float* G, const float* Y, const mmg_prologue_t* pro, const float* mean, const float* rstd,
                                double* sums, int64_t M, int N, void* ws, size_t ws_bytes, void* stream);

// workspace: <= 768 partial rows of a fused producer, or (fallback) one row of sums + the separate statistics pass
extern "C" size_t mmg_next_bn_ws_bytes(int64_t M, int N) {
  if (M < 0 || N <= 0) return 0;
  const size_t a = (size_t)768 * 2 * N * sizeof(double) + 256;
  const size_t b = (size_t)2 * N * sizeof(double) + 512 + mmg_col_reduce2_ws_bytes(M, N);
  return a > b ? a : b;
}

// internal: validates the descriptor and fills the device view; *partial = the aligned workspace
extern "C" int mmg_next_bn_dev(const mmg_next_bn_t* next, int64_t M, int N, const char* what, NextBnDev* d, double** partial) {
  MMG_CHECK_ARG(next->y && next->pro && next->mean && next->rstd && next->sums, "%s: next-BatchNorm descriptor with a null field", what);
  MMG_CHECK_ARG(!next->pro->scale || next->pro->shift, "%s: next BatchNorm: scale without shift", what);
  MMG_CHECK_ARG(next->ws && next->ws_bytes >= mmg_next_bn_ws_bytes(M, N), "%s: next BatchNorm: workspace too small", what);
  d->Y = next->y; d->mean = next->mean; d->rstd = next->rstd;
  d->pr = mmg_pro_dev(next->pro);
  *partial = (double*)(((uintptr_t)next->ws + 255) & ~(uintptr_t)255);
  return MMG_OK;
}
// internal: partial[rows][2][N] -> sums (or sums += when the descriptor says so), fixed order
extern "C" int mmg_next_bn_finish(const mmg_next_bn_t* next, const double* partial, int N, int rows, void* stream) {
  return mmg_partial_sum_add(partial, next->sums, 2 * N, rows, next->accumulate ? next->sums : nullptr, stream);
}
// internal: the producer has no fused form for this shape -- the separate statistics pass over its finished output G
extern "C" int mmg_next_bn_fallback(const float* G, int64_t M, int N, const mmg_next_bn_t* next, const char* what, void* stream) {
  NextBnDev d;
  double* tmp;
  int rc = mmg_next_bn_dev(next, M, N, what, &d, &tmp);
  if (rc) return rc;
  unsigned char* rest = (unsigned char*)(tmp + 2 * N);
  const size_t used = (size_t)(rest - (unsigned char*)next->ws);
  rc = mmg_bn_bwd_stats(G, next->y, next->pro, next->mean, next->rstd, next->accumulate ? tmp : next->sums, M, N, rest,
                        next->ws_bytes - used, stream);
  if (rc || !next->accumulate) return rc;
  return mmg_partial_sum_add(tmp, next->sums, 2 * N, 1, next->sums, stream);
}
// relu / none only in the fused epilogues (the fallback takes every activation mmg_bn_bwd_stats does)
static inline bool next_bn_fusable(const mmg_next_bn_t* next) {
  return next->pro && (next->pro->relu == MMG_ACT_NONE || next->pro->relu == MMG_ACT_RELU);
}

extern "C" size_t mmg_linear_fwd_stats_ws_bytes(int64_t M, int N) {
  if (M < 0 || N <= 0) return 0;
  const size_t a = (size_t)768 * 2 * N * sizeof(double) + 256;       // <= 768 partial rows from the GEMM epilogue
  const size_t b = mmg_col_reduce2_ws_bytes(M, N);                   // fallback: a separate column reduction
  return a > b ? a : b;
}

extern "C" int mmg_linear_fwd_stats(const float* X, const mmg_prologue_t* pro, const float* W, const float* bias,
                                    float* Y, int64_t M, int N, int K, int flags, double* col_sums, void* ws,
                                    size_t ws_bytes, void* stream);

extern "C" int mmg_linear_fwd(const float* X, const mmg_prologue_t* pro, const float* W, const float* bias, float* Y,
                              int64_t M, int N, int K, int flags, void* stream) {
  return mmg_linear_fwd_stats(X, pro, W, bias, Y, M, N, K, flags, nullptr, nullptr, 0, stream);
}

static int linear_fwd_stats_impl(const float* X, const mmg_prologue_t* pro, const float* W, const float* bias,
                                 float* Y, int64_t M, int N, int K, int flags, double* col_sums, void* ws,
                                 size_t ws_bytes, void* stream, const mmg_bn_fin_t* fin) {
  const int accumulate = flags;           // the launchers forward the whole flag word
  if (col_sums) {
    MMG_CHECK_ARG(ws && ws_bytes >= mmg_linear_fwd_stats_ws_bytes(M, N), "linear_fwd_stats: workspace too small");
    MMG_CHECK_ARG(M > 0, "linear_fwd_stats: M must be positive");
  }
  double* partial = col_sums ? (double*)(((uintptr_t)ws + 255) & ~(uintptr_t)255) : nullptr;
  bool stats_done = false;
  MMG_CHECK_ARG(M >= 0, "linear_fwd: M < 0");
  MMG_CHECK_ARG((flags & ~(MMG_LIN_ACCUMULATE | MMG_LIN_W_KN)) == 0, "linear_fwd: unknown flag bits");
  MMG_CHECK_ARG(K == 64 || K == 128 || K == 256, "linear_fwd: K=%d unsupported (64|128|256)", K);
  MMG_CHECK_ARG(N > 0 && N % 64 == 0 && N <= 4096, "linear_fwd: N=%d must be a multiple of 64", N);
  if (M == 0) return MMG_OK;
  MMG_CHECK_ARG(X && W && Y, "linear_fwd: null buffer");
  MMG_CHECK_ARG(!pro || !pro->scale || pro->shift, "linear_fwd: prologue scale without shift");
  MMG_CHECK_ARG(!pro || pro->relu == MMG_ACT_NONE || pro->relu == MMG_ACT_RELU, "linear_fwd: the prologue takes relu only");
  hipStream_t st = (hipStream_t)stream;
  const ProDev pr = mmg_pro_dev(pro);
  if (M <= 512) {          // vocab-side tables
    if (K == 64) launch_small<64>(X, pr, W, bias, Y, M, N, accumulate, st);
    else if (K == 128) launch_small<128>(X, pr, W, bias, Y, M, N, accumulate, st);
    else launch_small<256>(X, pr, W, bias, Y, M, N, accumulate, st);
  } else if (K <= 128 || N % 128 == 0) {
    // exact-product 6-term bf16 split on the bf16 matrix cores
    int rc = 0;
    if (K == 64) {
      if (N % 128 == 0) rc = launch_fwd_x6<64, 4>(X, pr, W, bias, Y, M, N, accumulate, st, partial);
      else rc = launch_fwd_x6<64, 2>(X, pr, W, bias, Y, M, N, accumulate, st, partial);
    } else if (K == 128) {
      if (N % 128 == 0) rc = launch_fwd_x6<128, 4>(X, pr, W, bias, Y, M, N, accumulate, st, partial);
      else rc = launch_fwd_x6<128, 2>(X, pr, W, bias, Y, M, N, accumulate, st, partial);
    } else if (N % 256 == 0) {
      rc = launch_fwd_h3_k256(X, pr, W, bias, Y, M, N, accumulate, st, partial);     // one workgroup spans 256 columns
    } else {
      rc = launch_fwd_x6<256, 4>(X, pr, W, bias, Y, M, N, accumulate, st, partial);
    }
    if (rc) return rc;
    if (col_sums) {     // partial[gy][2][N] -> col_sums[2][N]
      const int rows = (K == 256 && N % 256 == 0) ? (int)fwd_k256_rows(M, N)
                                                  : (int)fwd_x6_rows(M, N, N % 128 == 0 ? 128 : 64, K);
      int rc2 = mmg_partial_sum_bn(partial, col_sums, N, rows, fin, stream);     // (+ the BatchNorm fold when asked for)
      if (rc2) return rc2;
      stats_done = true;
    }
  } else {
    // K = 256 with a 64-wide output (the heads' first layer at 256-d)
    int rc = launch_fwd_x6<256, 2>(X, pr, W, bias, Y, M, N, accumulate, st, partial);
    if (rc) return rc;
    if (col_sums) {
      int rc2 = mmg_partial_sum_bn(partial, col_sums, N, (int)fwd_x6_rows(M, N, 64, K), fin, stream);
      if (rc2) return rc2;
      stats_done = true;
    }
  }
  MMG_CHECK_LAUNCH("linear_fwd");
  if (col_sums && !stats_done) {          // small-M / fp32 kernels: a separate pass over Y
    int rc3 = mmg_col_reduce2(Y, nullptr, col_sums, M, N, ws, ws_bytes, stream);
    if (rc3 || !fin) return rc3;
    return mmg_bn_finalize(col_sums, fin->count, fin->gamma, fin->beta, fin->running_mean, fin->running_var, 1, fin->n_updates,
                           fin->momentum, fin->eps, fin->scale, fin->shift, fin->mean, fin->rstd, N, stream);
  }
  return MMG_OK;
}

extern "C" int mmg_linear_fwd_stats(const float* X, const mmg_prologue_t* pro, const float* W, const float* bias,
                                    float* Y, int64_t M, int N, int K, int flags, double* col_sums, void* ws,
                                    size_t ws_bytes, void* stream) {
  return linear_fwd_stats_impl(X, pro, W, bias, Y, M, N, K, flags, col_sums, ws, ws_bytes, stream, nullptr);
}

extern "C" int mmg_linear_fwd_stats_bn(const float* X, const mmg_prologue_t* pro, const float* W, const float* bias,
                                       float* Y, int64_t M, int N, int K, int flags, double* col_sums, void* ws,
                                       size_t ws_bytes, const mmg_bn_fin_t* fin, void* stream) {
  MMG_CHECK_ARG(col_sums && fin && fin->count > 0 && fin->scale && fin->shift, "linear_fwd_stats_bn: col_sums and a fold descriptor are required");
  return linear_fwd_stats_impl(X, pro, W, bias, Y, M, N, K, flags, col_sums, ws, ws_bytes, stream, fin);
}

extern "C" int mmg_linear_fwd_next_bn(const float* X, const mmg_prologue_t* pro, const float* W, const float* bias, float* Y,
                                      int64_t M, int N, int K, int flags, const mmg_next_bn_t* next, void* stream) {
  if (!next) return mmg_linear_fwd(X, pro, W, bias, Y, M, N, K, flags, stream);
  const ProDev pr = mmg_pro_dev(pro);
  const bool fused = M > 512 && N % 128 == 0 && N <= 4096 && (K == 64 || K == 128) && (flags & ~MMG_LIN_W_KN) == 0 &&
                     !(pr.scale || pr.relu || pr.p > 0.f) && next_bn_fusable(next);
  if (!fused) {
    int rc0 = mmg_linear_fwd(X, pro, W, bias, Y, M, N, K, flags, stream);
    return rc0 ? rc0 : mmg_next_bn_fallback(Y, M, N, next, "linear_fwd_next_bn", stream);
  }
  MMG_CHECK_ARG(X && W && Y, "linear_fwd_next_bn: null buffer");
  NextBnDev nb;
  double* partial;
  int rc = mmg_next_bn_dev(next, M, N, "linear_fwd_next_bn", &nb, &partial);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (K == 64) rc = launch_fwd_x6_v<64, 4, false, false, false, true>(X, pr, W, bias, Y, M, N, flags, st, partial, nullptr, 0.f, nb);
  else rc = launch_fwd_x6_v<128, 4, false, false, false, true>(X, pr, W, bias, Y, M, N, flags, st, partial, nullptr, 0.f, nb);
  if (rc) return rc;
  MMG_CHECK_LAUNCH("linear_fwd_next_bn");
  return mmg_next_bn_finish(next, partial, N, (int)fwd_x6_rows(M, N, 128, K), stream);
}

extern "C" int mmg_linear_fwd_l2norm_supported(int64_t M, int N, int K) {
  return (M > 512 && ((N == 128 && (K == 64 || K == 128)) || (N == 64 && (K == 64 || K == 128)))) ? 1 : 0;
}

extern "C" int mmg_linear_fwd_l2norm(const float* X, const mmg_prologue_t* pro, const float* W, const float* bias, float* Y,
                                     float* rnorm, int64_t M, int N, int K, float eps, void* stream) {
  MMG_CHECK_ARG(mmg_linear_fwd_l2norm_supported(M, N, K), "linear_fwd_l2norm: M=%lld N=%d K=%d unsupported (M > 512, N and K in {64,128})",
                (long long)M, N, K);
  MMG_CHECK_ARG(X && W && Y && rnorm, "linear_fwd_l2norm: null buffer");
  MMG_CHECK_ARG(!pro || !pro->scale || pro->shift, "linear_fwd_l2norm: prologue scale without shift");
  MMG_CHECK_ARG(!pro || pro->relu == MMG_ACT_NONE || pro->relu == MMG_ACT_RELU, "linear_fwd_l2norm: the prologue takes relu only");
  const ProDev pr = mmg_pro_dev(pro);
  const bool has_pro = pr.scale || pr.relu || pr.p > 0.f;
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (N == 128) {
    if (K == 128) rc = has_pro ? launch_fwd_x6_v<128, 4, true, false, true>(X, pr, W, bias, Y, M, N, 0, st, nullptr, rnorm, eps)
                               : launch_fwd_x6_v<128, 4, false, false, true>(X, pr, W, bias, Y, M, N, 0, st, nullptr, rnorm, eps);
    else rc = has_pro ? launch_fwd_x6_v<64, 4, true, false, true>(X, pr, W, bias, Y, M, N, 0, st, nullptr, rnorm, eps)
                      : launch_fwd_x6_v<64, 4, false, false, true>(X, pr, W, bias, Y, M, N, 0, st, nullptr, rnorm, eps);
  } else {
    if (K == 128) rc = has_pro ? launch_fwd_x6_v<128, 2, true, false, true>(X, pr, W, bias, Y, M, N, 0, st, nullptr, rnorm, eps)
                               : launch_fwd_x6_v<128, 2, false, false, true>(X, pr, W, bias, Y, M, N, 0, st, nullptr, rnorm, eps);
    else rc = has_pro ? launch_fwd_x6_v<64, 2, true, false, true>(X, pr, W, bias, Y, M, N, 0, st, nullptr, rnorm, eps)
                      : launch_fwd_x6_v<64, 2, false, false, true>(X, pr, W, bias, Y, M, N, 0, st, nullptr, rnorm, eps);
  }
  if (rc) return rc;
  MMG_CHECK_LAUNCH("linear_fwd_l2norm");
  return MMG_OK;
}

extern "C" int mmg_linear_bnbwd_supported(int64_t M, int N, int K) {
  return (M > 512 && (K == 64 || K == 128) && (N == 64 || N == 128)) ? 1 : 0;
}

extern "C" int mmg_linear_bnbwd_next_bn(const float* G, const float* Y, const mmg_prologue_t* pro, const float* mean,
                                        const float* rstd, const double* sums, double inv_count, float* dbeta, float* dgamma,
                                        const float* W, float* dZ, float* dX, int64_t M, int N, int K,
                                        const mmg_next_bn_t* next, void* stream);

extern "C" int mmg_linear_bnbwd(const float* G, const float* Y, const mmg_prologue_t* pro, const float* mean,
                                const float* rstd, const double* sums, double inv_count, float* dbeta, float* dgamma,
                                const float* W, float* dZ, float* dX, int64_t M, int N, int K, void* stream) {
  return mmg_linear_bnbwd_next_bn(G, Y, pro, mean, rstd, sums, inv_count, dbeta, dgamma, W, dZ, dX, M, N, K, nullptr, stream);
}

extern "C" int mmg_linear_bnbwd_next_bn(const float* G, const float* Y, const mmg_prologue_t* pro, const float* mean,
                                        const float* rstd, const double* sums, double inv_count, float* dbeta, float* dgamma,
                                        const float* W, float* dZ, float* dX, int64_t M, int N, int K,
                                        const mmg_next_bn_t* next, void* stream) {
  MMG_CHECK_ARG(mmg_linear_bnbwd_supported(M, N, K), "linear_bnbwd: M=%lld N=%d K=%d unsupported (M > 512, K and N in {64,128})",
                (long long)M, N, K);
  MMG_CHECK_ARG(G && Y && W && dZ && dX, "linear_bnbwd: null buffer");
  MMG_CHECK_ARG(!pro || !pro->scale || (pro->shift && mean && rstd), "linear_bnbwd: BatchNorm fold without shift / mean / rstd");
  MMG_CHECK_ARG(!pro || pro->relu == MMG_ACT_NONE || pro->relu == MMG_ACT_RELU, "linear_bnbwd: relu only");
  MMG_CHECK_ARG(!sums || (pro && pro->scale), "linear_bnbwd: sums without a BatchNorm fold");
  const ProDev pr = mmg_pro_dev(pro);
  BnBwdDev bb{Y, mean, rstd, sums, inv_count, dZ, dbeta, dgamma, 0.f, nullptr, nullptr, 0};
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (next && K == 128 && N == 128 && next_bn_fusable(next)) {
    NextBnDev nb;
    double* partial;
    rc = mmg_next_bn_dev(next, M, N, "linear_bnbwd", &nb, &partial);
    if (rc) return rc;
    rc = launch_bnbwd_x6<128, 4, 0, true>(G, bb, pr, W, dX, M, st, mmg_pro_dev(nullptr), nb, partial);
    if (rc) return rc;
    MMG_CHECK_LAUNCH("linear_bnbwd");
    return mmg_next_bn_finish(next, partial, N, (int)bnbwd_x6_rows(M, K), stream);
  }
  if (K == 128) rc = N == 128 ? launch_bnbwd_x6<128, 4>(G, bb, pr, W, dX, M, st) : launch_bnbwd_x6<128, 2>(G, bb, pr, W, dX, M, st);
  else rc = N == 128 ? launch_bnbwd_x6<64, 4>(G, bb, pr, W, dX, M, st) : launch_bnbwd_x6<64, 2>(G, bb, pr, W, dX, M, st);
  if (rc) return rc;
  MMG_CHECK_LAUNCH("linear_bnbwd");
  return next ? mmg_next_bn_fallback(dX, M, N, next, "linear_bnbwd", stream) : MMG_OK;
}

extern "C" int mmg_linear_bnbwd2(const float* G, const float* G2, const float* Y, const mmg_prologue_t* pro,
                                 const mmg_prologue_t* pro2, const float* mean, const float* rstd, const double* sums,
                                 double inv_count, float* dbeta, float* dgamma, const float* W, float* dZ, float* dX,
                                 int64_t M, int N, int K, void* stream) {
  MMG_CHECK_ARG(mmg_linear_bnbwd_supported(M, N, K) && K == 128 && N == 128,
                "linear_bnbwd2: M=%lld N=%d K=%d unsupported (M > 512, K = N = 128)", (long long)M, N, K);
  MMG_CHECK_ARG(G && G2 && Y && W && dZ && dX && pro && pro2, "linear_bnbwd2: null buffer");
  MMG_CHECK_ARG(!pro->scale || (pro->shift && mean && rstd), "linear_bnbwd2: BatchNorm fold without shift / mean / rstd");
  MMG_CHECK_ARG(pro->relu == MMG_ACT_NONE || pro->relu == MMG_ACT_RELU, "linear_bnbwd2: relu only");
  MMG_CHECK_ARG(!sums || pro->scale, "linear_bnbwd2: sums without a BatchNorm fold");
  const ProDev pr = mmg_pro_dev(pro), pr2 = mmg_pro_dev(pro2);
  BnBwdDev bb{Y, mean, rstd, sums, inv_count, dZ, dbeta, dgamma, 0.f, G2, nullptr, 0};
  int rc = launch_bnbwd_x6<128, 4, 2>(G, bb, pr, W, dX, M, (hipStream_t)stream, pr2);
  if (rc) return rc;
  MMG_CHECK_LAUNCH("linear_bnbwd2");
  return MMG_OK;
}

extern "C" int mmg_linear_bnbwd_rows_next_bn(const float* G_rows, const int32_t* row_pos, int64_t n_sel, const float* Y,
                                             const mmg_prologue_t* pro, const float* mean, const float* rstd,
                                             const double* sums, double inv_count, float* dbeta, float* dgamma, const float* W,
                                             float* dZ, float* dX, int64_t M, int N, int K, const mmg_next_bn_t* next,
                                             void* stream);

extern "C" int mmg_linear_bnbwd_rows(const float* G_rows, const int32_t* row_pos, int64_t n_sel, const float* Y,
                                     const mmg_prologue_t* pro, const float* mean, const float* rstd, const double* sums,
                                     double inv_count, float* dbeta, float* dgamma, const float* W, float* dZ, float* dX,
                                     int64_t M, int N, int K, void* stream) {
  return mmg_linear_bnbwd_rows_next_bn(G_rows, row_pos, n_sel, Y, pro, mean, rstd, sums, inv_count, dbeta, dgamma, W, dZ, dX, M,
                                       N, K, nullptr, stream);
}

extern "C" int mmg_linear_bnbwd_rows_next_bn(const float* G_rows, const int32_t* row_pos, int64_t n_sel, const float* Y,
                                             const mmg_prologue_t* pro, const float* mean, const float* rstd,
                                             const double* sums, double inv_count, float* dbeta, float* dgamma, const float* W,
                                             float* dZ, float* dX, int64_t M, int N, int K, const mmg_next_bn_t* next,
                                             void* stream) {
  MMG_CHECK_ARG(mmg_linear_bnbwd_supported(M, N, K) && K == 128 && N == 128,
                "linear_bnbwd_rows: M=%lld N=%d K=%d unsupported (M > 512, K = N = 128)", (long long)M, N, K);
  MMG_CHECK_ARG(Y && W && dZ && dX && pro && row_pos, "linear_bnbwd_rows: null buffer");
  MMG_CHECK_ARG(n_sel >= 0 && n_sel * (int64_t)K * 4 < (int64_t)0x7FFFFFFF && (G_rows || n_sel == 0), "linear_bnbwd_rows: bad row list");
  MMG_CHECK_ARG(!pro->scale || (pro->shift && mean && rstd), "linear_bnbwd_rows: BatchNorm fold without shift / mean / rstd");
  MMG_CHECK_ARG(pro->relu == MMG_ACT_NONE || pro->relu == MMG_ACT_RELU, "linear_bnbwd_rows: relu only");
  MMG_CHECK_ARG(!sums || pro->scale, "linear_bnbwd_rows: sums without a BatchNorm fold");
  const ProDev pr = mmg_pro_dev(pro);
  BnBwdDev bb{Y, mean, rstd, sums, inv_count, dZ, dbeta, dgamma, 0.f, nullptr, row_pos, n_sel};
  if (next && next_bn_fusable(next)) {
    NextBnDev nb;
    double* partial;
    int rc0 = mmg_next_bn_dev(next, M, N, "linear_bnbwd_rows", &nb, &partial);
    if (rc0) return rc0;
    rc0 = launch_bnbwd_x6<128, 4, 3, true>(G_rows ? G_rows : Y, bb, pr, W, dX, M, (hipStream_t)stream, mmg_pro_dev(nullptr), nb, partial);
    if (rc0) return rc0;
    MMG_CHECK_LAUNCH("linear_bnbwd_rows");
    return mmg_next_bn_finish(next, partial, N, (int)bnbwd_x6_rows(M, K), stream);
  }
  int rc = launch_bnbwd_x6<128, 4, 3>(G_rows ? G_rows : Y, bb, pr, W, dX, M, (hipStream_t)stream);
  if (rc) return rc;
  MMG_CHECK_LAUNCH("linear_bnbwd_rows");
  return next ? mmg_next_bn_fallback(dX, M, N, next, "linear_bnbwd_rows", stream) : MMG_OK;
}

extern "C" int mmg_linear_l2bwd_next_bn(const float* G, const float* out, const float* rnorm, const float* W, float* dZ,
                                        float* dX, int64_t M, int N, int K, float eps, const mmg_next_bn_t* next, void* stream);

extern "C" int mmg_linear_l2bwd(const float* G, const float* out, const float* rnorm, const float* W, float* dZ, float* dX,
                                int64_t M, int N, int K, float eps, void* stream) {
  return mmg_linear_l2bwd_next_bn(G, out, rnorm, W, dZ, dX, M, N, K, eps, nullptr, stream);
}

extern "C" int mmg_linear_l2bwd_next_bn(const float* G, const float* out, const float* rnorm, const float* W, float* dZ,
                                        float* dX, int64_t M, int N, int K, float eps, const mmg_next_bn_t* next, void* stream) {
  MMG_CHECK_ARG(mmg_linear_bnbwd_supported(M, N, K), "linear_l2bwd: M=%lld N=%d K=%d unsupported (M > 512, K and N in {64,128})",
                (long long)M, N, K);
  MMG_CHECK_ARG(G && out && rnorm && W && dZ && dX, "linear_l2bwd: null buffer");
  const ProDev pr = mmg_pro_dev(nullptr);
  BnBwdDev bb{out, rnorm, nullptr, nullptr, 0.0, dZ, nullptr, nullptr, eps, nullptr, nullptr, 0};
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (next && K == 128 && N == 128 && next_bn_fusable(next)) {
    NextBnDev nb;
    double* partial;
    rc = mmg_next_bn_dev(next, M, N, "linear_l2bwd", &nb, &partial);
    if (rc) return rc;
    rc = launch_bnbwd_x6<128, 4, 1, true>(G, bb, pr, W, dX, M, st, mmg_pro_dev(nullptr), nb, partial);
    if (rc) return rc;
    MMG_CHECK_LAUNCH("linear_l2bwd");
    return mmg_next_bn_finish(next, partial, N, (int)bnbwd_x6_rows(M, K), stream);
  }
  if (K == 128) rc = N == 128 ? launch_bnbwd_x6<128, 4, 1>(G, bb, pr, W, dX, M, st) : launch_bnbwd_x6<128, 2, 1>(G, bb, pr, W, dX, M, st);
  else rc = N == 128 ? launch_bnbwd_x6<64, 4, 1>(G, bb, pr, W, dX, M, st) : launch_bnbwd_x6<64, 2, 1>(G, bb, pr, W, dX, M, st);
  if (rc) return rc;
  MMG_CHECK_LAUNCH("linear_l2bwd");
  return next ? mmg_next_bn_fallback(dX, M, N, next, "linear_l2bwd", stream) : MMG_OK;
}

extern "C" size_t mmg_linear_wgrad_ws_bytes(int64_t M, int N, int K) {
  if (M < 0 || N <= 0 || K <= 0 || N % 64 || K % 64) return 0;
  WgradPlan p = plan_wgrad(M, N, K);
  return (size_t)p.n_split * ((size_t)N * K + N) * 4 + 256;       // + N: the optional bias sums behind every slab
}

static int linear_wgrad_impl(const float* dY, const float* X, const mmg_prologue_t* pro, float* dW, float* dbias,
                             int64_t M, int N, int K, int accumulate, void* ws, size_t ws_bytes, void* stream,
                             mmg_wgrad_reduce_t* job) {
  if (job) { job->slab = nullptr; job->n4 = 0; job->n_split = 0; job->dW = dW; job->dbias = dbias; job->nk4 = (int64_t)N * K / 4; job->accumulate = accumulate; }
  MMG_CHECK_ARG(M >= 0 && N > 0 && K > 0 && N % 64 == 0 && K % 64 == 0, "linear_wgrad: N=%d K=%d must be multiples of 64", N, K);
  MMG_CHECK_ARG(dW, "linear_wgrad: dW is null");
  hipStream_t st = (hipStream_t)stream;
  if (M == 0) {
    if (!accumulate) {
      MMG_CHECK_HIP(mmg_zero_async(dW, (size_t)N * K * 4, st), "linear_wgrad(memset)");
      if (dbias) MMG_CHECK_HIP(mmg_zero_async(dbias, (size_t)N * 4, st), "linear_wgrad(memset)");
    }
    return MMG_OK;
  }
  MMG_CHECK_ARG(dY && X && ws, "linear_wgrad: null buffer");
  MMG_CHECK_ARG(!pro || pro->relu == MMG_ACT_NONE || pro->relu == MMG_ACT_RELU, "linear_wgrad: the prologue takes relu only");
  const size_t need = mmg_linear_wgrad_ws_bytes(M, N, K);
  if (ws_bytes < need) {
    mmg_set_error("linear_wgrad: workspace %zu < %zu", ws_bytes, need);
    return MMG_E_WS;
  }
  WgradPlan p = plan_wgrad(M, N, K);
  float* slab = (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  const ProDev pr = mmg_pro_dev(pro);
  dim3 grid((unsigned)p.n_tiles, (unsigned)p.n_split);
  const int direct = p.n_split == 1 ? (accumulate ? 2 : 1) : 0;     // small M: no slab, no reduce launch
  float* target = direct ? dW : slab;
  const int64_t stride = (int64_t)N * K + (dbias ? N : 0);
  {
    // eight waves (two per SIMD: one stages while the other multiplies) wherever the tile has 8 sub-tiles
    const int64_t rps = p.rows_per_split;
    // role-specialised kernel for the plain 128 x 128 case; with a prologue the X stagers would also carry the dropout
    // hashes and the symmetric kernel is faster (53 vs 64 us)
    const bool has_pro = pr.scale || pr.relu || pr.p > 0.f;
    int rc = 0;
    if (p.TN == 128 && p.TK == 128 && !has_pro) rc = launch_wgrad_ws(grid, st, dY, X, target, M, N, K, direct, stride, dbias);
    else if (p.TN == 128 && p.TK == 128) rc = launch_wgrad_x6<128, 128, 4, 2>(grid, st, dY, X, pr, target, M, N, K, rps, direct, stride, dbias);
    else if (p.TN == 128) rc = launch_wgrad_x6<128, 64, 4, 2>(grid, st, dY, X, pr, target, M, N, K, rps, direct, stride, dbias);
    else if (p.TK == 128) rc = launch_wgrad_x6<64, 128, 2, 4>(grid, st, dY, X, pr, target, M, N, K, rps, direct, stride, dbias);
    else rc = launch_wgrad_x6<64, 64, 2, 2>(grid, st, dY, X, pr, target, M, N, K, rps, direct, stride, dbias);
    if (rc) return rc;
  }
  if (!direct) {
    if (job) {                      // the caller sums the slabs later, together with those of other layers
      job->slab = slab; job->n4 = stride / 4; job->n_split = p.n_split;
    } else {
      MMG_LAUNCH(MMG_PROBE_LINEAR_WGRAD_REDUCE, M, N, K, 0, (mmg_k_reduce_slabs<EpiStore>),
                 dim3((unsigned)((stride / 4 + 15) / 16)), dim3(256), 0, st, slab, stride / 4, p.n_split,
                 EpiStore{dW, accumulate, dbias, (int64_t)N * K / 4});
    }
  }
  MMG_CHECK_LAUNCH("linear_wgrad");
  return MMG_OK;
}

extern "C" int mmg_linear_wgrad(const float* dY, const float* X, const mmg_prologue_t* pro, float* dW, float* dbias,
                                int64_t M, int N, int K, int accumulate, void* ws, size_t ws_bytes, void* stream) {
  return linear_wgrad_impl(dY, X, pro, dW, dbias, M, N, K, accumulate, ws, ws_bytes, stream, nullptr);
}

extern "C" int mmg_linear_wgrad_is_direct(int64_t M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0 || N % 64 || K % 64) return 1;
  return plan_wgrad(M, N, K).n_split == 1 ? 1 : 0;
}

extern "C" int mmg_linear_wgrad_deferred(const float* dY, const float* X, const mmg_prologue_t* pro, float* dW, float* dbias,
                                         int64_t M, int N, int K, int accumulate, void* ws, size_t ws_bytes, void* stream,
                                         mmg_wgrad_reduce_t* job) {
  MMG_CHECK_ARG(job, "linear_wgrad_deferred: job is null");
  return linear_wgrad_impl(dY, X, pro, dW, dbias, M, N, K, accumulate, ws, ws_bytes, stream, job);
}

namespace {
struct WgradReduceTable { mmg_wgrad_reduce_t j[MMG_WGRAD_REDUCE_MAX]; };
// the slab sums of several weight gradients in ONE launch: blockIdx.y = job, the body of mmg_k_reduce_slabs<EpiStore>
__global__ __launch_bounds__(256) void k_wgrad_reduce_group(WgradReduceTable tb) {
  __shared__ mmg_f4 part[16][16];
  const mmg_wgrad_reduce_t jb = tb.j[blockIdx.y];
  const int e = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int64_t i4 = (int64_t)blockIdx.x * 16 + e;
  if ((int64_t)blockIdx.x * 16 >= jb.n4) return;               // (uniform per block)
  mmg_f4 acc = {0.f, 0.f, 0.f, 0.f};
  if (i4 < jb.n4) {
    const mmg_f4* base = reinterpret_cast<const mmg_f4*>(jb.slab) + i4;
    int sidx = g;
    for (; sidx + 7 * 16 < jb.n_split; sidx += 8 * 16) {
      mmg_f4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = base[(size_t)(sidx + u * 16) * jb.n4];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; sidx < jb.n_split; sidx += 16) acc += base[(size_t)sidx * jb.n4];
  }
  part[g][e] = acc;
  __syncthreads();
  if (g == 0 && i4 < jb.n4) {
    mmg_f4 t = part[0][e];
#pragma unroll
    for (int q = 1; q < 16; ++q) t += part[q][e];
    EpiStore{jb.dW, jb.accumulate, jb.dbias, jb.nk4}(i4, t);
  }
}
}  // namespace

extern "C" int mmg_wgrad_reduce_group(const mmg_wgrad_reduce_t* jobs, int n_jobs, void* stream) {
  MMG_CHECK_ARG(jobs && n_jobs >= 1 && n_jobs <= MMG_WGRAD_REDUCE_MAX, "wgrad_reduce_group: 1..%d jobs", MMG_WGRAD_REDUCE_MAX);
  WgradReduceTable tb;
  int64_t max_n4 = 0;
  int n = 0;
  for (int j = 0; j < n_jobs; ++j) {
    if (!jobs[j].slab) continue;                                // (a small-M launch wrote its gradient directly)
    MMG_CHECK_ARG(jobs[j].dW && jobs[j].n4 > 0 && jobs[j].n_split > 0, "wgrad_reduce_group: bad job %d", j);
    for (int q = 0; q < n; ++q)
      MMG_CHECK_ARG(tb.j[q].dW != jobs[j].dW, "wgrad_reduce_group: two jobs of one launch write the same gradient (job %d)", j);
    tb.j[n++] = jobs[j];
    if (jobs[j].n4 > max_n4) max_n4 = jobs[j].n4;
  }
  if (n == 0) return MMG_OK;
  MMG_LAUNCH(MMG_PROBE_LINEAR_WGRAD_REDUCE, 0, 0, 0, 0, (k_wgrad_reduce_group), dim3((unsigned)((max_n4 + 15) / 16), (unsigned)n),
             dim3(256), 0, (hipStream_t)stream, tb);
  MMG_CHECK_LAUNCH("wgrad_reduce_group");
  return MMG_OK;
}
