/*
 * mmgnn.h -- C ABI of libmmgnn.so: MI355X (gfx950) kernels for the heterogeneous-GNN
 * message-passing hot path of AdalineL/Multi-Modal-GNN.
 *
 * The reference is pure Python; its "FFI" for this path is the set of ATen / PyG operators
 * that src/model.py dispatches.  Each entry point below names the reference expression it
 * replaces (paths relative to the reference repo).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions (all entry points):
 *   - extern "C", returns 0 on success, <0 on error (MMG_E_*); the message is available from
 *     mmg_last_error() (thread-local).
 *   - The CALLER owns every buffer: device pointers to contiguous, 16-byte aligned memory.
 *     No allocation, no synchronisation, no global mutable state inside; everything is
 *     enqueued on `stream` (a hipStream_t passed as void*; NULL = the null stream).
 *   - Indices are int32 inside (row/col counts are checked to be < 2^31); the edge_index
 *     handed to mmg_csr_build is the reference's int64 [2,E] row-major tensor.
 *   - All features are fp32.  D (feature width) must be 64, 128 or 256.
 *   - Workspace: *_ws_bytes() twins return the scratch size a call needs.
 */
#ifndef MMGNN_H
#define MMGNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMG_OK 0
#define MMG_E_ARG (-1)     /* bad argument (shape, null pointer, unsupported D) */
#define MMG_E_LAUNCH (-2)  /* HIP launch / runtime error */
#define MMG_E_WS (-3)      /* workspace too small */

#define MMG_MAX_REL 4      /* relations fused in one aggregate launch */

int mmg_version(void);
const char* mmg_last_error(void);

/* A HIP stream of the library's own (hipStreamNonBlocking, on the current device) -- outside every stream pool of the
 * host framework, so nothing the host code captures into a hipGraph ever lands on it.  The sharded host side
 * (mmgnn/dist.py) issues its EAGER RCCL all-reduces there: torch's process-group watchdog polls the end events of eager
 * collectives with hipEventQuery, and HIP refuses that query -- and invalidates the capture -- while the stream the
 * event was recorded on is capturing (profiles/probes/rccl_event_cache_abort.py). */
int mmg_stream_create(void** stream_out);
int mmg_stream_destroy(void* stream);

/* ---------------------------------------------------------------------------------------
 * CSR construction (consumes the tensors of
 * src/graph_build.py:476-586).  Stable sort of edge ids by edge_index[sort_row]:
 *   rowptr[n_rows+1], col[E] = edge_index[1-sort_row][perm], perm[E] = original edge id.
 * Bit-exact with torch.sort(stable=True) + bincount + cumsum.
 * A key outside [0, n_rows) is never used as an address: it sorts behind the last row and is not counted, so
 * rowptr[n_rows] < n_edges tells the caller that the input was invalid (graph_build.py:618-633 validates up front).
 * ------------------------------------------------------------------------------------- */
size_t mmg_csr_build_ws_bytes(int64_t n_edges, int64_t n_rows);
int mmg_csr_build(const int64_t* edge_index, int64_t n_edges, int64_t n_rows, int sort_row,
                  int32_t* rowptr, int32_t* col, int32_t* perm,
                  void* ws, size_t ws_bytes, void* stream);

/* deg[i] = rowptr[i+1]-rowptr[i]; inv[i] = 1/max(deg,1)  (PyG mean aggregation's clamp;
 * also torch.bincount of src/model.py:297-298 when called on the has_lab CSR) */
int mmg_row_degree(const int32_t* rowptr, int64_t n_rows, int32_t* deg, float* inv_deg, void* stream);
/* in-degree of the column side: cnt[j] = #edges with col == j; inv[j] = 1/max(cnt,1) */
int mmg_col_degree(const int32_t* col, int64_t n_edges, int64_t n_cols, int32_t* cnt, float* inv_cnt,
                   void* stream);

/* ---------------------------------------------------------------------------------------
 * Sparse aggregates (replace PyG SAGEConv's gather + scatter-mean, call site
 * src/model.py:125-131,256, and their backward).  All relations share the ROW axis
 * (patients, CSR-by-patient), so up to MMG_MAX_REL relations are fused per launch.
 *
 * gather:   out[i,:] (+)= sum_r  rowscale_r[i] * sum_{k in row_r(i)} table_r[col_r[k], :]
 *           (vocab -> patient forward; patient -> vocab backward)
 * scatter:  out_r[j,:]  = colscale_r[j] * sum_{k: col_r[k]=j} rowscale_r[row(k)] * x[row(k), :]
 *           (patient -> vocab forward; vocab -> patient backward)
 * rowscale / colscale entries may be NULL (= 1).
 * ------------------------------------------------------------------------------------- */
typedef struct {
  const int32_t* rowptr;   /* [n_rows+1] */
  const int32_t* col;      /* [E] */
  const float* rowscale;   /* [n_rows] or NULL */
  const float* colscale;   /* [n_cols] or NULL */
  const float* table;      /* gather: [n_cols, D] source rows */
  float* out;              /* scatter: [n_cols, D] destination rows */
  int32_t n_cols;
  uint32_t flags;          /* MMG_REL_SIMPLE: no (row, col) pair occurs twice -- enables the 0/1 indicator
                              (bf16-split MFMA) kernels; multigraphs take the counting fp32 path */
  const uint64_t* mask_t;  /* NULL, or the relation's adjacency as bit planes (mmg_rel_mask_build): word
                              [row / 64][col][(row % 64 / 8) & 1] has bit 16 * (row % 64 / 16) + 4 + row % 8 set
                              iff (row, col) is an edge (four 16-bit fields, one per 16 rows, each holding 8 row
                              bits << 4); [ceil(n_rows / 64)][pad32(n_cols)][2] words.  Simple relations only. */
  const uint16_t* mask_r;  /* NULL, or the same adjacency row-major for the gather: field
                              [row][(col % 16 / 8) & 1][col / 16] = (8 item bits of cols 16 (col/16) + 8 half + 0..7)
                              << 4; [n_rows][2][pad32(n_cols) / 16] uint16. */
} mmg_rel_t;
#define MMG_REL_SIMPLE 1u

/* One-off per static graph: the bit-plane form of a CSR-by-row relation (40 B per patient at the eICU vocab
 * instead of 4 B per edge).  The scatter kernels expand indicator fragments for the matrix cores straight from
 * these words.  mask_t and mask_r must each hold mmg_rel_mask_words(n_rows, n_cols) uint64 (8-byte units for both);
 * they are zeroed here.  Either may be NULL (not built). */
size_t mmg_rel_mask_words(int64_t n_rows, int32_t n_cols);
int mmg_rel_mask_build(const int32_t* rowptr, const int32_t* col, int64_t n_rows, int32_t n_cols,
                       uint64_t* mask_t, uint16_t* mask_r, void* stream);

int mmg_gather_rows(const mmg_rel_t* rels, int n_rel, int64_t n_rows, int D,
                    float* out, int accumulate, void* stream);
/* the same, plus col_sums[2,D] (fp64) = (sum_rows out, sum_rows out^2) of the FINAL output: the batch statistics of
 * the per-type BatchNorm that follows the HeteroConv sum (src/model.py:258-262), from the gather epilogue */
/* Training-mode BatchNorm1d fold (what mmg_bn_finalize computes from the column sums: scale / shift / mean / rstd, running
 * statistics advanced n_updates times) taken in the SAME launch that sums the producers' partial statistics rows:
 * mmg_linear_fwd_stats_bn / mmg_gather_rows_stats_bn = mmg_linear_fwd_stats / mmg_gather_rows_stats + mmg_bn_finalize
 * (training = 1) with one launch less (col_sums still receives the sums).  Single-GPU form: a patient-sharded run has to
 * all-reduce the sums between the two. */
typedef struct {
  int64_t count;
  const float* gamma; const float* beta;      /* [N], nullable: 1 / 0 */
  float* running_mean; float* running_var;    /* [N], nullable: not advanced */
  int n_updates;
  float momentum; float eps;
  float* scale; float* shift; float* mean; float* rstd;   /* [N] outputs; mean / rstd nullable */
} mmg_bn_fin_t;
size_t mmg_gather_rows_stats_ws_bytes(int64_t n_rows, int D);
int mmg_gather_rows_stats(const mmg_rel_t* rels, int n_rel, int64_t n_rows, int D, float* out, int accumulate,
                          double* col_sums, void* ws, size_t ws_bytes, void* stream);
int mmg_gather_rows_stats_bn(const mmg_rel_t* rels, int n_rel, int64_t n_rows, int D, float* out, int accumulate,
                             double* col_sums, void* ws, size_t ws_bytes, const mmg_bn_fin_t* fin, void* stream);

size_t mmg_scatter_rows_ws_bytes(const mmg_rel_t* rels, int n_rel, int64_t n_rows, int D);
int mmg_scatter_rows(const mmg_rel_t* rels, int n_rel, int64_t n_rows, int D,
                     const float* x, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * Dense layers: fp32 products as an exact six-term bf16 split on the bf16 matrix cores
 * (v_mfma_f32_32x32x16_bf16, fp32 accumulation; 2e-6 of an fp64 reference).  Replace torch.nn.Linear /
 * BatchNorm1d / ReLU / Dropout / F.normalize of src/model.py:93-105,229-232,258-269 and the
 * lin_l / lin_r of SAGEConv.
 *
 * Prologue applied to X on load (all optional):
 *   x' = dropout( relu( x * scale[k] + shift[k] ) )      scale/shift = folded BatchNorm
 * ------------------------------------------------------------------------------------- */
#define MMG_ACT_NONE 0
#define MMG_ACT_RELU 1
#define MMG_ACT_LEAKY_RELU 2
#define MMG_ACT_ELU 3
typedef struct {
  const float* scale;      /* [K] or NULL: no affine */
  const float* shift;      /* [K] */
  int relu;                /* activation after the affine: MMG_ACT_NONE / _RELU / _LEAKY_RELU (slope 0.01) / _ELU (alpha 1)
                              -- the three src/model.py:145-152 accepts.  The dense kernels' prologue takes NONE / RELU
                              only (patient_transform and the heads are ReLU by construction, model.py:93-103,373-386);
                              the materialising / backward kernels (mmg_affine_act_drop*, mmg_bn_bwd_*) take all four. */
  float drop_p;            /* 0 = no dropout */
  uint64_t seed;           /* dropout RNG: keep(seed, site, global_row*K + k) */
  uint32_t site;
  int64_t row_offset;      /* global index of row 0 (patient sharding) */
  const uint64_t* seed_ptr;/* optional DEVICE pointer: when non-NULL the seed is read from it at run time
                              (lets a captured hipGraph draw fresh masks on every replay) */
} mmg_prologue_t;

/* Y[M,N] = prologue(X)[M,K] . W[N,K]^T (+ bias[N]) (+ Y if MMG_LIN_ACCUMULATE).
 * MMG_LIN_W_KN: W is stored [K,N] (Y = X . W) -- the data-gradient GEMMs of the backward pass read the forward
 * weight in place instead of a transposed copy. */
#define MMG_LIN_ACCUMULATE 1
#define MMG_LIN_W_KN 2
int mmg_linear_fwd(const float* X, const mmg_prologue_t* pro, const float* W, const float* bias,
                   float* Y, int64_t M, int N, int K, int flags, void* stream);
/* the same, plus col_sums[2,N] (fp64) = (sum_m Y, sum_m Y^2): the batch statistics of the BatchNorm that follows
 * (src/model.py:95,99), taken in the GEMM epilogue instead of a second pass over Y */
/* the same layer followed by the row L2 normalisation (patient_transform's last Linear + F.normalize(p=2, dim=1),
 * src/model.py:103,232) in one kernel: Y = y / max(|y|_2, eps) per row, rnorm[row] = 1 / max(|y|_2, eps) (what
 * mmg_l2norm_fwd returns; mmg_l2norm_bwd takes exactly these two).  Supported for M > 512, N and K in {64, 128}. */
int mmg_linear_fwd_l2norm_supported(int64_t M, int N, int K);
int mmg_linear_fwd_l2norm(const float* X, const mmg_prologue_t* pro, const float* W, const float* bias, float* Y,
                          float* rnorm, int64_t M, int N, int K, float eps, void* stream);
size_t mmg_linear_fwd_stats_ws_bytes(int64_t M, int N);
int mmg_linear_fwd_stats(const float* X, const mmg_prologue_t* pro, const float* W, const float* bias,
                         float* Y, int64_t M, int N, int K, int flags, double* col_sums, void* ws,
                         size_t ws_bytes, void* stream);
int mmg_linear_fwd_stats_bn(const float* X, const mmg_prologue_t* pro, const float* W, const float* bias, float* Y,
                            int64_t M, int N, int K, int flags, double* col_sums, void* ws, size_t ws_bytes,
                            const mmg_bn_fin_t* fin, void* stream);

/* dW[N,K] (+)= dY[M,N]^T . prologue(X)[M,K]   (reduction over the M rows);
 * dbias (nullable, [N]) (+)= the column sums of dY -- the bias gradient of the same layer, from the same pass */
size_t mmg_linear_wgrad_ws_bytes(int64_t M, int N, int K);
int mmg_linear_wgrad(const float* dY, const float* X, const mmg_prologue_t* pro, float* dW, float* dbias,
                     int64_t M, int N, int K, int accumulate, void* ws, size_t ws_bytes, void* stream);
/* The weight gradient is the fixed-order sum of per-workgroup partial slabs; that sum is a launch of a few microseconds
 * behind every layer.  mmg_linear_wgrad_deferred leaves the slabs in `ws` (which then has to stay untouched) and fills
 * `job`; mmg_wgrad_reduce_group sums the slabs of up to MMG_WGRAD_REDUCE_MAX layers in ONE launch (nobody reads a weight
 * gradient before the optimizer).  Two jobs of one launch must not write the same dW (an accumulating second
 * contribution goes into a later launch); job.slab == NULL (a small-M launch wrote dW directly) is skipped. */
#define MMG_WGRAD_REDUCE_MAX 16
typedef struct {
  const float* slab; int64_t n4; int n_split;     /* n_split slabs of n4 float4 each */
  float* dW; float* dbias; int64_t nk4;           /* float4s [0, nk4) -> dW, the rest -> dbias */
  int accumulate;
} mmg_wgrad_reduce_t;
int mmg_linear_wgrad_deferred(const float* dY, const float* X, const mmg_prologue_t* pro, float* dW, float* dbias,
                              int64_t M, int N, int K, int accumulate, void* ws, size_t ws_bytes, void* stream,
                              mmg_wgrad_reduce_t* job);
int mmg_wgrad_reduce_group(const mmg_wgrad_reduce_t* jobs, int n_jobs, void* stream);
/* 1: a launch of this shape writes / accumulates dW itself (one row range: no slabs) -- an accumulating call of that kind
 * must not overtake a deferred job of the same gradient */
int mmg_linear_wgrad_is_direct(int64_t M, int N, int K);

/* column reductions over rows: out[0,:] = sum_m A[m,:], out[1,:] = sum_m A[m,:]*B[m,:] (fp64 out) */
size_t mmg_col_reduce2_ws_bytes(int64_t M, int N);
int mmg_col_reduce2(const float* A, const float* B, double* out, int64_t M, int N,
                    void* ws, size_t ws_bytes, void* stream);

/* BatchNorm statistics -> folded scale/shift (+ running-stat update, momentum 0.1, unbiased var)
 *   sums[2,N] fp64 = (sum y, sum y^2) over `count` rows (already all-reduced when sharded).
 *   training != 0: mean/var from sums; running_* updated `n_updates` times (F7 double update).
 *   training == 0: scale/shift from running stats.
 *   Outputs: scale[N] = gamma*rstd, shift[N] = beta - mean*scale, mean[N], rstd[N]. */
int mmg_bn_finalize(const double* sums, int64_t count, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, int training, int n_updates,
                    float momentum, float eps, float* scale, float* shift, float* mean, float* rstd,
                    int N, void* stream);

/* out = dropout(relu(y*scale + shift))  materialised                                */
int mmg_affine_act_drop(const float* Y, const mmg_prologue_t* pro, float* out, int64_t M, int N,
                        void* stream);
/* The same for a subset of rows:  out[s, :] = dropout(relu(y[rows[s], :]*scale + shift)),  s < n_sel, the dropout
 * mask being the one of row rows[s] of the full tensor.  (src/model.py:294 runs encode_nodes a first time only to feed
 * tabular_mlp, which model.py:312-322 applies to the low-degree patients: everything after the last BatchNorm of that
 * pass -- model.py:99-103 -- runs on those rows alone.) */
int mmg_affine_act_drop_rows(const float* Y, const mmg_prologue_t* pro, const int64_t* rows, int64_t n_sel,
                             float* out, int N, void* stream);

/* Backward through dropout -> relu -> affine(BN):
 *   g_out = g * keepmask/(1-p) * [y*scale+shift > 0]
 * pass 1 (stats):  sums[0,:] = sum g_out ; sums[1,:] = sum g_out * xhat,  xhat = (y-mean)*rstd
 * pass 2 (apply):  dy = scale * (g_out - c0[k] - xhat*c1[k]),  c0 = sums0/count, c1 = sums1/count
 *                  (eval mode: sums = NULL, c0 = c1 = 0).  `sums` is the (all-reduced, when sharded) output of pass 1
 *                  and inv_count = 1/count; dbeta / dgamma (nullable, [N] float) receive sums0 / sums1, the gradients
 *                  of the BatchNorm bias / weight.  accumulate != 0: dY += (the two encode_nodes passes of a training
 *                  step share their first layer: the second pass adds its gradient to the first one's). */
int mmg_bn_bwd_stats(const float* G, const float* Y, const mmg_prologue_t* pro, const float* mean,
                     const float* rstd, double* sums, int64_t M, int N, void* ws, size_t ws_bytes,
                     void* stream);
int mmg_bn_bwd_apply(const float* G, const float* Y, const mmg_prologue_t* pro, const float* mean,
                     const float* rstd, const double* sums, double inv_count, float* dbeta, float* dgamma,
                     float* dY, int64_t M, int N, int accumulate, void* stream);

/* mmg_bn_bwd_apply folded into the data-gradient GEMM that follows it (autograd of nn.Linear behind BatchNorm1d + relu +
 * dropout, src/model.py:93-101,258-269):  dZ = pass 2 above at (G, Y)  and  dX[M,N] = dZ[M,K] . W  with W stored [K,N] (the
 * forward weight in place, as MMG_LIN_W_KN), in ONE pass over G and Y -- dZ is written once for the weight gradient of
 * the same layer.  Arguments as mmg_bn_bwd_apply (sums NULL: eval mode; pro->scale NULL: no BatchNorm, relu / dropout only);
 * relu only.  mmg_linear_bnbwd_supported: M > 512, K and N in {64, 128} (one workgroup spans all N columns). */
int mmg_linear_bnbwd_supported(int64_t M, int N, int K);
int mmg_linear_bnbwd(const float* G, const float* Y, const mmg_prologue_t* pro, const float* mean, const float* rstd,
                     const double* sums, double inv_count, float* dbeta, float* dgamma, const float* W, float* dZ,
                     float* dX, int64_t M, int N, int K, void* stream);

/* ... and mmg_bn_bwd_apply2 (two upstream gradients, own dropout masks) the same way; K = N = 128. */
int mmg_linear_bnbwd2(const float* G, const float* G2, const float* Y, const mmg_prologue_t* pro, const mmg_prologue_t* pro2,
                      const float* mean, const float* rstd, const double* sums, double inv_count, float* dbeta,
                      float* dgamma, const float* W, float* dZ, float* dX, int64_t M, int N, int K, void* stream);

/* ... and the row-list form (mmg_bn_bwd_apply with G = NULL + mmg_bn_bwd_apply_rows): G_rows [n_sel, K] holds the listed
 * rows of the upstream gradient back to back, row_pos[row] (int32, [M]) = position of `row` in that list or -1. */
int mmg_linear_bnbwd_rows(const float* G_rows, const int32_t* row_pos, int64_t n_sel, const float* Y,
                          const mmg_prologue_t* pro, const float* mean, const float* rstd, const double* sums,
                          double inv_count, float* dbeta, float* dgamma, const float* W, float* dZ, float* dX, int64_t M,
                          int N, int K, void* stream);

/* mmg_l2norm_bwd folded into the data-gradient GEMM of the linear in front of the normalisation the same way:
 * dZ = rnorm * (G - out * <G, out>) (0 dot product where the norm was clamped), dX = dZ . W, W stored [K,N].  Shapes as
 * mmg_linear_bnbwd_supported. */
int mmg_linear_l2bwd(const float* G, const float* out, const float* rnorm, const float* W, float* dZ, float* dX,
                     int64_t M, int N, int K, float eps, void* stream);

/* Two upstream gradients through the SAME BatchNorm + ReLU, each with its own dropout mask (pro2: only its dropout fields
 * are used) -- the two encode_nodes passes of a training step (src/model.py:294 and :301 -> :251) see the same
 * Linear + BatchNorm1d in front of their first Dropout (model.py:93-96), i.e. they share that layer:
 * g_out = g_out(G; pro) + g_out(G2; pro2), one statistics pass and one apply pass instead of two of each. */
int mmg_bn_bwd_stats2(const float* G, const float* G2, const float* Y, const mmg_prologue_t* pro,
                      const mmg_prologue_t* pro2, const float* mean, const float* rstd, double* sums, int64_t M, int N,
                      void* ws, size_t ws_bytes, void* stream);
int mmg_bn_bwd_apply2(const float* G, const float* G2, const float* Y, const mmg_prologue_t* pro,
                      const mmg_prologue_t* pro2, const float* mean, const float* rstd, const double* sums,
                      double inv_count, float* dbeta, float* dgamma, float* dY, int64_t M, int N, void* stream);

/* The same backward for an upstream gradient that is zero outside a short list of rows (G_rows [n_sel, N] holds the rows
 * rows[s]; masks are those of the ORIGINAL rows):  mmg_bn_bwd_stats_rows gives the sums of pass 1 from the listed rows
 * alone; mmg_bn_bwd_apply with G = NULL writes the dense part of pass 2; mmg_bn_bwd_apply_rows adds scale * g_out to the
 * listed rows of dY (rows must be distinct). */
size_t mmg_bn_bwd_stats_rows_ws_bytes(int N);
int mmg_bn_bwd_stats_rows(const float* G_rows, const float* Y, const int64_t* rows, int64_t n_sel,
                          const mmg_prologue_t* pro, const float* mean, const float* rstd, double* sums, int N,
                          void* ws, size_t ws_bytes, void* stream);
int mmg_bn_bwd_apply_rows(const float* G_rows, const float* Y, const int64_t* rows, int64_t n_sel,
                          const mmg_prologue_t* pro, float* dY, int N, void* stream);

/* The statistics pass of a BatchNorm backward taken from the kernel that PRODUCES its upstream gradient.
 * mmg_bn_bwd_stats (and _stats2) read the [M, N] upstream gradient and the [M, N] pre-BatchNorm activation once more; a
 * producer that is handed this descriptor sums them from its output tile while it is still in registers and only reads
 * Y.  `sums` receives exactly what mmg_bn_bwd_stats(G = the producer's output, y, pro, mean, rstd) would, summed in a
 * fixed order (16 rows in fp32, the rest in fp64); with accumulate != 0 that is ADDED to `sums` -- the two sums are
 * linear in G, so two producers whose outputs go through the same BatchNorm with their own dropout masks (the two
 * encode_nodes passes of a training step, mmg_bn_bwd_stats2) each add their share.
 * Autograd of  nn.Linear -> BatchNorm1d -> ReLU -> Dropout  chains (src/model.py:93-101, 258-269): the gradient a layer
 * hands down is the upstream gradient of the BatchNorm below it.
 *   mmg_linear_fwd_next_bn          dX = dY . W  (MMG_LIN_W_KN) of a plain linear, e.g. the heads' first layer; no prologue,
 *                                   no accumulate;  M > 512, N % 128 == 0, K in {64, 128}
 *   mmg_linear_l2bwd_next_bn        dX of mmg_linear_l2bwd;  K = N = 128
 *   mmg_linear_bnbwd_next_bn        dX of mmg_linear_bnbwd;  K = N = 128
 *   mmg_linear_bnbwd_rows_next_bn   dX of mmg_linear_bnbwd_rows;  K = N = 128
 *   mmg_gather_rows_next_bn         the final `out` of mmg_gather_rows (accumulate or not);  the bit-plane layouts, D >= 128
 * next == NULL: the plain entry point.  A shape (or activation) outside the list above runs the producer followed by
 * the separate statistics pass: the result is defined for everything the plain entry point accepts. */
typedef struct {
  const float* y;                 /* [M, N] pre-BatchNorm activation of the layer below */
  const mmg_prologue_t* pro;      /* its fold (scale / shift), activation (none | relu) and dropout */
  const float* mean;              /* [N] */
  const float* rstd;              /* [N] */
  double* sums;                   /* [2, N] out (in / out with accumulate) */
  int accumulate;
  void* ws;                       /* >= mmg_next_bn_ws_bytes(M, N) */
  size_t ws_bytes;
} mmg_next_bn_t;
size_t mmg_next_bn_ws_bytes(int64_t M, int N);
int mmg_linear_fwd_next_bn(const float* X, const mmg_prologue_t* pro, const float* W, const float* bias, float* Y, int64_t M,
                           int N, int K, int flags, const mmg_next_bn_t* next, void* stream);
int mmg_linear_l2bwd_next_bn(const float* G, const float* out, const float* rnorm, const float* W, float* dZ, float* dX,
                             int64_t M, int N, int K, float eps, const mmg_next_bn_t* next, void* stream);
int mmg_linear_bnbwd_next_bn(const float* G, const float* Y, const mmg_prologue_t* pro, const float* mean, const float* rstd,
                             const double* sums, double inv_count, float* dbeta, float* dgamma, const float* W, float* dZ,
                             float* dX, int64_t M, int N, int K, const mmg_next_bn_t* next, void* stream);
int mmg_linear_bnbwd_rows_next_bn(const float* G_rows, const int32_t* row_pos, int64_t n_sel, const float* Y,
                                  const mmg_prologue_t* pro, const float* mean, const float* rstd, const double* sums,
                                  double inv_count, float* dbeta, float* dgamma, const float* W, float* dZ, float* dX,
                                  int64_t M, int N, int K, const mmg_next_bn_t* next, void* stream);
int mmg_gather_rows_next_bn(const mmg_rel_t* rels, int n_rel, int64_t n_rows, int D, float* out, int accumulate,
                            const mmg_next_bn_t* next, void* stream);

/* Measurement hook (bench.py).  After mmg_probe_arm(n) the next n launches of the big kernels (from any host thread: the
 * backward of a step runs on the autograd engine's thread) carry a HIP start / stop event pair on the kernel itself (hipExtLaunchKernelGGL: the kernel's own begin / end
 * timestamps on its stream -- what rocprofv3 reports -- not a pair of extra queue entries around it).  mmg_probe_read
 * waits for them, disarms the hook and returns how many entries it wrote (<= cap):
 *   ms = kernel duration, tag = MMG_PROBE_* family, (M, N, K) = the launch's shape: rows / output width / inner width for
 *   the dense kernels; patient rows / D / total vocab rows of the fused relations for the aggregates; pairs / 0 / 0 for
 *   the heads;  flags: 1 = accumulate, 4 = prologue, 8 = rowscale.
 * The hook is the library's only process-wide mutable state (mutex-guarded); the product path never arms it. */
#define MMG_PROBE_LINEAR_FWD 1
#define MMG_PROBE_LINEAR_WGRAD 2
#define MMG_PROBE_LINEAR_WGRAD_REDUCE 3
#define MMG_PROBE_GATHER 4
#define MMG_PROBE_SCATTER 5
#define MMG_PROBE_SCATTER_REDUCE 6
#define MMG_PROBE_PAIR_FWD 7
#define MMG_PROBE_PAIR_BWD 8
#define MMG_PROBE_BN_BWD_STATS 9
#define MMG_PROBE_BN_BWD_APPLY 10
#define MMG_PROBE_ELEMENTWISE 11
int mmg_probe_arm(int n_launches);
#define MMG_PROBE_NAME_LEN 128
/* names (nullable): cap * MMG_PROBE_NAME_LEN bytes; entry i receives the instantiated kernel symbol of launch i, e.g.
 * "k_linear_bnbwd_x6<128, 4, 0>" -- the name rocprofv3 lists the same launch under. */
int mmg_probe_read(float* ms, int* tag, int64_t* M, int* N, int* K, int* flags, char* names, int cap);

/* Row L2 normalisation, F.normalize(p=2, dim=1, eps): out = z / max(||z||, eps); rnorm = 1/max(..) */
int mmg_l2norm_fwd(const float* Z, float* out, float* rnorm, int64_t M, int N, float eps, void* stream);
/* dz = rnorm * (g - out * <out, g>)   (rows whose norm hit eps: dz = g * rnorm)     */
int mmg_l2norm_bwd(const float* G, const float* out, const float* rnorm, float* dZ, int64_t M, int N,
                   float eps, void* stream);

/* Weighted, masked regression loss over the prediction pairs and its gradient in ONE pass
 * (src/train.py:366-386 of the reference: mean(|p - y| * w[lab]) over the supervision subset):
 *   loss = inv_den * sum_k sup[k] * w[k] * (|p_k - y_k|  or  (p_k - y_k)^2)        (fp64 accumulation)
 *   dpred[k] = inv_den * sup[k] * w[k] * (sign(p_k - y_k)  or  2 (p_k - y_k))
 * sup / w may be NULL (= 1).  loss_type 0 = mae, 1 = mse, 2 = huber with delta 1 (compute_regression_loss,
 * src/model.py:579-612: 0.5 d^2 for |d| <= 1, |d| - 0.5 beyond).  `loss` is ONE double on the device.
 * inv_den_ptr (nullable, DEVICE): when non-NULL the normaliser is read from it at run time instead of `inv_den` -- the
 * reference divides by the size of the per-epoch supervision subset (.mean() over pred[mask], train.py:366-386), which
 * changes every epoch while a captured hipGraph keeps its launch arguments. */
size_t mmg_pair_loss_ws_bytes(int64_t n);
int mmg_pair_loss(const float* pred, const float* y, const float* w, const float* sup, int64_t n, double inv_den,
                  const double* inv_den_ptr, int loss_type, float* dpred, double* loss, void* ws, size_t ws_bytes,
                  void* stream);

/* The per-epoch supervision subset (src/train.py:150-176: supervision_mask = torch.rand(n) < mask_fraction, redrawn every
 * epoch from a wall-clock seed) drawn on the device: sup[k] = 1.0f with probability `fraction` (quantised to 1/65536),
 * else 0.0f, from the counter RNG keyed on (seed, a site of its own, ids[k] or k) -- partition-invariant when ids holds
 * global pair ids.  seed_ptr (nullable, DEVICE) overrides `seed` at run time (the dropout seed stream a captured step
 * advances).  count / inv_den (nullable, DEVICE doubles) receive the subset size and 1 / max(size, 1) -- the normaliser
 * mmg_pair_loss reads through inv_den_ptr.  sup == NULL: count only -- a patient-sharded rank draws the mask of ITS pairs
 * (ids = their global ids) and counts the subset of ALL n_global pairs (ids NULL) itself: the draw is a pure function of
 * (seed, id), so every rank gets the global size without a collective. */
size_t mmg_sup_mask_ws_bytes(int64_t n);
int mmg_sup_mask_draw(const uint64_t* seed_ptr, uint64_t seed, const int64_t* ids, int64_t n, float fraction, float* sup,
                      double* count, double* inv_den, void* ws, size_t ws_bytes, void* stream);

/* keep-mask of the dropout RNG, for injected-mask parity tests: mask[i] in {0,1}     */
int mmg_dropout_mask(uint64_t seed, const uint64_t* seed_ptr, uint32_t site, int64_t first_elem, int64_t n_elems,
                     float p, uint8_t* mask, void* stream);

/* ---------------------------------------------------------------------------------------
 * Degree-gated dual edge head (src/model.py:305-333, EdgeRegressionHead :342-396).
 * The first Linear(2D,64) is split: A = x_P . W1[:, :D]^T (per patient), B = x_lab . W1[:, D:]^T + b1
 * (per lab) -- computed by mmg_linear_fwd -- so that per pair
 *   h1 = drop(relu(A[pi] + B[li])); h2 = drop(relu(W2 h1 + b2)); pred = W3 h2 + b3
 * edge_predictor runs on the final embeddings, tabular_mlp on the initial ones; the head is
 * chosen per pair by deg[pi] < degree_threshold (src/model.py:312-315).
 * ------------------------------------------------------------------------------------- */
typedef struct {
  const float* A;          /* [n_patients, 64] */
  const float* B;          /* [n_labs, 64] (bias b1 folded in) */
  const float* W2;         /* [32, 64] */
  const float* b2;         /* [32] */
  const float* W3;         /* [32] */
  const float* b3;         /* [1] */
} mmg_head_t;

typedef struct {
  float* dA;               /* [n_patients, 64]  accumulated (+=) */
  float* dB;               /* [n_labs, 64]      accumulated (+=) */
  float* dW2;              /* [32, 64] += */
  float* db2;              /* [32] += */
  float* dW3;              /* [32] += */
  float* db3;              /* [1] += */
} mmg_head_grad_t;

/* One launch evaluates ONE head on the pairs whose gate matches `want_low`
 * (want_low = 1: deg[pi] < degree_threshold -> tabular_mlp; 0: the GNN edge_predictor);
 * other pairs are left untouched in pred / contribute nothing to the gradients.
 * pair_id (nullable) = original position of each pair, used only to key the dropout RNG so that
 * a permuted (patient-sorted) pair list draws the same masks.  seed_ptr (nullable, device): overrides
 * `seed` at run time (hipGraph replays).
 * io_perm (nullable, device): the kernels work on a patient-SORTED pair list; io_perm[k] is the position of sorted
 * pair k in the caller's order -- pred is written to pred[io_perm[k]] and dpred read from dpred[io_perm[k]], so no
 * separate permutation pass over the predictions / their gradient is needed.
 * sel / n_sel (both nullable, device): a compacted list of pair positions built by mmg_pair_select -- only
 * sel[0 .. *n_sel) are visited (in list order) and `n_pairs` is then an upper bound of *n_sel that sizes the
 * launch.  Forward: the per-head lists (static per pair set) replace the predicated sweep over all pairs.
 * Backward: pairs whose upstream gradient is exactly 0 (the ~80 % of train pairs outside the supervision
 * subset, src/train.py:366-370) contribute exactly 0 to every gradient and are skipped. */
/* Sizes: n_total = length of pi / li / pair_id / io_perm / pred / dpred; n_patients = rows of A and entries of deg;
 * n_labs = rows of B; n_pairs <= n_total sizes the launch (the list bound, or n_total without a list).  Every indexed
 * access inside the kernels is range-checked against these (buffer descriptors): an index outside its array reads 0 /
 * is not written, it can never become an address.  Limits: n_total < 2^29, n_patients < 2^24, n_labs < 2^24. */
int mmg_pair_head_fwd(const mmg_head_t* head, const int32_t* pi, const int32_t* li,
                      const int32_t* deg, int degree_threshold, int want_low, int64_t n_pairs, int64_t n_total,
                      int64_t n_patients, int n_labs,
                      float drop_p, uint64_t seed, const uint64_t* seed_ptr, const int64_t* pair_id,
                      float* pred, const int32_t* sel, const int32_t* n_sel, const int64_t* io_perm, void* stream);
/* What autograd would save for the backward of the head (model.py:388-396), per pair k of the pair arrays, written by
 * mmg_pair_head_fwd_save for every pair it visits and read by mmg_pair_head_bwd_saved for every pair IT visits (a subset:
 * same head, same gate, pairs with a non-zero upstream gradient):
 *   h1_bits[2 k + w] bit j = [h1[32 w + j] > 0] -- the first layer's activation survived ReLU and dropout, so
 *                            h1 = bit ? (A[pi] + B[li]) / (1 - p) : 0 exactly;
 *   h2[32 k + u]           = the second layer's activation after ReLU and dropout (its sign pattern is the mask).
 * The backward then needs no RNG, no second-layer product and no epilogue arithmetic per pair (136 B per pair instead);
 * NULL `saved` = the plain entry points: the backward recomputes -- in the forward's own order, so both variants return
 * the same bits.  Up to 64 labs on the backward side (beyond: recomputed). */
typedef struct {
  uint32_t* h1_bits;      /* [n_entries, 2] */
  float* h2;              /* [n_entries, 32] */
  int by_position;        /* != 0: entry = position in the pair list `sel` instead of the pair index k -- dense, streamed
                           * writes and reads; the backward must then run over the SAME list as the forward */
  int64_t n_entries;      /* rows of both buffers: >= n_total (indexed by pair), >= n_pairs -- the launch bound of the list
                           * -- when by_position; checked by both entry points (MMG_ERR_ARG) */
} mmg_pair_saved_t;
int mmg_pair_head_fwd_save(const mmg_head_t* head, const int32_t* pi, const int32_t* li,
                           const int32_t* deg, int degree_threshold, int want_low, int64_t n_pairs, int64_t n_total,
                           int64_t n_patients, int n_labs,
                           float drop_p, uint64_t seed, const uint64_t* seed_ptr, const int64_t* pair_id,
                           float* pred, const int32_t* sel, const int32_t* n_sel, const int64_t* io_perm,
                           const mmg_pair_saved_t* saved, void* stream);
/* Backward: the weight-side gradients (dW2, db2, dW3, db3, dB) leave every workgroup as ONE partial slab in `ws` and are
 * summed in fixed order (bitwise reproducible); dA rows are flushed per patient run (pairs sorted by patient: a row
 * receives at most two partial sums unless one patient holds more than 32 listed pairs). */
size_t mmg_pair_head_bwd_ws_bytes(int64_t n_pairs, int n_labs);
int mmg_pair_head_bwd(const mmg_head_t* head, const mmg_head_grad_t* grad,
                      const int32_t* pi, const int32_t* li, const int32_t* deg, int degree_threshold,
                      int want_low, int64_t n_pairs, int64_t n_total, int64_t n_patients, int n_labs, float drop_p,
                      uint64_t seed,
                      const uint64_t* seed_ptr, const int64_t* pair_id, const float* dpred,
                      const int32_t* sel, const int32_t* n_sel, const int64_t* io_perm,
                      void* ws, size_t ws_bytes, void* stream);
int mmg_pair_head_bwd_saved(const mmg_head_t* head, const mmg_head_grad_t* grad,
                            const int32_t* pi, const int32_t* li, const int32_t* deg, int degree_threshold,
                            int want_low, int64_t n_pairs, int64_t n_total, int64_t n_patients, int n_labs, float drop_p,
                            uint64_t seed,
                            const uint64_t* seed_ptr, const int64_t* pair_id, const float* dpred,
                            const int32_t* sel, const int32_t* n_sel, const int64_t* io_perm,
                            const mmg_pair_saved_t* saved, void* ws, size_t ws_bytes, void* stream);

/* Stable two-way compaction of pair positions by head: position k goes to sel_low if deg[pi[k]] < threshold,
 * else to sel_high -- and only if dpred is NULL or dpred[k] != 0.  Order inside a list = pair order (pairs sorted
 * by patient stay sorted).  counts[0], counts[1] (device) = list lengths.  sel_low / sel_high: capacity n each.
 * dpred is read through io_perm (nullable); dpred_sorted (nullable, [n]) receives dpred in pair order -- the ONE
 * random pass over the gradient -- for mmg_pair_head_bwd to read sequentially (with io_perm = NULL). */
size_t mmg_pair_select_ws_bytes(int64_t n_pairs);
int mmg_pair_select(const int32_t* pi, const int32_t* deg, int degree_threshold, const float* dpred,
                    const int64_t* io_perm, float* dpred_sorted, int64_t n_pairs, int32_t* sel_low, int32_t* sel_high,
                    int32_t* counts, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * Grouped launches for the vocab side (tables of 50 .. 200 rows: every lin_l / lin_r of a SAGEConv into a vocab type, the
 * transformed tables the patient-side gather reads, their weight / data gradients -- src/model.py:125-131,256 -- is a few
 * microseconds of work behind a launch).  One launch runs up to MMG_SMALL_MAX independent problems of one (N, K).
 *   mmg_small_fwd_group  : Y[M,N] (+)= X[M,K] . W^T (+ X2[M,K] . W2^T) + bias ;  flags as mmg_linear_fwd
 *                          (MMG_LIN_ACCUMULATE, MMG_LIN_W_KN: both W and W2 stored [K,N]);  exact fp32 products
 *   mmg_small_wgrad_group: dW[N,K] (+)= dY[M,N]^T . X[M,K] ;  dbias[N] (nullable) (+)= column sums of dY
 * M <= 4096 per problem (M = 0: skipped by the forward, zero / untouched gradient by the weight gradient).
 * ------------------------------------------------------------------------------------- */
#define MMG_SMALL_MAX 8
typedef struct {
  const float* X; const float* W;      /* [M,K], [N,K] (or [K,N]) */
  const float* X2; const float* W2;    /* optional second term, both NULL or both set */
  const float* bias;                   /* [N] or NULL */
  float* Y;                            /* [M,N] */
  int64_t M;
  int flags;
} mmg_small_fwd_t;
typedef struct {
  const float* dY; const float* X;     /* [M,N], [M,K] */
  float* dW; float* dbias;             /* [N,K], [N] or NULL */
  int64_t M;
  int accumulate;
} mmg_small_wgrad_t;
int mmg_small_fwd_group(const mmg_small_fwd_t* probs, int n_probs, int N, int K, void* stream);
int mmg_small_wgrad_group(const mmg_small_wgrad_t* probs, int n_probs, int N, int K, void* stream);

/* The per-type epilogue of a HeteroConv layer (src/model.py:258-269: BatchNorm1d -> activation -> Dropout) for every
 * SMALL node type in one launch, and its backward in another.  Per problem (M <= 4096 rows):
 *   forward : training: batch statistics (fp64 sums) -> scale / shift / mean / rstd written to stats_out[4,N] (the layout
 *             mmg_bn_finalize produces), running statistics advanced once (unbiased variance, `momentum`);  eval: folded
 *             from the running statistics;  gamma == NULL: no BatchNorm.  out = dropout(act(Y * scale + shift)).
 *   backward: g' = G * act'(.) * keep / (1 - p);  training: dY = scale (g' - mean g' - xhat mean(g' xhat)), dbeta = sum g',
 *             dgamma = sum g' xhat;  eval: dY = scale g';  scale == NULL: dY = g'.
 * Dropout masks are the ones mmg_affine_act_drop draws for the same (seed, site, row_offset). */
typedef struct {
  const float* Y; float* out;           /* [M,N] */
  const float* gamma; const float* beta; float* running_mean; float* running_var;   /* [N]; gamma NULL = no BatchNorm */
  float* stats_out;                     /* [4,N]: scale | shift | mean | rstd */
  int64_t M;
  int training;
  int act;                              /* MMG_ACT_* */
  float drop_p;
  uint64_t seed; uint32_t site; int64_t row_offset; const uint64_t* seed_ptr;
} mmg_small_bn_t;
typedef struct {
  const float* G; const float* Y; float* dY;                                  /* [M,N] */
  const float* scale; const float* shift; const float* mean; const float* rstd;   /* [N]; scale NULL = no BatchNorm */
  float* dbeta; float* dgamma;                                                 /* [N], nullable */
  int64_t M;
  int training;
  int act;
  float drop_p;
  uint64_t seed; uint32_t site; int64_t row_offset; const uint64_t* seed_ptr;
} mmg_small_bn_bwd_t;
int mmg_small_bn_act_group(const mmg_small_bn_t* probs, int n_probs, int N, float momentum, float eps, void* stream);
int mmg_small_bn_bwd_group(const mmg_small_bn_bwd_t* probs, int n_probs, int N, void* stream);

/* ---------------------------------------------------------------------------------------
 * Optimizer step (src/train.py:219,390: torch.optim.Adam(model.parameters()).step()) and small vector sums.
 * mmg_adam_step updates EVERY parameter in one launch: p / m / v are flat fp32 buckets holding the tensors back to back
 * at offsets[0 .. n_tensors] (ascending, offsets[n_tensors] = end); grads[i] (HOST array of device pointers; NULL = the
 * tensor received no gradient and is left untouched, as torch skips a .grad of None) is read where the backward kernels
 * wrote it.  Arithmetic of torch.optim.Adam (amsgrad / maximize off, L2 weight decay added to the gradient).  `step`
 * (device float, the number of steps taken so far) is advanced by a one-thread launch behind the update -- hipGraph
 * replays keep counting.  `ticket` (device uint32) is no longer used (the last-workgroup scheme it served cost a
 * device-scope fence per workgroup) and stays in the signature for the ABI.
 * mmg_vec_sums: dst_j = src_j0 (+ src_j1 + src_j2 + src_j3), fixed order, <= MMG_SUM_MAX_JOBS jobs per launch (the
 * three lin_r weights / lin_l biases that share x_patient in a HeteroConv layer, src/model.py:125-131; the gradient
 * contributions of a parameter that is used twice).  A job is a [len / cols, cols] matrix with its own row stride on
 * every side (cols = 0: a flat vector), so the same launch also takes the two halves of an edge head's first-layer
 * weight W1[:, :D] | W1[:, D:] apart (src/model.py:375-382: the head's Linear(2D, H)) and puts their gradients together.
 * mmg_counters_add: *counters[i] += incs[i] for <= MMG_SUM_MAX_JOBS * 4 int64 counters (BatchNorm1d.num_batches_tracked
 * of every layer in one launch).  mmg_seed_advance: state[1] += 1 step of a SplitMix64 stream, state[0] = its output
 * (< 2^62) -- the dropout seed the kernels read through mmg_prologue_t.seed_ptr, advanced inside a captured step.
 * mmg_fill_zero: zero-fill by a kernel on `stream` (bytes of any count / alignment).  Not hipMemsetAsync: recorded into a
 * hipGraph its node replays a pattern other than the recorded zero on this ROCm (profiles/probes/hipgraph_memset_node.py).
 * ------------------------------------------------------------------------------------- */
#define MMG_ADAM_MAX_TENSORS 96
#define MMG_SUM_MAX_JOBS 8
int mmg_adam_step(float* p, float* m, float* v, const float* const* grads, const int32_t* offsets, int n_tensors,
                  float lr, float beta1, float beta2, float eps, float weight_decay, float* step, uint32_t* ticket,
                  void* stream);
/* The same step with its hyper-parameters read from DEVICE memory at run time: hyper[5] = (lr, beta1, beta2, eps,
 * weight_decay).  A captured hipGraph keeps its launch arguments, so a learning-rate scheduler (src/train.py:271-291:
 * ReduceLROnPlateau / StepLR acting on optimizer.param_groups) would be silently ignored by mmg_adam_step under replay;
 * the caller refreshes `hyper` between replays instead. */
int mmg_adam_step_dev(float* p, float* m, float* v, const float* const* grads, const int32_t* offsets, int n_tensors,
                      const float* hyper, float* step, uint32_t* ticket, void* stream);
typedef struct {
  float* dst;
  const float* src[4];
  int n_src;               /* 1..4 */
  int len;                 /* elements */
  int cols;                /* 0: flat;  > 0: rows of `cols` elements (len % cols == 0) with the strides below */
  int ld_dst;              /* row stride of dst (elements) */
  int ld_src[4];           /* row stride of every source */
} mmg_sum_job_t;
int mmg_vec_sums(const mmg_sum_job_t* jobs, int n_jobs, void* stream);
#define MMG_COUNTERS_MAX 32
int mmg_counters_add(int64_t* const* counters, const int64_t* incs, int n, void* stream);
int mmg_seed_advance(uint64_t* state /* [2]: seed | stream position */, void* stream);
int mmg_fill_zero(void* ptr, size_t bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * Evaluation reducers on the device (src/evaluate.py:36-82 metrics, :417-440 per-lab +-3 sigma winsorisation, :89-141
 * per-lab rows, :237-342 stratifications).  Every figure evaluate_model reports is a segment sum over the prediction
 * pairs (segment = lab index, patient-degree bucket, lab-frequency bucket; seg[k] outside [0, n_seg) = not counted), so
 * the predictions stay on the device and [n_seg, 8] doubles come back.
 *   mmg_seg_moments: moments[s] = (n, sum r, sum r^2), r = pred - target.
 *   mmg_seg_metrics: residuals clipped to mean +- n_sigma * std of their segment (population std, segments with > 1
 *     sample; n_sigma <= 0 or moments NULL: none), adjusted prediction = target + clipped residual (written to
 *     pred_out when non-NULL); sums[s] = (n, sum |e|, sum e^2, sum t, sum t^2, sum |e/t| over t != 0, count(t != 0),
 *     count(clipped)) with e = t - adjusted prediction.  fp64 accumulation.  n_seg <= 2048.
 * ------------------------------------------------------------------------------------- */
size_t mmg_seg_reduce_ws_bytes(int64_t n, int n_seg);
int mmg_seg_moments(const float* pred, const float* target, const int64_t* seg, int64_t n, int n_seg,
                    double* moments, void* ws, size_t ws_bytes, void* stream);
int mmg_seg_metrics(const float* pred, const float* target, const int64_t* seg, int64_t n, int n_seg,
                    const double* moments, float n_sigma, float* pred_out, double* sums, void* ws, size_t ws_bytes,
                    void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MMGNN_H */
