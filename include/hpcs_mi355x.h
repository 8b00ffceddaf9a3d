/* libhpcs_mi355x.so -- C ABI of the MI355X (gfx950) kernels behind the HPCS operator API.
 *
 * The reference (TheCrossProduct/HPCS) has no FFI boundary: its hot path is plain
 * torch modules/functions.  This library is the native layer the build adds UNDER
 * that Python surface; each entry point names the reference
 * code it replaces (paths relative to the reference repository root).
 *
 * Conventions (all entry points):
 *   - raw DEVICE pointers to contiguous fp32 / int64 / int32 buffers owned by the caller;
 *   - enqueue-only on `stream` (a hipStream_t passed as void*): no allocation, no host
 *     sync, no global state; re-entrant;
 *   - returns HPCS_OK or a negative hpcs_status; HIP launch errors come back as
 *     HPCS_ERR_HIP - hipError_t;
 *   - tensors keep the reference's layouts: vector-neuron features are [B, C, 3, L]
 *     with L = N (points) or N*k (edges), L contiguous.
 */
#ifndef HPCS_MI355X_H
#define HPCS_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* hpcs_stream_t;

enum hpcs_status {
    HPCS_OK = 0,
    HPCS_ERR_SHAPE = -1,       /* negative/zero/inconsistent dimension */
    HPCS_ERR_UNSUPPORTED = -2, /* e.g. k > 64 */
    HPCS_ERR_WORKSPACE = -3,   /* workspace too small */
    HPCS_ERR_NULL = -4,        /* required pointer is NULL */
    HPCS_ERR_HIP = -1000       /* HPCS_ERR_HIP - hipError_t */
};

int hpcs_version(void);

/* ---- a1: knn(x, k)  hpcs/nn/dgcnn/utils/vn_dgcnn_util.py:4-10 ------------------------------
 * x [B,D,N] -> idx [B,N,k] int64: the k largest of  (-|xj|^2 - (-2 xi.xj)) - |xi|^2  per row,
 * nearest first, ties by ascending index.  Rounding sequence = torch CPU. k <= 64. */
/* ws (hpcs_knn_workspace bytes, may be NULL): |x|^2 scratch that enables the MFMA kernel for D in {3,63}, k<=40. */
size_t hpcs_knn_workspace(int B, int N);
int hpcs_knn(const float* x, int B, int D, int N, int k, int64_t* idx, void* ws, size_t ws_bytes,
             hpcs_stream_t stream);

/* ---- a2: get_graph_feature  hpcs/nn/dgcnn/utils/vn_dgcnn_util.py:13-41 ----------------------
 * x [B,C,3,N], idx [B,N,k] (cloud-local) -> out [B,2C,3,N,k] = (x[nbr]-x[ctr], x[ctr]).
 * bwd: gx [B,C,3,N] is overwritten.  N <= 1024, N*k <= 20480, k % 4 == 0: the scatter-add runs as a GATHER over the
 * transposed graph, which is built in `ws` on every call (no atomics, fixed summation order, bit-reproducible);
 * otherwise the neighbour part goes through a point-major scratch with row-contiguous fp32 atomics.
 * `ws`: hpcs_edge_gather_bwd_workspace bytes (covers both). */
int hpcs_edge_gather_fwd(const float* x, const int64_t* idx, int B, int C, int N, int k, float* out,
                         hpcs_stream_t stream);
size_t hpcs_edge_gather_bwd_workspace(int B, int C, int N);
int hpcs_edge_gather_bwd(const float* gout, const int64_t* idx, int B, int C, int N, int k, float* gx, void* ws,
                         size_t ws_bytes, hpcs_stream_t stream);

/* ---- channel mixing (nn.Linear over the channel axis / Conv1d k=1) -----------------------------
 * replaces map_to_feat / map_to_dir of vn_layers.py:14-24,48-77 and the Conv1d's of
 * vn_dgcnn_partseg.py:40-57.  fp32 MFMA (v_mfma_f32_32x32x2_f32), exact k-ordered fma chain.
 *
 * hpcs_gemm_tn:  C[b] (M x Nc) (+)= At^T (At is K x M, row stride lda) * Bm[b] (K x Nc, row stride ldb)
 *                + rowbias[b][m] (optional).  Batch index b = bo*n_inner + bi addresses
 *                Bm + bo*strideB_o + bi*strideB_i (same for C), so a [B,C,3,N] tensor can be
 *                walked as 3B matrices of N columns.
 * hpcs_gemm_nt:  dW (M x Kx, row stride lddw) (+)= sum_b G[b] (M x Nc) * X[b]^T (Kx x Nc).
 *                Deterministic split reduction through `ws` (hpcs_gemm_nt_workspace bytes). */
int hpcs_gemm_tn(const float* At, int lda, const float* Bm, int ldb, int64_t strideB_o, int64_t strideB_i,
                 float* C, int ldc, int64_t strideC_o, int64_t strideC_i, int M, int Nc, int K, int n_outer,
                 int n_inner, const float* rowbias, int accumulate, hpcs_stream_t stream);
/* hpcs_gemm_tn_x3: the same product with the same arguments, evaluated on the BF16 matrix cores at fp32 accuracy: every
 * fp32 operand splits exactly into three bf16 pieces and six piece products (exact in fp32, accumulated in fp32 by the MFMA)
 * carry the product to 2^-23 |a||b| (csrc/gemm_x3.hip).  Not bit-identical to the fma chain of hpcs_gemm_tn -- as close to
 * the exact product (tests).  Used for the large dense layers (M > 64); kNN and the EdgeConv mixes keep the fma chain.
 * Its loads address one operand matrix with 32-bit byte offsets: HPCS_ERR_UNSUPPORTED when (K + 16) * lda * 4 or
 * (K + 16) * ldb * 4 reaches 2^31 (hpcs_gemm_tn routes such shapes to the fma chain itself). */
int hpcs_gemm_tn_x3(const float* At, int lda, const float* Bm, int ldb, int64_t strideB_o, int64_t strideB_i,
                    float* C, int ldc, int64_t strideC_o, int64_t strideC_i, int M, int Nc, int K, int n_outer,
                    int n_inner, const float* rowbias, int accumulate, hpcs_stream_t stream);
/* prep_weights: W = [W1 (M1 x K, row stride ld1); W2 (M2 x K, row stride ld2, optional)] -> both GEMM operands in one
 * launch: At [K, lda >= M] = W^T and Wp [M, ldp >= K] = W, zero padded (lda, ldp multiples of 4 for 16-byte loads). */
int hpcs_prep_weights(const float* W1, int ld1, int M1, const float* W2, int ld2, int M2, int K, float* At, int lda,
                      float* Wp, int ldp, hpcs_stream_t stream);
/* EdgeConv weights at point level (W [x_nbr - x_ctr; x_ctr] = Wa x_nbr + (Wb - Wa) x_ctr, vn_dgcnn_util.py:33-39 + vn_layers.py:48-77):
 * graph_wuv: W [Ctot, 2C] (row stride ldw; difference | centre columns) -> Wuv [2Ctot, C] = [Wa ; Wb - Wa].
 * graph_wgrad_fold: gWuv [2Ctot, C] -> dW = [gU - gV | gV] ADDED into gWf [Cout, 2C] (rows < Cout) and gWd [Ctot-Cout, 2C]
 * (either may be NULL: those rows are skipped). */
int hpcs_graph_wuv(const float* W, int ldw, int Ctot, int C, float* Wuv, hpcs_stream_t stream);
int hpcs_graph_wgrad_fold(const float* gWuv, int Ctot, int C, int Cout, float* gWf, float* gWd, hpcs_stream_t stream);
/* prep_weights_multi: the same for n weight matrices in ONE launch (a training step re-lays out every layer's weights once,
 * after the optimiser moved them): desc_dev[n] on the device, pointers as 64-bit integers. */
typedef struct {
    uint64_t W1, W2, At, Wp;      /* W2 = 0: no second block */
    int32_t ld1, M1, ld2, M2, K, lda, ldp, pad_;
} hpcs_prep_desc;
int hpcs_prep_weights_multi(const hpcs_prep_desc* desc_dev, int n, hpcs_stream_t stream);
/* hpcs_gemm_tn_graph: PD[b] (M x 3*N*k) = At^T (M x 2C) @ E[b], E = get_graph_feature(feat, idx=idx) (vn_dgcnn_util.py:10-37:
 * rows [0,C) = x_j - x_i, rows [C,2C) = x_i) formed inside the operand loader -- the [B,2C,3,N,k] edge tensor never exists in
 * HBM.  Same values and the same k-ordered fma chain as hpcs_edge_gather_fwd followed by hpcs_gemm_tn (bit-identical output).
 * feat [B,C,3,N], idx [B,N,k] int64 neighbour ids in [0,N); M <= 64, 2C <= 64, k % 4 == 0, lda % 4 == 0 (else
 * HPCS_ERR_UNSUPPORTED). */
size_t hpcs_gemm_tn_graph_workspace(int B, int C, int N);   /* bytes of `ws`: a point-major copy of feat */
int hpcs_gemm_tn_graph(const float* At, int lda, const float* feat, const int64_t* idx, int B, int C, int N, int k, int M,
                       float* PD, void* ws, size_t ws_bytes, hpcs_stream_t stream);
size_t hpcs_gemm_nt_workspace(int M, int Kx, int Nc, int n_outer, int n_inner);
/* hpcs_gemm_nt_fused: the same product with the split reduction INSIDE the launch (last-arriving workgroup per slice of
 * splits, then per tile; fixed summation order; no follow-up reduce launches).  `tickets`: hpcs_gemm_nt_tickets() int32
 * words, ZERO before the first use (every call leaves them zero again; the step takes them from its once-per-step zeroed
 * arena).  Same `ws`. */
size_t hpcs_gemm_nt_tickets(int M, int Kx, int Nc, int n_outer, int n_inner);
int hpcs_gemm_nt_fused(const float* G, int ldg, int64_t strideG_o, int64_t strideG_i, const float* X, int ldx,
                       int64_t strideX_o, int64_t strideX_i, int M, int Kx, int Nc, int n_outer, int n_inner, float* dW,
                       int lddw, int accumulate, void* ws, size_t ws_bytes, int32_t* tickets, size_t n_tickets,
                       hpcs_stream_t stream);
int hpcs_gemm_nt(const float* G, int ldg, int64_t strideG_o, int64_t strideG_i, const float* X, int ldx,
                 int64_t strideX_o, int64_t strideX_i, int M, int Kx, int Nc, int n_outer, int n_inner, float* dW,
                 int lddw, int accumulate, void* ws, size_t ws_bytes, hpcs_stream_t stream);

/* ---- a3: VNLinearLeakyReLU / VNBatchNorm / VNLeakyReLU  vn_layers.py:27-77,112-132 -----------
 * PD [B, Cout+Cd, 3, L]: rows [0,Cout) = p = W_feat x, rows [Cout,Cout+Cd) = d = W_dir x
 * (Cd = Cout, or 1 for share_nonlinearity).  flags: HPCS_VN_BN | HPCS_VN_LRELU.
 *   norm_stats : sums[2*Cout] (double; sum n, sum n^2 with n = |p|+1e-6), overwritten
 *   bn_finalize: batch stats -> mean/rstd (+ running-stat update, momentum 0.1, unbiased var);
 *                training=0 -> mean/rstd from the running stats
 *   apply_fwd  : out [B,Cout,3,L]  (pool_k>0: mean over groups of pool_k consecutive l -> [B,Cout,3,L/pool_k],
 *                i.e. mean_pool of vn_layers.py:152-153 fused)
 *   bwd_reduce : red[2*Cout] (double; sum g_nb = dbeta, sum g_nb*xhat = dgamma), overwritten
 *   bwd_apply  : GPD [B,Cout+Cd,3,L] = (dL/dp, dL/dd) */
#define HPCS_VN_BN 1
#define HPCS_VN_LRELU 2
/* Accumulators `sums` / `red` (fp64, reached with atomics): *_zeroed != 0 promises the caller already zero-filled them
 * (e.g. slices of one arena cleared once per step), otherwise each call clears its own. */
int hpcs_vn_norm_stats(const float* PD, int B, int Cout, int Ctot, int64_t L, double* sums, int sums_zeroed, hpcs_stream_t stream);
int hpcs_bn_finalize(const double* sums, int C, int64_t count, float eps, float momentum, float* running_mean,
                     float* running_var, float* mean, float* rstd, int training, hpcs_stream_t stream);
/* Statistics + finalisation in ONE launch (training mode): the last workgroup of the statistics kernel turns the sums into
 * mean / rstd and updates the running statistics (what hpcs_bn_finalize does in a launch of its own).  `sums` [2C] doubles and
 * `counter` [C] uint32 tickets (one per channel: the last workgroup OF A CHANNEL finalises it) must arrive zeroed. */
int hpcs_vn_norm_stats_finalize(const float* PD, int B, int Cout, int Ctot, int64_t L, double* sums, uint32_t* counter, float eps,
                                float momentum, float* running_mean, float* running_var, float* mean, float* rstd,
                                hpcs_stream_t stream);
int hpcs_vn_norm_stats_k2_finalize(const float* E, const float* W2, int ldw, int B, int Cout, int Ctot, int64_t L, double* sums,
                                   uint32_t* counter, float eps, float momentum, float* running_mean, float* running_var,
                                   float* mean, float* rstd, hpcs_stream_t stream);
int hpcs_bn_stats_finalize(const float* Y, int B, int C, int64_t L, double* sums, uint32_t* counter, float eps, float momentum,
                           float* running_mean, float* running_var, float* mean, float* rstd, hpcs_stream_t stream);
int hpcs_vn_apply_fwd(const float* PD, const float* mean, const float* rstd, const float* gamma,
                      const float* beta, int B, int Cout, int Cd, int64_t L, int pool_k, float slope, int flags,
                      float* out, hpcs_stream_t stream);
int hpcs_vn_bwd_reduce(const float* PD, const float* gout, const float* mean, const float* rstd,
                       const float* gamma, const float* beta, int B, int Cout, int Cd, int64_t L, int pool_k,
                       float slope, int flags, double* red, int red_zeroed, hpcs_stream_t stream);
/* ggamma / gbeta (nullable, both or neither; here and in hpcs_bn_act_bwd_apply): when given, the BatchNorm parameter
 * gradients -- the finished reductions red[Cout..2Cout) / red[0..Cout) themselves -- are ADDED into them by the same
 * launch (one writer per channel), instead of costing two more launches per layer. */
int hpcs_vn_bwd_apply(const float* PD, const float* gout, const float* mean, const float* rstd,
                      const float* gamma, const float* beta, const double* red, int training, int B, int Cout,
                      int Cd, int64_t L, int pool_k, float slope, int flags, float* GPD, float* ggamma, float* gbeta,
                      hpcs_stream_t stream);
/* The same four passes for a layer with TWO input channels (conv1: the coordinate edge feature, vn_dgcnn_partseg.py:65-66):
 * E [B,2,3,L] is the layer's INPUT and W2 [Cout+Cd rows of pitch ldw >= 2] its stacked weight [W_feat; W_dir]; the
 * pre-activations PD[b,ch,a,l] = fma(W2[ch][1], E[b,1,a,l], W2[ch][0] * E[b,0,a,l]) -- the k-ordered fma chain hpcs_gemm_tn
 * evaluates, bit for bit -- are formed on the fly, so the 42-channel PD of a 2-channel input (330 MB at B=32, N=1024,
 * k=20, touched five times per step) never exists in HBM. */
int hpcs_vn_norm_stats_k2(const float* E, const float* W2, int ldw, int B, int Cout, int Ctot, int64_t L, double* sums,
                          int sums_zeroed, hpcs_stream_t stream);
int hpcs_vn_apply_fwd_k2(const float* E, const float* W2, int ldw, const float* mean, const float* rstd,
                         const float* gamma, const float* beta, int B, int Cout, int Cd, int64_t L, int pool_k,
                         float slope, int flags, float* out, hpcs_stream_t stream);
int hpcs_vn_bwd_reduce_k2(const float* E, const float* W2, int ldw, const float* gout, const float* mean,
                          const float* rstd, const float* gamma, const float* beta, int B, int Cout, int Cd, int64_t L,
                          int pool_k, float slope, int flags, double* red, int red_zeroed, hpcs_stream_t stream);
int hpcs_vn_bwd_apply_k2(const float* E, const float* W2, int ldw, const float* gout, const float* mean,
                         const float* rstd, const float* gamma, const float* beta, const double* red, int training, int B,
                         int Cout, int Cd, int64_t L, int pool_k, float slope, int flags, float* GPD, float* ggamma,
                         float* gbeta, hpcs_stream_t stream);
/* ---- EdgeConv "gather" layers: get_graph_feature + VNLinearLeakyReLU without the N*k tensors ----------------
 * The first VN layer of an EdgeConv block sees (x[nbr]-x[ctr], x[ctr]) (vn_dgcnn_util.py:33-39), so
 * W [x_nbr - x_ctr; x_ctr] = W_a x_nbr + (W_b - W_a) x_ctr.  UV [B, 2*Ctot, 3, N] holds the point-level products
 * U = W_a F (channels [0,Ctot)) and V = (W_b - W_a) F (channels [Ctot,2Ctot)) from ONE k-times-smaller GEMM; the
 * kernels form PD[b,ch,:,n*k+j] = U[b,ch,:,idx[b,n,j]] + V[b,ch,:,n] on the fly.  Same semantics as the
 * materialised entry points above with L = N*k; pool != 0 fuses the mean over the k neighbours.
 * edge_grad_reduce: GPD [B,Ctot,3,N*k] -> gUV (gU = scatter-add over idx, gV = sum over k); also the first step of the
 * point-level backward of the MATERIALISED graph layers (reduce over the graph, then k-times smaller GEMMs).  Runs as a
 * gather over the transposed graph (built in `ws` per call; hub points are summed by whole waves) when k % 4 == 0 and
 * either N <= 1024, N*k <= 20480, or N <= 2048 and the cloud splits into <= 4 slabs of N/S source points with exactly
 * 20480 edges each (N = 2048, k = 40); else with row-contiguous fp32 atomics. */
int hpcs_vn_edge_norm_stats(const float* UV, const int64_t* idx, int B, int Cout, int Ctot, int N, int k,
                            double* sums, int sums_zeroed, hpcs_stream_t stream);
int hpcs_vn_edge_apply_fwd(const float* UV, const int64_t* idx, const float* mean, const float* rstd,
                           const float* gamma, const float* beta, int B, int Cout, int Cd, int N, int k, int pool,
                           float slope, int flags, float* out, hpcs_stream_t stream);
int hpcs_vn_edge_bwd_reduce(const float* UV, const int64_t* idx, const float* gout, const float* mean,
                            const float* rstd, const float* gamma, const float* beta, int B, int Cout, int Cd, int N,
                            int k, int pool, float slope, int flags, double* red, int red_zeroed, hpcs_stream_t stream);
int hpcs_vn_edge_bwd_apply(const float* UV, const int64_t* idx, const float* gout, const float* mean,
                           const float* rstd, const float* gamma, const float* beta, const double* red, int training,
                           int B, int Cout, int Cd, int N, int k, int pool, float slope, int flags, float* GPD,
                           hpcs_stream_t stream);
size_t hpcs_edge_grad_reduce_workspace(int B, int Ctot, int N);
int hpcs_edge_grad_reduce(const float* GPD, const int64_t* idx, int B, int Ctot, int N, int k, float* gUV, void* ws,
                          size_t ws_bytes, hpcs_stream_t stream);
/* The transposed graph depends on the kNN indices only: hpcs_edge_graph_transpose builds it into `ws` (same size query) as soon
 * as idx exists -- e.g. during the forward pass on another stream -- and hpcs_edge_grad_reduce_prebuilt runs the reduction from
 * it.  Both return -2 (unsupported) for geometries / alignments the transposed-graph kernels do not serve: use
 * hpcs_edge_grad_reduce then. */
int hpcs_edge_graph_transpose(const int64_t* idx, int B, int Ctot, int N, int k, void* ws, size_t ws_bytes, hpcs_stream_t stream);
int hpcs_edge_grad_reduce_prebuilt(const float* GPD, int B, int Ctot, int N, int k, float* gUV, void* ws, size_t ws_bytes,
                                   hpcs_stream_t stream);

/* ---- Conv1d(k=1)+BatchNorm1d(+LeakyReLU)(+Dropout mask)  vn_dgcnn_partseg.py:40-57,97-102 ------
 * Y [B,C,L] is the Conv1d output (hpcs_gemm_tn).  slope = 1 means no activation; mask (nullable,
 * [B,C,L]) is the already scaled dropout multiplier.  With mask == NULL and 0 < drop_p < 1 the multiplier is
 * generated in the kernels from a counter hash of (drop_seed, element offset): the same call arguments give the
 * same mask in forward and backward, so it is never stored (nn.Dropout, vn_dgcnn_partseg.py:46,50).
 * *_seed_dev (nullable, here and in hpcs_so3_rotate_points / hpcs_hyphc_sampled): when given, the kernel reads the seed
 * from that device word instead of the by-value argument, so a captured hipGraph can be replayed with a new seed per
 * step (the step driver refreshes the word with one host->device copy before each replay). */
int hpcs_bn_stats(const float* Y, int B, int C, int64_t L, double* sums, int sums_zeroed, hpcs_stream_t stream);
int hpcs_bn_act_fwd(const float* Y, const float* mean, const float* rstd, const float* gamma, const float* beta,
                    const float* mask, uint64_t drop_seed, const uint64_t* drop_seed_dev, float drop_p, int B, int C,
                    int64_t L, float slope, float* out,
                    hpcs_stream_t stream);
int hpcs_bn_act_bwd_reduce(const float* Y, const float* gout, const float* mean, const float* rstd,
                           const float* gamma, const float* beta, const float* mask, uint64_t drop_seed,
                           const uint64_t* drop_seed_dev, float drop_p,
                           int B, int C, int64_t L, float slope, double* red, int red_zeroed, hpcs_stream_t stream);
int hpcs_bn_act_bwd_apply(const float* Y, const float* gout, const float* mean, const float* rstd,
                          const float* gamma, const float* beta, const float* mask, uint64_t drop_seed,
                          const uint64_t* drop_seed_dev, float drop_p,
                          const double* red, int training, int B, int C, int64_t L, float slope, float* gY,
                          float* ggamma, float* gbeta, hpcs_stream_t stream);
/* The multiplier tensor that (drop_seed, drop_p) stands for: out[i] = 1/(1-p) or 0 (parity tests). */
int hpcs_dropout_mask(uint64_t drop_seed, float drop_p, int64_t n, float* out, hpcs_stream_t stream);

/* ---- a4/a5/a6 glue of VN_DGCNN_partseg.forward  vn_dgcnn_partseg.py:79-95 ---------------------
 * concat_mean   : x [B,C,3,N] -> out [B,2C,3,N] = (x, mean_N(x) expanded)          (:82-83)
 * frame_project : einsum('bijm,bjkm->bikm', x [B,C,3,N], z [B,3,3,N])               (:85, vn_layers.py:197)
 * rowmax        : x [rows,L] -> max over L (+ int32 argmax)                          (:87)
 * rowsum        : x [rows,L] -> sum over L */
int hpcs_concat_mean_fwd(const float* x, int B, int C, int N, float* out, hpcs_stream_t stream);
int hpcs_concat_mean_bwd(const float* gout, int B, int C, int N, float* gx, hpcs_stream_t stream);
int hpcs_frame_project_fwd(const float* x, const float* z, int B, int C, int N, float* out, hpcs_stream_t stream);
int hpcs_frame_project_bwd(const float* x, const float* z, const float* gout, int B, int C, int N, float* gx,
                           float* gz, hpcs_stream_t stream);
/* project_max: fused frame_project + max over N for cat(x [B,C,3,N], xconst [B,Cc,3] expanded) (vn_dgcnn_partseg.py
 * :82-87): out/arg [B, C+Cc, 3]; the projected [B,C+Cc,3,N] tensor is never written.  bwd is sparse (argmax only):
 * gx [B,C,3,N] and gz [B,3,3,N] are overwritten (zero-filled, then atomics), gxconst [B,Cc,3] overwritten.
 * gx == NULL skips the x part; gz == NULL skips the z and xconst parts; gx_accumulate != 0 adds the sparse x terms
 * to the gradient already in gx instead of zero-filling it (so another producer's dense gradient can be the target). */
int hpcs_project_max_fwd(const float* x, const float* xconst, const float* z, int B, int C, int Cc, int N, float* out,
                         int32_t* arg, hpcs_stream_t stream);
int hpcs_project_max_bwd(const float* x, const float* xconst, const float* z, const int32_t* arg, const float* gout,
                         int B, int C, int Cc, int N, float* gx, int gx_accumulate, float* gxconst, float* gz,
                         hpcs_stream_t stream);
int hpcs_rowmax_fwd(const float* x, int64_t rows, int64_t L, float* out, int32_t* arg, hpcs_stream_t stream);
int hpcs_rowmax_bwd(const float* gout, const int32_t* arg, int64_t rows, int64_t L, float* gx,
                    hpcs_stream_t stream);
int hpcs_rowsum(const float* x, int64_t rows, int64_t L, float scale, float* out, hpcs_stream_t stream);
/* VNMaxPool  vn_layers.py:135-149: x [B,C,3,N,k], d = W x [B,C,3,N,k] -> out [B,C,3,N], arg [B,C,N] */
int hpcs_vn_maxpool_fwd(const float* x, const float* d, int B, int C, int N, int k, float* out, int32_t* arg,
                        hpcs_stream_t stream);
int hpcs_vn_maxpool_bwd(const float* gout, const int32_t* arg, int B, int C, int N, int k, float* gx,
                        hpcs_stream_t stream);

/* ---- a7 + Poincare toolbox  hpcs/utils/poincare.py:16-136, hpcs/utils/math.py:61-82 -----------
 * Row-wise over the last dim E of [rows,E]. */
int hpcs_expmap0_fwd(const float* u, int64_t rows, int E, float* out, hpcs_stream_t stream);
int hpcs_expmap0_bwd(const float* u, const float* gout, int64_t rows, int E, float* gu, hpcs_stream_t stream);
int hpcs_expmap_fwd(const float* u, const float* p, int64_t rows, int E, float* out, hpcs_stream_t stream);
int hpcs_project(const float* x, int64_t rows, int E, float* out, hpcs_stream_t stream);
int hpcs_mobius_add(const float* x, const float* y, int64_t rows, int E, float* out, hpcs_stream_t stream);
int hpcs_hyp_dist_o(const float* x, int64_t rows, int E, float* out, hpcs_stream_t stream);
/* a11: hyp_lca(a,b,return_coord=False)  hpcs/distances/lca.py:8-52 -> dist [rows]; optional coord [rows,E] */
int hpcs_hyp_lca_fwd(const float* a, const float* b, int64_t rows, int E, float* dist, float* coord,
                     hpcs_stream_t stream);
int hpcs_hyp_lca_bwd(const float* a, const float* b, const float* gdist, int64_t rows, int E, float* ga, float* gb,
                     hpcs_stream_t stream);
/* backward of the coordinate output (return_coord=True is differentiable in the reference, hpcs/distances/lca.py:37-52):
 * gcoord [rows,E] = dL/dproj -> ga, gb [rows,E] */
int hpcs_hyp_lca_coord_bwd(const float* a, const float* b, const float* gcoord, int64_t rows, int E, float* ga, float* gb,
                           hpcs_stream_t stream);
/* a15: HyperbolicDistance.compute_mat  hpcs/distances/poincare.py:144-176 -> sim [M,M'] */
int hpcs_hyperbolic_sim_mat(const float* x, const float* y, int M, int Mp, int E, float* out, hpcs_stream_t stream);
/* a15: HyperbolicDistance.pairwise_distance  hpcs/distances/poincare.py:178-187 -> sim [M] of the row-aligned pairs */
int hpcs_hyperbolic_sim_pairs(const float* x, const float* y, int64_t M, int E, float* out, hpcs_stream_t stream);
/* HyperbolicLCA.compute_mat / pairwise_distance  hpcs/distances/poincare.py:190-237 (E == 2 only: the reference reads the
 * points as complex numbers): min(d(o,x), d(o,y), d(o, midpoint of the geodesic x-y)) on projected points.
 * pairs == 0: out [M, Mp]; pairs != 0 (M == Mp): row-aligned pairs, out [M]. */
int hpcs_hyperbolic_lca(const float* x, const float* y, int M, int Mp, int E, int pairs, float* out, hpcs_stream_t stream);

/* ---- a8-a14: miner and losses -----------------------------------------------------------------
 * All matrix-free: the (B*N)^2 cosine matrix of hpcs/loss/ultrametric_loss.py:65 is never built.
 *
 * l2_rows        : xn = x / max(|x|,1e-12)  (PML BaseDistance normalisation; hpcs/distances/cosine.py:6-8)
 * cosine_mat     : 0.5*(1 + xn yn^T) [M,M']  (API completeness: CosineSimilarity.compute_mat, cosine.py:10-13)
 * class_index    : labels [M] int64 in [0,C) -> order [M] int32 (members grouped by class, ascending index),
 *                  offsets [C+1] int32                          (loss_and_miner_utils.py:13-21)
 * sample_triplets: per class plan (host-computed: anchor-triplet offsets, k_l) -> a,p,n [T] int64.
 *                  draws != NULL: use host draws (parity mode: identical to the reference's CPU
 *                  torch.randint stream); draws == NULL: device counter-based RNG (seed).   (:23-71)
 * triplet_filter : keep[t] from margin = cos(a,p)-cos(a,n) and (margin, kind); stable compaction
 *                  -> a',p',n' [T'] and count on device      (triplet_margin_miner.py:13-38)
 * hyphc          : per-triplet HypHC term + closed-form mean of the similarity matrix
 *                  (ultrametric_loss.py:57-93)
 * cosface        : PML CosFaceLoss (mirror: hierarchical_cosface_loss.py:57-87)
 * triplet_margin : TripletMarginLoss + AvgNonZeroReducer      (triplet_margin_loss.py:34-65) */
int hpcs_l2_rows_fwd(const float* x, int64_t rows, int E, float* xn, float* inv_norm, hpcs_stream_t stream);
int hpcs_l2_rows_bwd(const float* xn, const float* inv_norm, const float* gxn, int64_t rows, int E, float* gx,
                     int accumulate, hpcs_stream_t stream);
int hpcs_cosine_mat(const float* xn, const float* yn, int M, int Mp, int E, float* out, hpcs_stream_t stream);
size_t hpcs_class_index_workspace(int M, int C);
int hpcs_class_index(const int64_t* labels, int M, int C, int32_t* order, int32_t* offsets, int32_t* counts_ws,
                     size_t ws_bytes, hpcs_stream_t stream);
int hpcs_sample_triplets(const int32_t* order, const int32_t* offsets, const int64_t* cls_toff,
                         const int32_t* cls_k, int C, int M, int64_t T, const int64_t* draws_pos,
                         const int64_t* draws_neg, uint64_t seed, int64_t* a, int64_t* p, int64_t* n,
                         hpcs_stream_t stream);
#define HPCS_TRIPLETS_ALL 0
#define HPCS_TRIPLETS_EASY 1
#define HPCS_TRIPLETS_SEMIHARD 2
#define HPCS_TRIPLETS_HARD 3
size_t hpcs_triplet_filter_workspace(int64_t T);
int hpcs_triplet_filter(const float* xn, int E, const int64_t* a, const int64_t* p, const int64_t* n, int64_t T,
                        float margin, int kind, int64_t* a_out, int64_t* p_out, int64_t* n_out, int64_t* count,
                        void* ws, size_t ws_bytes, hpcs_stream_t stream);
/* hyphc: walks min(*count, T_cap) triplets.  kind < 0: every triplet counts (already mined).  kind >= 0
 * (HPCS_TRIPLETS_*): the miner's margin rule is applied inline -- no compaction pass -- fwd writes the number of
 * kept triplets to kept[1], and the mean is over those (pass it back as norm_count).  fwd writes loss_sum[1]
 * (double; sum over kept triplets of sum(w) - sum(w*softmax(d/temp))).  bwd accumulates into gxn [M,E] (grad wrt
 * unit rows; row-contiguous fp32 atomics) and writes gscale[1] (double); gloss = device scalar dL/d(mean term). */
int hpcs_hyphc_fwd(const float* xn, int E, const int64_t* a, const int64_t* p, const int64_t* n,
                   const int64_t* count, int64_t T_cap, const float* scale, float temperature, float margin,
                   int kind, double* loss_sum, int64_t* kept, hpcs_stream_t stream);
/* bwd with loss_sum/kept != NULL is the ONE-PASS mode used in training: called from the forward with *gloss = 1 and
 * *norm_count = 1 it returns the loss sum, the kept count and the UNSCALED gradient in a single sweep over the
 * triplets (the loss is linear in its upstream gradient, so backward is a rescale by gloss/kept). */
int hpcs_hyphc_bwd(const float* xn, int E, const int64_t* a, const int64_t* p, const int64_t* n,
                   const int64_t* count, int64_t T_cap, const float* scale, float temperature, float margin,
                   int kind, const int64_t* norm_count, const float* gloss, float* gxn, double* gscale,
                   double* loss_sum, int64_t* kept, hpcs_stream_t stream);
/* hyphc_sampled: the training-path form for the device RNG.  Triplets are generated INSIDE the kernel from
 * (class index, per-class plan, seed) -- no index arrays, no sampler/filter pass -- the margin rule `kind` is applied
 * inline, and loss_sum[1], kept[1] and (gxn != NULL) the UNSCALED gradient gxn [M,E] (zero-filled by the call's own
 * zero launch, then accumulated into) + gscale[1] come out of one sweep.  Positive-row gradients are accumulated in an LDS table per
 * workgroup.  E must be 16, 32 or 64 (the host mirror zero-pads other widths up to the next of these: zero columns change
 * no dot product and receive zero gradient).  a_dbg/p_dbg/n_dbg (nullable, [T]) receive the generated triplets (tests);
 * a triplet the miner's rule dropped is exported with a_dbg = ~anchor (negative).
 * T sizes the grid only and must be >= cls_toff[C], the number of triplets the kernel actually samples (read on the
 * device): a captured launch sized for the largest batch serves smaller ones, surplus workgroups exit at once. */
int hpcs_hyphc_sampled(const float* xn, const int64_t* labels, const int32_t* order, const int32_t* offsets,
                       const int64_t* cls_toff, const int32_t* cls_k, int C, int M, int E, int64_t T, uint64_t seed,
                       const uint64_t* seed_dev, const float* scale, float temperature, float margin, int kind,
                       double* loss_sum, int64_t* kept, float* gxn, double* gscale, int64_t* a_dbg, int64_t* p_dbg,
                       int64_t* n_dbg, hpcs_stream_t stream);
/* mean of the M x M similarity in closed form: 0.5 + 0.5*|sum_i xn_i|^2/M^2.  colsum [E] double. */
int hpcs_colsum(const float* xn, int64_t rows, int E, double* colsum, hpcs_stream_t stream);
/* scalar tails of the sampled HypHC term (hpcs/loss/ultrametric_loss.py:27-40 mean over triplets + the closed-form mean
 * similarity): out[0] = loss_sum/kept + 0.5 + 0.5*|colsum|^2/M^2;  backward, in place on the unscaled gradient:
 * gxn[i,:] = colsum/M^2 * g + gxn[i,:] * (g / max(kept,1)), gscale[0] = gscale_raw * (g / max(kept,1)) (gscale NULL: skipped). */
int hpcs_hyphc_finalize(const double* loss_sum, const int64_t* kept, const double* colsum, int64_t M, int E, float* out,
                        hpcs_stream_t stream);
int hpcs_hyphc_grad_finalize(const float* g, const int64_t* kept, const double* colsum, const double* gscale_raw, int64_t M,
                             int E, float* gxn, float* gscale, hpcs_stream_t stream);
/* cosface: plain CosFace = one level of singleton branches (branch_id NULL, n_levels 1).  Hierarchical CosFace
 * (next row f2: hierarchical_loss + HierarchicalCosFaceLoss, hpcs/loss/hierarchical_cosface_loss.py:9-87):
 * branch_id [n_levels, C] int32 gives each class its branch per level; loss_row = sum_l -log(sum_{c in branch_l(y)} p_c).
 * fwd writes logits [M,C], loss_rows [M], lse_rows [M].
 * bwd: gxn [M,E] and gwn [C,E] are overwritten (zero-filled by the call, then summed into); gloss = device scalar dL/d(mean loss). */
int hpcs_cosface_fwd(const float* xn, const float* wn, const int64_t* labels, const int32_t* branch_id, int n_levels,
                     int M, int E, int C, float margin, float scale, float* logits, float* loss_rows,
                     float* lse_rows, hpcs_stream_t stream);
int hpcs_cosface_bwd(const float* xn, const float* wn, const int64_t* labels, const float* logits,
                     const float* lse_rows, const int32_t* branch_id, int n_levels, int M, int E, int C, float scale,
                     const float* gloss, float* gxn, float* gwn, hpcs_stream_t stream);
/* hierarchical_loss(probabilities, targets, hierarchy_list)  hpcs/loss/hierarchical_cosface_loss.py:9-28  as a
 * stand-alone op on already-softmaxed probabilities prob [M,C]: loss_rows[i] = sum_l -log(sum_{c in branch_l(y_i)} prob[i,c])
 * (the caller means over M: F.nll_loss's reduction).  bwd: gprob [M,C] overwritten with gloss/M * d loss_row/d prob. */
int hpcs_hier_nll_fwd(const float* prob, const int64_t* labels, const int32_t* branch_id, int n_levels, int M, int C,
                      float* loss_rows, hpcs_stream_t stream);
int hpcs_hier_nll_bwd(const float* prob, const int64_t* labels, const int32_t* branch_id, int n_levels, int M, int C,
                      const float* gloss, float* gprob, hpcs_stream_t stream);
/* ---- next row f3: in-step metrics  hpcs/models/base_hyp_hc.py:88-99,126-130 -----------------------------------------
 * The reference feeds softmax(get_logits(...)) to torchmetrics (pinned 0.10.1 in hpcs-env.yaml, but the constructor calls
 * need >= 0.11: Accuracy(task='multiclass', top_k=1) = micro accuracy; MulticlassJaccardIndex = macro IoU over the
 * classes that occur in predictions or targets).  torchmetrics is a third-party dependency absent from this image:
 * its published algorithm is restated (parity unpinned).  logits [M,C] = hpcs_cosface_fwd's; counts [3*C] int64
 * (true positives, predictions, targets per class) overwritten; acc_iou [2] = {accuracy, IoU}. */
int hpcs_class_metrics(const float* logits, const int64_t* labels, int M, int C, int64_t* counts, float* acc_iou,
                       hpcs_stream_t stream);
int hpcs_triplet_margin_fwd(const float* xn, int E, const int64_t* a, const int64_t* p, const int64_t* n,
                            const int64_t* count, int64_t T_cap, float margin, double* sum_and_nnz,
                            hpcs_stream_t stream);
int hpcs_triplet_margin_bwd(const float* xn, int E, const int64_t* a, const int64_t* p, const int64_t* n,
                            const int64_t* count, int64_t T_cap, float margin, const double* sum_and_nnz,
                            const float* gloss, float* gxn, hpcs_stream_t stream);

/* ---- next row f1: hpcs.optim.RAdam.step  hpcs/optim/radam.py:59-134 ---------------------------
 * One fused launch over a table of tensors.  desc[i] = {param, grad, exp_avg, exp_avg_sq, rows, cols}
 * (device pointers as uint64, rows*cols elements, `cols` = size of the last dim = the "point").
 * step_sizes[i] = lr*sqrt(1-b2^t)/(1-b1^t) with the reference's per-parameter double increment of
 * group["step"] (host-computed).  Applies egrad2rgrad, moment updates, expmap retraction, project
 * (row-norm clip 1-4e-3) and parallel transport of exp_avg.  lanes_per_row = 8 (vector points), 64 (wide rows: a wave per row) or 1 (tables
 * whose tensors all have cols == 1: same arithmetic, one lane per row). */
typedef struct {
    uint64_t param, grad, exp_avg, exp_avg_sq;
    int64_t rows;
    int32_t cols;
    float step_size;
} hpcs_radam_desc;
int hpcs_radam_step(const hpcs_radam_desc* desc_dev, int n_tensors, int64_t total_rows_hint, int lanes_per_row,
                    float beta1, float beta2, float eps, float weight_decay, hpcs_stream_t stream);

/* ---- caller-side augmentation at the head of the step: hpcs/models/shapenet_hyp_hc.py:63-73 ------------------
 * One rotation per cloud (pytorch3d random_rotations: normalised Gaussian quaternion; drawn from a counter RNG keyed
 * by `seed`, or taken from rot_in [B,3,3] when given), points [B,N,3] -> x_out [B,3,N] = transpose(points @ R);
 * rot_out [B,3,3] (optional) receives the matrices used. */
int hpcs_so3_rotate_points(const float* points, int B, int N, uint64_t seed, const uint64_t* seed_dev,
                           const float* rot_in, float* x_out,
                           float* rot_out, hpcs_stream_t stream);

/* ---- next row f4: test-time decode ---------------------------------------------------------------------------------
 * hpcs_complete_linkage replaces BaseSimilarityHypHC._decode_linkage (hpcs/models/base_hyp_hc.py:81-86) for a whole
 * batch: Z[b] = scipy.cluster.hierarchy.linkage(leaves[b], method='complete', metric='cosine'), bit-identical
 * (fp64 cosine distances in scipy's rounding sequence, nearest-neighbour chain, stable sort by height, union-find
 * labels).  leaves [B,N,E] fp32 = project(normalize_embeddings(x_poincare)); Z [B,N-1,4] fp64 rows
 * (cluster a, cluster b, height, size); status [B] int32: 0 ok, bit 0 = a non-finite distance (scipy raises
 * ValueError), bit 2 = chain step bound exceeded (cannot happen for finite input).
 * N <= 2048.  ws: hpcs_complete_linkage_workspace bytes (the B*N*N fp64 distance matrices; destroyed). */
size_t hpcs_complete_linkage_workspace(int B, int N);
int hpcs_complete_linkage(const float* leaves, int B, int N, int E, double* Z, int32_t* status, void* ws,
                          size_t ws_bytes, hpcs_stream_t stream);
/* hpcs_optimal_k replaces get_optimal_k(y, linkage_matrix, index) (hpcs/utils/scores.py:141-177, remap_labels
 * :124-139) for a whole batch: for k = 1..K_true+4, y_pred = fcluster(Z, k, 'maxclust') - 1 with scipy's cluster
 * numbering; index_mode 0 ('iou'): per-part arg-max IoU remap then one-hot Jaccard; 1 ('ri'): adjusted Rand index;
 * the first k with the strictly best score wins.  y [B,N] int64 labels in [0, num_labels), num_labels <= 1024, at
 * most 64 distinct per cloud; best_pred [B,N] int32 (-1 when no k scored > 0), best_k [B] int32, best_score [B] fp64;
 * status [B]: 0 ok, 2 = label out of range, 3 = too many parts for this kernel. */
int hpcs_optimal_k(const double* Z, const int64_t* y, int B, int N, int num_labels, int index_mode,
                   int32_t* best_pred, int32_t* best_k, double* best_score, int32_t* status, hpcs_stream_t stream);

/* flat fp32 helpers used by the data-parallel wrapper (gradient bucket pack/unpack+scale) */
int hpcs_scale_inplace(float* x, int64_t n, float s, hpcs_stream_t stream);

/* ---- EdgeConv edge-level VN layers without their pre-activation tensor (csrc/vn_recompute.hip) -------------------------
 * VNLinearLeakyReLU on an edge tensor (hpcs/nn/dgcnn/utils/vn_layers.py:48-77, VNBatchNorm :112-132, mean_pool :150-153;
 * conv2 / conv4 of hpcs/nn/dgcnn/vn_dgcnn_partseg.py:66-75): every pass re-forms PD = [Wf; Wd] X on the matrix cores from
 * X [B,Cin,3,L] instead of reading a materialised PD [B,2*Cout,3,L].  Wp = the stacked weight [2*Cout, ldw] (row-major,
 * feature rows then direction rows); BN + VN-LeakyReLU, Cd = Cout <= 24, Cin <= 24, L % 16 == 0, pool_k in {0} or a
 * multiple of 4 up to 64 dividing L (hpcs_vn_rc_supported).  Results equal the materialised path's (same fp32 fma chains). */
int hpcs_vn_rc_supported(int Cin, int Cout, int Cd, int64_t L, int pool_k, int flags);
/* sums[2*Cout] (zeroed doubles) += sum |p|+eps, sum (|p|+eps)^2; counter != NULL (one zeroed uint32): the last workgroup
 * finalises mean / rstd / running statistics (training-mode BatchNorm), like hpcs_vn_norm_stats_finalize. */
int hpcs_vn_rc_stats(const float* X, const float* Wp, int ldw, int B, int Cin, int Cout, int64_t L, double* sums,
                     uint32_t* counter, float eps, float momentum, float* running_mean, float* running_var, float* mean,
                     float* rstd, hpcs_stream_t stream);
/* out [B,Cout,3,L], or [B,Cout,3,L/pool_k] = mean over pool_k consecutive positions (pool_k > 0) */
int hpcs_vn_rc_apply_fwd(const float* X, const float* Wp, int ldw, const float* mean, const float* rstd, const float* gamma,
                         const float* beta, int B, int Cin, int Cout, int64_t L, int pool_k, float slope, float* out,
                         hpcs_stream_t stream);
/* red[2*Cout] (zeroed doubles) += the two BatchNorm backward sums (as hpcs_vn_bwd_reduce); gout as the forward's out */
int hpcs_vn_rc_bwd_reduce(const float* X, const float* Wp, int ldw, const float* gout, const float* mean, const float* rstd,
                          const float* gamma, const float* beta, int B, int Cin, int Cout, int64_t L, int pool_k, float slope,
                          double* red, hpcs_stream_t stream);
/* GPD [B,2*Cout,3,L] = (dL/dp, dL/dd) (written once, for the weight-gradient GEMM) and gX [B,Cin,3,L] = Wp^T GPD from the
 * same registers; ggamma / gbeta (nullable, both or neither): the BatchNorm parameter gradients are added there. */
int hpcs_vn_rc_bwd_apply(const float* X, const float* Wp, int ldw, const float* gout, const float* mean, const float* rstd,
                         const float* gamma, const float* beta, const double* red, int training, int B, int Cin, int Cout,
                         int64_t L, int pool_k, float slope, float* GPD, float* gX, float* ggamma, float* gbeta,
                         hpcs_stream_t stream);

/* ---- diagnostics (no reference counterpart): a stream-ordered watchpoint on a small device region ------------------
 * hpcs_debug_snapshot copies `nbytes` (multiple of 4) of the watched region into a shadow buffer; hpcs_debug_watch
 * compares region and shadow and records in status[0] the `seq` of the first call that saw them differ (status:
 * int32[8], zeroed by the caller; [1] word offset, [2] found, [3] expected, [4] differing words, [5] next word,
 * [6] checks so far, [7] [6] at the first mismatch).  hpcs_amd/_lib.py enqueues one watch after every launch when a
 * watch is armed (HPCS_WATCH), also inside hipGraph captures; nothing is allocated. */
/* One launch zero-fills up to four device buffers (sizes in bytes, multiples of 4; NULL skipped): the library never issues
 * hipMemset nodes. */
int hpcs_zero_buffers(void* p0, size_t n0, void* p1, size_t n1, void* p2, size_t n2, void* p3, size_t n3, hpcs_stream_t stream);
int hpcs_debug_snapshot(const void* src, void* dst, int nbytes, hpcs_stream_t stream);
int hpcs_debug_watch(const void* watched, const void* shadow, int nbytes, int seq, int32_t* status,
                     hpcs_stream_t stream);
/* Node census of a captured hipGraph (host-only): counts16[t] = nodes of hipGraphNodeType t (0 kernel, 1 memcpy, 2 memset,
 * 3 host, 4 child graph, 5 empty, ...), child graphs included.  Returns the node total or a negative status.  The step
 * driver refuses a capture holding memset nodes (their arguments do not survive a second graph on this runtime). */
int hpcs_debug_graph_census(void* graph, int64_t* counts16);
/* Tests only: replace one value of the library's tuning table (csrc/common.h HpcsTuning: the HPCS_* switches, read from the
 * environment once at first use and immutable afterwards).  name = the environment variable's name; value = its integer
 * code (HPCS_KNN_KERNEL: 0 auto, 1 tiled, 2 queue, 4 bq, 5 bq2). */
int hpcs_debug_set_tuning(const char* name, int value);

#ifdef __cplusplus
}
#endif
#endif /* HPCS_MI355X_H */
