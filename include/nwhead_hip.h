/*
 * nwhead_hip.h -- C ABI of libnwhead_hip.so: the MI355X (gfx950) implementation of the
 * Nadaraya-Watson head hot path of alanqrwang/nwhead @ 2024_08_07.
 *
 * The reference has no FFI; its seam is the Python operator
 *     NWHead.forward(x, sx, sy)            nwhead/nw.py:266-289
 * built from                               nwhead/kernel.py:13-44   (score functions)
 * plus util/metric.py:23-50 (support_influence) and the implicit autograd of the above
 * (loss.backward(), train.py:414).  Each entry point below names the reference lines it
 * replaces.  INTEGRATION.md shows the ctypes binding a reference maintainer would add.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer into HIP memory owned by the caller, row-major, dense;
 *     float = IEEE fp32, labels = int64 (torch.long), sizes = int64;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls only enqueue
 *     work on it: they never synchronise, never allocate, never free;
 *   - return value: NW_OK (0) or a negative nw_status; nothing is thrown;
 *   - stateless and thread-safe: safe from any host thread with its own stream; every option of a call is an argument
 *     of that call (nw_fwd_opts), and the library never reads the process environment.  The one piece of process state is
 *     the table of diagnostic knobs (nw_debug_set, nw_debug_tile_timing): timing experiments, unset in normal use;
 *   - scratch memory is supplied by the caller; ask nw_*_workspace_bytes() for the size.
 */
#ifndef NWHEAD_HIP_H
#define NWHEAD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NW_ABI_VERSION 2

/* score functions = the reference's kernel modules, nwhead/kernel.py:80-97 (get_kernel) */
typedef enum {
    NW_SCORE_EUCLIDEAN = 0,   /* -cdist(x, y)                           kernel.py:13-15 */
    NW_SCORE_HYPERSPHERE = 1, /* -cdist(normalize(x), normalize(y))     kernel.py:17-21 */
    NW_SCORE_COSINE = 2,      /* normalize(x) . normalize(y)            kernel.py:23-28 */
    NW_SCORE_DOT = 3,         /* x . y                                  kernel.py:30-33 */
    NW_SCORE_CLIP = 4         /* exp(logit_scale) * cosine              kernel.py:35-44 */
} nw_score_kind;

typedef enum {
    NW_OK = 0,
    NW_ERR_INVALID_ARG = -1,  /* null pointer, negative size, label array missing ...          */
    NW_ERR_UNSUPPORTED = -2,  /* unknown score kind                                             */
    NW_ERR_WORKSPACE = -3,    /* workspace smaller than nw_*_workspace_bytes()                  */
    NW_ERR_LAUNCH = -4,       /* hipLaunchKernel / hipMemsetAsync reported an error             */
    NW_ERR_NO_DEVICE = -5     /* no gfx950 device visible to this process                       */
} nw_status;

int nw_abi_version(void);
const char *nw_status_string(int status);
/* 0 when a HIP device is visible and its arch is gfx950, else NW_ERR_NO_DEVICE. */
int nw_device_check(void);

/* ---------------------------------------------------------------------------------------------
 * Scores only.  Replaces kernel(x, y): nwhead/kernel.py:13-44 as called from nwhead/nw.py:283
 * (x unsqueezed to (B,1,d), y = support) and from NWNet.get_neighbors nwhead/nw.py:248 and
 * KNN.__call__ nwhead/utils.py:187 (2-D queries x bank).
 *   q        (B,d)
 *   s        (N,d) when sup_batched == 0, (B,N,d) when sup_batched != 0
 *   scores   (B,N) out
 *   logit_scale_dev  device pointer to the CLIP log-scale scalar (kernel.py:38); only read for
 *                    NW_SCORE_CLIP; may be NULL otherwise
 * cdist regime (torch picks the matmul form when N > 25, else the direct difference form): the
 * same switch is made here so that exact-zero distances behave as in the reference.
 * ------------------------------------------------------------------------------------------- */
int nw_scores_f32(const float *q, const float *s, float *scores,
                  int64_t B, int64_t N, int64_t d,
                  int kind, const float *logit_scale_dev, int sup_batched, void *stream);

/* Squared L2 norm of every row of a dense (rows,d) matrix: n2[r] = sum_k x[r,k]^2.  The pow(2).sum(-1)
 * half of torch.cdist's matmul form (and of F.normalize, nwhead/kernel.py:19-20), hoisted out of the
 * hot loop for operands that do not change between calls (the precomputed support bank). */
int nw_row_norm2_f32(const float *x, float *n2, int64_t rows, int64_t d, void *stream);

/* Split-fp16 form of a dense (rows,d) fp32 matrix, d % 32 == 0 (NW_ERR_UNSUPPORTED otherwise):
 *   out_split (rows,d) floats-worth of bytes: every 128-byte chunk of a row = [32 x h | 32 x l] fp16 with
 *             h + l = x * 2^e to 2^-23 relative, e per row such that the row's largest magnitude lands
 *             in [2^13, 2^14) (fp16 normal range whatever the feature scale);
 *   row_scale (rows,) = 2^-e;   row_norm2 (rows,) = sum_k x_k^2 of the original values.
 * Like nw_row_norm2_f32 this belongs to precompute() (nwhead/nw.py:118-125): the bank is split once. */
int nw_split_rows_f16x2(const float *x, float *out_split, float *row_scale, float *row_norm2,
                        int64_t rows, int64_t d, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Forward.  Replaces NWHead.forward nwhead/nw.py:266-289:
 *     one_hot(sy) -> kernel scores -> softmax over supports -> bmm with one-hot -> log(. + 1e-12)
 *   sy          (N,) or, when labels_batched != 0, (B,N); values outside [0,C) contribute nothing
 *   s_norm2     optional (N,): squared row norms of s as written by nw_row_norm2_f32.  The resident
 *               bank of 'full' inference (NWNet.precompute, nwhead/nw.py:118-125) caches them so
 *               that the hot loop is matrix-core work only; NULL = computed inside the kernel.
 *               Only read for a shared 2-D support.
 *   s_split, s_scale   optional, d % 32 == 0 only: the support in split-fp16 form as written by
 *               nw_split_rows_f16x2 (with s_norm2 from the same call).  Selects the fast path of
 *               'full' inference: dot products on the fp16 matrix cores at fp32-grade accuracy
 *               (x = h + l, three fp16 MFMAs per product block; error ~1e-7 * sum|a_k b_k|, the
 *               level of an fp32 FMA chain).  The queries are split inside the call.  NULL = fp32
 *               matrix cores on the original operands.
 *   out         (B,C) log-probabilities
 *   scores_out  optional (B,N): raw scores (saved for backward / neighbour search)
 *   lse_out     optional (B,):  log sum_j exp(score_bj)  (saved for backward)
 *   weights_out optional (B,N): softmax weights  (the `sweights` of util/metric.py:23)
 *   workspace   nw_fwd_workspace_bytes(B,N,d,C) bytes of scratch (may be NULL if that is 0)
 * ------------------------------------------------------------------------------------------- */
/* Options of one forward call (NULL = all defaults).  struct_size = sizeof(nw_fwd_opts) of the caller's header.
 *   tables / tables_bytes  run tables of a RESIDENT bank (labels that do not change between calls), built once with
 *               nw_bank_tables_build(sy, N, C, ...) for the very sy, N and C of this call: on large launches the forward
 *               walks the bank in tiles of 128 supports and needs, per tile, the runs of equal consecutive labels; without
 *               tables it builds them in its workspace on every call (~5 us + a kernel boundary)
 *   tables_sy / tables_N   the label array and row count the tables were built from.  The forward uses the tables only
 *               when these are the call's own sy and N (anything else -- another label array, a shorter struct of an older
 *               header -- and it builds the tables itself, as without the option)
 *   persistent_wgs  workgroups of the persistent tile kernel, a multiple of 8; 0 = one per CU.  Sharded inference passes
 *               CUs - 8 (one CU per XCD left to the concurrent RCCL kernel)
 *   force_split nonzero: the split-fp16 path whenever the bank is prepared, also below ~2e8 multiply-adds */
typedef struct nw_fwd_opts {
    uint32_t struct_size;
    int32_t persistent_wgs;
    int32_t force_split;
    int32_t reserved;
    const void *tables;
    size_t tables_bytes;
    const int64_t *tables_sy;
    int64_t tables_N;
} nw_fwd_opts;

size_t nw_fwd_workspace_bytes(int64_t B, int64_t N, int64_t d, int64_t C);
int nw_fwd_f32(const float *q, const float *s, const int64_t *sy, const float *s_norm2,
               const float *s_split, const float *s_scale,
               float *out, float *scores_out, float *lse_out, float *weights_out,
               void *workspace, size_t workspace_bytes,
               int64_t B, int64_t N, int64_t d, int64_t C,
               int kind, const float *logit_scale_dev,
               int sup_batched, int labels_batched, const nw_fwd_opts *opts, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Sharded 'full' inference (new capability): per-shard partials of the same
 * forward over this rank's slice of the support bank, and the merge.
 *   m    (B,)   max_j score            (-inf for an empty shard)
 *   den  (B,)   sum_j exp(score - m)
 *   num  (B,C)  sum_{j: sy_j = c} exp(score - m)
 * nw_merge_finalize_f32 takes the partials of G shards and writes
 *   out = log(num / den + 1e-12) after rescaling every shard to the common max.
 *   Shard g's arrays start at m + g*stride_m, den + g*stride_den, num + g*stride_num (strides in
 *   floats; pass B, B, B*C for dense (G,B) (G,B) (G,B,C) stacks, or the common row length when the
 *   three sections of each shard are packed in one all-gathered buffer).
 *   class_lo (device, (G,)) / C_local: optional class windows.  A contiguous slice of a class-sorted
 *   bank only holds the classes [class_lo[g], class_lo[g] + C_local); its partial forward is then run
 *   with labels shifted by class_lo[g] and C = C_local, num is (B, C_local) per shard, and the rows
 *   that cross xGMI shrink ~G-fold.  NULL = every shard carries all C classes.
 * ------------------------------------------------------------------------------------------- */
int nw_fwd_partial_f32(const float *q, const float *s, const int64_t *sy, const float *s_norm2,
                       const float *s_split, const float *s_scale,
                       float *m, float *den, float *num,
                       void *workspace, size_t workspace_bytes,
                       int64_t B, int64_t N, int64_t d, int64_t C,
                       int kind, const float *logit_scale_dev, const nw_fwd_opts *opts, void *stream);
int nw_merge_finalize_f32(const float *m, const float *den, const float *num, float *out,
                          int64_t G, int64_t B, int64_t C,
                          int64_t stride_m, int64_t stride_den, int64_t stride_num,
                          const int64_t *class_lo, int64_t C_local, void *stream);

/* Run tables of a RESIDENT bank for nw_fwd_opts.tables:
 *     tables = malloc(nw_bank_tables_bytes(N));  nw_bank_tables_build(sy, N, C, tables, bytes, stream);   (once)
 *     opts.tables = tables; opts.tables_bytes = bytes;  nw_fwd_f32(..., &opts, stream)                     (per call)
 * Labels outside [0, C) are kept out of the tables like everywhere else (they contribute nothing). */
size_t nw_bank_tables_bytes(int64_t N);
int nw_bank_tables_build(const int64_t *sy, int64_t N, int64_t C, void *tables, size_t tables_bytes, void *stream);


/* ---------------------------------------------------------------------------------------------
 * Backward.  Replaces the autograd graph the reference builds through nwhead/nw.py:276-289 and
 * nwhead/kernel.py:13-44 (loss.backward(), train.py:414), including torch's zero sub-gradient at
 * distance 0 (_euclidean_dist_backward / cdist_backward mask).
 *   scores, lse   as saved by nw_fwd_f32
 *   out, gout     (B,C) forward output and its incoming gradient
 *   gq            (B,d) out;  gs (N,d) or (B,N,d) out;  glogit_scale optional scalar out (CLIP)
 *   workspace     nw_bwd_workspace_bytes(...) bytes
 * ------------------------------------------------------------------------------------------- */
size_t nw_bwd_workspace_bytes(int64_t B, int64_t N, int64_t d, int64_t C, int kind, int sup_batched);
int nw_bwd_f32(const float *q, const float *s, const int64_t *sy,
               const float *scores, const float *lse, const float *out, const float *gout,
               float *gq, float *gs, float *glogit_scale,
               void *workspace, size_t workspace_bytes,
               int64_t B, int64_t N, int64_t d, int64_t C,
               int kind, const float *logit_scale_dev,
               int sup_batched, int labels_batched, void *stream);
/* The same with the supports' bank as prepared for the forward (nw_split_rows_f16x2 of these very rows: s_split,
 * s_scale, s_norm2; all three or none; shared (N,d) supports only).  On large shapes -- nw_bwd_uses_split(...) != 0:
 * shared supports, d % 32 == 0, a row of N coefficients fits in LDS -- the two products of the backward
 * (gq = A s + 2 rq q, gs = A^T q + 2 rs s) run on the fp16 matrix cores with split-row operands like the forward
 * (three fp16 MFMAs per fp32 multiply-add, fp32 accumulation); without a bank the rows are split inside the call.
 * A training step builds the bank once, hands it to nw_fwd_f32 (scores saved for the backward) and to this call.
 * The caller's s_split (exactly N * d floats, no tail required) is used when d % 64 == 0; for d % 64 == 32 the rows
 * are split again into the workspace, whose copy carries the tail the product kernel's last column tile reads. */
int nw_bwd_uses_split(int64_t B, int64_t N, int64_t d, int64_t C, int sup_batched);
int nw_bwd_bank_f32(const float *q, const float *s, const float *s_norm2, const float *s_split,
                    const float *s_scale, const int64_t *sy,
                    const float *scores, const float *lse, const float *out, const float *gout,
                    float *gq, float *gs, float *glogit_scale,
                    void *workspace, size_t workspace_bytes,
                    int64_t B, int64_t N, int64_t d, int64_t C,
                    int kind, const float *logit_scale_dev,
                    int sup_batched, int labels_batched, void *stream);


/* ---------------------------------------------------------------------------------------------
 * support_influence.  Replaces util/metric.py:23-50, vectorised over the query batch:
 *   infl[b,j] = log((p - p*w_bj) / (p - w_bj*[sy_j == qy_b])),  p = probs[b, qy_b]
 *   probs (B,C) = exp(out);  qy (B,) int64;  w (B,N) softmax weights;  sy (N,) int64
 * Every product/difference/quotient is rounded separately (no FMA contraction) so the +inf / NaN
 * pattern of the reference's one-shot classes is reproduced.
 * ------------------------------------------------------------------------------------------- */
int nw_support_influence_f32(const float *probs, const int64_t *qy, const float *w,
                             const int64_t *sy, float *infl,
                             int64_t B, int64_t N, int64_t C, void *stream);

/* Softmax over supports + label aggregation + log of a GIVEN (B,N) score matrix, and its gradient: the tail of
 * NWHead.forward (nwhead/nw.py:285-289) for score functions that are not built into the kernels (the reference
 * accepts any callable kernel module, nw.py:256-264: the scores then come from that module, on the device, through
 * torch autograd; this is the rest).  sy (N,) or (B,N) when labels_batched.
 *   nw_aggregate_f32      out (B,C), optional lse_out (B,), optional weights_out (B,N)
 *   nw_aggregate_bwd_f32  gscores (B,N) = W * (dW - sum_j W dW), dW_j = gout[b, sy_j] * exp(-out[b, sy_j]) */
int nw_aggregate_f32(const float *scores, const int64_t *sy, float *out, float *lse_out, float *weights_out,
                     int64_t B, int64_t N, int64_t C, int labels_batched, void *stream);
int nw_aggregate_bwd_f32(const float *scores, const int64_t *sy, const float *lse, const float *out, const float *gout,
                         float *gscores, int64_t B, int64_t N, int64_t C, int labels_batched, void *stream);

/* Forward + support_influence in one call: nw_fwd_f32's arguments plus
 *   qy (B,) int64 query labels, infl_out (B,N).
 * The fused tile kernel writes the raw scores into infl_out, the merge gives out (log-probabilities) and the
 * log-sum-exp, and ONE in-place pass turns scores into influences with w = exp(score - lse), p = exp(out[b, qy_b])
 * (util/metric.py:23-50 applied to the head's own outputs, README.md:101-132).  The (B,N) matrix crosses HBM three
 * times (scores out, scores in, influences out) instead of five (scores, weights written and read, influences), and
 * no softmax-weight matrix exists.  (Recomputing the scores instead of storing them would cost a second pass of the
 * tile kernel: 13 us against the 4 us of the round trip at B = 256, N = 10000.)  Shared 2-D supports only.
 * lse_out optional (B,).  workspace: nw_fwd_workspace_bytes(B,N,d,C). */
int nw_fwd_influence_f32(const float *q, const float *s, const int64_t *sy, const float *s_norm2,
                         const float *s_split, const float *s_scale, const int64_t *qy,
                         float *out, float *lse_out, float *infl_out,
                         void *workspace, size_t workspace_bytes,
                         int64_t B, int64_t N, int64_t d, int64_t C,
                         int kind, const float *logit_scale_dev, const nw_fwd_opts *opts, void *stream);

/* ---------------------------------------------------------------------------------------------
 * The k best supports per query.  Replaces the full descending argsort that the reference cuts to
 * its first k columns (KNN.__call__, nwhead/utils.py:185-193; 'knn' / 'hnsw' support modes):
 *   idx_out[b][0..k) = argsort(scores[b], descending, stable)[0..k)      (ties: lower index first)
 *   scores   (B,N) fp32 (e.g. from nw_scores_f32)
 *   idx_out  (B,k) int64;  val_out optional (B,k) fp32: the selected scores, best first
 * 1 <= k <= min(N, 1024); larger k returns NW_ERR_UNSUPPORTED (callers sort the whole row then).
 * ------------------------------------------------------------------------------------------- */
int nw_topk_f32(const float *scores, int64_t *idx_out, float *val_out,
                int64_t B, int64_t N, int64_t k, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Eval-mode BatchNorm (+ ReLU) of the pre-activation backbones as one pass: out = max(x * scale[c] +
 * shift[c], 0).  Replaces the BatchNorm2d -> ReLU pairs in front of the convolutions of
 * model/densenet.py:33-60 (_DenseLayer norm1/relu1), :82-91 (_Transition) and :139 + :160 (norm5 + relu)
 * at inference, where scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale.
 *   x      n planes-of-c: element (i, ch, p) at x[i * x_batch_stride + ch * hw + p]  (the first c channels
 *          of a wider NCHW slab qualify: x_batch_stride >= c * hw)
 *   out    (n, c, hw) contiguous;  scale, shift (c,);  relu != 0 applies the max
 * ------------------------------------------------------------------------------------------- */
int nw_scale_shift_relu_f32(const float *x, const float *scale, const float *shift, float *out,
                            int64_t n, int64_t c, int64_t hw, int64_t x_batch_stride, int relu,
                            void *stream);
/* The same followed by a 2x2 / stride-2 average pool (floor sizes): out (n, c, h/2, w/2).  The transitions of the
 * DenseNets (norm - relu - conv 1x1 - avgpool, model/densenet.py:82-91, model/densenet3.py:25-35): the pool commutes
 * with the bias-free 1x1 convolution, so the folded inference copy pools first and convolves a quarter of the pixels. */
int nw_scale_shift_relu_avgpool2_f32(const float *x, const float *scale, const float *shift, float *out,
                                     int64_t n, int64_t c, int64_t h, int64_t w, int64_t x_batch_stride,
                                     int relu, void *stream);

/* out[r][c] = act(x[r][c] + bias[c] [+ residual[r][c]]) for a channels-last (NHWC) activation seen as (rows = n h w,
 * c): the folded BatchNorm bias, the identity of a ResNet block (model/resnet.py:58-66) and the ReLU behind a bias-free
 * convolution in one pass; c % 4 == 0, 16-byte aligned pointers, in place allowed (out == x). */
int nw_bias_act_nhwc_f32(const float *x, const float *bias, const float *residual, int relu, float *out,
                         int64_t rows, int64_t c, void *stream);

/* 1x1 convolution of the folded inference backbones with its neighbours fused in, on the fp32 matrix cores
 * (v_mfma_f32_16x16x4_f32: exact fp32 multiply-adds).  Replaces, in DenseNet's dense layers and transitions
 * (model/densenet.py:33-60 norm1-relu1-conv1-norm2-relu2, :82-91 norm-relu-conv) and CIFAR_DenseNet's
 * (model/densenet3.py:10-35), the scale-shift-ReLU pass, the GEMM, the bias add and the ReLU pass:
 *     out[n, co, p] = post( bias[co] + sum_ci W[co, ci] * pre(x[n, ci, p]) )
 *   x          (n, >= cin, hw) fp32, plane contiguous, batch stride x_batch_stride floats (a channel prefix of a
 *              dense-block slab is fine)
 *   pre_scale / pre_shift  optional (cin,): pre(v) = a_ci v + b_ci (eval-mode BatchNorm), then max(., 0) if pre_relu
 *   w_t        (cin rounded up to 16, cout) fp32: the weight TRANSPOSED, rows past cin ZERO (made once, when the
 *              inference copy is folded); cout % 4 == 0.  cout % 128 == 0 and hw % 4 == 0 select the LDS-DMA kernel
 *   workspace  nw_conv1x1_workspace_bytes(n, cin, cout, hw) bytes (partial tiles when K is split over workgroups:
 *              small planes); may be NULL when that is 0
 *   bias       optional (cout,) (the BatchNorm that FOLLOWS the convolution, folded); post_relu: max(., 0)
 *   out        (n, cout, hw), batch stride out_batch_stride floats */
size_t nw_conv1x1_workspace_bytes(int64_t n, int64_t cin, int64_t cout, int64_t hw);
int nw_conv1x1_f32(const float *x, int64_t x_batch_stride, const float *pre_scale, const float *pre_shift,
                   int pre_relu, const float *w_t, const float *bias, int post_relu, float *out,
                   int64_t out_batch_stride, void *workspace, size_t workspace_bytes,
                   int64_t n, int64_t cin, int64_t cout, int64_t hw, void *stream);

/* 3x3 convolution, stride 1, padding 1, as an implicit GEMM on the fp32 matrix cores with bias / residual / ReLU fused
 * in; the output may be a channel window of a wider tensor (out_batch_stride), e.g. a DenseNet block's slab.  Replaces
 * the 3x3 convolutions of the folded inference backbones (model/densenet.py:41-45 conv2, model/densenet3.py:10-22,
 * model/resnet.py:31-66 BasicBlock conv + folded BatchNorm + ReLU [+ identity]).
 *     out[n, co, y, x] = post( bias[co] + sum_{ci,ky,kx} W[co,ci,ky,kx] in[n, ci, y+ky-1, x+kx-1] [+ residual[n, co, y, x]] )
 *   x         (n, >= cin, H, W) fp32, planes contiguous, batch stride x_batch_stride floats
 *   w_t       (ceil(cin/8), 9, 8, cout) fp32: the weight re-laid out once at fold time,
 *             w_t[c / 8][3 ky + kx][c % 8][co] = W[co][c][ky][kx], channels past cin ZERO; cout % 32 == 0
 *   bias      optional (cout,); residual optional (n, cout, H, W) with its batch stride; post_relu: max(., 0) */
size_t nw_conv3x3_workspace_bytes(int64_t n, int64_t cin, int64_t cout, int64_t H, int64_t W);
int nw_conv3x3_f32(const float *x, int64_t x_batch_stride, const float *w_t, const float *bias,
                   const float *residual, int64_t res_batch_stride, int post_relu, float *out,
                   int64_t out_batch_stride, void *workspace, size_t workspace_bytes,
                   int64_t n, int64_t cin, int64_t cout, int64_t H, int64_t W, void *stream);
/* Workgroups nw_conv3x3_f32 launches for a shape (0: unsupported): 32-256 pixels x 32-128 output channels per
 * workgroup, 32 x 64 tiles with the K range shared by the workgroup's waves when the larger tiles would leave the
 * chip idle (small planes), their K range split over up to 8 workgroups when there are many input channels (7x7
 * planes: partial tiles in `workspace`, nw_conv3x3_workspace_bytes, added in order by a second kernel).  The folded
 * backbones keep MIOpen below 192. */
int64_t nw_conv3x3_workgroups(int64_t n, int64_t cin, int64_t cout, int64_t H, int64_t W);


/* ---------------------------------------------------------------------------------------------
 * Convolution of the backbones on the fp16 matrix cores at fp32-grade accuracy (csrc/conv_nhwc.hip): an implicit
 * GEMM over fp32 NHWC (torch channels_last) activations.  Replaces F.conv2d at model/resnet.py:31-66 (BasicBlock's
 * 3x3), :147-156 / :178-190 (strided 3x3 and the 1x1 projection of a stage's first block), model/densenet.py:33-60
 * (conv1 1x1, conv2 3x3), :82-91 (transition 1x1); a stride-1 convolution's data gradient (train.py:414) is the same
 * call on the flipped, transposed weight.
 *     y[n, yo, xo, co] = post( bias[co] + sum_{ky,kx,ci} W[co,ky,kx,ci] x[n, s yo + ky - pad, s xo + kx - pad, ci] [+ residual] )
 *   x         (n, H, W, Cin) fp32, Cin % 32 == 0 -- or few channels with KW * Cin <= 32 (the stems: Cin = 4 after
 *             nw_to_nhwc_pad_f32 is the fast form); the weight is then the (Cout, KH, 32) matrix [co][ky][kx * Cin + ci],
 *             zero-padded, through the same split
 *   amax_in   the amax record of x: NW_AMAX_SLOTS floats whose maximum is an upper bound on max|x| (nw_absmax_f32, or the
 *             amax_out of the call that wrote x; partial maxima per producing workgroup: no atomics, nothing to clear):
 *             the activations are split into fp16 pairs on the way into LDS with ONE power of two per tensor
 *   w_split, w_scale   the weight as (Cout, KH*KW*Cin) rows -- the bytes of a channels_last (Cout, Cin, KH, KW) tensor --
 *             through nw_split_rows_f16x2 (its third output, the row norms, is not used); Cout % 32 == 0
 *   bias      optional (Cout,);  residual optional (n, Ho, Wo, Cout);  relu != 0: max(., 0) (NaN kept)
 *   y         (n, Ho, Wo, Cout) fp32;  amax_out optional: the amax record of y (NW_AMAX_SLOTS floats, all written)
 *   ldx, ldy  floats between consecutive pixels of x / y (0: Cin / Cout, dense): a channel window of a wider NHWC tensor
 *             as input or output (a dense block's slab, model/densenet.py:62-80, takes each layer's 32 channels in place);
 *             residual stays dense
 * nw_conv2d_nhwc_supported: 1 when the shape is served (else nw_conv2d_nhwc_f16x2 returns NW_ERR_UNSUPPORTED).
 * ------------------------------------------------------------------------------------------- */
#define NW_AMAX_SLOTS 256
int nw_absmax_f32(const float *x, int64_t count, float *amax_out, void *stream);
/* x (n, c, hw) fp32 with element strides (stride_n, stride_c, stride_p) -- NCHW or channels_last -- to y (n, hw, cp)
 * channels-last with the channels c .. cp-1 zero (cp % 4 == 0), and amax_out <- max|x|: the network input as the
 * 4-channel NHWC tensor the 7x7 stems (model/resnet.py:147, model/densenet.py:118) read pixel by aligned pixel. */
int nw_to_nhwc_pad_f32(const float *x, float *y, float *amax_out, int64_t n, int64_t c, int64_t hw, int64_t cp,
                       int64_t stride_n, int64_t stride_c, int64_t stride_p, void *stream);
int nw_conv2d_nhwc_supported(int64_t n, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int64_t KH, int64_t KW,
                             int64_t stride, int64_t pad);
int nw_conv2d_nhwc_f16x2(const float *x, const float *amax_in, const float *w_split, const float *w_scale,
                         const float *bias, const float *residual, int relu, float *y, float *amax_out,
                         int64_t n, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int64_t KH, int64_t KW,
                         int64_t stride, int64_t pad, int64_t ldx, int64_t ldy, float *moments, void *stream);
/* moments (nullable, Cin % 32 == 0): the convolution also leaves BatchNorm's batch statistics of y, so the BatchNorm that
 * follows (model/densenet.py:33-60: conv1 -> norm2; the next layers' norm1 over conv2's channels) needs no pass over y:
 * per group g of output pixels and channel co, moments[(k G + g) Cout + co] = k 0: pixels in the group, 1: their mean,
 * 2: the sum of squared deviations from it, 3: their minimum, 4: their maximum; G = nw_conv2d_nhwc_moments_groups(same shape
 * arguments) groups (5 G Cout floats), merged by nw_bn_nhwc_moments_from_partials_f32 / nw_bn_nhwc_prep_from_partials_f32. */
/* Round 4 -- BatchNorm + ReLU in front of a convolution, applied by the convolution's loaders on the way into LDS
 * (model/densenet.py:36-45: norm1 -> relu1 -> conv1, norm2 -> relu2 -> conv2 without the tensors in between;
 * :86-90 likewise): y = conv(relu((x - mean) a + beta)), a = gamma / sqrt(var + eps).
 *   pre      3 Cin floats: mean | a | beta of the Cin channels the convolution reads (nw_bn_nhwc_prep_f32 /
 *            nw_bn_nhwc_prep_from_partials_f32 make it; an inference caller fills it from the running statistics)
 *   amax_in  raw_records == 0: ONE amax record bounding |relu((x - mean) a + beta)| (the same two entries leave the exact
 *            bound; any upper bound is legal).  raw_records = n > 0 (an inference caller without batch statistics): n
 *            consecutive amax records of the RAW tensor x -- those of its producers: a dense block's input and every layer's
 *            output so far -- and the kernel derives the bound itself, max_c |a_c| (A + |mean_c|) + max(beta_c, 0) with A the
 *            records' maximum
 *   bias / relu: the inference epilogue (a folded BatchNorm behind the convolution); moments: as nw_conv2d_nhwc_f16x2.
 * `moments` of this and of nw_conv2d_nhwc_f16x2 hold FIVE rows per group since round 4: count, mean, M2, minimum,
 * maximum (5 G Cout floats). */
int nw_conv2d_nhwc_bnrelu_f16x2(const float *x, const float *pre, const float *amax_in, int64_t raw_records,
                                const float *w_split, const float *w_scale, const float *bias, int relu, float *y,
                                float *amax_out, int64_t n,
                                int64_t H, int64_t W, int64_t Cin, int64_t Cout, int64_t KH, int64_t KW, int64_t stride,
                                int64_t pad, int64_t ldx, int64_t ldy, float *moments, void *stream);
int64_t nw_conv2d_nhwc_moments_groups(int64_t n, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int64_t KH, int64_t KW,
                                      int64_t stride, int64_t pad);
/* The data gradient of a convolution whose input was relu(batch_norm(x)) (model/densenet.py:33-60: norm - relu - conv):
 * y = dL/d relu(bn(x)) is computed as nw_conv2d_nhwc_f16x2 does (no bias / residual / ReLU), and the epilogue also leaves
 * BatchNorm's backward statistics: per pixel group and channel, partials[(k G + group) Cout + co] = k 0: sum g, 1: sum g xhat,
 * with g = y where the forward's bn(x) = (x - mean) gamma invstd + beta was positive, xhat = (x - mean) invstd; G =
 * nw_conv2d_nhwc_moments_groups(same shape arguments).  x: (pixels, >= Cout) fp32 with row stride ldx, the tensor the
 * BatchNorm read (channel c of x is channel c of y).  nw_bn_relu_nhwc_train_bwd_from_partials_f32 takes it from there. */
typedef struct nw_conv_bnstat {
    const float *x; int64_t ldx;
    const float *mean, *invstd, *gamma, *beta;   /* (Cout,) each */
    float *partials;                             /* 2 G Cout floats */
} nw_conv_bnstat;
int nw_conv2d_nhwc_bnstat_f16x2(const float *x, const float *amax_in, const float *w_split, const float *w_scale, float *y,
                                float *amax_out, int64_t n, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int64_t KH,
                                int64_t KW, int64_t stride, int64_t pad, int64_t ldx, int64_t ldy,
                                const nw_conv_bnstat *bnstat, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Training-mode BatchNorm2d (+ ReLU) in front of / behind the backbones' convolutions, forward and
 * backward, one kernel each (torch: batch-norm kernels + a relu kernel each way).  Same call sites as
 * nw_scale_shift_relu_f32 plus the conv -> bn -> relu pairs of model/resnet.py:31-66; the backward is what
 * autograd derives for relu(batch_norm(x)) at train.py:414.
 *   forward : batch statistics per channel over (n, hw); y = max(gamma (x - mean) / sqrt(var + eps) + beta, 0);
 *             running_mean/var (nullable) updated in place with `momentum` (unbiased variance), the int64
 *             num_batches_tracked counter (nullable) incremented, save_mean / save_invstd (c,) written for the
 *             backward
 *   backward: dx (n, c, hw), dgamma (c,), dbeta (c,) from dy (n, c, hw) contiguous; the ReLU mask is recomputed
 *             from x, nothing else is saved
 *   residual (nullable, (n, c, hw) contiguous): y = max(bn(x) + residual, 0), the tail of a ResNet block
 *             (model/resnet.py:58-66, :100-108); the backward then also writes dresidual (the masked dy)
 *   acc (nullable, backward): a second gradient of x, element (i, ch, p) at acc[i * acc_batch_stride + ch * hw + p],
 *             added into dx (the running concatenation of a dense block, model/densenet.py:62-80, feeds this
 *             BatchNorm and the next concatenation: autograd would add the two with a strided kernel of its own)
 *   x element (i, ch, p) at x[i * x_batch_stride + ch * hw + p]; y, dy, dx contiguous
 * ------------------------------------------------------------------------------------------- */
int nw_bn_relu_train_fwd_f32(const float *x, const float *residual, const float *gamma, const float *beta, float *running_mean,
                             float *running_var, float *y, float *save_mean, float *save_invstd,
                             int64_t *num_batches_tracked, int64_t n, int64_t c, int64_t hw,
                             int64_t x_batch_stride, float momentum, float eps, int relu, void *stream);
int nw_bn_relu_train_bwd_f32(const float *x, const float *residual, const float *dy, const float *gamma,
                             const float *beta, const float *save_mean, const float *save_invstd, float *dx,
                             float *dresidual, float *dgamma, float *dbeta, const float *acc,
                             int64_t acc_batch_stride, int64_t n, int64_t c, int64_t hw,
                             int64_t x_batch_stride, int relu, void *stream);

/* Every convolution weight of a network to the split-row operands nw_conv2d_nhwc_f16x2 reads, in ONE launch (the weights
 * change with every optimizer step, train.py:414-415).  jobs: device array of njobs x 10 int64 --
 *   [0] address of a torch-contiguous (Cout, Cin, KH, KW) fp32 weight, [1] offset (floats) of the operand in split_base,
 *   [2] offset of its row scales in scale_base, [3] index of its first row among all rows (ascending), [4] rows, [5] cols
 *   (floats per row, % 32 == 0), [6] Cin, [7] Cout, [8] KH * KW, [9] KW | mode << 32 with
 *   mode 0: the forward operand, rows = Cout, row co = [tap][ci];  mode 1: the data-gradient operand of a stride-1
 *   convolution, rows = Cin, row ci = [flipped tap][co];  mode 2: few input channels (Cin <= 4), rows = Cout,
 *   row co = [ky][32 floats: kx * 4 + ci, zero-padded] (the operand of the stems over a 4-channel input). */
int nw_split_conv_weights_f16x2(const int64_t *jobs, int64_t njobs, int64_t total_rows, float *split_base,
                                float *scale_base, void *stream);

/* Weight gradient of a stride-1 'same' convolution (1x1, or 3x3 with padding 1) over channels-last activations on the fp16
 * matrix cores (csrc/conv_wgrad.hip): what autograd derives for the weight of F.conv2d at model/densenet.py:33-60, :82-91
 * and model/resnet.py:31-66 in loss.backward() (train.py:414).
 *     dw[co, ky, kx, ci] = sum_{n,y,x} gy[n, y, x, co] * x[n, y + ky - pad, x + kx - pad, ci]
 *   x (n, H, W, Cin), gy (n, H, W, Cout) fp32 with their amax records; Cin % 8 == 0, Cout % 8 == 0; ldx, ldg: floats
 *   between consecutive pixels of x / gy (0: dense)
 *   dw (Cout, KH, KW, Cin) fp32: the bytes of a channels_last (Cout, Cin, KH, KW) tensor
 *   workspace: nw_conv2d_nhwc_wgrad_workspace_bytes(...) (partial tiles of the position chunks, added in order)
 * nw_conv2d_nhwc_wgrad_supported: 1 when the shape is served, else the call returns NW_ERR_UNSUPPORTED. */
int nw_conv2d_nhwc_wgrad_supported(int64_t n, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int64_t KH, int64_t KW,
                                   int64_t stride, int64_t pad);
size_t nw_conv2d_nhwc_wgrad_workspace_bytes(int64_t n, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int64_t KH,
                                            int64_t KW, int64_t stride, int64_t pad);
int nw_conv2d_nhwc_wgrad_f16x2(const float *x, const float *amax_x, const float *gy, const float *amax_g, float *dw,
                               void *workspace, size_t workspace_bytes, int64_t n, int64_t H, int64_t W, int64_t Cin,
                               int64_t Cout, int64_t KH, int64_t KW, int64_t stride, int64_t pad, int64_t ldx, int64_t ldg,
                               void *stream);
/* Several independent weight gradients in as few launches as possible (all 3x3 problems in one kernel, all 1x1 problems in
 * another, 32 per launch): the layers of a dense block on 14x14 / 7x7 maps give a dozen workgroups each, and their weight
 * gradients are not on the backward's critical path -- collected per block and run together they fill the chip.  Each job
 * as in nw_conv2d_nhwc_wgrad_f16x2; workspace: nw_conv2d_nhwc_wgrad_batch_workspace_bytes(jobs, njobs) (every job its own
 * part).  Nothing is launched if any job is refused. */
typedef struct nw_wgrad_job {
    const float *x, *amax_x, *gy, *amax_g;
    float *dw;
    int64_t n, H, W, Cin, Cout, KH, KW, stride, pad, ldx, ldg;
    int64_t out_oihw;   /* != 0: dw in torch's contiguous (Cout, Cin, KH, KW) layout instead of (Cout, KH, KW, Cin) */
    const float *pre_x; /* nullable: 3 Cin floats mean | a | beta -- the x operand is relu((x - mean) a + beta), as the forward
                           convolution read it (nw_conv2d_nhwc_bnrelu_f16x2); amax_x then bounds THAT tensor */
    /* rowrun_stride = s > 0 (round 4: the strided few-channel stems, model/densenet.py:114-116, model/resnet.py:147): the weight
     * gradient of a KH x KW convolution over a 4-channel NHWC input x (in_H x in_W pixels; RGB zero-padded to four channels) as
     * ONE 1x1 problem over gy's grid (H, W = output size, KH = KW = 1, stride 1, pad 0) with Cin = 32 KH "channels": channel
     * 32 ky + 4 kx + ci of output pixel (yo, xo) is x[s yo + row0 + ky][s xo + col0 + kx][ci] (zero outside the image; row0 =
     * col0 = -padding).  dw (Cout, 32 KH): [co][32 ky + 4 kx + ci]; the columns with kx >= KW or ci = 3 are not part of dW. */
    int64_t rowrun_stride, in_H, in_W, row0, col0;
} nw_wgrad_job;
size_t nw_conv2d_nhwc_wgrad_batch_workspace_bytes(const nw_wgrad_job *jobs, int64_t njobs);
int nw_conv2d_nhwc_wgrad_batch_f16x2(const nw_wgrad_job *jobs, int64_t njobs, void *workspace, size_t workspace_bytes,
                                     void *stream);
/* device address of 32 bytes of zeros the convolution kernels read in place of pixels that do not exist (internal) */
const void *nw_conv_zero_page(void);

/* ---------------------------------------------------------------------------------------------
 * The same pair over channels-last activations (csrc/bn_nhwc.hip), for the training path whose convolutions run in
 * nw_conv2d_nhwc_f16x2: x is (rows = n h w, c) with row stride ldx >= c floats (a channel prefix of a wider NHWC
 * tensor qualifies), c % 4 == 0 and c <= 2560 (NW_ERR_UNSUPPORTED beyond: the per-channel factors live in LDS), y / dy / dx
 * dense (rows, c).  Moments per row chunk merged in a fixed order (Chan),
 * deterministic; amax_out (nullable): the amax record (NW_AMAX_SLOTS floats) of y / dx for the convolution that reads
 * it next.  acc (nullable, backward): a second gradient of x with row stride ldacc, added into dx; lddx: row stride of
 * dx (0: dense); dx may be acc itself (the gradient slab of a dense block accumulates in place).
 * workspace: nw_bn_nhwc_workspace_bytes(rows, c).
 * ------------------------------------------------------------------------------------------- */
size_t nw_bn_nhwc_workspace_bytes(int64_t rows, int64_t c);
int nw_bn_relu_nhwc_train_fwd_f32(const float *x, int64_t ldx, const float *gamma, const float *beta, float *running_mean,
                                  float *running_var, float *y, float *save_mean, float *save_invstd,
                                  int64_t *num_batches_tracked, float *amax_out, void *workspace, size_t workspace_bytes,
                                  int64_t rows, int64_t c, float momentum, float eps, int relu, void *stream);
/* The forward in phases, for a caller that has the batch statistics from elsewhere (a dense block: a channel's statistics
 * are the same for every later layer's norm1, and a convolution's epilogue leaves the moments of what it writes):
 *   nw_bn_nhwc_moments_f32                mean, 1/sqrt(var + eps), biased var of the c channels of x (c,) each
 *   nw_bn_nhwc_moments_from_partials_f32  the same from the groups a convolution left (nw_conv2d_nhwc_f16x2's `moments`)
 *   nw_bn_relu_nhwc_apply_f32             y = act((x - mean) gamma invstd + beta) + the amax record of y; updates THIS
 *                                         layer's running statistics (momentum; unbiased var) and step counter */
int nw_bn_nhwc_moments_f32(const float *x, int64_t ldx, int64_t rows, int64_t c, float eps, float *mean, float *invstd,
                           float *var, void *workspace, size_t workspace_bytes, void *stream);
int nw_bn_nhwc_moments_from_partials_f32(float *partials, int64_t groups, int64_t c, float eps, float *mean,
                                         float *invstd, float *var, void *stream);
/* Round 4: what a convolution that applies the BatchNorm itself needs (nw_conv2d_nhwc_bnrelu_f16x2).
 *   nw_bn_nhwc_moments_minmax_f32        nw_bn_nhwc_moments_f32 + every channel's minimum and maximum
 *                                        (workspace: nw_bn_nhwc_minmax_workspace_bytes)
 *   nw_bn_nhwc_prep_from_partials_f32    statistics (mean, invstd, var, vmin, vmax) from a convolution's 5-row `moments`
 *                                        and, with gamma != NULL, the table `tab` (3 c floats mean | a | beta), the exact
 *                                        bound on |act((x - mean) a + beta)| as an amax record, the layer's running
 *                                        statistics (nullable) and step counter (nullable), in the same launch
 *   nw_bn_nhwc_prep_f32                  the table and bound from statistics that exist (rows = samples per channel) */
int nw_bn_nhwc_moments_minmax_f32(const float *x, int64_t ldx, int64_t rows, int64_t c, float eps, float *mean, float *invstd,
                                  float *var, float *vmin, float *vmax, void *workspace, size_t workspace_bytes, void *stream);
size_t nw_bn_nhwc_minmax_workspace_bytes(int64_t rows, int64_t c);
int nw_bn_nhwc_prep_from_partials_f32(float *partials, int64_t groups, int64_t c, float eps, float *mean, float *invstd,
                                      float *var, float *vmin, float *vmax, const float *gamma, const float *beta,
                                      float *running_mean, float *running_var, int64_t *num_batches_tracked, float momentum,
                                      int relu, float *tab, float *amax_out, void *stream);
/* ... the same where the fresh channels are a WINDOW [offset, offset + c) of wider statistics arrays (a dense block's slab:
 * mean .. vmax are the slab-wide arrays) and the BatchNorm to prepare (gamma, beta, running statistics: offset + c channels) also
 * reads the n_old <= offset channels in front of it, whose statistics exist: one launch does both (rows: samples per channel). */
int nw_bn_nhwc_prep_window_from_partials_f32(float *partials, int64_t groups, int64_t c, int64_t offset, int64_t n_old, int64_t rows,
                                             float eps, float *mean, float *invstd, float *var, float *vmin, float *vmax,
                                             const float *gamma, const float *beta, float *running_mean, float *running_var,
                                             int64_t *num_batches_tracked, float momentum, int relu, float *tab, float *amax_out,
                                             void *stream);
int nw_bn_nhwc_prep_f32(const float *mean, const float *invstd, const float *var, const float *vmin, const float *vmax,
                        const float *gamma, const float *beta, float *running_mean, float *running_var,
                        int64_t *num_batches_tracked, float momentum, int relu, int64_t rows, int64_t c, float *tab,
                        float *amax_out, void *stream);
int nw_bn_relu_nhwc_apply_f32(const float *x, int64_t ldx, const float *mean, const float *invstd, const float *var,
                              const float *gamma, const float *beta, float *running_mean, float *running_var,
                              int64_t *num_batches_tracked, float momentum, float *y, float *amax_out, int64_t rows,
                              int64_t c, int relu, void *stream);
/* The backward from the statistics a data-gradient convolution left (nw_conv2d_nhwc_bnstat_f16x2): the groups are summed
 * (-> dgamma, dbeta), then dx as in nw_bn_relu_nhwc_train_bwd_f32 (relu != 0).  workspace: 2 c floats. */
int nw_bn_relu_nhwc_train_bwd_from_partials_f32(const float *x, int64_t ldx, const float *dy, const float *gamma,
                                                const float *beta, const float *save_mean, const float *save_invstd,
                                                const float *partials, int64_t groups, float *dx, float *dgamma,
                                                float *dbeta, const float *acc, int64_t ldacc, int64_t lddx,
                                                float *amax_out, void *workspace, size_t workspace_bytes, int64_t rows,
                                                int64_t c, void *stream);
/* The ImageNet stems at inference in one kernel (csrc/stem_pool.hip; model/resnet.py:147, :200-203, model/densenet.py:114-120 with
 * the BatchNorm folded into weight and bias): y = maxpool3x3/2/1(relu(conv7x7/2/3(x) + bias)), 64 output channels; the
 * 112 x 112 map between the two never reaches memory.
 *   x4 (n, H, W, 4) fp32: the 4-channel padded input of nw_to_nhwc_pad_f32, amax_in its amax record;
 *   w_split / w_scale: the (64, 7 x 32) few-channel operand of nw_split_rows_f16x2 (k = 32 ky + 4 kx + ci, as nw_conv2d_nhwc_f16x2
 *     takes it for a 4-channel input); bias (64,);
 *   y (n, Hp, Wp, >= 64) with row stride ldy floats (0: 64), Hp = ((H - 1) / 2 + 1 - 1) / 2 + 1; amax_out (nullable): its record.
 * Same values as nw_conv2d_nhwc_f16x2(bias, relu) + nw_maxpool3x3s2_nhwc_f32 up to fp32 summation order. */
int nw_stem7x7s2_relu_maxpool_supported(int64_t n, int64_t H, int64_t W, int64_t cout);
int nw_stem7x7s2_relu_maxpool_f16x2(const float *x4, const float *amax_in, const float *w_split, const float *w_scale,
                                    const float *bias, float *y, float *amax_out, int64_t n, int64_t H, int64_t W, int64_t ldy,
                                    void *stream);
/* An eval-mode DenseNet transition's BatchNorm + ReLU with its 2 x 2 average pool in front of the bias-free 1 x 1 convolution
 * (model/densenet.py:83-91 runs norm -> relu -> conv -> pool; pool and convolution commute): y = avgpool2x2(relu((x - mean) a + beta)),
 * tab = mean | a | beta (c floats apart), rows of x / y may be strided (ldx, ldy; 0: c); amax_out (nullable): the record of y. */
int nw_bn_relu_avgpool2x2_nhwc_f32(const float *x, int64_t ldx, const float *tab, float *y, int64_t ldy, float *amax_out,
                                   int64_t n, int64_t h, int64_t w, int64_t c, void *stream);
/* The end of a residual block in training (model/resnet.py:60-66, :100-108: out = relu(bn(.) + identity)) over two tensors of
 * equal layout, `count` floats each (a multiple of 4), 16-byte aligned: nw_add_relu_f32 writes out = relu(a + b) (x < 0 ? 0 : x:
 * keeps a NaN) and out's amax record; nw_relu_bwd_f32 writes dx = g where out > 0, else 0 -- the gradient of BOTH summands --
 * and dx's amax record.  amax_out nullable. */
int nw_add_relu_f32(const float *a, const float *b, float *out, float *amax_out, int64_t count, void *stream);
int nw_relu_bwd_f32(const float *out, const float *g, float *dx, float *amax_out, int64_t count, void *stream);
/* Backward of `norm1 -> relu1 -> conv1` of a dense layer (model/densenet.py:36-40; the 1x1 bottleneck convolution) in two
 * streaming passes, without the convolution's data gradient ever reaching memory (csrc/bn_dgrad.hip, round 4):
 *   du (rows, k) fp32, dense: dL/d(conv1 output), k = conv1's output channels (multiple of 32, <= 128); amax_du: its amax record;
 *   w_split / w_scale: conv1's DATA-GRADIENT operand (the (c, k) matrix W^T in nw_split_rows_f16x2 form: what
 *     nw_conv2d_nhwc_f16x2 takes to compute the data gradient of a 1x1 convolution);
 *   x (rows, >= c) with row stride ldx: the tensor norm1 read; tab: norm1's table mean | a | beta (rows tab_stride floats apart,
 *     a = gamma invstd: nw_bn_nhwc_prep_*); invstd (c,);
 *   g (rows, >= c) with row stride ldg: dL/dx is ADDED into g[:, :c] in place; amax_out (nullable): amax record of the result;
 *   dgamma, dbeta (c,): written.  workspace: nw_bn_dgrad1x1_workspace_bytes(rows, c).
 * Equals nw_conv2d_nhwc_f16x2 (data gradient) + nw_bn_relu_nhwc_train_bwd_f32(acc = dx = g) up to fp32 summation order.
 * NW_ERR_UNSUPPORTED for c % 32 != 0 or k outside {32, 64, 96, 128} (callers keep the two-step path for those). */
size_t nw_bn_dgrad1x1_workspace_bytes(int64_t rows, int64_t c);
int nw_bn_dgrad1x1_bwd_f16x2(const float *du, const float *amax_du, const float *w_split, const float *w_scale,
                             const float *x, int64_t ldx, const float *tab, int64_t tab_stride, const float *invstd,
                             float *g, int64_t ldg, float *amax_out, float *dgamma, float *dbeta, void *workspace,
                             size_t workspace_bytes, int64_t rows, int64_t c, int64_t k, void *stream);
int nw_bn_relu_nhwc_train_bwd_f32(const float *x, int64_t ldx, const float *dy, const float *gamma, const float *beta,
                                  const float *save_mean, const float *save_invstd, float *dx, float *dgamma,
                                  float *dbeta, const float *acc, int64_t ldacc, int64_t lddx, float *amax_out,
                                  void *workspace, size_t workspace_bytes, int64_t rows, int64_t c, int relu, void *stream);

/* ---------------------------------------------------------------------------------------------
 * The pools of the backbones' training path over channels-last fp32 activations (csrc/pool_nhwc.hip): the 2 x 2 / 2
 * average pool of a DenseNet transition (reference model/densenet.py:83-91, nn.AvgPool2d(2, 2): an odd last row / column
 * is dropped) and the 3 x 3 / 2 pad 1 max pool of the stems (model/densenet.py:114, model/resnet.py:147; torch's scan:
 * a later value of the window wins when it is greater or NaN).  x (n, h, w, c), c % 4 == 0, rows may be strided (ld* >= c
 * floats, 0: dense): a pool can read from / write into a channel prefix of a wider NHWC tensor.  The max pool leaves the
 * winning tap (ky * 3 + kx) of every output value in `tap` ((n, ho, wo, c) bytes, dense; nullable: inference), which is all its backward needs
 * beside gy; both backwards are gathers over the input pixels (no atomics, deterministic) and write every value of gx.
 * ------------------------------------------------------------------------------------------- */
int nw_avgpool2x2_nhwc_f32(const float *x, int64_t ldx, float *y, int64_t ldy, int64_t n, int64_t h, int64_t w, int64_t c,
                           void *stream);
int nw_avgpool2x2_nhwc_bwd_f32(const float *gy, int64_t ldgy, float *gx, int64_t ldgx, int64_t n, int64_t h, int64_t w,
                               int64_t c, void *stream);
int nw_maxpool3x3s2_nhwc_f32(const float *x, int64_t ldx, float *y, int64_t ldy, unsigned char *tap, int64_t n, int64_t h,
                             int64_t w, int64_t c, void *stream);
int nw_maxpool3x3s2_nhwc_bwd_f32(const float *gy, int64_t ldgy, const unsigned char *tap, float *gx, int64_t ldgx, int64_t n,
                                 int64_t h, int64_t w, int64_t c, void *stream);

/* ---------------------------------------------------------------------------------------------
 * The optimizer step of the training harness (reference train.py:243-247: torch.optim.SGD(momentum 0.9, weight decay,
 * nesterov)) for all parameter tensors of a network in a few launches (csrc/sgd.hip; 96 tensors per launch, their addresses
 * in the kernel arguments).  Per element, torch's formula with dampening 0:
 *   g = grad + weight_decay p;  buf = init_buf ? g : momentum buf + g;  p -= lr (nesterov ? g + momentum buf : buf)
 * momentum == 0: no buffer is read or written (momentum_buf may be NULL).  init_buf != 0: the first step of these
 * parameters (torch clones the gradient into the buffer).  fp32, any length; tensors 16-byte aligned take the vector path.
 * ------------------------------------------------------------------------------------------- */
typedef struct nw_sgd_param {
    float *param;
    const float *grad;
    float *momentum_buf;
    int64_t n;
} nw_sgd_param;
int nw_sgd_step_f32(const nw_sgd_param *params, int64_t nparams, float lr, float momentum, float weight_decay, int nesterov,
                    int init_buf, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Diagnostics (no reference counterpart): device time of the TILE kernel alone -- the kernel the
 * roofline is quoted on (nw_fused_kernel / nw_fused_f16p_kernel), without the small kernels around
 * it (query split, run tables, merge).  While enabled, every forward brackets its tile-kernel launch
 * with two HIP events on the launch stream; nw_debug_tile_timing_read waits for them, returns their
 * summed elapsed time and the number of launches since the last read, and clears both.
 * ------------------------------------------------------------------------------------------- */
int nw_debug_tile_timing(int enable);
/* Diagnostic knobs (timing experiments; process-wide; never needed in normal use).  Names: pvar, qg, tile_rs, merge_mq,
 * merge_per_query, merge_no_global_tables, persistent_any_rs, no_persistent, split_queries, bwd_no_mfma, bwd_split,
 * coeff_threads, xgemm_wgs, xgemm_nbuf, split_lbits, conv_gather, conv_max_wgs, wgrad_min_stages, conv_skip_cfgs,
 * wgrad_batch_wgs, bn_inline_fin, conv_moments_per_tile, conv_force_cfg.  The Python layer forwards
 * the NW_<NAME> environment variables once, when it loads the library; the library itself never reads the environment. */
int nw_debug_set(const char *name, int value);
int nw_debug_tile_timing_read(double *total_us, int64_t *launches);

#ifdef __cplusplus
}
#endif
#endif /* NWHEAD_HIP_H */
