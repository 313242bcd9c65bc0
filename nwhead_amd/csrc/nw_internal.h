// Internal declarations shared by the .hip translation units of libnwhead_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nwhead_hip.h"

#define NW_LOG_EPS 1e-12f   /* nwhead/nw.py:289 */
#define NW_NORM_EPS 1e-12f  /* F.normalize default eps, nwhead/kernel.py:19-20 */

#define NW_CHECK_LAUNCH()                          \
    do {                                           \
        if (hipGetLastError() != hipSuccess) return NW_ERR_LAUNCH; \
    } while (0)

namespace nw {

// scores (B,N) <- q (B,d), s (N,d) | (B,N,d)
int launch_scores(const float* q, const float* s, float* scores, int64_t B, int64_t N, int64_t d,
                  int kind, const float* logit_scale_dev, int sup_batched, hipStream_t st);

// softmax over supports + per-class aggregation of one (B,N) score matrix
//   final:   out (B,C), optional lse (B,), optional weights (B,N)
//   partial: m (B,), den (B,), num (B,C)
// slice_ws (nullable, slice_ws_floats): scratch of aggregate_slices(B, N) * B * (2 + C) floats -- with it, few queries over
// long rows are aggregated by several workgroups per query (partials + merge)
int aggregate_slices(int64_t B, int64_t N);
int launch_aggregate(const float* scores, const int64_t* sy, int labels_batched, float* out,
                     float* lse, float* weights, float* m, float* den, float* num, int64_t B,
                     int64_t N, int64_t C, hipStream_t st, float* slice_ws = nullptr, size_t slice_ws_floats = 0);

int launch_merge(const float* m, const float* den, const float* num, float* out, int64_t G,
                 int64_t B, int64_t C, int64_t sm, int64_t sd, int64_t sn, const int64_t* class_lo,
                 int64_t CL, hipStream_t st);

// influence.hip: softmax weights / influences from a score matrix and the forward's log-sum-exp (in place allowed)
int launch_weights_from_scores(const float* scores, const float* lse, float* weights, int64_t B, int64_t N, hipStream_t st);
int launch_influence(const float* probs_or_logp, const int64_t* qy, const float* w_or_scores, const int64_t* sy,
                     const float* lse, float* infl, int64_t B, int64_t N, int64_t C, hipStream_t st);

// squared row norms of a (rows,d) matrix (backward.hip)
int launch_rownorm2(const float* x, float* n2, int64_t rows, int64_t d, hipStream_t st);

// bwd_split.hip: the backward's two products on the fp16 matrix cores (split-row operands)
struct XgemmPlan {
    int nchunks, k_chunk;
};
constexpr size_t XGEMM_TAIL_BYTES = 512;   // readable bytes every operand buffer needs past its last row
XgemmPlan xgemm_plan(int64_t M, int64_t Nn, int64_t K);
// A K-split product's reduction (out = f(m) sum_chunks part + 2 rowscale Xo), to be run by extra workgroups of a LATER
// launch_xgemm on the same stream instead of a kernel of its own (blocks == 0: nothing pending).
struct XgemmReduce {
    const float* part;
    const float* fac;
    const float* gfac;
    const float* rowscale;
    const float* Xo;
    float* out;
    int64_t M, Nn;
    int nchunks, fac_inverse, blocks;
};
int launch_xgemm(bool x_km, const float* X, int64_t ldx, int64_t x_rows, const float* Y, int64_t ldy, int64_t y_rows,
                 float* part, const float* fac, int fac_inverse, const float* gfac, const float* rowscale,
                 const float* Xo, float* out, int64_t M, int64_t Nn, int64_t K, hipStream_t st,
                 XgemmReduce* defer = nullptr, const XgemmReduce* pending = nullptr);

// Options of the forward call in progress on this thread (nw_fwd_opts of the ABI, set for the duration of the call by
// the entry point): explicit arguments, not process state.
struct FwdOpts {
    const char* tables = nullptr;   // run tables of the call's labels (nw_bank_tables_build), or null
    size_t tables_bytes = 0;
    const int64_t* tables_sy = nullptr;   // ... the label array and row count they were built from
    int64_t tables_N = -1;
    int persistent_wgs = 0;         // workgroups of the persistent tile kernel (0: one per CU)
    int force_split = 0;            // split-fp16 path at every size
};
const FwdOpts& fwd_opts();
// Diagnostic knobs (nw_debug_set; timing experiments, never needed in normal use).  KNOB_UNSET when not set.  The library
// itself never reads the environment: the Python layer forwards NW_* variables once, at load time (_lib.py).
constexpr int KNOB_UNSET = -2147483647 - 1;
enum Knob { KNOB_PVAR, KNOB_QG, KNOB_TILE_RS, KNOB_MERGE_MQ, KNOB_MERGE_PER_QUERY, KNOB_MERGE_NO_GLOBAL_TABLES,
            KNOB_PERSISTENT_ANY_RS, KNOB_NO_PERSISTENT, KNOB_SPLIT_QUERIES, KNOB_BWD_NO_MFMA, KNOB_BWD_SPLIT,
            KNOB_COEFF_THREADS, KNOB_XGEMM_WGS, KNOB_XGEMM_NBUF, KNOB_SPLIT_LBITS, KNOB_CONV_GATHER, KNOB_CONV_MAX_WGS,
            KNOB_WGRAD_MIN_STAGES, KNOB_CONV_SKIP_CFGS, KNOB_WGRAD_BATCH_WGS, KNOB_BN_INLINE_FIN, KNOB_CONV_MOMENTS_PER_TILE,
            KNOB_CONV_FORCE_CFG, KNOB_COUNT };
int knob(int id);

// fused forward (fused.hip)
int launch_topk(const float* scores, int64_t* idx, float* vals, int64_t B, int64_t N, int64_t k, hipStream_t st);  // topk.hip
int tile_timer_enable(bool on);
int tile_timer_read(double* total_us, int64_t* launches);
int pick_rs(int64_t B, int64_t N, int64_t d, bool f16 = false);
bool fused_eligible(const float* q, const float* s, int64_t B, int64_t N, int64_t d, int64_t C);
size_t fused_workspace_bytes(int64_t B, int64_t N, int64_t d, int64_t C = 0);
int launch_split_rows(const float* x, float* out, float* scale, float* norm2, int64_t rows, int64_t d,
                      hipStream_t st);
int launch_fused(const float* q, const float* s, const int64_t* sy, const float* s_norm2,
                 const float* s_scale,
                 const float* logit_scale_dev, float* out, float* scores, float* lse, float* m,
                 float* den, float* num,
                 void* workspace, size_t workspace_bytes, int64_t B, int64_t N, int64_t d, int64_t C,
                 int kind, hipStream_t st);

// bn_nhwc.hip, for bn_dgrad.hip: the BatchNorm backward finalize over G groups of partial sums
int bn_bwd_finalize_groups(const float* part, int G, int C, float inv_m, float* dgamma, float* dbeta, float* k, hipStream_t st);

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// Block-wide reductions for 256-thread workgroups (4 waves); `red` is >= 8 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
    const int nw_ = (blockDim.x + 63) >> 6;
    for (int w = 0; w < nw_; ++w) r += red[w];
    return r;
}
// counted wait for all but the wave's N youngest vector-memory operations; every counted wait of the library goes through
// here so that the 6-bit field is checked at compile time (an overflow once hung the test suite: DESIGN.md 4.7h)
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = -INFINITY;
    const int nw_ = (blockDim.x + 63) >> 6;
    for (int w = 0; w < nw_; ++w) r = fmaxf(r, red[w]);
    return r;
}

// exp(x) for x <= 0 in ~6 VALU instructions (libm expf is ~20; in an fp32-MFMA kernel every VALU
// instruction is matrix-pipe time).  exp(x) = 2^(x*log2e): the product is split into its rounded
// value t and its rounding error e (one FMA), 2^t comes from v_exp_f32 (<= 1 ulp) and the error is
// folded back as 2^t * (1 + e*ln2).  Relative error ~2e-7 over [-100, 0]; exact 0 for -inf.
__device__ __forceinline__ float fast_exp_neg(float x) {
    const float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.925963033500011e-8f;
    const float t = x * L2E_HI;
    const float e = __builtin_fmaf(x, L2E_HI, -t) + x * L2E_LO;
    const float r = __builtin_amdgcn_exp2f(t);
    return (x == -INFINITY) ? 0.f : __builtin_fmaf(r * e, 0.693147182464599609375f, r);
}

// sqrt(max(x,0)) with the hardware v_sqrt_f32 (<= 1 ulp) instead of the ~12-instruction correctly
// rounded sequence; x is a squared distance, 1 ulp of its root is below the fp32 spacing the
// reference's own result has.
__device__ __forceinline__ float fast_sqrt_pos(float x) { return __builtin_amdgcn_sqrtf(fmaxf(x, 0.f)); }

// Reductions over the four lanes {i, i+16, i+32, i+48} of a wave (the four k-groups of a 16x16 MFMA
// accumulator column), result in all four.  gfx950's v_permlane{32,16}_swap are plain VALU ops
// (mov + swap + op = 3 issue slots per step); __shfl_xor goes through ds_bpermute and its LDS round
// trip (~120 cycles per step, exposed when one wave per SIMD runs an epilogue).
__device__ __forceinline__ float group4_sum(float x) {
    typedef unsigned u2_ __attribute__((ext_vector_type(2)));
    u2_ r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float group4_min(float x);
__device__ __forceinline__ float group4_max(float x) {
    typedef unsigned u2_ __attribute__((ext_vector_type(2)));
    u2_ r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

__device__ __forceinline__ float group4_min(float x) {
    typedef unsigned u2_ __attribute__((ext_vector_type(2)));
    u2_ r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = fminf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fminf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// score from (dot, |q|^2, |s|^2); shared by the MFMA and the generic kernels
template <int KIND>
__device__ __forceinline__ float score_from_dot(float dot, float qn2, float sn2, float scale) {
    if (KIND == NW_SCORE_DOT) return dot;
    if (KIND == NW_SCORE_EUCLIDEAN) return -fast_sqrt_pos(qn2 + sn2 - 2.f * dot);
    const float nq = fmaxf(sqrtf(qn2), NW_NORM_EPS), ns = fmaxf(sqrtf(sn2), NW_NORM_EPS);
    const float c = dot / (nq * ns);
    if (KIND == NW_SCORE_COSINE) return c;
    if (KIND == NW_SCORE_CLIP) return scale * c;
    // hypersphere: -||x/|x| - y/|y|||, matmul form like torch's cdist for N > 25
    const float a = qn2 / (nq * nq), b = sn2 / (ns * ns);
    return -fast_sqrt_pos(a + b - 2.f * c);
}


// The same scores, factored for the tile epilogues: with per-support-row factors (K, Base), per-query
// factors (Cq, Bq) and x = acc * K * Cq + (Base + Bq), the score in BASE-2 units (u = score * log2 e) is
// finish(x).  `acc` is the raw accumulator (for split-fp16 operands the dot product times 2^(e_q + e_s);
// sscale / qscale = 2^-e of the row, 1 for fp32 operands).  Everything that depends on one row only --
// square roots, reciprocals, the normalisation of the cosine-type kernels -- is done once per row here
// instead of once per (query, support) pair; 1/x is v_rcp_f32 (1 ulp), well inside the 1e-5 bar.
template <int KIND>
struct ScoreFactors {
    static constexpr float L2E = 1.44269504088896340736f;
    static constexpr bool DIST = (KIND == NW_SCORE_EUCLIDEAN || KIND == NW_SCORE_HYPERSPHERE);
    static constexpr bool NORMALISED = (KIND == NW_SCORE_HYPERSPHERE || KIND == NW_SCORE_COSINE || KIND == NW_SCORE_CLIP);
    static __device__ __forceinline__ float inv_norm(float n2) {  // 1 / max(|x|, eps), F.normalize's eps
        return __builtin_amdgcn_rcpf(fmaxf(__builtin_amdgcn_sqrtf(n2), NW_NORM_EPS));
    }
    static __device__ __forceinline__ void support(float n2, float sscale, float& K, float& Base) {
        const float in = NORMALISED ? inv_norm(n2) : 1.f;
        K = sscale * in * (DIST ? -2.f * L2E * L2E : L2E);
        Base = DIST ? (NORMALISED ? n2 * in * in : n2) * (L2E * L2E) : 0.f;
    }
    static __device__ __forceinline__ void query(float n2, float qscale, float clip_scale, float& Cq, float& Bq) {
        const float in = NORMALISED ? inv_norm(n2) : 1.f;
        Cq = qscale * in * (KIND == NW_SCORE_CLIP ? clip_scale : 1.f);
        Bq = DIST ? (NORMALISED ? n2 * in * in : n2) * (L2E * L2E) : 0.f;
    }
    static __device__ __forceinline__ float finish(float x) { return DIST ? -fast_sqrt_pos(x) : x; }
    // the same without the sign for the distance kernels (u = -finish_abs(x)): the persistent epilogue
    // carries distances and takes their MINIMUM, which saves one negation per pair
    static __device__ __forceinline__ float finish_abs(float x) { return DIST ? fast_sqrt_pos(x) : x; }
};

}  // namespace nw
