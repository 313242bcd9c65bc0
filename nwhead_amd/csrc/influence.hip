// influence.hip -- support_influence, vectorised over the query batch (gfx950 / MI355X only).
//
// Replaces util/metric.py:23-50, which loops over queries in Python with a host sync per query:
//     p = softmax[b, qy_b];  ind_j = [sy_j == qy_b];  infl[b,j] = log((p - p*w_bj)/(p - w_bj*ind_j))
// Pure streaming, HBM-bound: 4*B*N bytes in, 4*B*N out, labels 8*N (L2-resident across rows).
// Every operation is rounded on its own (__fmul_rn / __fsub_rn / __fdiv_rn: no FMA contraction), so
// the near-zero residue p - w of a one-shot class lands on the same side of zero as in the
// reference and the +inf / NaN / huge-finite outcomes are reproduced.
#include "nw_internal.h"

namespace nw {
namespace {

// The two differences are rounded exactly as the reference rounds them (their SIGN and ZERO decide
// between a finite value, +inf and NaN); the quotient and the logarithm only have to be accurate:
// v_rcp_f32 and v_log_f32 (<= 1-2 ulp) keep IEEE's special cases (x/0 = +-inf, 0/0 = NaN,
// log(negative) = NaN, log(0) = -inf, log(inf) = inf) at a quarter of the instruction count.
__device__ __forceinline__ float infl_one(float p, float w, bool same) {
    const float nume = __fsub_rn(p, __fmul_rn(p, w));
    const float deno = __fsub_rn(p, same ? w : 0.f);   // w * ind with ind in {0,1} is exact
    const float ratio = __fmul_rn(nume, __builtin_amdgcn_rcpf(deno));
    return __builtin_amdgcn_logf(ratio) * 0.693147180559945309417f;
}

// One thread = one 16-byte column chunk (4 supports) of IROWS query rows: the chunk's labels are read once and
// compared per row, the IROWS weight loads are independent (all in flight), every access is a full 1 KB per wave.
// (One row per workgroup re-read the int64 labels -- 2 bytes per byte of weights -- from L2 for every row:
//  7.5 us per 20.6 MB call when the rows stream from HBM.)
// SCORES: `w` holds raw scores, the weight is exp(score - lse[b]) (the forward's log-sum-exp) and p is
// exp(logp[b, qy_b]): the influence straight from the fused forward's score matrix, in place if infl == w.
constexpr int IROWS = 4;
typedef float infl_f4 __attribute__((ext_vector_type(4)));
typedef long long infl_l2 __attribute__((ext_vector_type(2)));
// Round 4: at BASELINE's shape (256 x 10000, 20.6 MB) the kernel is a handful of memory latencies
// long, and the first version spent most of them in its prologue -- per row a scalar load of qy[b], a wait, a
// (conditional) scalar load of probs[b, qy_b], a wait: eight dependent round trips in front of the first vector load.
// Now the first chunk's label and weight loads are issued at the top, the IROWS labels of the query rows come with one
// load, the IROWS probabilities with independent unconditional loads (index clamped, value selected afterwards), so the
// prologue is two overlapped round trips under the weights' own latency; weights and influences are streamed with
// non-temporal accesses (read once / written once).
template <bool SCORES>
__global__ __launch_bounds__(256) void nw_influence_kernel(
    const float* __restrict__ probs, const int64_t* __restrict__ qy, const float* w,
    const int64_t* __restrict__ sy, const float* __restrict__ lse, float* infl, int64_t B, int64_t N, int64_t C) {
    const int64_t b0 = (int64_t)blockIdx.y * IROWS;
    const int64_t n4 = N >> 2;
    const bool vec = ((N & 3) == 0) && (((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(infl)) & 15) == 0);
    const bool full = b0 + IROWS <= B;            // (the last row group of a batch may be short)
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    infl_l2 ya = {0, 0}, yb = {0, 0};
    infl_f4 v[IROWS];
    auto load_chunk = [&](int64_t jj) {
        const infl_l2* y = reinterpret_cast<const infl_l2*>(sy + 4 * jj);
        ya = y[0];
        yb = y[1];
#pragma unroll
        for (int r = 0; r < IROWS; ++r)
            if (full || b0 + r < B) v[r] = __builtin_nontemporal_load(reinterpret_cast<const infl_f4*>(w + (b0 + r) * N) + jj);
    };
    // program order = issue order: the rows' labels (scalar loads), the first chunk (vector loads, independent of them),
    // then the probabilities, whose addresses need the labels.  (Loading the IROWS rows of `probs` whole and handing
    // p = probs[b, qy_b] over through LDS removes the dependent load, but every column block re-reads the rows -- 10 %
    // more bytes at 256 x 10000 x 200: 6.2 us against 5.8.)
    int64_t q[IROWS];
    float p[IROWS], l[IROWS];
#pragma unroll
    for (int r = 0; r < IROWS; ++r) q[r] = qy[full ? b0 + r : min(b0 + r, B - 1)];
    bool have = vec && j < n4;
    if (have) load_chunk(j);
#pragma unroll
    for (int r = 0; r < IROWS; ++r) {
        const int64_t b = full ? b0 + r : min(b0 + r, B - 1);
        const bool ok = (uint64_t)q[r] < (uint64_t)C;
        const float pv = probs[b * C + (ok ? q[r] : 0)];       // always loaded: four independent loads, one wait
        const float pe = ok ? pv : (SCORES ? -INFINITY : 0.f);
        p[r] = SCORES ? expf(pe) : pe;
        l[r] = SCORES ? lse[b] : 0.f;
    }
    if (vec) {
        while (have) {
            infl_f4 o[IROWS];
#pragma unroll
            for (int r = 0; r < IROWS; ++r) {
                if (!(full || b0 + r < B)) continue;
                infl_f4 x = v[r];
                if (SCORES) x = infl_f4{expf(x[0] - l[r]), expf(x[1] - l[r]), expf(x[2] - l[r]), expf(x[3] - l[r])};
                o[r][0] = infl_one(p[r], x[0], ya[0] == q[r]);
                o[r][1] = infl_one(p[r], x[1], ya[1] == q[r]);
                o[r][2] = infl_one(p[r], x[2], yb[0] == q[r]);
                o[r][3] = infl_one(p[r], x[3], yb[1] == q[r]);
            }
            const int64_t jc = j;
            j += stride;
            have = j < n4;
            if (have) load_chunk(j);              // (N > 65536: more than one chunk per thread)
#pragma unroll
            for (int r = 0; r < IROWS; ++r)
                if (full || b0 + r < B) __builtin_nontemporal_store(o[r], reinterpret_cast<infl_f4*>(infl + (b0 + r) * N) + jc);
        }
    } else {
        for (; j < N; j += stride) {
            const int64_t y = sy[j];
#pragma unroll
            for (int r = 0; r < IROWS; ++r) {
                if (b0 + r >= B) continue;
                float x = w[(b0 + r) * N + j];
                if (SCORES) x = expf(x - l[r]);
                infl[(b0 + r) * N + j] = infl_one(p[r], x, y == q[r]);
            }
        }
    }
}

// weights[b, j] = exp(scores[b, j] - lse[b]): the softmax weights from the fused forward's score matrix (in place
// when weights == scores)
__global__ __launch_bounds__(256) void nw_weights_from_scores_kernel(const float* scores, const float* __restrict__ lse,
                                                                     float* weights, int64_t B, int64_t N) {
    const int64_t b0 = (int64_t)blockIdx.y * IROWS;
    const int64_t n4 = N >> 2;
    const bool vec = ((N & 3) == 0) && (((reinterpret_cast<uintptr_t>(scores) | reinterpret_cast<uintptr_t>(weights)) & 15) == 0);
    float l[IROWS];
#pragma unroll
    for (int r = 0; r < IROWS; ++r) l[r] = lse[min(b0 + r, B - 1)];
    if (vec) {
        for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n4; j += (int64_t)gridDim.x * 256) {
            float4 v[IROWS];
#pragma unroll
            for (int r = 0; r < IROWS; ++r)
                if (b0 + r < B) v[r] = reinterpret_cast<const float4*>(scores + (b0 + r) * N)[j];
#pragma unroll
            for (int r = 0; r < IROWS; ++r)
                if (b0 + r < B)
                    reinterpret_cast<float4*>(weights + (b0 + r) * N)[j] =
                        make_float4(expf(v[r].x - l[r]), expf(v[r].y - l[r]), expf(v[r].z - l[r]), expf(v[r].w - l[r]));
        }
    } else {
        for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < N; j += (int64_t)gridDim.x * 256)
#pragma unroll
            for (int r = 0; r < IROWS; ++r)
                if (b0 + r < B) weights[(b0 + r) * N + j] = expf(scores[(b0 + r) * N + j] - l[r]);
    }
}

inline unsigned col_blocks(int64_t N) {
    int64_t gx = (N / 4 + 255) / 256;
    return (unsigned)(gx < 1 ? 1 : (gx > 64 ? 64 : gx));
}

}  // namespace
}  // namespace nw

namespace nw {
int launch_weights_from_scores(const float* scores, const float* lse, float* weights, int64_t B, int64_t N, hipStream_t st) {
    for (int64_t b0 = 0; b0 < B; b0 += 65535 * IROWS) {  // grid.y limit: walk the batch in slabs
        const int64_t nb = (B - b0 < 65535 * IROWS) ? (B - b0) : 65535 * IROWS;
        hipLaunchKernelGGL(nw_weights_from_scores_kernel, dim3(col_blocks(N), (unsigned)((nb + IROWS - 1) / IROWS)), dim3(256), 0, st,
                           scores + b0 * N, lse + b0, weights + b0 * N, nb, N);
    }
    NW_CHECK_LAUNCH();
    return NW_OK;
}
int launch_influence(const float* probs_or_logp, const int64_t* qy, const float* w_or_scores, const int64_t* sy,
                     const float* lse, float* infl, int64_t B, int64_t N, int64_t C, hipStream_t st) {
    for (int64_t b0 = 0; b0 < B; b0 += 65535 * IROWS) {
        const int64_t nb = (B - b0 < 65535 * IROWS) ? (B - b0) : 65535 * IROWS;
        const dim3 grid(col_blocks(N), (unsigned)((nb + IROWS - 1) / IROWS));
        if (lse)
            hipLaunchKernelGGL(nw_influence_kernel<true>, grid, dim3(256), 0, st, probs_or_logp + b0 * C, qy + b0,
                               w_or_scores + b0 * N, sy, lse + b0, infl + b0 * N, nb, N, C);
        else
            hipLaunchKernelGGL(nw_influence_kernel<false>, grid, dim3(256), 0, st, probs_or_logp + b0 * C, qy + b0,
                               w_or_scores + b0 * N, sy, nullptr, infl + b0 * N, nb, N, C);
    }
    NW_CHECK_LAUNCH();
    return NW_OK;
}
}  // namespace nw

extern "C" int nw_support_influence_f32(const float* probs, const int64_t* qy, const float* w,
                                        const int64_t* sy, float* infl, int64_t B, int64_t N,
                                        int64_t C, void* stream) {
    if (B < 0 || N < 0 || C < 0) return NW_ERR_INVALID_ARG;
    if (B == 0 || N == 0) return NW_OK;
    if (!probs || !qy || !w || !sy || !infl) return NW_ERR_INVALID_ARG;
    return nw::launch_influence(probs, qy, w, sy, nullptr, infl, B, N, C, static_cast<hipStream_t>(stream));
}
