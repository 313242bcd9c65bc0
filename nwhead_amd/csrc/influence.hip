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

__global__ __launch_bounds__(256) void nw_influence_kernel(
    const float* __restrict__ probs, const int64_t* __restrict__ qy, const float* __restrict__ w,
    const int64_t* __restrict__ sy, float* __restrict__ infl, int64_t N, int64_t C) {
    const int64_t b = blockIdx.y;
    const int64_t q = qy[b];
    const float p = ((uint64_t)q < (uint64_t)C) ? probs[b * C + q] : 0.f;
    const float* wr = w + b * N;
    float* orow = infl + b * N;
    const bool vec = ((N & 3) == 0) && (((reinterpret_cast<uintptr_t>(wr) | reinterpret_cast<uintptr_t>(orow)) & 15) == 0);
    if (vec) {
        const int64_t n4 = N >> 2;
        for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n4; j += (int64_t)gridDim.x * 256) {
            const float4 v = reinterpret_cast<const float4*>(wr)[j];
            const int64_t* y = sy + 4 * j;
            float4 o;
            o.x = infl_one(p, v.x, y[0] == q);
            o.y = infl_one(p, v.y, y[1] == q);
            o.z = infl_one(p, v.z, y[2] == q);
            o.w = infl_one(p, v.w, y[3] == q);
            reinterpret_cast<float4*>(orow)[j] = o;
        }
    } else {
        for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < N; j += (int64_t)gridDim.x * 256)
            orow[j] = infl_one(p, wr[j], sy[j] == q);
    }
}

}  // namespace
}  // namespace nw

extern "C" int nw_support_influence_f32(const float* probs, const int64_t* qy, const float* w,
                                        const int64_t* sy, float* infl, int64_t B, int64_t N,
                                        int64_t C, void* stream) {
    if (B < 0 || N < 0 || C < 0) return NW_ERR_INVALID_ARG;
    if (B == 0 || N == 0) return NW_OK;
    if (!probs || !qy || !w || !sy || !infl) return NW_ERR_INVALID_ARG;
    if (B > 65535) {
        // grid.y limit: walk the batch in slabs
        for (int64_t b0 = 0; b0 < B; b0 += 65535) {
            const int64_t nb = (B - b0 < 65535) ? (B - b0) : 65535;
            const int rc = nw_support_influence_f32(probs + b0 * C, qy + b0, w + b0 * N, sy,
                                                    infl + b0 * N, nb, N, C, stream);
            if (rc != NW_OK) return rc;
        }
        return NW_OK;
    }
    int64_t gx = (N / 4 + 255) / 256;
    if (gx < 1) gx = 1;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(nw::nw_influence_kernel, dim3((unsigned)gx, (unsigned)B), dim3(256), 0,
                       static_cast<hipStream_t>(stream), probs, qy, w, sy, infl, N, C);
    NW_CHECK_LAUNCH();
    return NW_OK;
}
